set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
(time timeout 900 python -m pytest tests/test_gpu_pipe.py -x -q) > gpurun_out/s2/pipetests.log 2>&1
for pipe in 0 -1 0 -1; do
  for c in c2 c2i c5; do
    echo "PIPE=$pipe $c" >> gpurun_out/s2/ab.txt
    if [ "$pipe" = "-1" ]; then timeout 300 python tools/run_case.py $c 300 >> gpurun_out/s2/ab.txt 2>&1; else GPUNTT_PIPE=$pipe timeout 300 python tools/run_case.py $c 300 >> gpurun_out/s2/ab.txt 2>&1; fi
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s2/kt -o c2 -- python $GRAFT_REPO_ROOT/tools/run_case.py c2 100 > $GRAFT_REPO_ROOT/gpurun_out/s2/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/s2/kt > gpurun_out/s2/kt_summary.txt 2>&1
find gpurun_out/s2/kt -name "*.db" -size +20M -delete
