set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
(time timeout 900 python -m pytest tests/test_gpu_pipe.py -x -q) > gpurun_out/s3/pipetests.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s3/kt_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/s3/kt_c3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py gpurun_out/s3/kt_c3 > gpurun_out/s3/kt_c3_summary.txt 2>&1
find gpurun_out/s3 -name "*.db" -size +20M -delete
