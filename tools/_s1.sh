set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/s1/gputests.log 2>&1
python bench.py > gpurun_out/s1/bench_c2.json 2> gpurun_out/s1/bench_c2.err
python bench.py --config c3 > gpurun_out/s1/bench_c3.json 2> gpurun_out/s1/bench_c3.err
python bench.py --config c3 --api plan > gpurun_out/s1/bench_c3_plan.json 2> gpurun_out/s1/bench_c3_plan.err
for k in 8 9 10 11 12; do
  for c in merge:64:24:4 merge:64:20:64 merge:64:24:4:inv merge:64:18:256; do
    echo "CONTIG_K=$k $c" >> gpurun_out/s1/contigk.txt
    GPUNTT_CONTIG_K=$k python tools/run_case.py $c 10 >> gpurun_out/s1/contigk.txt 2>&1
  done
done
