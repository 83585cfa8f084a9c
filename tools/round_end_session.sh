#!/bin/bash
# The measurement session whose files go to profiles/ at the end of a round (one MI355X):
#   bash tools/round_end_session.sh [outdir]        (default gpurun_out/round_end)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/${1:-gpurun_out/round_end}"
mkdir -p "$O"
cd "$R"
python bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench_line.err"
python bench.py --steps 20 --warmup 5 --api plan --no-traffic --no-cpu-baseline > "$O/bench_line_plan.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --direction inv > "$O/bench_line_c2i.json" 2> "$O/bench_line_c2i.err"
python bench.py --config c3 --steps 5 --warmup 2 > "$O/bench_line_c3.json" 2> "$O/bench_line_c3.err"
python bench.py --config c4 --steps 50 --warmup 10 > "$O/bench_line_c4.json" 2> "$O/bench_line_c4.err"
python bench.py --config c5 --steps 20 --warmup 5 > "$O/bench_line_c5.json" 2> "$O/bench_line_c5.err"
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the C2 and C3 bench commands
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/kt_c2" -o kt -- python "$R/bench.py" --steps 100 --warmup 10 --no-traffic --no-cpu-baseline --no-power > /dev/null 2>&1)
python tools/rocprof_summary.py "$O/kt_c2" > "$O/bench_kernel_stats.txt" 2>&1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/kt_c3" -o kt -- python "$R/bench.py" --config c3 --steps 5 --warmup 2 --no-traffic --no-cpu-baseline --no-power > /dev/null 2>&1)
python tools/rocprof_summary.py "$O/kt_c3" > "$O/c3_kernel_stats.txt" 2>&1
rm -rf "$O/kt_c2" "$O/kt_c3"
# north_star's table: every ring 2^12 .. 2^24, both algorithms, both word sizes
python bench.py --sweep --sweep-bits 64 > "$O/sweep_u64.jsonl" 2> "$O/sweep_u64.err"
python bench.py --sweep --sweep-bits 32 > "$O/sweep_u32.jsonl" 2> "$O/sweep_u32.err"
python bench.py --sweep --sweep-bits 64 --direction inv > "$O/sweep_u64_inv.jsonl" 2> "$O/sweep_u64_inv.err"
python bench.py --sweep --sweep-bits 32 --direction inv > "$O/sweep_u32_inv.jsonl" 2> "$O/sweep_u32_inv.err"
python tools/bench_small_dropin.py > "$O/small_dropin.txt" 2>/dev/null
python tools/bench_4step_small.py > "$O/4step_small_calls.txt" 2>/dev/null
python tools/bench_batch1.py > "$O/batch1.txt" 2>&1
ls -la "$O"
