#!/bin/bash
# The measurement session whose files go to profiles/ at the end of a round (one MI355X).  Started through
# tools/run_round_end.sh, which refuses a dirty tree and hands the commit over (the GPU box has no .git):
#   SESSION_HEAD=<sha> bash tools/round_end_session.sh [outdir]        (default gpurun_out/round_end)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$R/${1:-gpurun_out/round_end}"
mkdir -p "$O"
cd "$R"
if [ -z "$SESSION_HEAD" ]; then echo "SESSION_HEAD not set: start this through tools/run_round_end.sh" >&2; exit 2; fi
{ echo "head $SESSION_HEAD"; echo "date $(date -u +%FT%TZ)"; sha256sum gpu-ntt_amd/lib/libgpuntt.so bench.py | sed 's/^/sha256 /'; } > "$O/SESSION.txt"
# the driver's form: headline + every other BASELINE config in one run
python bench.py --steps 20 --warmup 5 > "$O/bench_line.json" 2> "$O/bench_line.err"
python bench.py --steps 20 --warmup 5 --api plan --no-traffic --no-cpu-baseline --no-other-configs > "$O/bench_line_plan.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --direction inv > "$O/bench_line_c2i.json" 2> "$O/bench_line_c2i.err"
python bench.py --config c3 --steps 5 --warmup 2 > "$O/bench_line_c3.json" 2> "$O/bench_line_c3.err"
python bench.py --config c4 --steps 50 --warmup 10 > "$O/bench_line_c4.json" 2> "$O/bench_line_c4.err"
python bench.py --config c4 --steps 50 --warmup 10 --direction inv > "$O/bench_line_c4i.json" 2> "$O/bench_line_c4i.err"
python bench.py --config c5 --steps 20 --warmup 5 > "$O/bench_line_c5.json" 2> "$O/bench_line_c5.err"
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the C2, C3 and C4 bench commands
for cfg in c2 c3 c4; do
  extra="--steps 100 --warmup 10"; [ $cfg = c3 ] && extra="--steps 5 --warmup 2"
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/kt_$cfg" -o kt -- python "$R/bench.py" --config $cfg $extra --no-traffic --no-cpu-baseline --no-power --no-other-configs --no-shard-overheads > /dev/null 2>&1)
  python tools/rocprof_summary.py "$O/kt_$cfg" > "$O/${cfg}_kernel_stats.txt" 2>&1
  rm -rf "$O/kt_$cfg"
done
mv "$O/c2_kernel_stats.txt" "$O/bench_kernel_stats.txt"
# SQ counters of the two headline-class kernels on this library
# (tools/pmc_sq.sh takes its output name relative to gpurun_out/)
REL="${1:-gpurun_out/round_end}"; REL="${REL#gpurun_out/}"
{ sha256sum gpu-ntt_amd/lib/libgpuntt.so | sed 's/^/# library sha256 /'; bash tools/pmc_sq.sh "$REL/pmc_c2" c2 5 2>&1; } > "$O/pmc_c2.txt"; rm -rf "$O/pmc_c2" "$O"/pmc_c2.p*.log
{ sha256sum gpu-ntt_amd/lib/libgpuntt.so | sed 's/^/# library sha256 /'; bash tools/pmc_sq.sh "$REL/pmc_c4" c4full 5 2>&1; } > "$O/pmc_c4.txt"; rm -rf "$O/pmc_c4" "$O"/pmc_c4.p*.log
{ sha256sum gpu-ntt_amd/lib/libgpuntt.so | sed 's/^/# library sha256 /'; bash tools/pmc_sq.sh "$REL/pmc_c4i" merge:32:14:8192:inv 5 2>&1; } > "$O/pmc_c4i.txt"; rm -rf "$O/pmc_c4i" "$O"/pmc_c4i.p*.log
# north_star's table: every ring 2^12 .. 2^24, both algorithms, both word sizes, both directions
python bench.py --sweep --sweep-bits 64 > "$O/sweep_u64.jsonl" 2> "$O/sweep_u64.err"
python bench.py --sweep --sweep-bits 32 > "$O/sweep_u32.jsonl" 2> "$O/sweep_u32.err"
python bench.py --sweep --sweep-bits 64 --direction inv > "$O/sweep_u64_inv.jsonl" 2> "$O/sweep_u64_inv.err"
python bench.py --sweep --sweep-bits 32 --direction inv > "$O/sweep_u32_inv.jsonl" 2> "$O/sweep_u32_inv.err"
python tools/bench_small_dropin.py > "$O/small_dropin.txt" 2>/dev/null
python tools/bench_4step_small.py > "$O/4step_small_calls.txt" 2>/dev/null
python tools/bench_batch1.py > "$O/batch1.txt" 2>&1
make -s -C tests/cpp > /dev/null 2>&1
{ tests/cpp/_bin/bench_multi_device c2 20 5; tests/cpp/_bin/bench_multi_device c4 50 10; } > "$O/cpp_bench_multi_device.jsonl" 2>&1
ls -la "$O"
