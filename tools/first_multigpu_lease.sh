#!/bin/bash
# The first lease of a multi-GPU node as ONE command with a verdict.  Nothing in this repo has ever run on two physical
# GPUs (every earlier lease was a 1-GPU box), so the steps go from "does RCCL come up" to the scaling table,
# each under its own timeout, each with its output kept, and a failing step does not stop the ones behind it.
#     bash tools/first_multigpu_lease.sh            run on this node (GPUS="2 4 8" by default, capped at the device count)
#     bash tools/first_multigpu_lease.sh --list     print the command list only (what tests/test_dist_shard.py parses)
# Output: gpurun_out/multigpu/NN_<step>.log + gpurun_out/multigpu/VERDICT.txt (one line per step: PASS / FAIL / TIMEOUT).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT" || exit 2
export HSA_ENABLE_IPC_MODE_LEGACY="${HSA_ENABLE_IPC_MODE_LEGACY:-0}"
OUT="${MULTIGPU_OUT:-gpurun_out/multigpu}"
GPUS="${GPUS:-2 4 8}"
PORT_BASE="${PORT_BASE:-29610}"
TR="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"

# step list: "<timeout s>|<name>_x<ranks>|<command>"   ({P} = rendezvous port, filled in per step)
STEPS=()
for N in $GPUS; do
  STEPS+=("180|rccl_smoke_x$N|$TR --nproc-per-node $N --master-port {P} tools/rccl_smoke.py")
done
for N in $GPUS; do
  STEPS+=("600|bench_c2_x$N|$TR --nproc-per-node $N --master-port {P} bench.py --gpus $N --steps 20 --warmup 5")
done
for N in $GPUS; do
  STEPS+=("600|bench_c4_strong_x$N|$TR --nproc-per-node $N --master-port {P} bench.py --gpus $N --config c4 --steps 50 --warmup 10")
done
for N in $GPUS; do
  STEPS+=("1500|sweep_x$N|$TR --nproc-per-node $N --master-port {P} bench.py --gpus $N --sweep")
done
for N in $GPUS; do
  STEPS+=("300|digests_x$N|$TR --nproc-per-node $N --master-port {P} tools/multigpu_digests.py")
done
# the C++ product on every device of the node: one host thread per device, drop-in call + NTTPlan, every polynomial checked
for N in $GPUS; do
  STEPS+=("300|cpp_multi_device_x$N|make -s -C tests/cpp && tests/cpp/_bin/example_multi_device 16 512 $N 8")
done
# ... and TIMED: resident shards, one host thread per device, HIP events, max over the devices -- the scaling table of the
# product without Python or RCCL (tests/cpp/bench_multi_device.cpp prints one JSON line per device count, 1 included)
STEPS+=("600|cpp_bench_c2_x1|make -s -C tests/cpp && tests/cpp/_bin/bench_multi_device c2 20 5 1,2,4,8")
STEPS+=("600|cpp_bench_c4_x1|make -s -C tests/cpp && tests/cpp/_bin/bench_multi_device c4 50 10 1,2,4,8")
# the single-GPU reference points the scaling table is read against
STEPS+=("600|bench_c2_x1|python bench.py --gpus 1 --steps 20 --warmup 5")
STEPS+=("600|bench_c4_strong_x1|python bench.py --gpus 1 --config c4 --steps 50 --warmup 10")

expand() {  # $1 = step string, $2 = index -> the command with {P} filled in
  echo "$1" | cut -d'|' -f3- | sed "s/{P}/$((PORT_BASE + $2))/g"
}

if [ "$1" = "--list" ]; then
  i=0
  for s in "${STEPS[@]}"; do
    printf '%s|%s|%s\n' "$(echo "$s" | cut -d'|' -f1)" "$(echo "$s" | cut -d'|' -f2)" "$(expand "$s" $i)"
    i=$((i + 1))
  done
  exit 0
fi

NDEV="$(python -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)"
mkdir -p "$OUT"
: > "$OUT/VERDICT.txt"
echo "devices on this node: $NDEV" | tee -a "$OUT/VERDICT.txt"
i=0
for s in "${STEPS[@]}"; do
  tmo="$(echo "$s" | cut -d'|' -f1)"; name="$(echo "$s" | cut -d'|' -f2)"
  n="$(echo "$name" | sed -n 's/.*_x\([0-9]*\)$/\1/p')"
  cmd="$(expand "$s" $i)"
  log="$OUT/$(printf '%02d' $i)_$name.log"
  i=$((i + 1))
  if [ "$n" -gt "$NDEV" ]; then
    echo "SKIP    $name (needs $n devices)" | tee -a "$OUT/VERDICT.txt"
    continue
  fi
  echo "\$ $cmd" > "$log"
  t0=$(date +%s)
  timeout --kill-after=20 "$tmo" bash -c "$cmd" >> "$log" 2>&1
  rc=$?
  dt=$(( $(date +%s) - t0 ))
  if [ $rc -eq 0 ]; then st=PASS; elif [ $rc -eq 124 ] || [ $rc -eq 137 ]; then st=TIMEOUT; else st="FAIL(rc=$rc)"; fi
  # the one-line results of the step, if it printed any
  res="$(grep -E '^(RCCL_SMOKE|MULTIGPU_DIGESTS|All Correct on|\{"metric"|\{"program")' "$log" | tail -n 1 | cut -c1-220)"
  printf '%-8s %-22s %4ss  %s\n' "$st" "$name" "$dt" "$res" | tee -a "$OUT/VERDICT.txt"
done
# scaling table from the bench lines (value = whole-job NTT/s)
python - "$OUT" <<'PY' | tee -a "$OUT/VERDICT.txt"
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "*_bench_*.log"))):
    name = os.path.basename(f).split("_", 1)[1][:-4]
    for line in open(f):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            rows[name] = (d["n_gpus"], d["value"], d["ms_per_step"])
for cfg in ("bench_c2", "bench_c4_strong"):
    base = rows.get(cfg + "_x1")
    for name, (n, v, ms) in sorted(rows.items(), key=lambda kv: kv[1][0]):
        if name.startswith(cfg + "_x"):
            eff = (v / (n * base[1])) if base else float("nan")
            print("%-20s n_gpus=%d  value=%.4g NTT/s  ms_per_step=%.4f  efficiency_vs_x1=%.3f" % (cfg, n, v, ms, eff))
PY
# the same table from the C++ program's lines
python - "$OUT" <<'PY' | tee -a "$OUT/VERDICT.txt"
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_cpp_bench_*.log"))):
    rows = [json.loads(l) for l in open(f) if l.startswith('{"program"')]
    base = next((r for r in rows if r["n_gpus"] == 1), None)
    for r in rows:
        eff = r["value_ntt_per_s"] / (r["n_gpus"] * base["value_ntt_per_s"]) if base else float("nan")
        print("cpp %-4s n_gpus=%d  value=%.4g NTT/s  ms_per_step=%.4f  efficiency_vs_x1=%.3f  bit_exact=%s"
              % (r["config"], r["n_gpus"], r["value_ntt_per_s"], r["ms_per_step"], eff, r["bit_exact_vs_NTTCPU"]))
PY
grep -q -E '^(FAIL|TIMEOUT)' "$OUT/VERDICT.txt" && exit 1
exit 0
