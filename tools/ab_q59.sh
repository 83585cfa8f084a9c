#!/bin/bash
# A/B of the "hi32(q) = 2^27" form of the quotient product (VERDICT r3 #4a): the product library against
# tools/_exp_q59.so (same sources, lazy_u64_fwd31.hip compiled with -DGPUNTT_EXP_Q59), alternating in ONE session.
#   bash tools/ab_q59.sh > gpurun_out/ab_q59.txt
# The experiment library is not kept in the tree (tools/_exp_* is git-ignored); rebuild it with
#   cd gpu-ntt_amd/csrc && mkdir -p _obj_q59 && cp -p _obj/*.o _obj_q59/ && \
#   hipcc -O3 -std=c++17 -fPIC -I../../include -I. --offload-arch=gfx950 -Wno-unused-result -ffp-contract=off \
#         -DGPUNTT_EXP_Q59 -c lazy_u64_fwd31.hip -o _obj_q59/lazy_u64_fwd31.o && \
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_exp_q59.so _obj_q59/*.o -Wl,-rpath,/opt/rocm/lib
# Result of round 4 (profiles/r04_c2_last_levers.txt): 0.9 % slower, not taken.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "# bit-exactness of the experiment build (Merge forward tests, pool primes = the 31 q kernels)"
GPUNTT_LIB=$PWD/tools/_exp_q59.so python -m pytest tests/test_gpu_merge.py -q -m gpu -k "forward_inverse_all_sizes or golden_vectors or full_size_c2" 2>&1 | tail -n 2
for rep in 1 2 3 4; do
  for lib in product q59; do
    if [ $lib = q59 ]; then export GPUNTT_LIB=$PWD/tools/_exp_q59.so; else unset GPUNTT_LIB; fi
    python bench.py --steps 400 --warmup 50 --no-traffic --no-cpu-baseline --no-power 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep %-8s C2 ms_per_step %.4f  call_ms(HIP events) %.4f' % ('$lib', d['ms_per_step'], d['roofline']['call_ms_hip_events']))"
  done
done
unset GPUNTT_LIB
echo "# per-kernel averages (rocprofv3 --kernel-trace --stats), one run each"
for lib in product q59; do
  if [ $lib = q59 ]; then export GPUNTT_LIB=$PWD/tools/_exp_q59.so; else unset GPUNTT_LIB; fi
  rm -rf /tmp/prof_$lib; (R=$PWD; cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o p -- python $R/bench.py --steps 200 --warmup 20 --no-traffic --no-cpu-baseline --no-power > /dev/null 2>&1)
  echo "## $lib"; python tools/rocprof_summary.py /tmp/prof_$lib 2>/dev/null | head -n 8
done
