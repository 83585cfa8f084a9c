python -m pytest tests/test_gpu_merge.py tests/test_gpu_4step.py tests/test_gpu_round5.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for lib in libgpuntt_base.so libgpuntt.so; do
  export GPUNTT_LIB=$PWD/gpu-ntt_amd/lib/$lib
  echo "== $lib"
  python bench.py --config c4 --steps 50 --warmup 10 --no-traffic --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c4', d['ms_per_step'], d['roofline']['frac'])"
  python bench.py --sweep --sweep-bits 32 --direction fwd --no-cpu-baseline --no-traffic --sweep-kinds merge 2>/dev/null | python -c "
import sys,json
r=[]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r.append('%s%d %.4f'%(d['algo'][0],d['log2N'],d['ms_per_step']))
print('fwd',' '.join(r))"
done
done
