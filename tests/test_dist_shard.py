"""world_size-2 `gloo` test of the batch-shard layer (gpu-ntt_amd/dist.py) on CPU.

The product has no CPU transform, so inside this test the per-rank "transform" is the oracle;
what is being tested is the partition (shard_range incl. RNS alignment), the barrier +
MAX-reduced timing bracket bench.py uses, and the digest gather: sharded result == unsharded."""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import load_pkg
g = load_pkg()
import importlib
dist_mod = importlib.import_module("gpu_ntt_amd.dist")
from oracle import oracle as O
dist, rank, world = dist_mod.init_process_group("gloo")
assert world == 2 and dist is not None
P = O.Port(64)
logn, batch, mc = 8, 24, 3
n = 1 << logn
prms = [P.merge_params(logn, O.X_N_plus, f) for f in FACTORS]
lo, hi = g.shard_range(batch, rank, world, mc)
assert lo % mc == 0 and hi % mc == 0
# every rank regenerates only its own polynomials (global index p decides seed and modulus)
x = np.concatenate([P.splitmix(100 + p, 0, n, prms[p % mc]["mod"][0]) for p in range(lo, hi)])
state = {}
def step():
    state["y"] = np.concatenate([P.merge_ntt(x[i * n:(i + 1) * n], prms[(lo + i) % mc])
                                 for i in range(hi - lo)])
wall = dist_mod.timed_region(step, 2, dist, None)
assert wall > 0
digs = dist_mod.gather_digests(state["y"], dist)
if rank == 0:
    print("DIGESTS", " ".join(digs), "SPAN", lo, hi, flush=True)
# the end-to-end leg (batch starts and ends on rank 0): scatter -> per-rank transform -> gather
import torch
shard_len = (hi - lo) * n
full = None
if rank == 0:
    full_np = np.concatenate([P.splitmix(100 + p, 0, n, prms[p % mc]["mod"][0]) for p in range(batch)])
    full = torch.from_numpy(full_np.view(np.int64).copy())
shard_in = torch.empty(shard_len, dtype=torch.int64)
shard_out = torch.empty(shard_len, dtype=torch.int64)
def run_shard(a, b):
    xa = a.numpy().view(np.uint64)
    y = np.concatenate([P.merge_ntt(xa[i * n:(i + 1) * n], prms[(lo + i) % mc]) for i in range(hi - lo)])
    b.copy_(torch.from_numpy(y.view(np.int64).copy()))
gathered, times = dist_mod.scatter_transform_gather(dist, rank, world, full, shard_in, shard_out, run_shard, None)
assert set(times) == {"scatter_s", "transform_s", "gather_s"} and all(v > 0 for v in times.values())
if rank == 0:
    print("E2E", hashlib.sha256(gathered.numpy().tobytes()).hexdigest(), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharded_equals_unsharded(tmp_path):
    from oracle import oracle as O
    P = O.Port(64)
    logn, batch, mc = 8, 24, 3
    n = 1 << logn
    factors = []
    for lg in (12, 13, 14):
        prm = P.fourstep_params(lg, with_W=False)
        q = prm["mod"][0]
        psi = pow(prm["psi"], 1 << (lg - logn), q)
        factors.append((q, psi * psi % q, psi))
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\nFACTORS = %r\n" % (ROOT, factors) + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29531",
                        str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGESTS")][0].split()
    digs = line[1:3]
    # unsharded reference run in this process
    prms = [P.merge_params(logn, O.X_N_plus, f) for f in factors]
    full = [P.merge_ntt(P.splitmix(100 + p, 0, n, prms[p % mc]["mod"][0]), prms[p % mc])
            for p in range(batch)]
    want = [hashlib.sha256(np.concatenate(full[0:12]).tobytes()).hexdigest(),
            hashlib.sha256(np.concatenate(full[12:24]).tobytes()).hexdigest()]
    assert digs == want
    e2e = [l for l in r.stdout.splitlines() if l.startswith("E2E")][0].split()[1]
    assert e2e == hashlib.sha256(np.concatenate(full).tobytes()).hexdigest()


def test_bench_starts_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under
    torch.distributed.run; on a host without a GPU the ranks rendezvous over gloo and print the dry-run line
    (value null): launch, sharding and timing plumbing are exercised before an 8-GPU lease ever is"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--config", "c5"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    import torch
    if torch.cuda.is_available():
        assert d["n_gpus"] == 2  # a real two-GPU run (or a launch error above)
        return
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 3
    # weak scaling: every rank its own 512 polynomials, shard bounds multiples of the 8 primes
    assert d["config"]["shards"] == [[0, 512], [512, 1024]]


def test_first_multigpu_lease_script_parses_and_names_existing_programs():
    """tools/first_multigpu_lease.sh --list: the command list of the first multi-GPU lease -- RCCL smoke,
    headline at 2 / 4 / 8 ranks, the strongly scaled c4, the sweep, the digest check, the C++ multi-device example, the 1-GPU
    reference points; every
    step has a timeout, a unique rendezvous port, a rank count that matches its name and a program that exists"""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "first_multigpu_lease.sh"), "--list"], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    steps = [l.split("|", 2) for l in r.stdout.splitlines() if l.strip()]
    assert len(steps) == 22
    names = [s[1] for s in steps]
    order = [n.rsplit("_x", 1)[0] for n in names]
    assert order[:3] == ["rccl_smoke"] * 3, "the RCCL smoke test must come first"
    for kind in ("rccl_smoke", "bench_c2", "bench_c4_strong", "sweep", "digests"):
        assert [n for n in names if n.startswith(kind + "_x")][:3] == [kind + "_x%d" % k for k in (2, 4, 8)]
    assert "bench_c2_x1" in names and "bench_c4_strong_x1" in names
    ports = set()
    for tmo, name, cmd in steps:
        assert int(tmo) >= 60
        n = int(name.rsplit("_x", 1)[1])
        words = cmd.split()
        if name.startswith("cpp_bench_"):  # the timed C++ program: every device count of the node in one run
            assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "bench_multi_device.cpp"))
            assert words[-1] == "1,2,4,8" and words[-4] in ("c2", "c4")
            continue
        if name.startswith("cpp_multi_device"):  # the C++ product: one process, one host thread per device
            assert os.path.exists(os.path.join(ROOT, "tests", "cpp", "example_multi_device.cpp"))
            assert words[-2] == str(n), "asks for as many devices as the step's name says"
            continue
        prog = [w for w in words if w.endswith(".py")][0]
        assert os.path.exists(os.path.join(ROOT, prog)), prog
        if n > 1:
            assert words[words.index("--nproc-per-node") + 1] == str(n)
            assert words[words.index("--master-addr") + 1] == "127.0.0.1"
            port = words[words.index("--master-port") + 1]
            assert port.isdigit() and port not in ports
            ports.add(port)
            if prog == "bench.py":
                assert words[words.index("--gpus") + 1] == str(n)


def test_rccl_smoke_steps_over_gloo_two_ranks():
    """tools/rccl_smoke.py -- the first thing the multi-GPU lease runs -- with --backend gloo on CPU tensors: the same
    init / all_reduce / broadcast / scatter / gather / digest steps through gpu-ntt_amd/dist.py's helpers, payloads checked"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tools", "rccl_smoke.py"), "--backend",
                        "gloo", "--words", "4096"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_SMOKE ")][0]
    d = json.loads(line[len("RCCL_SMOKE "):])
    assert d["ok"] is True and d["world"] == 2 and d["backend"] == "gloo"
    assert all(d[k] > 0 for k in ("broadcast_s", "scatter_s", "gather_s"))


def test_nccl_process_group_is_created_with_the_rank_device(monkeypatch):
    """dist.init_process_group("nccl", "cuda:<LOCAL_RANK>") must reach torch.distributed.init_process_group with backend
    "nccl" (= RCCL on ROCm) AND device_id set -- eager communicator creation bound to the rank's GPU, which is what a
    one-process-per-GPU RCCL job needs.  No GPU here: torch.distributed is mocked at that call."""
    import importlib
    import torch
    import torch.distributed as td
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    load_pkg()
    dm = importlib.import_module("gpu_ntt_amd.dist")
    seen = {}

    def fake_init(backend, rank=None, world_size=None, **kw):
        seen.update(backend=backend, rank=rank, world_size=world_size, **kw)

    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "5")
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    monkeypatch.setattr(td, "is_initialized", lambda: False)
    monkeypatch.setattr(td, "init_process_group", fake_init)
    dist, rank, world = dm.init_process_group("nccl", "cuda:5")
    assert (rank, world) == (5, 8) and dist is td
    assert seen["backend"] == "nccl" and seen["rank"] == 5 and seen["world_size"] == 8
    assert seen["device_id"] == torch.device("cuda:5")
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
