"""GPU: the sharded-filter phases (mp_pf_shard_*) and the torch.distributed orchestration.
 (a) world = 1: every shard kernel runs, results must equal the unsharded HIP filter bit for bit;
 (b) two ranks sharing the one GPU of the test box (gloo + host staging as the transport; the bench uses
     nccl/RCCL with device tensors through the same code): equals ONE filter of the CPU checker."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_sharded_equals_unsharded():
    import modppl_amd
    from modppl_amd.distributed import ShardedParticleSystem

    ys = O.lgssm_observations(10)
    n, seed = 50000, 21
    a = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    b = ShardedParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed, exchange="exact")
    a.init_step(None, ys[:1])
    b.init_step(None, ys[:1])
    for t in range(1, 10):
        assert a.effective_sample_size(fresh=True) == b.effective_sample_size(fresh=True)
        assert a.resample() == b.resample()
        assert np.array_equal(a.parents, b.parents)
        assert np.array_equal(a.states(), b.states())
        assert a.effective_sample_size() == b.effective_sample_size()
        a.step(ys[t:t + 1])
        b.step(ys[t:t + 1])
    assert np.array_equal(a.log_weights, b.log_weights)
    assert a.log_marginal_likelihood_estimate() == b.log_marginal_likelihood_estimate()


WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MP_ROOT"])
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
import modppl_amd
from modppl_amd.distributed import ShardedParticleSystem
from tests import oracle_lib as O
rank, world = dist.get_rank(), dist.get_world_size()
N, T, seed, D = int(os.environ["MP_N"]), 8, 5, int(os.environ["MP_D"])
if D == 1:
    model, okind, params = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), 1, O.LGSSM_PARAMS
    obs = O.lgssm_observations(T).reshape(T, 1)
else:
    params = np.array([D, 0.9, 0.05, 1.0, 0.5, 1.0]); model, okind = modppl_amd.lgssm_band_model(D), 5
    obs = np.random.default_rng(1).normal(0, 1.2, size=(T, D))
pf = ShardedParticleSystem(model, N, seed, host_staging=True, exchange="exact")   # both ranks on cuda:0
ref = O.OraclePF(okind, D, D, params, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA, threads=4) if rank == 0 else None
def gather(a):
    out = [None] * world
    dist.all_gather_object(out, a)
    return np.concatenate(out)
pf.init_step(None, obs[:1])
if ref: ref.init_step(obs[:1])
ok = True
for t in range(1, T):
    L = pf.resample()
    par, x = gather(pf.parents), gather(pf.states())
    if ref:
        ok &= L == ref.resample()
        ok &= bool(np.array_equal(par, ref.parents())) and bool(np.array_equal(x, ref.state()))
    pf.step(obs[t:t + 1])
    if ref: ref.step(obs[t:t + 1])
lml = pf.log_marginal_likelihood_estimate()
if ref:
    ok &= lml == ref.log_marginal_likelihood_estimate()
    print("RESULT", json.dumps({"ok": bool(ok), "lml": lml}))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n,d", [(40960, 1), (8192, 16)])  # shards are tile-aligned (2048)
def test_two_ranks_one_gpu_equal_single_filter(tmp_path, n, d):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MP_ROOT=ROOT, MP_N=str(n), MP_D=str(d), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT")]
    assert line, res.stdout[-2000:] + res.stderr[-2000:]
    assert '"ok": true' in line[0], line[0]


NCCL_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MP_ROOT"])
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import modppl_amd
from modppl_amd.distributed import ShardedParticleSystem
from tests import oracle_lib as O
ys = O.lgssm_observations(6)
n, seed = 1 << 16, 9
a = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
b = ShardedParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed, exchange="exact")   # device tensors through RCCL
a.init_step(None, ys[:1]); b.init_step(None, ys[:1])
ok = True
for t in range(1, 6):
    ok &= a.resample() == b.resample()
    ok &= bool(np.array_equal(a.parents, b.parents)) and bool(np.array_equal(a.states(), b.states()))
    a.step(ys[t:t + 1]); b.step(ys[t:t + 1])
ok &= a.log_marginal_likelihood_estimate() == b.log_marginal_likelihood_estimate()
want_fb = os.environ.get("MP_EXPECT_FALLBACK", "0") == "1"
ok &= (b.fallbacks > 0) == want_fb
print("RESULT ok" if ok else "RESULT mismatch", b.fallbacks)
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["fixed", "fixed_overflow", "variable"])
def test_rccl_collectives_world_of_one(tmp_path, mode):
    """The bench's transport (backend nccl = RCCL, device tensors) with every collective forced in a world of one, on the
    shared stream: fixed-capacity equal-split all-to-alls; the same with a capacity too small (every resample falls back
    to the exact-size phases without committing anything); exact-size phases only."""
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER)
    extra = {"fixed": {}, "fixed_overflow": {"MP_SHARD_CAP": "1000", "MP_EXPECT_FALLBACK": "1"}, "variable": {"MP_SHARD_FIXED": "0"}}[mode]
    from tests.conftest import diag_env

    env = diag_env(dict(os.environ, MP_ROOT=ROOT, MP_SHARD_ALWAYS_COLLECTIVE="1", **extra))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "RESULT ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("d,cap_frac,peek", [(1, 1.0, True), (4, 0.2, True), (1, 0.3, False), (4, 0.2, False)])
def test_fixed_capacity_exchange_two_shards_in_process(d, cap_frac, peek):
    """World = 2 through the fixed-capacity phases, the collectives done by hand between two handles on the one GPU:
    sub-segment layout, counts in the headers, rows adopted lazily.  Equal to ONE unsharded filter bit for bit.
    peek=False never reads states between steps, so every propagate takes its inputs straight from the exchange buffer."""
    import ctypes as C

    import torch

    import modppl_amd
    from modppl_amd.distributed import HipShardEngine

    n, world, seed, T = 8192, 2, 77, 6
    N = n * world
    cap = int(n * cap_frac)
    if d == 1:
        model = modppl_amd.lgssm_model(*O.LGSSM_PARAMS)
        obs = O.lgssm_observations(T).reshape(T, 1)
    else:
        model = modppl_amd.lgssm_band_model(d)
        obs = np.random.default_rng(3).normal(0, 1.2, size=(T, d))
    one = modppl_amd.ParticleSystem(model, N, seed)
    eng = [HipShardEngine(model, n, N, r * n, seed) for r in range(world)]
    nt = n // 2048
    dev = eng[0].device
    tiles = [torch.zeros(3 * nt, dtype=torch.int64, device=dev) for _ in range(world)]
    req_out = [torch.zeros(world * 8 * (cap + 1) * 2, dtype=torch.int64, device=dev) for _ in range(world)]
    rows_out = [torch.zeros(world * 8 * cap * (d + 1), dtype=torch.float64, device=dev) for _ in range(world)]
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def sync():
        for e in eng:
            e.synchronize()
        torch.cuda.synchronize()

    one.init_step(None, obs[:1])
    for e in eng:
        e.init_step(None, obs[:1])
    for t in range(1, T):
        for r, e in enumerate(eng):
            e.shard_tiles_packed(ptr(tiles[r]))
        sync()
        tiles_all = torch.cat(tiles)                                   # all_gather_into_tensor
        torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
        for r, e in enumerate(eng):
            e.shard_route_fixed(0, ptr(tiles_all), world, r, cap, ptr(req_out[r]))
        sync()
        seg = 8 * (cap + 1) * 2                                        # all_to_all_single, equal splits
        req_in = [torch.cat([req_out[s][r * seg:(r + 1) * seg] for s in range(world)]) for r in range(world)]
        torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
        for r, e in enumerate(eng):
            e.shard_resolve_fixed(ptr(req_in[r]), world, cap, ptr(rows_out[r]))
        sync()
        seg = 8 * cap * (d + 1)
        rows_in = [torch.cat([rows_out[s][r * seg:(r + 1) * seg] for s in range(world)]) for r in range(world)]
        torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
        res = [e.shard_commit_fixed(ptr(rows_in[r]), True) for r, e in enumerate(eng)]
        L = one.resample()
        assert all(done for done, _ in res)
        assert all(v == L for _, v in res)
        if peek:
            assert np.array_equal(np.concatenate([e.parents() for e in eng]), one.parents)
            assert np.array_equal(np.concatenate([e.states() for e in eng]), one.states())
        one.step(obs[t:t + 1])
        for e in eng:
            e.step(obs[t:t + 1])
    assert np.array_equal(np.concatenate([e.log_weights() for e in eng]), one.log_weights)
    assert np.array_equal(np.concatenate([e.states() for e in eng]), one.states())
    for r, e in enumerate(eng):
        e.shard_tiles_packed(ptr(tiles[r]))
    sync()
    tiles_all = torch.cat(tiles)
    torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
    assert eng[0].shard_query_packed(ptr(tiles_all), world)[0] == one.log_marginal_likelihood_estimate()


def test_fixed_capacity_overflow_commits_nothing():
    """Capacity far too small for two shards: every rank reports MP_ERR_CAPACITY (the flag travels in the request
    headers) and nothing is committed — states, weights and the resample counter are untouched."""
    import ctypes as C

    import torch

    import modppl_amd
    from modppl_amd.distributed import HipShardEngine

    n, world, seed, cap = 4096, 2, 3, 64
    N = n * world
    model = modppl_amd.lgssm_model(*O.LGSSM_PARAMS)
    obs = O.lgssm_observations(2).reshape(2, 1)
    eng = [HipShardEngine(model, n, N, r * n, seed) for r in range(world)]
    for e in eng:
        e.init_step(None, obs[:1])
    before = [(e.states().copy(), e.log_weights().copy()) for e in eng]
    nt, dev = n // 2048, eng[0].device
    tiles = [torch.zeros(3 * nt, dtype=torch.int64, device=dev) for _ in range(world)]
    req_out = [torch.zeros(world * 8 * (cap + 1) * 2, dtype=torch.int64, device=dev) for _ in range(world)]
    rows_out = [torch.zeros(world * 8 * cap * 2, dtype=torch.float64, device=dev) for _ in range(world)]
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def sync():
        for e in eng:
            e.synchronize()
        torch.cuda.synchronize()

    for r, e in enumerate(eng):
        e.shard_tiles_packed(ptr(tiles[r]))
    sync()
    tiles_all = torch.cat(tiles)
    torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
    for r, e in enumerate(eng):
        e.shard_route_fixed(0, ptr(tiles_all), world, r, cap, ptr(req_out[r]))
    sync()
    seg = 8 * (cap + 1) * 2
    req_in = [torch.cat([req_out[s][r * seg:(r + 1) * seg] for s in range(world)]) for r in range(world)]
    torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
    for r, e in enumerate(eng):
        e.shard_resolve_fixed(ptr(req_in[r]), world, cap, ptr(rows_out[r]))
    sync()
    seg = 8 * cap * 2
    rows_in = [torch.cat([rows_out[s][r * seg:(r + 1) * seg] for s in range(world)]) for r in range(world)]
    torch.cuda.synchronize()   # made on torch's stream, read by kernels on the engines' own streams
    res = [e.shard_commit_fixed(ptr(rows_in[r]), True) for r, e in enumerate(eng)]
    assert all(not done for done, _ in res)
    for e, (x, w) in zip(eng, before):
        assert np.array_equal(e.states(), x)
        assert np.array_equal(e.log_weights(), w)


def test_collapsed_weights_fall_back_and_capacity_grows():
    """An observation far in the tail leaves a handful of particles with all the weight: one sub-segment of the
    fixed-capacity exchange would need (nearly) every draw.  That resample is repeated with exact sizes, the capacity
    grows for the next ones, and the results stay those of the unsharded filter."""
    import modppl_amd
    from modppl_amd.distributed import ShardedParticleSystem

    n, seed = 1 << 16, 13
    ys = np.array([0.1, 9.5, 9.0, 0.3, 0.2])
    a = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed)
    b = ShardedParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), n, seed, exchange="exact")
    cap0 = b.cap
    a.init_step(None, ys[:1])
    b.init_step(None, ys[:1])
    for t in range(1, len(ys)):
        a.step(ys[t:t + 1])
        b.step(ys[t:t + 1])
        assert a.effective_sample_size(fresh=True) == b.effective_sample_size(fresh=True)
        assert a.resample() == b.resample()
        assert np.array_equal(a.parents, b.parents) and np.array_equal(a.states(), b.states())
    assert b.fallbacks >= 1 and b.cap > cap0
    assert a.log_marginal_likelihood_estimate() == b.log_marginal_likelihood_estimate()
