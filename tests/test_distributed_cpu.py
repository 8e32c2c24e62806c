"""CPU, world_size 2, gloo: the sharded-filter orchestration (routing of draws to owner ranks, count
exchange, particle exchange, scalar finalisation) gives bit for bit the results of ONE filter.
The local compute engine here is the CPU checker (injected by the test; the product default is HIP)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MP_ROOT"])
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
import modppl_amd
from modppl_amd.distributed import ShardedParticleSystem
from tests import oracle_lib as O
rank, world = dist.get_rank(), dist.get_world_size()
kind = os.environ["MP_MODEL"]
N, T, seed = int(os.environ["MP_N"]), int(os.environ["MP_T"]), 77
if kind == "lgssm":
    model, okind, params, ds, do = modppl_amd.lgssm_model(*O.LGSSM_PARAMS), 1, O.LGSSM_PARAMS, 1, 1
    obs = O.lgssm_observations(T).reshape(T, 1)
else:
    D = 4
    params = np.array([D, 0.9, 0.05, 1.0, 0.5, 1.0]); model, okind, ds, do = modppl_amd.lgssm_band_model(D), 5, D, D
    obs = np.random.default_rng(1).normal(0, 1.2, size=(T, D))
pf = ShardedParticleSystem(model, N, seed, engine_cls=O.OracleShardEngine)
ref = O.OraclePF(okind, ds, do, params, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA) if rank == 0 else None
n = N // world
sl = slice(rank * n, (rank + 1) * n)
def gather(a):
    out = [None] * world
    dist.all_gather_object(out, a)
    return np.concatenate(out)
pf.init_step(None, obs[:1])
if ref: ref.init_step(obs[:1])
ok = True
for t in range(1, T):
    ess = pf.effective_sample_size(fresh=True)
    L = pf.resample()
    par, x = gather(pf.parents), gather(pf.states())
    if ref:
        ok &= ess == ref.effective_sample_size(1)
        ok &= L == ref.resample()
        ok &= bool(np.array_equal(par, ref.parents())) and bool(np.array_equal(x, ref.state()))
        ok &= pf.effective_sample_size() == ref.effective_sample_size(0)
    if t % 3 == 0:   # let weights accumulate over two steps now and then
        pf.step(obs[t:t + 1]); 
        if ref: ref.step(obs[t:t + 1])
    pf.step(obs[t:t + 1])
    if ref: ref.step(obs[t:t + 1])
lw = gather(pf.log_weights)
lml = pf.log_marginal_likelihood_estimate()
if ref:
    ok &= bool(np.array_equal(lw, ref.log_weights())) and lml == ref.log_marginal_likelihood_estimate()
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


@pytest.mark.parametrize("model,n,t", [("lgssm", 8192, 8), ("band", 4096, 6)])  # shards are tile-aligned (2048)
def test_two_rank_filter_equals_single_filter(tmp_path, model, n, t):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MP_ROOT=ROOT, MP_MODEL=model, MP_N=str(n), MP_T=str(t), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT")]
    assert line, res.stdout[-2000:] + res.stderr[-2000:]
    assert '"ok": true' in line[0], line[0]
