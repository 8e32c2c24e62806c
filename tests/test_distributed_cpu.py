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
pf = ShardedParticleSystem(model, N, seed, engine_cls=O.OracleShardEngine, exchange="exact")
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


# ---- exchange="owned": offspring stay with the owner of their parent, only the surplus travels ----------------------
OWNED_WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["MP_ROOT"])
import numpy as np, torch, torch.distributed as dist
dist.init_process_group("gloo")
import modppl_amd
from modppl_amd.distributed import ShardedParticleSystem
from tests import oracle_lib as O
from tests.owned_ref import OwnedReference
rank, world = dist.get_rank(), dist.get_world_size()
N, T, seed = int(os.environ["MP_N"]), int(os.environ["MP_T"]), 78
scheme = int(os.environ["MP_SCHEME"])
if os.environ["MP_MODEL"] == "lgssm":
    model = modppl_amd.lgssm_model(*O.LGSSM_PARAMS)
    obs = O.lgssm_observations(T).reshape(T, 1)
else:
    model = modppl_amd.lgssm_band_model(4)
    obs = np.random.default_rng(1).normal(0, 1.2, size=(T, 4))
pf = ShardedParticleSystem(model, N, seed, engine_cls=O.OracleShardEngine, exchange=os.environ.get("MP_EXCHANGE", "owned"))
ref = OwnedReference(model, N, seed, world) if rank == 0 else None
def gather(a):
    out = [None] * world
    dist.all_gather_object(out, a)
    return np.concatenate(out)
pf.init_step(None, obs[:1])
if ref: ref.init_step(None, obs[:1])
ok, moved = True, 0
for t in range(1, T):
    L = pf.resample(scheme)
    par, x = gather(pf.parents), gather(pf.states())
    if ref:
        ok &= L == ref.resample(int(os.environ.get("MP_REF_SCHEME", scheme)))
        ok &= bool(np.array_equal(par, ref.parents())) and bool(np.array_equal(x, ref.states()))
        moved += sum(max(c - N // world, 0) for c in ref.counts)
    pf.step(obs[t:t + 1])
    if ref: ref.step(obs[t:t + 1])
lw = gather(pf.log_weights)
lml = pf.log_marginal_likelihood_estimate()
if ref:
    ok &= bool(np.array_equal(lw, ref.log_weights()))
    print("RESULT", json.dumps({"ok": bool(ok), "lml": lml, "moved": int(moved)}))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("model,n,t,scheme,nproc", [("lgssm", 8192, 8, 0, 2), ("band", 4096, 5, 0, 2), ("lgssm", 4096, 5, 1, 2),
                                                     ("lgssm", 8192, 8, 3, 2), ("band", 6144, 5, 3, 3)])
def test_two_rank_owner_keeps_exchange(tmp_path, model, n, t, scheme, nproc):
    """gloo, 2 (3) ranks, exact-size all-to-all of the surplus == the same protocol with all shards in one process; scheme 3 = the
    split multinomial (exchange="split" with the default scheme)"""
    script = tmp_path / "worker.py"
    script.write_text(OWNED_WORKER)
    env = dict(os.environ, MP_ROOT=ROOT, MP_MODEL=model, MP_N=str(n), MP_T=str(t), MP_SCHEME=str(scheme), OMP_NUM_THREADS="1")
    if scheme == 3:   # the worker calls resample(scheme) with the plain multinomial; the exchange turns it into the split one
        env.update(MP_EXCHANGE="split", MP_SCHEME="0", MP_REF_SCHEME="3")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT")]
    assert line, res.stdout[-2000:] + res.stderr[-2000:]
    assert '"ok": true' in line[0], line[0]
    import json
    assert json.loads(line[0][len("RESULT"):])["moved"] > 0    # the exchange really carried rows


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_owner_keeps_is_the_single_filters_resample_placed_by_owner(world, scheme):
    """The parents are the single filter's draws (same multiset); the placement follows the rule of include/modppl_hip.h,
    restated here with numpy from the single filter's parent vector."""
    import modppl_amd
    from tests import oracle_lib as O
    from tests.owned_ref import OwnedReference, owned_placement
    N, T, seed = 16384, 4, 5
    obs = O.lgssm_observations(T).reshape(T, 1)
    model = modppl_amd.lgssm_model(*O.LGSSM_PARAMS)
    one = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    ref = OwnedReference(model, N, seed, world)
    one.init_step(obs[:1])
    ref.init_step(None, obs[:1])
    assert np.array_equal(one.log_weights(), ref.log_weights())
    x_before = one.state().reshape(N, 1)
    L1 = one.resample(scheme)
    L = ref.resample(scheme)
    assert L == L1
    x_exp, par_exp = owned_placement(one.parents(), x_before, N // world, world)
    assert np.array_equal(ref.parents(), par_exp)
    assert np.array_equal(ref.states().reshape(N, 1), x_exp)
    assert np.array_equal(np.sort(ref.parents()), np.sort(one.parents()))
    assert sum(ref.counts) == N


def test_owned_plan_conserves_rows():
    from modppl_amd.distributed import ShardedParticleSystem as S
    rng = np.random.default_rng(3)
    for w in (1, 2, 3, 8):
        n = 1000
        for _ in range(50):
            c = rng.multinomial(n * w, rng.dirichlet(np.ones(w) * rng.choice([0.1, 1, 50])))
            a = np.array(S.owned_plan(list(c), n))
            assert np.array_equal(a.sum(1), np.maximum(c - n, 0)) and np.array_equal(a.sum(0), np.maximum(n - c, 0))
            assert (np.diag(a) == 0).all()


@pytest.mark.parametrize("exchange", ["owned", "exact", "split"])
def test_sharded_maybe_resample_follows_the_fresh_ess(exchange):
    """ESS-triggered resampling over shards (a world of one here, the checker as local engine): the decisions and results of
    one filter that is resampled whenever its fresh ESS drops below the threshold."""
    import modppl_amd
    from modppl_amd.distributed import ShardedParticleSystem
    from tests import oracle_lib as O
    N, T, seed, frac = 4096, 10, 11, 0.6
    obs = O.lgssm_observations(T).reshape(T, 1)
    pf = ShardedParticleSystem(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), N, seed, engine_cls=O.OracleShardEngine, exchange=exchange)
    one = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    pf.init_step(None, obs[:1])
    one.init_step(obs[:1])
    n_res = 0
    for t in range(1, T):
        did, ess, ltw = pf.maybe_resample(frac)
        ess1 = one.effective_sample_size(1)
        assert ess == ess1 and did == (ess1 < frac * N)
        if did:
            n_res += 1
            assert ltw == one.resample()
            assert np.array_equal(pf.states().reshape(-1), one.state().reshape(-1))
        pf.step(obs[t:t + 1])
        one.step(obs[t:t + 1])
    assert 0 < n_res < T - 1
    assert pf.log_marginal_likelihood_estimate() == one.log_marginal_likelihood_estimate()


# ---- the split multinomial (MP_RESAMPLE_MULTINOMIAL_SPLIT): counts per rank first, then every rank its own draws ------------
def test_split_multinomial_in_a_world_of_one_is_the_single_filters_resample():
    """rank 0's stream is the single filter's and its share of the mass is all of it: same parents, bit for bit"""
    import modppl_amd
    from tests import oracle_lib as O
    from tests.owned_ref import OwnedReference
    N, T, seed = 6144, 5, 21
    obs = O.lgssm_observations(T).reshape(T, 1)
    one = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
    ref = OwnedReference(modppl_amd.lgssm_model(*O.LGSSM_PARAMS), N, seed, 1)
    one.init_step(obs[:1])
    ref.init_step(None, obs[:1])
    for t in range(1, T):
        assert ref.resample(3) == one.resample(0)
        assert np.array_equal(ref.parents(), one.parents()) and np.array_equal(ref.states().reshape(-1), one.state().reshape(-1))
        ref.step(obs[t:t + 1])
        one.step(obs[t:t + 1])
    assert np.array_equal(ref.log_weights(), one.log_weights())


@pytest.mark.parametrize("world", [2, 4, 8])
def test_split_multinomial_structure_and_law(world):
    """Counts = the splitting tree's over the rank masses; a rank's kept offspring have parents on that rank; L, ESS and log-ML are
    the single filter's (they do not depend on the draws); and over many seeds the offspring per TILE follow the weights
    (chi-square against N x tile mass, the masses restated in numpy from the log-weights)."""
    import modppl_amd
    from scipy import stats
    from tests import oracle_lib as O
    from tests.owned_ref import OwnedReference
    n, T = 2048, 3
    N = n * world
    obs = O.lgssm_observations(T).reshape(T, 1).copy()
    obs[1] = 4.0                       # uneven shard masses
    model = modppl_amd.lgssm_model(*O.LGSSM_PARAMS)
    chi2, dof = 0.0, 0
    for seed in range(30):
        one = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, N, seed, O.VARIANT_CANONICAL | O.VARIANT_SOA)
        ref = OwnedReference(model, N, seed, world)
        for f in (lambda: (one.init_step(obs[:1]), ref.init_step(None, obs[:1])), lambda: (one.step(obs[1:2]), ref.step(obs[1:2]))):
            f()
        lw = one.log_weights()
        L1 = one.resample(0)
        assert ref.resample(3) == L1                     # the normalisation is the job's, whatever the sampler
        assert ref.eng[0].ess_reference() == one.effective_sample_size(0)
        counts = np.array(ref.counts, dtype=np.int64)
        assert counts.sum() == N
        par = ref.parents().astype(np.int64).reshape(world, n)
        for r in range(world):
            keep = min(counts[r], n)
            assert (par[r, :keep] // n == r).all()       # kept offspring: parents on the owner
        # every offspring's parent, wherever it was placed: counts per rank as drawn
        assert np.array_equal(np.bincount(par.reshape(-1) // n, minlength=world), counts)
        w = np.exp(lw - lw.max())
        p_tile = w.reshape(-1, 2048).sum(1) / w.sum()
        got = np.bincount(par.reshape(-1) // 2048, minlength=world)
        exp = N * p_tile
        chi2 += ((got - exp) ** 2 / exp).sum()
        dof += world - 1
    assert chi2 < stats.chi2.ppf(1 - 1e-6, dof), (chi2, dof)
    assert chi2 > stats.chi2.ppf(1e-6, dof), (chi2, dof)   # (and not suspiciously regular either)
