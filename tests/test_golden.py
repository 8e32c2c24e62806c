"""Committed golden vectors (tests/golden/, made by make_golden.py from the literal / canonical oracle).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces the canonical ones bit for bit."""
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    z = np.load(os.path.join(G, "lgssm_c1_n1000_t50.npz"))
    meta = json.load(open(os.path.join(G, "lgssm_c1_n1000_t50.json")))
    return z, meta


def test_oracle_reproduces_golden():
    z, meta = load()
    assert np.array_equal(O.lgssm_observations(meta["T"]), z["ys"])
    assert O.kalman_log_ml(z["ys"]) == meta["kalman_log_ml"]
    u = np.empty(16)
    O.load().oracle_u01_stream(meta["seed"], 3, 7, 1, 2, 16, O.dptr(u))
    assert u.tolist() == meta["u01_stream_seed_slot3_step7_dom1_site2"]
    for variant, key in ((0, "lit"), (O.VARIANT_CANONICAL | O.VARIANT_SOA, "can")):
        pf = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, meta["n"], meta["seed"], variant)
        pf.init_step(z["ys"][:1])
        for t in range(1, meta["T"]):
            assert pf.resample() == z[f"{key}_L"][t - 1]
            assert np.array_equal(pf.parents(), z[f"{key}_parents"][t - 1])
            pf.step(z["ys"][t:t + 1])
        assert pf.log_marginal_likelihood_estimate() == meta[f"{key}_lml"]
        assert np.array_equal(pf.state()[:, 0], z[f"{key}_final_x"])
    assert meta["index_mismatches_literal_vs_canonical"] == 0
    assert abs(meta["lit_lml"] - meta["can_lml"]) <= 1e-12 * abs(meta["lit_lml"])


@pytest.mark.gpu
def test_hip_reproduces_golden():
    import modppl_amd

    z, meta = load()
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(*meta["params"]), meta["n"], meta["seed"])
    pf.init_step(None, z["ys"][:1])
    for t in range(1, meta["T"]):
        assert pf.resample() == z["can_L"][t - 1]
        assert pf.effective_sample_size() == z["can_ess"][t - 1]
        assert np.array_equal(pf.parents, z["can_parents"][t - 1])
        assert np.array_equal(pf.parents, z["lit_parents"][t - 1])   # and index-identical to the literal (reference) arithmetic
        pf.step(z["ys"][t:t + 1])
    assert pf.log_marginal_likelihood_estimate() == meta["can_lml"]
    assert np.array_equal(pf.states()[:, 0], z["can_final_x"])
    assert np.array_equal(pf.log_weights, z["can_final_logw"])
    assert abs(pf.log_marginal_likelihood_estimate() - meta["lit_lml"]) <= 1e-12 * abs(meta["lit_lml"])


# ---- second fixture: MH (three proposals + regen), pointed MH, importance on the reference's models, systematic /
# stratified parents, spiral / bearings / banded filters (tests/golden/paths_small.npz) -----------------------------------
def _paths():
    z = np.load(os.path.join(G, "paths_small.npz"))
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    exp = {k: z[k] for k in z.files if not k.startswith("in_")}
    return inp, exp


def _compare(got, exp):
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k


def test_oracle_reproduces_paths_golden():
    from tests.golden import make_golden as M

    inp, exp = _paths()
    for k, v in M.paths_inputs().items():
        assert np.array_equal(v, inp[k]), k
    _compare(M.oracle_paths(inp), exp)


@pytest.mark.gpu
def test_hip_reproduces_paths_golden():
    """The device against the committed vectors alone (no checker in the loop)."""
    import modppl_amd
    from modppl_amd import capi
    from tests.golden import make_golden as M

    inp, exp = _paths()
    kinds = {1: lambda p: modppl_amd.lgssm_model(*p), 2: lambda p: modppl_amd.spiral_model(), 4: lambda p: modppl_amd.bearings_model(*p),
             5: lambda p: modppl_amd.lgssm_band_model(int(p[0]), *p[1:]), 6: lambda p: modppl_amd.pointed_2d_model(list(p[:4]), list(p[4:])),
             7: lambda p: modppl_amd.line_model(list(p))}

    class Pf:
        def __init__(self, kind, ds, do, params, n, seed):
            self.pf = modppl_amd.ParticleSystem(kinds[kind](params), n, seed)

        def init_step(self, args0, obs):
            self.pf.init_step(args0, obs)

        def step(self, obs):
            self.pf.step(obs)

        def resample(self, scheme):
            return self.pf.resample(scheme=scheme)

        def parents_now(self):
            return self.pf.parents

        def states_now(self):
            return self.pf.states()

        def log_marginal_likelihood_estimate(self):
            return self.pf.log_marginal_likelihood_estimate()

    class Mh:
        def __init__(self, ys, n, seed, constrain=None):
            self.h = modppl_amd.HierarchicalChains(M.XS, ys, n, seed, constrain_is_linear=constrain)

        def mh(self, std, it):
            return self.h.mh(std, it)

        def mh_add_or_remove(self, it):
            return self.h.mh_add_or_remove(it)

        def regen_mh(self, sites, it, cycle):
            return self.h.regen_mh(sites, it, cycle)

        def states(self):
            return self.h.states()

    def importance(kind, ds, do, params, obs, n, m, seed):
        states, idx, lml = modppl_amd.importance_resampling(kinds[kind](params), [0.0] * ds, obs, n, m, seed)
        _, lnw, _ = modppl_amd.importance_sampling(kinds[kind](params), [0.0] * ds, obs, n, seed)
        return lml, lnw, idx, states

    got = M.paths_outputs(lambda ys, n, seed, constrain=None: Mh(ys, n, seed, constrain),
                          lambda n, seed: modppl_amd.PointedChains(M.BOUNDS, M.COV, [0.0, 0.0], n, seed), importance, Pf, inp)
    _compare(got, exp)


# mh / regen_mh over the registered functor models (tests/golden/mh_functor.npz) -------------------------------------------------
def _functor():
    z = np.load(os.path.join(G, "mh_functor.npz"))
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    exp = {k: z[k] for k in z.files if not k.startswith("in_")}
    return inp, exp


def test_oracle_reproduces_functor_golden():
    """the dynamic interpretation (tries) AND the product's static handlers compiled for the host, against the committed vectors"""
    from tests.golden import make_golden as M

    inp, exp = _functor()
    for k, v in M.functor_inputs().items():
        assert np.array_equal(v, inp[k]), k
    _compare(M.oracle_functor(inp), exp)

    class Static:
        def __init__(self, kind, params, cons, n, seed):
            self.s = O.HostStaticFunctionChains(kind, params, cons, n, seed)
            self.ns = {101: 20, 102: 26, 103: 13}[kind]

        def mh(self, kind, args, it):
            return self.s.mh(kind, args, it)

        def regen_mh(self, sites, it, cycle):
            return self.s.regen_mh(sites, it, cycle)

        def trace(self):
            return self.s.trace(self.ns)

    _compare(M.functor_outputs(Static, inp), exp)


@pytest.mark.gpu
def test_hip_reproduces_functor_golden():
    """The device's generic MH kernels against the committed vectors alone (no checker in the loop)."""
    import modppl_amd
    from tests.golden import make_golden as M

    inp, exp = _functor()

    class Dev:
        def __init__(self, kind, params, cons, n, seed):
            self.c = modppl_amd.FunctionChains(kind, params, cons, n, seed)

        def mh(self, kind, args, it):
            return self.c.mh(kind, args, it)

        def regen_mh(self, sites, it, cycle):
            return self.c.regen_mh(sites, it, cycle)

        def trace(self):
            return self.c.trace()

    _compare(M.functor_outputs(Dev, inp), exp)
