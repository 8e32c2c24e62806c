#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU restatement (oracle/).

The reference (Rust) cannot be built or run in this image, and its RNG cannot be seeded, so these vectors pin the
BUILD's own definitions (Philox stream, literal and canonical arithmetic) against regressions; the reference-derived
known answers live in tests/test_oracle_kats.py.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import oracle_lib as O  # noqa: E402


def pf_run(variant, n, T, seed, ys):
    pf = O.OraclePF(1, 1, 1, O.LGSSM_PARAMS, n, seed, variant)
    pf.init_step(ys[:1])
    Ls, ess, par = [], [], []
    for t in range(1, T):
        Ls.append(pf.resample())
        ess.append(pf.effective_sample_size())
        par.append(pf.parents().astype(np.uint32))
        pf.step(ys[t:t + 1])
    return dict(L=Ls, ess=ess, parents=np.array(par), final_x=pf.state()[:, 0], final_logw=pf.log_weights(),
                lml=pf.log_marginal_likelihood_estimate())


def main():
    T, n, seed = 50, 1000, 20241008
    ys = O.lgssm_observations(T, canonical=True)
    lit = pf_run(0, n, T, seed, ys)                      # structure-faithful engine, literal arithmetic (libm, fp64 CDF scan)
    can = pf_run(O.VARIANT_CANONICAL | O.VARIANT_SOA, n, T, seed, ys)
    np.savez_compressed(os.path.join(HERE, "lgssm_c1_n1000_t50.npz"), ys=ys,
                        lit_parents=lit["parents"], lit_final_x=lit["final_x"], lit_L=np.array(lit["L"]), lit_ess=np.array(lit["ess"]),
                        can_parents=can["parents"], can_final_x=can["final_x"], can_final_logw=can["final_logw"],
                        can_L=np.array(can["L"]), can_ess=np.array(can["ess"]))
    u = np.empty(16)
    O.load().oracle_u01_stream(seed, 3, 7, 1, 2, 16, O.dptr(u))
    meta = {"T": T, "n": n, "seed": seed, "params": O.LGSSM_PARAMS.tolist(), "kalman_log_ml": O.kalman_log_ml(ys),
            "lit_lml": lit["lml"], "can_lml": can["lml"], "u01_stream_seed_slot3_step7_dom1_site2": u.tolist(),
            "index_mismatches_literal_vs_canonical": int((lit["parents"] != can["parents"]).sum())}
    json.dump(meta, open(os.path.join(HERE, "lgssm_c1_n1000_t50.json"), "w"), indent=1)
    print(json.dumps(meta)[:300])


# ---- second fixture: the other paths (small sizes; canonical arithmetic, which the device implements) ---------------------
XS = np.arange(-5.0, 6.0)
BOUNDS, COV, NOISE = [-5.0, 5.0, -5.0, 5.0], [1.0, -0.6, -0.6, 2.0], [0.25, 0.0, 0.0, 0.25]
CANON_SOA = O.VARIANT_CANONICAL | O.VARIANT_SOA


def paths_inputs():
    rng = np.random.default_rng(20241008)
    return dict(
        mh_ys=0.3 + 0.4 * XS + 0.5 * XS * XS + 0.1 * rng.normal(size=XS.size),       # tests/mh.rs:84-87
        line_ys=(0.5 * XS - 1.0 + 0.1 * rng.normal(size=XS.size)).reshape(1, -1),     # tests/importance.rs:66-67
        spiral_obs=rng.normal(0.0, 0.5, size=(6, 2)),
        bearings_obs=np.arctan2(1.0 + 0.05 * np.arange(6), 1.0 + 0.1 * np.arange(6)).reshape(6, 1) + 0.02 * rng.normal(size=(6, 1)),
        band_obs=rng.normal(0.0, 1.2, size=(5, 4)))


def paths_outputs(mk_mh, mk_pointed, importance, mk_pf, inp):
    """One driver for the oracle (make / CPU test) and the HIP path (GPU test): the callables hide which engine runs."""
    out = {}
    h = mk_mh(inp["mh_ys"], 64, 31)
    acc = []
    for _ in range(3):                                      # the loop of tests/mh.rs:93-106, then regen_mh sweeps
        acc += [h.mh_add_or_remove(1), h.mh(0.1, 3), h.mh(0.01, 10)]
    out["mh_loop_states"], out["mh_loop_acc"] = h.states().copy(), np.array(acc, dtype=np.int64)
    h2 = mk_mh(inp["mh_ys"], 64, 32, constrain=False)
    out["regen_acc"] = np.array([h2.regen_mh([1, 2, 3], 9, True)], dtype=np.int64)
    out["regen_states"] = h2.states().copy()
    p = mk_pointed(64, 33)
    out["pointed_acc"] = np.array([p.mh(NOISE, 25)], dtype=np.int64)
    out["pointed_states"] = p.states().copy()
    out["is_pointed"] = importance(6, 2, 2, np.array(BOUNDS + COV), np.array([[0.0, 0.0]]), 512, 32, 34)
    out["is_line"] = importance(7, 2, 11, XS, inp["line_ys"], 512, 32, 35)
    for name, kind, ds, do, params, obs, scheme in (("sys", 1, 1, 1, O.LGSSM_PARAMS, O.lgssm_observations(5).reshape(5, 1), 1),
                                                    ("strat", 1, 1, 1, O.LGSSM_PARAMS, O.lgssm_observations(5).reshape(5, 1), 2),
                                                    ("spiral", 2, 2, 2, np.zeros(0), inp["spiral_obs"], 0),
                                                    ("bearings", 4, 4, 1, np.array([1.0, 1.0, 1.0, 0.1, 0.05, 0.02]), inp["bearings_obs"], 0),
                                                    ("band4", 5, 4, 4, np.array([4, 0.9, 0.05, 1.0, 0.5, 1.0]), inp["band_obs"], 0)):
        pf = mk_pf(kind, ds, do, params, 3000, 36)
        pf.init_step([0.0] * ds, obs[:1])
        par = []
        for t in range(1, len(obs)):
            pf.resample(scheme)
            par.append(np.asarray(pf.parents_now(), dtype=np.uint32).copy())
            pf.step(obs[t:t + 1])
        out[f"{name}_parents"] = np.array(par)
        out[f"{name}_x"] = pf.states_now().copy()
        out[f"{name}_lml"] = np.array([pf.log_marginal_likelihood_estimate()])
    flat = {}
    for k, v in out.items():
        if isinstance(v, tuple):
            flat[k + "_lml"], flat[k + "_lnw"], flat[k + "_idx"], flat[k + "_x"] = np.array([v[0]]), v[1], np.asarray(v[2], dtype=np.uint64), v[3]
        else:
            flat[k] = v
    return flat


class _OraclePfAdapter:
    def __init__(self, kind, ds, do, params, n, seed):
        self.pf = O.OraclePF(kind, ds, do, params, n, seed, CANON_SOA)

    def init_step(self, args0, obs):
        self.pf.init_step(obs, args0)

    def step(self, obs):
        self.pf.step(obs)

    def resample(self, scheme):
        return self.pf.resample(scheme)

    def parents_now(self):
        return self.pf.parents()

    def states_now(self):
        return self.pf.state()

    def log_marginal_likelihood_estimate(self):
        return self.pf.log_marginal_likelihood_estimate()


class _OracleMhAdapter:
    def __init__(self, ys, n, seed, constrain=None):
        self.h = O.OracleMH(XS, ys, n, seed, -1 if constrain is None else int(constrain), canonical=True)

    def mh(self, std, it):
        return self.h.mh(std, it)

    def mh_add_or_remove(self, it):
        return self.h.mh_add_or_remove(it)

    def regen_mh(self, sites, it, cycle):
        return self.h.regen_mh(sites, it, cycle)

    def states(self):
        return self.h.state()


class _OraclePointedAdapter:
    def __init__(self, n, seed):
        self.h = O.OraclePointedMH(BOUNDS, COV, [0.0, 0.0], n, seed, canonical=True)

    def mh(self, noise, it):
        return self.h.mh(noise, it)

    def states(self):
        return self.h.state()


def oracle_paths(inp):
    return paths_outputs(lambda ys, n, seed, constrain=None: _OracleMhAdapter(ys, n, seed, constrain), _OraclePointedAdapter,
                         lambda kind, ds, do, params, obs, n, m, seed: O.importance_resampling(kind, ds, do, params, obs, n, m, seed, CANON_SOA, args0=[0.0] * ds),
                         _OraclePfAdapter, inp)


def main_paths():
    inp = paths_inputs()
    out = oracle_paths(inp)
    np.savez_compressed(os.path.join(HERE, "paths_small.npz"), **{"in_" + k: v for k, v in inp.items()}, **out)
    print("paths_small.npz:", sorted(out)[:6], "...", len(out), "arrays")


# ---- third fixture: mh / regen_mh over the registered functor models (kinds 101, 102, 103 of modppl_amd/csrc/mp_mh_models.h), as the
# checker's dynamic interpretation of those functors runs them (tries, sample_at / trace_at / gc): tests/golden/mh_functor.npz ---------
def functor_inputs():
    rng = np.random.default_rng(20241009)
    xs2 = np.linspace(-3, 3, 10)
    ys2 = 0.7 * xs2 - 0.4 + 0.3 * rng.normal(size=10)
    ys2[2] += 9.0
    ys2[7] -= 8.0
    xs3 = np.linspace(-2, 2, 9)
    return dict(hier_ys=0.3 + 0.4 * XS + 0.5 * XS * XS + 0.1 * rng.normal(size=XS.size), rl_xs=xs2, rl_ys=ys2, sl_xs=xs3,
                sl_ys=-0.6 * xs3 + 0.8 + 0.7 * rng.normal(size=9))


def functor_outputs(mk, inp):
    """mk(kind, params, constraints, n_chains, seed) -> an object with mh / regen_mh / trace (FunctionChains' surface)"""
    out = {}
    h = mk(101, XS, {4 + k: y for k, y in enumerate(inp["hier_ys"])}, 96, 41)
    acc = []
    for _ in range(2):
        acc += [h.mh(2, [], 1), h.mh(1, [0.1], 3), h.regen_mh([1, 2, 3], 6, True), h.regen_mh([1, 2], 2, False)]
    acc += [h.regen_mh([], 1, False), h.mh(1, [0.05], 2)]
    out["hier_vals"], out["hier_present"] = h.trace()
    out["hier_acc"] = np.array(acc, dtype=np.int64)
    r = mk(102, inp["rl_xs"], {14 + k: y for k, y in enumerate(inp["rl_ys"])}, 96, 42)
    acc = []
    for _ in range(2):
        acc += [r.mh(1, [0.3], 2)] + [r.mh(2, [k], 1) for k in (0, 2, 7, 9)] + [r.regen_mh([2 + k for k in range(10)], 10, True), r.regen_mh([0], 2, False),
                                                                                 r.regen_mh([1, 3, 6], 2, False)]
    acc += [r.regen_mh([5], 2, False), r.regen_mh([], 1, False), r.mh(1, [0.2], 2)]
    out["rl_vals"], out["rl_present"] = r.trace()
    out["rl_acc"] = np.array(acc, dtype=np.int64)
    s = mk(103, inp["sl_xs"], {3 + k: y for k, y in enumerate(inp["sl_ys"])}, 96, 43)
    acc = []
    for _ in range(3):
        acc += [s.regen_mh([0], 2, False), s.mh(2, [0.2], 2), s.mh(1, [], 2), s.regen_mh([0, 1], 2, False), s.regen_mh([0, 2, 1], 3, True)]
    acc += [s.regen_mh([], 1, False), s.regen_mh([0], 3, False)]
    out["sl_vals"], out["sl_present"] = s.trace()
    out["sl_acc"] = np.array(acc, dtype=np.int64)
    return out


def oracle_functor(inp):
    return functor_outputs(lambda kind, params, cons, n, seed: O.OracleFunctionChains(kind, params, cons, n, seed, canonical=True), inp)


def main_functor():
    inp = functor_inputs()
    out = oracle_functor(inp)
    np.savez_compressed(os.path.join(HERE, "mh_functor.npz"), **{"in_" + k: v for k, v in inp.items()}, **out)
    print("mh_functor.npz:", sorted(out), {k: int(v.sum()) for k, v in out.items() if k.endswith("_acc")})


if __name__ == "__main__":
    main()
    main_paths()
    main_functor()
