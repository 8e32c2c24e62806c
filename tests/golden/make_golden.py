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


if __name__ == "__main__":
    main()
