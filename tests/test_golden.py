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
