"""The C++ host wrapper (modppl_amd/cpp/modppl.hpp): CPU — it compiles and links against libmodppl_hip.so;
GPU — the reference's SMC loop through it gives the values of the Python mirror (same C ABI underneath)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "smc_loop.cpp")
CSRC = os.path.join(ROOT, "modppl_amd", "csrc")


def build_exe(tmp_path):
    from modppl_amd import build as b

    b.build(force=False)
    exe = str(tmp_path / "smc_loop")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", SRC, "-o", exe, "-L" + CSRC, "-lmodppl_hip", "-Wl,-rpath," + CSRC,
           "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "warning" not in res.stderr, res.stderr[-3000:]
    return exe


def test_cpp_wrapper_compiles_and_links(tmp_path):
    build_exe(tmp_path)


@pytest.mark.gpu
def test_cpp_wrapper_runs_the_smc_loop(tmp_path):
    import modppl_amd
    exe = build_exe(tmp_path)
    n, seed = 4096, 7
    env = dict(os.environ)
    res = subprocess.run([exe, str(n), str(seed), "1"], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    ys = [0.31, -0.12, 0.58, 1.02, 0.44, -0.27]
    pf = modppl_amd.ParticleSystem(modppl_amd.lgssm_model(0.0, 1.0, 0.9, 0.5, 1.0), n, seed)
    pf.init_step(None, ys[:1])
    pf.resample()
    for y in ys[1:]:
        pf.step([y])
        pf.resample()
    want = "lml=%.17g x0=%.17g parents0=%u" % (pf.log_marginal_likelihood_estimate(), pf.states()[0, 0], pf.parents[0])
    ch = modppl_amd.HierarchicalChains([-1.0, 0.0, 1.0], [0.2, 0.9, 2.2], 256, seed)
    want2 = "accepted=%d,%d" % (ch.mh_add_or_remove(2), ch.mh(0.1, 3))
    lines = res.stdout.strip().splitlines()
    assert lines[0] == want and lines[1] == want2, (lines, want, want2)
    # the same chains as a registered functor model through the generic entry points, and the sharded filter's native resample
    # (world of one, RCCL collectives forced): a world of one IS the single filter
    assert lines[2] == want2.replace("accepted=", "accepted_fn=") + " sites=20", lines
    # generate / simulate on that handle and importance sampling over the declared-data functor: the Python mirror's values
    fc = modppl_amd.FunctionChains(101, [-1.0, 0.0, 1.0], {4: 0.2, 5: 0.9, 6: 2.2}, 256, seed)
    fc.mh(2, [], 2); fc.mh(1, [0.1], 3)
    w = fc.generate({4: 0.2, 5: 0.9, 6: 2.2}, rng_step=40)
    lj = fc.simulate(rng_step=41)
    tr, idx, lml = modppl_amd.fn_importance_resampling(105, [-1.0, 0.0, 1.0], {4: 0.2, 5: 0.9, 6: 2.2}, 2048, 8, seed)
    _, lnw, _ = modppl_amd.fn_importance_sampling(105, [-1.0, 0.0, 1.0], {4: 0.2, 5: 0.9, 6: 2.2}, 2048, seed, traces=False)
    want3 = "fn w0=%.17g lj0=%.17g lml=%.17g idx0=%d idx7=%d lnw0=%.17g sites=%d" % (w[0], lj[0], lml, idx[0], idx[7], lnw[0], tr.num_sites)
    assert [l for l in lines if l.startswith("fn ")] == [want3], (lines, want3)
    sharded = [l for l in lines if l.startswith("sharded ")]   # (RCCL prints its version banner to stdout in between)
    assert sharded == ["sharded " + want.split()[0]], (lines, want)
