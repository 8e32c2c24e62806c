"""Builds the gfx950 shared library of the hot path in-tree (modppl_amd/csrc/libmodppl_hip.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libmodppl_hip.so")
SOURCES = ["mp_pf.hip", "mp_mh.hip", "mp_probe.hip"]
HEADERS = ["mp_math.h", "mp_philox.h", "mp_dists.h", "mp_models.h", "mp_linalg.h", "mp_pf_kernels.h", "mp_pf_shard_kernels.h", os.path.join("..", "..", "include", "modppl_hip.h"),
           os.path.join("..", "..", "include", "modppl_hip_probe.h")]
# -ffp-contract=off: the only fused multiply-adds are the explicit fma() calls in mp_math.h, so the
# device evaluates exp/log with exactly the operations the CPU checker uses (bit-exact indices).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared"]


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not is_stale():
        return SO
    cmd = ["hipcc"] + FLAGS + ["-o", SO] + [os.path.join(CSRC, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
