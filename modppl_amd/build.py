"""Builds the gfx950 shared library of the hot path in-tree (modppl_amd/csrc/libmodppl_hip.so).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the snapshot.

Staleness is decided by CONTENT: a sha256 over every source, header and the flag list is stored next
to the library (`<lib>.srchash`); a library whose stamp differs from the tree's hash is rebuilt
where hipcc exists and refused (loudly) where it does not, so a binary that does not correspond to
the checked-out sources is never loaded silently.
"""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libmodppl_hip.so")
SO_STAMPS = os.path.join(CSRC, "libmodppl_hip_stamps.so")   # diagnostics build (-DMP_STAMPS), tools/stamp_probe.py only
SO_DIAG = os.path.join(CSRC, "libmodppl_hip_diag.so")       # the same sources with -DMP_DIAGNOSTICS: the A/B and test switches (csrc/mp_diag.h) exist
                                                            # in this build only; tests and tools that need one load it, the product never does
SOURCES = ["mp_pf.hip", "mp_mh.hip", "mp_probe.hip"]
# -ffp-contract=off: the only fused multiply-adds are the explicit fma() calls in mp_math.h, so the
# device evaluates exp/log with exactly the operations the CPU checker uses (bit-exact indices).
# -amdgpu-kernarg-preload-count=8: the first 8 dwords of a kernel's arguments (scalars and pointers only; the run stops at the
# first aggregate) arrive in SGPRs at wave launch instead of through the kernel-argument fetch — k_propagate's tile-scalar
# pointers sit there so that a drawing launch's first loads do not wait ~1 us for that fetch (-0.3 us per step, measured).
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-mllvm", "-amdgpu-kernarg-preload-count=8"]


def _dep_files():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".hpp"))]
    inc = os.path.join(HERE, "..", "include")
    deps += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    return deps


def source_hash(extra_flags=()):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(extra_flags)).encode())
    for d in _dep_files():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_path(so):
    return so + ".srchash"


def is_stale(so=SO, extra_flags=()):
    if not os.path.exists(so) or not os.path.exists(_stamp_path(so)):
        return True
    with open(_stamp_path(so)) as f:
        return f.read().strip() != source_hash(extra_flags)


def build(force=False, verbose=False, so=SO, extra_flags=()):
    """Builds `so` unless its stamp matches the tree.  Safe when several processes (one rank per GPU) find the same stale
    library at import: the compile runs under an exclusive lock next to the library, the staleness test is repeated inside
    the lock (the ranks that waited find a fresh library and return), and both the library and its stamp appear atomically
    (compiled / written to temporary names, then os.replace) — a reader never maps a half-written file, and a kill between
    the two leaves a library without a matching stamp, i.e. stale, not a stale library that looks fresh."""
    if not force and not is_stale(so, extra_flags):
        return so
    if shutil.which("hipcc") is None:
        raise RuntimeError(f"{so} does not correspond to the sources in this tree (content hash differs) and hipcc is not available to rebuild it")
    import fcntl

    with open(so + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale(so, extra_flags):
                return so   # another process built it while this one waited
            tmp = f"{so}.tmp.{os.getpid()}"
            # the translation units are compiled side by side (mp_pf.hip alone is two thirds of the serial time), then linked
            cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
            objs = [f"{tmp}.{i}.o" for i in range(len(SOURCES))]
            procs = [subprocess.Popen(["hipcc"] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                     for src, obj in zip(SOURCES, objs)]
            outs = [p.communicate() for p in procs]
            err = "".join(o + e for o, e in outs)
            ok = all(p.returncode == 0 for p in procs)
            res = None
            if ok:
                res = subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + objs, capture_output=True, text=True)
                err += res.stdout + res.stderr
                ok = res.returncode == 0
            for o in objs:
                if os.path.exists(o):
                    os.unlink(o)
            if not ok:
                if os.path.exists(tmp):
                    os.unlink(tmp)
                raise RuntimeError("hipcc failed:\n" + err)
            if os.path.exists(_stamp_path(so)):
                os.unlink(_stamp_path(so))   # (first the old stamp goes: from here until the new one lands the library counts as stale)
            os.replace(tmp, so)
            stmp = f"{_stamp_path(so)}.tmp.{os.getpid()}"
            with open(stmp, "w") as f:
                f.write(source_hash(extra_flags) + "\n")
            os.replace(stmp, _stamp_path(so))
            if verbose:
                print(err)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return so


def build_stamps(force=False):
    """the diagnostics library: same sources with per-workgroup clock stamps compiled in (never loaded by the product)"""
    return build(force=force, so=SO_STAMPS, extra_flags=("-DMP_STAMPS", "-DMP_DIAGNOSTICS"))


def build_diag(force=False):
    """the same library with its A/B and test switches enabled (csrc/mp_diag.h); never loaded by the product"""
    return build(force=force, so=SO_DIAG, extra_flags=("-DMP_DIAGNOSTICS",))


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "stamps":
        print(build_stamps(force=True))
    elif len(sys.argv) > 1 and sys.argv[1] == "diag":
        print(build_diag(force=True))
    else:
        print(build(force=True, verbose=True))
