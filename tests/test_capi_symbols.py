"""CPU: the C-ABI library loads without a GPU and exports every symbol include/*.h declares;
creating a handle without a device fails loudly (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for hdr in ("modppl_hip.h", "modppl_hip_probe.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from modppl_amd import capi

    L = capi.load()
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    for name in sorted(decl):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert decl == set(capi.SYMBOLS), (decl ^ set(capi.SYMBOLS))


def test_no_cpu_fallback_without_device():
    import modppl_amd
    from modppl_amd import capi

    L = capi.load()
    if L.mp_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(modppl_amd.ModpplError) as e:
        modppl_amd.ParticleSystem(modppl_amd.lgssm_model(), 128, 1)
    assert e.value.code == capi.MP_ERR_HIP


def test_product_never_imports_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "modppl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                for line in open(os.path.join(dirpath, f), errors="ignore"):
                    code = line.split("//")[0].split("#include")[-1] if "#include" in line else line.split("#")[0]
                    is_dep = ("#include" in line and "oracle" in code) or \
                             (re.search(r"^\s*(from|import)\s+.*oracle", line) is not None) or \
                             ("liboracle" in code) or ("oracle_lib" in code)
                    assert not is_dep, (os.path.join(dirpath, f), line)


def test_only_the_cpu_baseline_leg_of_bench_touches_the_oracle():
    """bench.py may use oracle/ in cpu_baseline() only; the tools not at all."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    lines = src.splitlines()
    uses = [i for i, l in enumerate(lines) if "oracle_lib" in l and not l.lstrip().startswith("#")]
    assert uses, "cpu_baseline should still import the checker"
    defs = [(i, l) for i, l in enumerate(lines) if l.startswith("def ")]
    for i in uses:
        owner = [name for j, name in defs if j <= i][-1]
        assert owner.startswith(("def _cpu_run", "def cpu_baseline")), (i + 1, owner)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            assert "oracle_lib" not in open(os.path.join(ROOT, "tools", f)).read(), f


def test_rust_sys_crate_declares_every_entry_point():
    """The source-only Rust FFI crate (no rustc in this image) must at least name every function of include/modppl_hip.h."""
    import re

    from modppl_amd import capi
    src = open(os.path.join(ROOT, "modppl_amd", "rust", "modppl-hip-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (mp_\w+)\(", src))
    wanted = {s for s in capi.SYMBOLS if not s.startswith("mp_probe")}
    assert wanted <= declared, sorted(wanted - declared)
