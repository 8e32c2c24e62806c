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


# ---- signatures, not just names: header <-> Rust -sys crate <-> ctypes binding ---------------------------------------------
_C2RUST = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "c_void", "char": "c_char",
           "mp_pf": "mp_pf", "mp_mh": "mp_mh", "mp_model_desc": "mp_model_desc", "mp_shard": "mp_shard", "mp_transport": "mp_transport"}


def _c_type_to_rust(t):
    """`const double*` -> `*const f64`, `mp_pf**` -> `*mut *mut mp_pf`, `uint64_t` -> `u64`."""
    import re

    t = t.strip()
    const = t.startswith("const ")
    base = re.sub(r"^const\s+", "", t)
    stars = base.count("*")
    base = base.replace("*", "").strip()
    r = _C2RUST[base]
    for k in range(stars):
        r = ("*const " if (const and k == 0) else "*mut ") + r
    return r


def _header_functions():
    import re

    src = open(os.path.join(ROOT, "include", "modppl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+\s*\*?)\s*(mp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M):
        args = " ".join(args.split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)(\w+)$", a)       # type, then the parameter name
                params.append(m.group(1).strip())
        out[name] = (ret.strip(), params)
    return out


def test_rust_sys_signatures_match_the_header():
    """argument count, order, width, signedness and constness of every entry point: the -sys crate is source only (no rustc in
    the image), so its declarations are checked against include/modppl_hip.h textually."""
    import re

    hdr = _header_functions()
    assert len(hdr) >= 40
    src = open(os.path.join(ROOT, "modppl_amd", "rust", "modppl-hip-sys", "src", "lib.rs")).read()
    rust = {}
    for name, args, ret in re.findall(r"pub fn (mp_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", src):
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if ":" in a]
        rust[name] = ((ret or "").strip(), params)
    for name, (ret, params) in hdr.items():
        assert name in rust, name
        rret, rparams = rust[name]
        assert rret == _c_type_to_rust(ret), (name, ret, rret)
        assert rparams == [_c_type_to_rust(p) for p in params], (name, params, rparams)
    # repr(C) struct layouts
    for struct, fields in (("mp_model_desc", ["kind: i32", "dim_state: i32", "dim_obs: i32", "n_params: i32", "params: *const f64"]),
                           ("mp_shard", ["n_global: u64", "slot_offset: u64"])):
        body = re.search(r"#\[repr\(C\)\]\s*pub struct %s \{(.*?)\}" % struct, src, flags=re.S).group(1)
        got = [" ".join(f.replace("pub ", "").split()) for f in body.split(",") if f.strip()]
        assert got == fields, (struct, got)
        cbody = re.search(r"typedef struct %s \{(.*?)\}" % struct, re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "modppl_hip.h")).read(), flags=re.S), flags=re.S).group(1)
        cfields = [f.strip() for f in cbody.split(";") if f.strip()]
        assert len(cfields) == len(fields)
        for cf, rf in zip(cfields, fields):
            m = re.match(r"^(.*?)(\w+)$", cf)
            assert rf == f"{m.group(2)}: {_c_type_to_rust(m.group(1))}", (cf, rf)


def test_ctypes_binding_matches_the_header():
    """the ctypes argtypes (what every GPU test calls through) have the header's argument counts and widths"""
    import ctypes as C

    from modppl_amd import capi

    L = capi.load()
    width = {"int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "double": C.c_double}
    for name, (ret, params) in _header_functions().items():
        fn = getattr(L, name)
        if not params:
            continue
        assert fn.argtypes is not None and len(fn.argtypes) == len(params), (name, len(params), fn.argtypes)
        for p, a in zip(params, fn.argtypes):
            if "*" in p:
                assert a in (C.c_void_p,) or issubclass(a, C._Pointer), (name, p, a)
            else:
                assert a is width[p.strip()], (name, p, a)


def test_every_host_kernel_stub_has_device_code(tmp_path):
    """each `__device_stub__<kernel>` the host side can launch has a gfx950 kernel of that name in the library's code objects
    (a template instantiated only in the host pass links, loads, and aborts at its first launch with "Cannot find Symbol")"""
    import subprocess

    from modppl_amd import build

    lib = build.build()
    nm = subprocess.run(["nm", "-C", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    raw = subprocess.run(["nm", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    stubs = set()
    for ln in raw.splitlines():
        m = re.match(r"_Z(\d+)__device_stub__(.*)", ln.split()[-1])   # _Z<len>__device_stub__<name><args> -> _Z<len-15><name><args>
        if m:
            stubs.add("_Z%d%s" % (int(m.group(1)) - len("__device_stub__"), m.group(2)))
    assert len(stubs) > 40 and nm
    fat = tmp_path / "fat.bin"
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, str(fat)], check=True)
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)]
    assert starts
    device = set()
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    for i, s in enumerate(starts):
        part = tmp_path / ("bundle%d.bin" % i)
        part.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / ("dev%d.co" % i)
        subprocess.run([bundler, "--unbundle", "--type=o", "--input=" + str(part), "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--output=" + str(co)], check=True)
        syms = subprocess.run(["readelf", "-sW", str(co)], capture_output=True, text=True, check=True).stdout
        device |= {ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln}
    missing = sorted(stubs - device)
    assert not missing, "host stubs without device code: %s" % missing[:5]


def test_the_product_library_reads_no_environment_switch():
    """Round 4's review: twenty getenv("MP_...") A/B and test switches lived in the library a host links.  They are in the DIAGNOSTICS
    build only now (csrc/mp_diag.h: mp_diag_env is a constant null without -DMP_DIAGNOSTICS): no product source calls getenv, and the
    switch names are not even strings of libmodppl_hip.so, while libmodppl_hip_diag.so — same sources, -DMP_DIAGNOSTICS — has them."""
    import re
    import subprocess

    from modppl_amd import build as B

    csrc = os.path.join(os.path.dirname(B.__file__), "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")) and f != "mp_diag.h":
            src = open(os.path.join(csrc, f)).read()
            assert not re.search(r"(?<![A-Za-z_])getenv\s*\(", src), f
    B.build()
    B.build_diag()
    names = [b"MP_K1_MT_FLAGS", b"MP_DEFERRED_LOOKUPS", b"MP_FUSED_DRAWS", b"MP_WALK_BISECT", b"MP_SHARD_OWNED_CAP", b"MP_HOST_MIRROR"]
    prod = subprocess.run(["strings", B.SO], capture_output=True).stdout
    diag = subprocess.run(["strings", B.SO_DIAG], capture_output=True).stdout
    for n in names:
        assert n not in prod, n
        assert n in diag, n
