"""VGPR / SGPR / scratch / LDS of every gfx950 kernel in the built library (from the code objects' metadata notes).
usage: python tools/kernel_resources.py [substring]        (MODPPL_HIP_LIB=path: that library instead of the tree's)"""
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from modppl_amd import build  # noqa: E402


def main():
    lib = os.environ.get("MODPPL_HIP_LIB") or build.build()
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        fat = td / "fat.bin"
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
        blob = fat.read_bytes()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for i, s in enumerate(starts):
            part = td / f"b{i}.bin"
            part.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = td / f"d{i}.co"
            subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(co)], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
                name = g("name")
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem)
                if want in dem:
                    print(f"{dem:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
