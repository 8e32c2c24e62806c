"""gfx950 disassembly of one kernel of the built library (or of MODPPL_HIP_LIB=path).
usage: python tools/disasm_kernel.py <mangled-name substring> [out.s]      e.g.  k_propagateI9mp_lgssm1Li1024E"""
import os
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    want = sys.argv[1]
    lib = os.environ.get("MODPPL_HIP_LIB")
    if not lib:
        from modppl_amd import build
        lib = build.build()
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
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", str(co)], capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n", out, re.M):
                if want in m.group(1) and not m.group(1).startswith("__"):
                    end = re.compile(r"^[0-9a-f]+ <[^>]+>:\n", re.M).search(out, m.end())
                    body = out[m.end():end.start() if end else len(out)]
                    if len(sys.argv) > 2:
                        Path(sys.argv[2]).write_text(body)
                        print(m.group(1), len(body.splitlines()), "lines ->", sys.argv[2])
                    else:
                        print(body)
                    return
    sys.exit(f"no kernel matching {want!r}")


if __name__ == "__main__":
    main()
