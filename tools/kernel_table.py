#!/usr/bin/env python3
"""VGPRs, scratch and static LDS of every kernel in a gfx950 .s file (hipcc -S --cuda-device-only):  python tools/kernel_table.py file.s [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, flags=re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: int(re.search(r"\." + k + r" (\d+)", body).group(1))
    rows.append((name, g("amdhsa_next_free_vgpr"), g("amdhsa_private_segment_fixed_size"), g("amdhsa_group_segment_fixed_size")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (n, v, s, l), d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt in d:
        print(f"{d[:64]:64s} vgpr {v:4d}  scratch {s:5d}  lds {l:6d}")
