#!/usr/bin/env python3
"""A/B builds of the library: tools/build_variant.py NAME [-DFLAG=VALUE ...]  ->  _ab/lib_NAME.so (git-ignored; travels to the GPU box).
Run one with MODPPL_HIP_LIB=$PWD/_ab/lib_NAME.so (tools/ab_libs.sh); bench.py marks such a line `library.override`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modppl_amd import build as b  # noqa: E402

name, flags = sys.argv[1], tuple(sys.argv[2:])
os.makedirs(os.path.join(os.path.dirname(b.HERE), "_ab"), exist_ok=True)
so = os.path.join(os.path.dirname(b.HERE), "_ab", f"lib_{name}.so")
print(b.build(force=True, so=so, extra_flags=flags), flags)
