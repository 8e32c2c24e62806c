#!/bin/bash
# same-box comparison of builds of the library on the other configs (C3, C5 shard, C4): tools/ab_models.sh lib1.so lib2.so ...
for L in "$@"; do
  echo "== $L"
  MODPPL_HIP_LIB=$PWD/$L timeout -k 10 300 python tools/model_bench.py --which ${MP_WHICH:-c3,c5,c4} 2>/dev/null | grep -v "^$" | cut -c1-330
done
