#!/usr/bin/env python3
"""Phase breakdown of k_propagate_mt (one workgroup per CU, two tiles: mp_pf_k1mt.h) from a raw stamp dump
(tools/stamp_probe.py --raw): mean shader cycles of wave 0 between stamps, and when the workgroups end."""
import sys

import numpy as np

b = np.load(sys.argv[1])
x = b[0]
x = x[x[:, 1] != 0].astype(np.int64)
seq = [(0, 17, "entry, tile-scalar loads, Philox A, table in LDS"), (17, 18, "draws of A and B: targets, tile walks, guide cells asked"),
       (18, 19, "deviates of A (order 0: under the guide gathers)"), (19, 2, "guide cells of A landed -> rows of A asked"),
       (2, 22, "deviates of B (order 1: of A too), under A's row gathers"), (22, 23, "rows of A landed, first walk loads, rows of B asked"),
       (23, 25, "rest of the walks: parents of A"), (25, 26, "model x 2 (A)"), (26, 20, "normalisation of A, compute only (2 barriers)"),
       (20, 27, "rows of B landed, parents of B"), (27, 28, "stores of A, model x 2 (B)"), (28, 4, "normalisation of B, stores, barrier, guides")]
tot = (x[:, 4] - x[:, 0]).mean()
print("k_propagate_mt: %d workgroups, wave 0 lifetime %.0f cycles = %.2f us real time, clock %.0f MHz" % (
    len(x), tot, ((x[:, 5] - x[:, 1]).mean()) / 100.0, np.median((x[:, 4] - x[:, 0]) / np.maximum(x[:, 5] - x[:, 1], 1) * 100)))
for a, c, nm in seq:
    d = (x[:, c] - x[:, a])
    print("  %-58s %8.0f cycles  %5.1f %%   (p10 %.0f, p90 %.0f)" % (nm, d.mean(), 100 * d.mean() / tot, np.percentile(d, 10), np.percentile(d, 90)))
rt0, rt1 = x[:, 1], x[:, 5]
t0 = rt0.min()
start, end = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0
print("workgroup starts  p0 %.2f p50 %.2f p100 %.2f us;  ends  p0 %.2f p50 %.2f p90 %.2f p100 %.2f us" % (
    start.min(), np.median(start), start.max(), end.min(), np.median(end), np.percentile(end, 90), end.max()))
