#!/usr/bin/env python3
"""Phase breakdown of k_propagate from a raw stamp dump (tools/stamp_probe.py --raw): mean shader cycles of wave 0 between stamps."""
import sys

import numpy as np

b = np.load(sys.argv[1])
x = b[0]
x = x[x[:, 1] != 0].astype(np.int64)
# (slots 16..24 exist only in launches that make their draws themselves)
names = [(0, 16, "entry, table loads issued, resample Philox"), (16, 17, "table copy landed + barrier 0"), (17, 18, "targets + tile walks (LDS)"),
         (18, 19, "guide cells -> start rows, draw stores"), (19, 20, "lookup of particle 0 (row pair)"), (20, 21, "attempt 0 of both deviates"),
         (21, 22, "retry rounds (wave-cooperative)"), (22, 2, "log / divide / sqrt -> z"), (2, 24, "lookup of particle 1 (row pair)"), (24, 3, "model x 2"), (3, 8, "wave max"), (8, 9, "barrier 1"), (9, 10, "exp + quantise"),
         (10, 11, "wave scan"), (11, 12, "LDS atomic + barrier 2"), (12, 7, "cross-wave offsets"), (7, 13, "scalars/ticket + row stores"),
         (13, 14, "guide build"), (14, 15, "barrier 3"), (15, 4, "guide store (+ table)")]
tot = (x[:, 4] - x[:, 0]).mean()
print("k_propagate: %d workgroups, wave 0 lifetime %.0f cycles = %.2f us real time" % (len(x), tot, ((x[:, 5] - x[:, 1]).mean()) / 100.0))
for a, c, nm in names:
    d = (x[:, c] - x[:, a])
    print("  %-32s %8.0f cycles  %5.1f %%   (p10 %.0f, p90 %.0f)" % (nm, d.mean(), 100 * d.mean() / tot, np.percentile(d, 10), np.percentile(d, 90)))
for k, nm in ((1, "k_draw_slots"),):
    y = b[k]
    y = y[y[:, 1] != 0].astype(np.int64)
    if len(y):
        print("%s: %d workgroups, span %.2f us, mean lifetime %.2f us, clock %.0f MHz" % (nm, len(y), (y[:, 5].max() - y[:, 1].min()) / 100.0, (y[:, 5] - y[:, 1]).mean() / 100.0,
                                                                                      np.median((y[:, 4] - y[:, 0]) / np.maximum(y[:, 5] - y[:, 1], 1) * 100)))
# who finishes when: the kernel ends with its slowest workgroup
rt0, rt1 = x[:, 1], x[:, 5]
t0 = rt0.min()
start, end = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0
print("workgroup starts  p0 %.2f p50 %.2f p100 %.2f us;  ends  p0 %.2f p50 %.2f p90 %.2f p100 %.2f us" % (
    start.min(), np.median(start), start.max(), end.min(), np.median(end), np.percentile(end, 90), end.max()))
nb = len(x)
print("first half of the grid (first workgroup of each CU): mean end %.2f us; second half: %.2f us" % (end[:nb // 2].mean(), end[nb // 2:].mean()))

# the start-up of a drawing launch, first against second workgroup of a CU (first / second half of the grid)
if x[:, 25].any():
    half = nb // 2
    seq = [(0, 25, "kernel arguments, struct (scalar loads)"), (25, 26, "tile-scalar loads issued"), (26, 27, "resample Philox block"),
           (27, 28, "tile scalars landed, wave max"), (28, 29, "barrier"), (29, 16, "exp, scan, barrier, offsets, LDS table"), (16, 17, "barrier 0")]
    print("start-up, mean cycles: first workgroup of a CU | second")
    for a, c, nm in seq:
        d = x[:, c] - x[:, a]
        print("  %-44s %8.0f | %8.0f" % (nm, d[:half].mean(), d[half:].mean()))
