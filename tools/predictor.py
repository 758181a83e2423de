#!/usr/bin/env python3
"""How well does the pilot pass predict the long chains?  Offline, from a dump of the diagnostic build:
    RT_STATS_DUMP=gpurun_out/c3_pixels.npz python tools/stats.py 10000 1200 800 64 32      (GPU box)
    python tools/predictor.py gpurun_out/c3_pixels.npz
`it` = main-loop iterations of every pixel (the truth), `pilot` = the pilot's bounce count per 2x2 block (tile x 16).  Prints,
for several selection rules, how many pixels they pick, the share of the pixels above 800 ... 2000 iterations among them (recall)
and how much of the pick is short.  RT_PILOT_LONG_SUM in csrc/rt_kernels.hip was chosen from this table."""
import sys
import numpy as np
from scipy.ndimage import uniform_filter, maximum_filter

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c3_pixels.npz")
it = d["it"].astype(np.int32); pilot = d["pilot"].astype(np.int32)
ny, nx = it.shape; tx, ty = nx // 8, ny // 8
P = np.zeros((ny // 2, nx // 2), np.int32)
for sub in range(16):
    P[(sub >> 2)::4, (sub & 3)::4] = pilot[:, sub].reshape(ty, tx)
S3 = np.rint(uniform_filter(P.astype(float), 3, mode="nearest") * 9).astype(int)      # the block and its eight neighbours
up = lambda Q: np.repeat(np.repeat(Q, 2, axis=0), 2, axis=1)
levels = (800, 1000, 1280, 1600, 2000)
print("pixels above %s iterations: %s" % (levels, [int((it >= k).sum()) for k in levels]))
def report(name, blocks):
    sel = up(blocks)
    r = [((it >= k) & sel).sum() / max(1, (it >= k).sum()) for k in levels]
    print("%-22s picks %6d px   recall %s   mean length %4.0f   shorter than 400: %.2f" % (
        name, sel.sum(), " ".join("%d:%.2f" % kv for kv in zip(levels, r)), it[sel].mean() if sel.any() else 0, (it[sel] < 400).mean() if sel.any() else 0))
for t in (30, 40, 50, 60): report("own >= %d" % t, P >= t)
for t in (50, 60, 70): report("max 3x3 >= %d" % t, maximum_filter(P, 3) >= t)
for t in (160, 180, 200, 220, 250, 300): report("sum 3x3 >= %d" % t, S3 >= t)
