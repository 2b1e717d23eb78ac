"""Diagnostic: one update of N features at wide P with a -DVIEKF_STAMPS build (VIEKF_LIB=variants/lib_stamps.so) and the s_memtime
stamps of filter 0, first group, measurement 8 of k_update_feat_panelsvc (clocks of the 100 MHz constant counter x 24 = shader clocks
are NOT assumed: deltas are printed in s_memtime ticks)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa

import vi_ekf_amd as v
from vi_ekf_amd import scene, capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc = scene.make_scene(B, N, 2, seed=3)
g = v.BatchVIEKF(B, N, sc["params"])
print(g.describe())
for i in range(N):
    g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
for rep in range(2):
    g.update_feat(sc["z"][rep], sc["slot"], sc["R"])
ws = np.zeros(256, dtype=np.uint64)
capi.lib().viekf_debug_read_ws.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
capi.check(capi.lib().viekf_debug_read_ws(g._h, C.c_void_p(ws.ctypes.data), 256))
t = ws.astype(np.int64)
print("second trip of the group loop (group 1's chain under group 0's pass); s_memtime ticks")
for w in range(8):
    q = 16 * w
    d = lambda i, j: int(t[q + j] - t[q + i])
    if w < 7:
        print("wave %d: preload barrier -> D done %6d | pass %6d | wait %6d || A %6d | C %6d" % (w, d(0, 1) if t[q + 1] else 0, d(2, 6), d(6, 3), d(3, 4), d(4, 5)))
    else:
        print("wave 7: preload + pending corrections %6d | chain %6d | pass %6d | wait %6d || A %6d | C %6d" % (d(0, 7), d(7, 2), d(2, 6), d(6, 3), d(3, 4), d(4, 5)))
print("inside the pass (first marks of each wave; A, B = the two register sets):")
print("        issue A | draw | issue B | k-loop A | row scale A | stores A | draw | issue A' | k-loop B | row scale B | stores B | draw | issue B' | k-loop A' | row scale A'")
for w in range(8):
    q = 128 + 16 * w
    print("wave %d: " % w + " ".join("%6d" % int(t[q + i + 1] - t[q + i]) for i in range(15)))
