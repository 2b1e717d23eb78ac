"""Diagnostic: runs steps with a -DVIEKF_STAMPS build (VIEKF_LIB=...) of the TILE family and prints the s_memtime deltas of block 0
(worker wave 0, lane 0 and the service wave, lane 0).   usage: python tools/stamps_tiles.py [N] [B]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa

import vi_ekf_amd as v
from vi_ekf_amd import scene, capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = scene.make_scene(B, N, 3, seed=3)
g = v.BatchVIEKF(B, N, sc["params"])
g.set_tuning(capi.TUNE_TILES, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
print(g.describe())
for i in range(N):
    g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
for rep in range(3):
    g.step(sc["u"][rep], sc["dt"], sc["z"][rep], sc["slot"], sc["R"])
ws = np.zeros(256, dtype=np.uint64)
capi.lib().viekf_debug_read_ws.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
capi.check(capi.lib().viekf_debug_read_ws(g._h, C.c_void_p(ws.ctypes.data), 256))
t = ws.astype(np.int64)
d = lambda a, b: int(t[b] - t[a])
print("worker0: load issue %d | B0 wait %d | prop setup %d | B3p wait %d | tile propagate %d | to B4q %d | B4q wait %d | body tiles+Pd+extract %d | Bp..B1 %d"
      % (d(0, 64) if t[0] else 0, d(64, 65), d(65, 66), d(66, 67), d(67, 68), d(68, 70), d(70, 71), d(71, 72), d(72, 73)))
print("service: dynamics %d | B0 wait %d | propagate interval %d | to Bp %d | first predict %d | B1 wait %d" % (d(0, 14), d(14, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5)))
for it in range(8):
    w = 80 + 4 * it
    s = 16 + 4 * it
    print("update %d  worker0: operands+MFMA issue %5d | fix-up %5d | extraction %5d | barrier %5d  (phase %5d)     service: column reads %5d | correction %5d | fix+predict %5d | barrier %5d"
          % (it, d(w, w + 1), d(w + 1, w + 2), d(w + 2, w + 3), d(w + 3, 112 + it), d(w, 112 + it) + (d(112 + it - 1, w) if it else d(73, w)),
             d(s - 1 if it else 5, s), d(s, s + 1), d(s + 1, s + 2), d(s + 2, s + 3)))
print("worker0: loop end -> store done %d ; whole worker %d ; whole service %d" % (d(74, 75), d(64, 75), d(0, 13)))
