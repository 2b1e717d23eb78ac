"""Diagnostic: runs ONE step with a -DVIEKF_STAMPS build (VIEKF_LIB=...) and prints the s_memtime deltas of block 0."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa

import vi_ekf_amd as v
from vi_ekf_amd import scene, capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc = scene.make_scene(B, N, 2, seed=3)
g = v.BatchVIEKF(B, N, sc["params"])
for i in range(N):
    g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
for rep in range(2):
    g.step(sc["u"][rep], sc["dt"], sc["z"][rep], sc["slot"], sc["R"])
# read the workspace of block 0 through a debug hook: the stamps live at d_ws[0..127]
ws = np.zeros(256, dtype=np.uint64)
capi.lib().viekf_debug_read_ws.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
capi.check(capi.lib().viekf_debug_read_ws(g._h, C.c_void_p(ws.ctypes.data), 256))
t = ws.astype(np.int64)
def d(a, b): return int(t[b] - t[a])
print("common prologue (62->63 incl barrier)", d(62, 63))
print("service: B0 wait", d(0, 1), " body phase", d(1, 2), " B1p wait", d(2, 3), " feature phase", d(3, 4), " B2p wait", d(4, 5),
      " body step+fix", d(5, 6), " B3p wait", d(6, 7), " B4p wait (GEMM)", d(7, 8), " first h_feat", d(8, 9), " B1 wait", d(9, 10))
print("worker : load->B0", d(63, 64), " B0 wait", d(64, 65), " to B3p", d(65, 66), " B3p wait", d(66, 67), " local3x3", d(67, 68),
      " GEMM", d(68, 69), " strips", d(69, 70), " B4p+extract->B1", d(70, 71))
# one barrier per update (B1).  service: 16+4it+{0: phase inputs loaded, 1: rotation vector ready, 2: gain rows of the next
# measurement written (before B1), 3: after B1}; 128+4it+{0,1}: around the manifold correction.  worker (thread 0):
# 80+4it+{1: phase top, 2: before B1, 3: after B1}; 160+4it+{0: blocks swept, 1: raw columns published + counted in}.
for it in range(8):
    s0 = 16 + 4 * it; w0 = 80 + 4 * it; q0 = 160 + 4 * it; b0 = 128 + 4 * it
    print("update %d  service: inputs %5d | correction %5d | fix+predict+poll+gain rows %5d | B1 wait %5d      worker0: blocks %5d | "
          "publish+count %5d | body columns %5d | B1 wait %5d" % (
              it, d(s0 - 1 if it else 10, s0), d(s0, b0 + 1), d(b0 + 1, s0 + 2), d(s0 + 2, s0 + 3),
              d(w0 + 1, q0), d(q0, q0 + 1), d(q0 + 1, w0 + 2), d(w0 + 2, w0 + 3)))
for it in range(8):
    b0 = 128 + 4 * it; s0 = 16 + 4 * it
    print("update %d  service detail: fix_depth %5d | predict+result %5d | poll %5d | sync+gain rows %5d" % (
        it, d(b0 + 1, b0 + 2), d(b0 + 2, b0 + 3), d(b0 + 3, 48 + it), d(48 + it, s0 + 2)))
it = 3
for w in range(7):
    q = 192 + 4 * w
    print("update 3 worker wave %d: blocks+publish %5d | body columns %5d | B1 wait %5d   (top skew vs wave 0: %d)" % (
        w, d(q, q + 2), d(q + 2, q + 1), d(q + 1, q + 3), int(t[q] - t[192])))
print("service tail: ", d(11, 12), d(12, 13), " worker store", d(72, 73), " total service", d(0, 13), " total worker", d(62, 73))
print("store: body cols %d" % d(72, 224))
for ch in range(3):
    q = 225 + 4 * ch
    print("  chunk %d: image writes %5d | S1 wait %5d | copy-out %5d | S2 wait %5d" % (ch, d(q - 1, q), d(q, q + 1), d(q + 1, q + 2), d(q + 2, q + 3)))
