import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import vi_ekf_amd as v
from vi_ekf_amd import scene
N=int(sys.argv[1]); B=2
sc = scene.make_scene(B, N, 2, seed=100+N)
g = v.BatchVIEKF(B, N, sc["params"])
g.set_kernel(2)
for i in range(N):
    g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
print("init ok", flush=True)
g.propagate(sc["u"][0], sc["dt"]); print("prop ok", flush=True)
r = g.update_feat(sc["z"][0], sc["slot"], sc["R"]); print("upd ok", r[0,:5], flush=True)
