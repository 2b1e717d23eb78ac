import sys, numpy as np
sys.path.insert(0, '.')
import vi_ekf_amd as v
from oracle import oracle as orc
from vi_ekf_amd import scene
from tests.test_gpu_parity import oracle_params, make_gpu
B, N = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 51
sc = scene.make_scene(B, N, 2, seed=100 + N)
for mode in ("prop", "prop+1upd", "step"):
    g = make_gpu(sc, B, N)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N): f.init_feature(sc["pix"][b, i], i, float("nan"))
        fs.append(f)
    if mode == "prop":
        g.propagate(sc["u"][0], sc["dt"])
        for b in range(B): fs[b].propagate(sc["u"][0, b], sc["dt"][b])
    else:
        M = 1 if mode == "prop+1upd" else N
        z = sc["z"][0][:, :M].copy(); sl = sc["slot"][:, :M].copy()
        res = g.step(sc["u"][0], sc["dt"], z, sl, sc["R"])
        for b in range(B):
            fs[b].run_steps(sc["u"][0, b][None], sc["dt"][b], z[b][None], sl[b], sc["R"])
    x = g.get_state(); P = g.get_covariance()
    xr = np.stack([f.x for f in fs]); Pr = np.stack([f.P for f in fs])
    dx = np.abs(x - xr); dP = np.abs(P - Pr)
    print(mode, "x err", dx.max(), "at", np.unravel_index(dx.argmax(), dx.shape), " P err", dP.max(), "at", np.unravel_index(dP.argmax(), dP.shape), "flags", g.get_status())
    if dx.max() > 1e-9:
        b = 0
        print("  body x err", dx[b, :17])
        print("  feat err max per feature", dx[b, 17:].reshape(N, 5).max(axis=1))
