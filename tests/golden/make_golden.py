"""Generates tests/golden/*.npz from the CPU oracle (oracle/viekf_oracle.c).

The reference holds no golden vectors and cannot be built here ("parity unpinned"),
so these fixtures are produced by our own fp64 restatement; tests/test_oracle_vs_twin.py cross-checks that
restatement against an independent numpy one.  Run from the repo root:  python tests/golden/make_golden.py
Inputs come from vi_ekf_amd.scene (seeded numpy Generator), so the .npz files carry both inputs and outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from vi_ekf_amd import scene  # noqa: E402

KEYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c", "p_b_c",
        "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")


def make(N, B, steps, seed):
    sc = scene.make_scene(B, N, steps, seed=seed)
    p = sc["params"]
    out = dict(N=N, B=B, steps=steps, seed=seed, pix=sc["pix"], u=sc["u"], z=sc["z"], slot=sc["slot"], dt=sc["dt"],
               R=sc["R"])
    for k in KEYS:
        out["param_" + k] = np.asarray(p[k], dtype=np.float64)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**{k: p[k] for k in KEYS})
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    out["x_init"] = np.stack([f.x.copy() for f in fs])
    out["Pdiag_init"] = np.stack([np.diag(f.P).copy() for f in fs])
    # Jacobians of filter 0 at the initial state with the first input (A, G blocks as the reference's dynamics())
    xdot, A, G = fs[0].dynamics(fs[0].x.copy(), sc["u"][0, 0])
    out["dyn_xdot"], out["dyn_A"], out["dyn_G"] = xdot, A, G
    h, H = fs[0].h(orc.FEAT, None, 0)
    out["h_feat0"], out["H_feat0"] = h[:2], H[:2]
    # one propagate, then the frame's updates, snapshot after each stage of step 0
    for b in range(B):
        fs[b].propagate(sc["u"][0, b], sc["dt"][b])
    out["x_prop"] = np.stack([f.x.copy() for f in fs])
    out["P_prop"] = np.stack([f.P.copy() for f in fs])
    res = np.zeros((steps, B, N), dtype=np.int32)
    for b in range(B):
        for m in range(N):
            res[0, b, m] = fs[b].update(orc.FEAT, sc["z"][0, b, m], sc["R"], True, int(sc["slot"][b, m]))
    out["x_step1"] = np.stack([f.x.copy() for f in fs])
    out["P_step1"] = np.stack([f.P.copy() for f in fs])
    for s in range(1, steps):
        for b in range(B):
            res[s, b] = fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], sc["z"][s, b][None], sc["slot"][b], sc["R"])[0]
    out["x_final"] = np.stack([f.x.copy() for f in fs])
    out["P_final"] = np.stack([f.P.copy() for f in fs])
    out["results"] = res
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for N, B, steps, seed in [(3, 2, 4, 11), (12, 2, 3, 12), (50, 1, 2, 13)]:
        d = make(N, B, steps, seed)
        np.savez_compressed(os.path.join(here, "step_N%d.npz" % N), **d)
        print("wrote step_N%d.npz" % N, {k: np.asarray(v).shape for k, v in d.items() if k.startswith(("x_", "P_"))})
