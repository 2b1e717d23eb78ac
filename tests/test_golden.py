"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle must still reproduce them bit-for-bit-ish (1e-13: same code, same compiler flags).
GPU: the HIP path (through the C ABI) must match them within the parity tolerance.
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c", "p_b_c",
        "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")


def load(N):
    d = np.load(os.path.join(HERE, "golden", "step_N%d.npz" % N))
    p = {k: (d["param_" + k] if d["param_" + k].ndim else d["param_" + k].item()) for k in KEYS}
    return d, p


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.mark.parametrize("N", [3, 12, 50])
def test_oracle_reproduces_golden(N):
    d, p = load(N)
    B, steps = int(d["B"]), int(d["steps"])
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**p)
        for i in range(N):
            f.init_feature(d["pix"][b, i], i)
        fs.append(f)
    assert rel(np.stack([f.x for f in fs]), d["x_init"]) < 1e-13
    xdot, A, G = fs[0].dynamics(fs[0].x.copy(), d["u"][0, 0])
    assert rel(A, d["dyn_A"]) < 1e-13 and rel(G, d["dyn_G"]) < 1e-13 and rel(xdot, d["dyn_xdot"]) < 1e-13
    h, H = fs[0].h(orc.FEAT, None, 0)
    assert rel(h[:2], d["h_feat0"]) < 1e-13 and rel(H[:2], d["H_feat0"]) < 1e-13
    # integer bookkeeping of the Jacobian blocks (reference test/jac_test.cpp:62-78): non-zero pattern of A
    nz = np.argwhere(np.abs(d["dyn_A"]) > 0)
    for r, c in nz:
        if r < 16:
            assert c < 16
        else:
            i = (r - 16) // 3
            assert (3 <= c < 6) or (12 <= c < 15) or (16 + 3 * i <= c < 19 + 3 * i)
    for b in range(B):
        fs[b].propagate(d["u"][0, b], d["dt"][b])
    assert rel(np.stack([f.P for f in fs]), d["P_prop"]) < 1e-13
    res = np.zeros((steps, B, N), dtype=np.int32)
    for b in range(B):
        for m in range(N):
            res[0, b, m] = fs[b].update(orc.FEAT, d["z"][0, b, m], d["R"], True, int(d["slot"][b, m]))
    assert rel(np.stack([f.P for f in fs]), d["P_step1"]) < 1e-13
    for s in range(1, steps):
        for b in range(B):
            res[s, b] = fs[b].run_steps(d["u"][s, b][None], d["dt"][b], d["z"][s, b][None], d["slot"][b], d["R"])[0]
    assert (res == d["results"]).all()
    assert rel(np.stack([f.x for f in fs]), d["x_final"]) < 1e-13
    assert rel(np.stack([f.P for f in fs]), d["P_final"]) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("N,kernel", [(3, 0), (12, 1), (12, 2), (50, 1), (50, 2)])
def test_hip_matches_golden(N, kernel):
    import vi_ekf_amd as v
    from tests.test_gpu_parity import assert_close
    d, p = load(N)
    B, steps = int(d["B"]), int(d["steps"])
    g = v.BatchVIEKF(B, N, dict(p, keyframe_overlap_threshold=0.8, name="golden"))
    if kernel:
        g.set_kernel(kernel)
    for i in range(N):
        g.init_feature(d["pix"][:, i, :].copy(), np.full(B, np.nan))
    assert_close(g.get_state(), d["x_init"], "x_init")
    g.propagate(d["u"][0], d["dt"])
    assert_close(g.get_state(), d["x_prop"], "x_prop")
    assert_close(g.get_covariance(), d["P_prop"], "P_prop")
    res = np.zeros((steps, B, N), dtype=np.int32)
    res[0] = g.update_feat(d["z"][0], d["slot"], d["R"])
    assert_close(g.get_state(), d["x_step1"], "x_step1")
    assert_close(g.get_covariance(), d["P_step1"], "P_step1")
    for s in range(1, steps):
        res[s] = g.step(d["u"][s], d["dt"], d["z"][s], d["slot"], d["R"])
    assert (res == d["results"]).all()
    assert_close(g.get_state(), d["x_final"], "x_final")
    assert_close(g.get_covariance(), d["P_final"], "P_final")
