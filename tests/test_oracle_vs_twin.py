"""The C oracle against the independent numpy restatement (oracle/np_twin.py) over multi-step runs."""
import numpy as np
import pytest

from oracle import np_twin, oracle as orc
from vi_ekf_amd import scene
from tests.helpers import jac_fixture, make_oracle, make_twin

KEYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c", "p_b_c",
        "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("N,steps,over", [(3, 8, {}), (7, 5, {}), (12, 3, dict(use_drag_term=0)),
                                          (5, 4, dict(use_partial_update=0, Qx=[1e-4] * 16, Qx_feat=[1e-5] * 3))])
def test_multistep_agreement(N, steps, over):
    sc = scene.make_scene(1, N, steps, seed=40 + N, params=over)
    p = {k: sc["params"][k] for k in KEYS}
    f = orc.OracleFilter(N).init(**p)
    t = np_twin.TwinFilter(N, **p)
    for i in range(N):
        f.init_feature(sc["pix"][0, i], i)
        t.init_feature(sc["pix"][0, i])
    assert rel(t.x, f.x) < 1e-14
    for s in range(steps):
        f.propagate(sc["u"][s, 0], sc["dt"][0])
        t.propagate(sc["u"][s, 0], sc["dt"][0])
        assert rel(t.P, f.P) < 1e-11
        for m in range(N):
            sl = int(sc["slot"][0, m])
            rf = f.update(orc.FEAT, sc["z"][s, 0, m], sc["R"], True, sl)
            rt = t.update_feat(sc["z"][s, 0, m], sc["R"], sl)
            assert rf == rt
    assert rel(t.x, f.x) < 1e-11
    assert rel(t.P, f.P) < 1e-11
    # P stays symmetric to rounding (the resident kernels mirror the body rows from the body columns)
    assert np.abs(f.P - f.P.T).max() <= 1e-12 * np.abs(f.P).max()


def test_jac_fixture_dynamics_and_h_agree():
    for seed in range(5):
        N = 4
        p, pix, dep, u = jac_fixture(N, 900 + seed)
        f, t = make_oracle(N, p, pix, dep), make_twin(N, p, pix, dep)
        xd, A, G = f.dynamics(f.x.copy(), u)
        xd2, A2, G2 = t.dynamics(f.x.copy(), u)
        assert rel(A2, A) < 1e-13 and rel(G2, G) < 1e-13 and rel(xd2, xd) < 1e-13
        for i in range(N):
            h, H = f.h(orc.FEAT, None, i)
            h2, H2 = t.h_feat(f.x.copy(), i)
            assert rel(h2, h[:2]) < 1e-13 and rel(H2, H[:2]) < 1e-12


def test_gate_and_fix_depth_agree():
    N = 3
    sc = scene.make_scene(1, N, 1, seed=77)
    p = {k: sc["params"][k] for k in KEYS}
    f = orc.OracleFilter(N).init(**p)
    t = np_twin.TwinFilter(N, **p)
    for i in range(N):
        f.init_feature(sc["pix"][0, i], i)
        t.init_feature(sc["pix"][0, i])
    f.x[17 + 4] = -0.2
    t.x[17 + 4] = -0.2
    f.propagate(sc["u"][0, 0], 0.004)
    t.propagate(sc["u"][0, 0], 0.004)
    assert rel(t.x, f.x) < 1e-12 and rel(t.P, f.P) < 1e-12
    z = sc["z"][0, 0, 0] + 4000.0
    assert f.update(orc.FEAT, z, sc["R"], True, int(sc["slot"][0, 0])) == orc.MEAS_GATED
    assert t.update_feat(z, sc["R"], int(sc["slot"][0, 0])) == 1
