"""The "structured" CPU flavour (block-sparse propagate, rank-2 FEAT update -- the formulation the HIP
kernels use, second cpu_baseline figure of bench.py) against the dense
reference-order oracle: same result codes, x and P equal to rounding."""
import numpy as np
import pytest

from oracle import oracle as orc
from vi_ekf_amd import scene

KEYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c",
        "p_b_c", "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")


def _filters(sc, B, N, nfeat):
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**{k: sc["params"][k] for k in KEYS})
        for i in range(nfeat):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    return fs


@pytest.mark.parametrize("N,nfeat,over", [
    (3, 3, {}), (12, 12, {}), (12, 7, dict(Qx=[1e-4] * 16, Qx_feat=[1e-5, 2e-5, 3e-5], use_drag_term=0)),
    (25, 25, dict(use_partial_update=0)), (50, 50, {})])
def test_structured_equals_dense(N, nfeat, over):
    B, steps = 3, 4
    sc = scene.make_scene(B, N, steps, seed=900 + N, params=over)
    u = np.ascontiguousarray(sc["u"].transpose(1, 0, 2))
    z = np.ascontiguousarray(sc["z"].transpose(1, 0, 2, 3))
    fd, fs = _filters(sc, B, N, nfeat), _filters(sc, B, N, nfeat)
    rd = orc.run_steps_mt(fd, 2, u, float(sc["dt"][0]), z, sc["slot"], sc["R"])
    rs = orc.run_steps_mt(fs, 2, u, float(sc["dt"][0]), z, sc["slot"], sc["R"], structured=True)
    assert (rd == rs).all()
    for a, b in zip(fd, fs):
        assert np.abs(a.x - b.x).max() <= 1e-11 * np.abs(a.x).max()
        assert np.abs(a.P - b.P).max() <= 1e-11 * np.abs(a.P).max()


def test_structured_gates_like_dense():
    """a wild pixel is gated (vi_ekf_meas.cpp:235-239) by both flavours, and nothing changes"""
    N = 6
    sc = scene.make_scene(1, N, 1, seed=5)
    sc["z"][0, 0, 2] += 400.0
    u = np.ascontiguousarray(sc["u"].transpose(1, 0, 2))
    z = np.ascontiguousarray(sc["z"].transpose(1, 0, 2, 3))
    fd, fs = _filters(sc, 1, N, N), _filters(sc, 1, N, N)
    rd = orc.run_steps_mt(fd, 1, u, float(sc["dt"][0]), z, sc["slot"], sc["R"])
    rs = orc.run_steps_mt(fs, 1, u, float(sc["dt"][0]), z, sc["slot"], sc["R"], structured=True)
    assert rd[0, 0, 2] == orc.MEAS_GATED and (rd == rs).all()
    assert np.abs(fd[0].P - fs[0].P).max() <= 1e-11 * np.abs(fd[0].P).max()
