"""Shared fixtures for the parity / property tests (test infrastructure)."""
import numpy as np

from oracle import np_twin, oracle as orc


def jac_fixture(N, seed):
    """Fixed-seed restatement of init_jacobians_test (reference test/jac_test.cpp:118-170).

    Returns (params dict, pixels [N,2], depths [N], u0 [6]).  The reference draws from
    rand()/Eigen::Random seeded by the wall clock; we draw the same distributions from a
    seeded numpy Generator.
    """
    r = np.random.default_rng(seed)
    U = lambda k: r.uniform(-1.0, 1.0, k)  # Eigen ::Random() is U[-1,1]
    x0 = np.zeros(17)
    x0[orc.xATT] = 1.0
    x0[orc.xMU] = 0.2
    x0[0:3] += U(3) * 100.0
    x0[3:6] += U(3) * 10.0
    x0[6:10] = orc.q_boxplus(x0[6:10], U(3))
    x0[10:13] += U(3) * 1.0
    x0[13:16] += U(3) * 0.5
    x0[16] += r.uniform(0, 1) * 0.05
    cam_center = np.array([320 - 25 + r.integers(0, 50), 240 - 25 + r.integers(0, 50)], dtype=float)
    focal_len = np.array([250 + r.uniform(0, 1) * 50, 250 + r.uniform(0, 1) * 50])
    qr = U(4)
    qr /= np.linalg.norm(qr)
    q_b_c = np.array([0.5, 0.5, 0.5, 0.5]) + qr  # NOT re-normalised, as in the reference (:146)
    p_b_c = U(3) * 0.5
    params = dict(x0=x0, P0=np.ones(16), Qx=np.ones(16), lam=np.ones(16), Qu=np.ones(6), P0_feat=np.ones(3),
                  Qx_feat=np.ones(3), lam_feat=np.ones(3), cam_center=cam_center, focal_len=focal_len,
                  q_b_c=q_b_c, p_b_c=p_b_c, q_b_u=np.array([1.0, 0, 0, 0]), min_depth=2.0, use_drag_term=True,
                  use_partial_update=True, use_keyframe_reset=True)
    pix = np.stack([r.integers(0, 640, N), r.integers(0, 480, N)], axis=1).astype(float)
    depth = 1.0 + r.uniform(0, 1, N) * 20.0
    u0 = np.concatenate([U(3), U(3)])
    return params, pix, depth, u0


def make_oracle(N, params, pix=None, depth=None):
    f = orc.OracleFilter(N).init(**params)
    if pix is not None:
        for i in range(len(pix)):
            f.init_feature(pix[i], i, depth[i] if depth is not None else float("nan"))
    return f


def make_twin(N, params, pix=None, depth=None):
    p = dict(params)
    t = np_twin.TwinFilter(N, **p)
    if pix is not None:
        for i in range(len(pix)):
            t.init_feature(pix[i], depth[i] if depth is not None else np.nan)
    return t


def apply_kernel(g, kernel):
    """kernel selection shared by the GPU tests: 0 automatic, 1 streaming family, 2 on-chip families, 3 and 5 the tile family (single, paired;
    opt-in), 4 on-chip without the tile family"""
    from vi_ekf_amd import capi
    if kernel in (3, 5):        # 3: one filter per workgroup, 5: the paired form (two filters per workgroup); both opt-in, the
        g.set_tuning(capi.TUNE_TILES, 2 if kernel == 3 else 3)        # resident family is the default
        assert ("k_step_tiles_pair" if kernel == 5 else "k_step_tiles<") in g.describe(), g.describe()
    elif kernel == 4:
        g.set_tuning(capi.TUNE_TILES, 0)
        g.set_kernel(2)
    elif kernel:
        g.set_kernel(kernel)
    return g
