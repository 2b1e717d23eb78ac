"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): <= 1e-6 relative on state/covariance floats; integer
results (meas_result codes, feature counts) bit-exact.  We assert the much tighter TOL below,
measured against max|ref| per array, and also the element-wise 1e-6 relative bar on every
entry that is not numerically zero.
"""
import numpy as np
import pytest

import vi_ekf_amd as v
from oracle import oracle as orc
from vi_ekf_amd import scene
from tests.helpers import jac_fixture

pytestmark = pytest.mark.gpu

TOL = 1e-9          # relative to max|ref| of the array
ELEM_RTOL = 1e-6    # north_star bar, element-wise, on entries above the noise floor


def assert_close(got, ref, what):
    ref = np.asarray(ref)
    got = np.asarray(got)
    scale = max(np.abs(ref).max(), 1e-300)
    err = np.abs(got - ref).max()
    assert err <= TOL * scale, "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)
    big = np.abs(ref) > 1e-9 * scale
    rel = np.abs(got - ref)[big] / np.abs(ref)[big]
    assert rel.size == 0 or rel.max() <= ELEM_RTOL, "%s: max elementwise rel err %.3e" % (what, rel.max())


def oracle_params(p):
    return dict(x0=p["x0"], P0=p["P0"], Qx=p["Qx"], lam=p["lam"], Qu=p["Qu"], P0_feat=p["P0_feat"],
                Qx_feat=p["Qx_feat"], lam_feat=p["lam_feat"], cam_center=p["cam_center"], focal_len=p["focal_len"],
                q_b_c=p["q_b_c"], p_b_c=p["p_b_c"], q_b_u=p["q_b_u"], min_depth=p["min_depth"],
                use_drag_term=p["use_drag_term"], use_partial_update=p["use_partial_update"],
                use_keyframe_reset=p["use_keyframe_reset"])


def run_oracle(sc, B, N, steps, nfeat=None):
    fs = []
    nfeat = N if nfeat is None else nfeat
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(nfeat):
            f.init_feature(sc["pix"][b, i], i, float("nan"))
        fs.append(f)
    res = np.zeros((steps, B, sc["slot"].shape[1]), dtype=np.int32)
    for s in range(steps):
        for b in range(B):
            res[s, b] = fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], sc["z"][s, b][None], sc["slot"][b], sc["R"])[0]
    x = np.stack([f.x.copy() for f in fs])
    P = np.stack([f.P.copy() for f in fs])
    return x, P, res


def make_gpu(sc, B, N, nfeat=None, kernel=0):
    """kernel: 0 automatic, 1 streaming family, 2 on-chip families, 3 the tile family (P as MFMA accumulator tiles) whatever the
    batch size -- it is otherwise chosen only for batches beyond one workgroup per CU --, 4 on-chip without the tile family"""
    from vi_ekf_amd import capi
    g = v.BatchVIEKF(B, N, sc["params"])
    if kernel in (3, 5):        # 3: one filter per workgroup, 5: the paired form (two filters per workgroup); both opt-in, the
        g.set_tuning(capi.TUNE_TILES, 2 if kernel == 3 else 3)        # resident family is the default
        assert ("k_step_tiles_pair" if kernel == 5 else "k_step_tiles<") in g.describe(), g.describe()
    elif kernel == 4:
        g.set_tuning(capi.TUNE_TILES, 0)
        g.set_kernel(2)
    elif kernel:
        g.set_kernel(kernel)
    nfeat = N if nfeat is None else nfeat
    for i in range(nfeat):
        ok = g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        assert (ok == 1).all()
    return g


# kernel 1 = streaming family (P in HBM/L2), 2 = resident family (P in registers, N in 8..50)
@pytest.mark.parametrize("B,N,steps,kernel", [(3, 3, 6, 1), (4, 12, 5, 1), (2, 25, 3, 1), (2, 50, 2, 1),
                                              (3, 1, 4, 2), (3, 2, 4, 2), (4, 8, 4, 2), (4, 12, 5, 2), (3, 17, 3, 2), (2, 25, 3, 2),
                                              (2, 33, 2, 2), (2, 41, 2, 2), (2, 49, 2, 2), (3, 50, 2, 2),
                                              # the tile family (P as fp64-MFMA accumulator tiles; NT = 11 tiles per side: N = 46 .. 50)
                                              (3, 50, 3, 3), (2, 46, 2, 3), (2, 47, 2, 3), (2, 48, 1, 3), (2, 49, 2, 3),
                                              # ... its paired form (two filters per workgroup half a phase out of step; an odd batch
                                              # leaves the last workgroup one filter)
                                              (3, 50, 3, 5), (4, 46, 2, 5), (1, 47, 2, 5), (2, 48, 1, 5), (5, 49, 2, 5),
                                              # N + 14 > 64 lanes: the body lanes on a second service wave (<6,6>, two service waves)
                                              (3, 51, 2, 2), (2, 57, 2, 2), (2, 64, 2, 2), (2, 60, 2, 0), (2, 63, 2, 2), (2, 59, 1, 2),
                                              # more than 64 features: features 64.. on the body wave's lanes (<7,6>, N <= 72)
                                              (2, 65, 2, 2), (2, 70, 2, 2), (2, 72, 2, 0), (2, 69, 1, 2),
                                              # eight blocks per thread (<8,6>, N <= 77: the end of the on-chip family)
                                              (2, 73, 2, 2), (2, 77, 2, 0)])
def test_step_parity(B, N, steps, kernel):
    sc = scene.make_scene(B, N, steps, seed=100 + N)
    x_ref, P_ref, res_ref = run_oracle(sc, B, N, steps)
    g = make_gpu(sc, B, N, kernel=kernel)
    res = np.zeros_like(res_ref)
    for s in range(steps):
        res[s] = g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    assert (res == res_ref).all()
    assert (g.get_len_features() == N).all()
    assert_close(g.get_state(), x_ref, "x")
    assert_close(g.get_covariance(), P_ref, "P")
    assert (g.get_status() & 1 == 0).all()


@pytest.mark.parametrize("N,kernel", [(12, 2), (50, 2), (57, 2), (30, 1), (70, 0), (50, 3), (47, 3), (50, 5)])
def test_general_lambda_on_the_bearing_components(N, kernel):
    """lambda_feat with entries != 1 on the bearing components: the fused kernel's general-Lambda instances (the reference's
    parameter files keep those at 1, which the ZU instances exploit; vi_ekf_meas.cpp:250-257)"""
    B, steps = 3, 3
    over = dict(lam_feat=[0.7, 0.85, 0.4], lam=[1.0] * 3 + [0.9] * 3 + [0.8] * 3 + [0.1] * 6 + [0.01])
    sc = scene.make_scene(B, N, steps, seed=900 + N, params=over)
    x_ref, P_ref, res_ref = run_oracle(sc, B, N, steps)
    g = make_gpu(sc, B, N, kernel=kernel)
    res = np.zeros_like(res_ref)
    for s in range(steps):
        res[s] = g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    assert (res == res_ref).all()
    assert_close(g.get_state(), x_ref, "x")
    assert_close(g.get_covariance(), P_ref, "P")


def test_init_state_matches_oracle():
    sc = scene.make_scene(2, 5, 1, seed=7)
    g = make_gpu(sc, 2, 5, nfeat=3)
    fs = [orc.OracleFilter(5).init(**oracle_params(sc["params"])) for _ in range(2)]
    for b in range(2):
        for i in range(3):
            fs[b].init_feature(sc["pix"][b, i], i)
    assert (g.get_len_features() == 3).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x after init_feature")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P after init_feature")


@pytest.mark.parametrize("N,nfeat,kernel", [(6, 4, 0), (12, 7, 0), (26, 20, 0), (50, 33, 3), (48, 48, 3), (46, 1, 3), (50, 33, 5), (47, 0, 5)])
def test_propagate_only_partial_features_and_qx(N, nfeat, kernel):
    """inactive slots still receive Qx_feat (reference vi_ekf.cpp:139-144,304); drag term off; Qx != 0"""
    B = 3
    over = dict(Qx=[1e-4] * 16, Qx_feat=[1e-5, 2e-5, 3e-5], use_drag_term=0)
    sc = scene.make_scene(B, N, 4, seed=11, params=over)
    g = make_gpu(sc, B, N, nfeat=nfeat, kernel=kernel)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(nfeat):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    for s in range(4):
        g.propagate(sc["u"][s], sc["dt"])
        for b in range(B):
            fs[b].propagate(sc["u"][s, b], sc["dt"][b])
    assert_close(g.get_state()[:, :17 + 5 * nfeat], np.stack([f.x for f in fs])[:, :17 + 5 * nfeat], "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.parametrize("N,kernel", [(4, 0), (9, 0), (9, 1), (20, 1), (20, 2), (53, 2), (70, 2), (76, 2), (100, 0), (50, 3), (46, 3), (50, 5), (46, 5)])
def test_update_gating_nan_invalid_and_full_update(N, kernel):
    """result codes: gated outlier, NaN pixel, out-of-range slot, skipped; Joseph-form (non-partial) update; both kernel
    families (the grouped streaming update has its own gate / skip paths inside a group)"""
    B = 4
    sc = scene.make_scene(B, N, 1, seed=21, params=dict(use_partial_update=0))
    g = make_gpu(sc, B, N, kernel=kernel)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    g.propagate(sc["u"][0], sc["dt"])
    for b in range(B):
        fs[b].propagate(sc["u"][0, b], sc["dt"][b])
    z = sc["z"][0].copy()
    slot = sc["slot"].copy()
    z[0, 1] += 5000.0          # outlier -> gated
    z[1, 2, 0] = np.nan        # NaN
    slot[2, 0] = -1            # skipped
    slot[3, 3] = N + 2         # invalid slot
    res = g.update_feat(z, slot, sc["R"])
    exp = np.zeros((B, N), dtype=np.int32)
    for b in range(B):
        for m in range(N):
            sl = slot[b, m]
            if sl < 0:
                exp[b, m] = -1
            elif sl >= N:
                exp[b, m] = 3
            elif np.isnan(z[b, m]).any():
                exp[b, m] = 2
            else:
                exp[b, m] = fs[b].update(orc.FEAT, z[b, m], sc["R"], True, sl)
    assert exp[0, 1] == 1
    assert (res == exp).all(), (res, exp)
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.parametrize("N,kernel", [(6, 1), (6, 2), (24, 1), (24, 2), (56, 2), (68, 2), (90, 0), (50, 3), (47, 3), (50, 5)])
def test_fix_depth_inside_the_updates(N, kernel):
    """fix_depth after an UPDATE (vi_ekf_meas.cpp:271): features that start just in front of the camera's infinity with a
    large depth variance correlated with the bearing are pushed to rho < 0 by noisy pixels -- the reset, the P(rho,rho) edit and the flag, in the middle
    of a frame's updates (the grouped update tracks that diagonal separately)"""
    B = 12
    sc = scene.make_scene(B, N, 1, seed=77 + N)
    g = make_gpu(sc, B, N, kernel=kernel)
    x = g.get_state()
    P = g.get_covariance()
    for f in range(0, N, 2):
        d = 16 + 3 * f
        x[:, 17 + 5 * f + 4] = 2e-3
        P[:, d + 2, d + 2] = 4.0
        P[:, d + 2, d] = P[:, d, d + 2] = 0.1          # depth / bearing correlation: the pixel residual moves rho
        P[:, d + 2, d + 1] = P[:, d + 1, d + 2] = -0.1
    g.set_state(x=x, P=P)
    rng = np.random.default_rng(5)
    z = sc["z"][0] + rng.normal(0.0, 1.5, sc["z"][0].shape)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        f.x[:] = x[b]
        f.P[:] = P[b]
        fs.append(f)
    res = g.step(sc["u"][0], sc["dt"], z, sc["slot"], sc["R"])
    hit = 0
    for b in range(B):
        ref = fs[b].run_steps(sc["u"][0, b][None], sc["dt"][b], z[b][None], sc["slot"][b], sc["R"])[0]
        assert (res[b] == ref).all()
    st = g.get_status()
    hit = int(((st & 4) != 0).sum())
    assert hit > 0, "no filter took the negative-depth branch: the test does not test what it says"
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.parametrize("N", [3, 10])
def test_fix_depth_branches(N):
    """force rho < 0 and rho > 1e2 (reference vi_ekf_helper.cpp:128-156): a rare branch needs its own test"""
    B = 2
    sc = scene.make_scene(B, N, 1, seed=31)
    g = make_gpu(sc, B, N)
    x = g.get_state()
    x[0, 17 + 4] = -0.3
    x[1, 17 + 5 + 4] = 250.0
    g.set_state(x=x)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        f.x[:] = x[b]
        fs.append(f)
    g.propagate(sc["u"][0], sc["dt"])
    for b in range(B):
        fs[b].propagate(sc["u"][0, b], sc["dt"][b])
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")
    assert g.get_status()[0] & 4


def test_jac_fixture_state_parity():
    """random large-state fixture of the reference's jac_test (test/jac_test.cpp:118-170), one step"""
    N = 6
    p, pix, dep, u = jac_fixture(N, 4242)
    f = orc.OracleFilter(N).init(**p)
    for i in range(N):
        f.init_feature(pix[i], i, dep[i])
    gp = dict(x0=p["x0"], P0=p["P0"], Qx=p["Qx"], lam=p["lam"], Qu=p["Qu"], P0_feat=p["P0_feat"], Qx_feat=p["Qx_feat"],
              lam_feat=p["lam_feat"], cam_center=p["cam_center"], focal_len=p["focal_len"], q_b_c=p["q_b_c"],
              p_b_c=p["p_b_c"], q_b_u=p["q_b_u"], min_depth=p["min_depth"], use_drag_term=1, use_partial_update=1,
              use_keyframe_reset=1)
    g = v.BatchVIEKF(1, N, gp)
    for i in range(N):
        g.init_feature(pix[i][None], np.array([dep[i]]))
    assert_close(g.get_state(), f.x[None], "x0")
    g.propagate(u[None], np.array([0.004]))
    f.propagate(u, 0.004)
    assert_close(g.get_state(), f.x[None], "x")
    assert_close(g.get_covariance(), f.P[None], "P")


def test_device_pointer_path_matches_host_path():
    torch = pytest.importorskip("torch")
    B, N, steps = 8, 12, 3
    sc = scene.make_scene(B, N, steps, seed=5)
    g1 = make_gpu(sc, B, N)
    g2 = make_gpu(sc, B, N)
    g2.use_torch_stream()
    dev = torch.device("cuda:0")
    R = torch.tensor(sc["R"], device=dev)
    dt = torch.tensor(sc["dt"], device=dev)
    slot = torch.tensor(sc["slot"], device=dev)
    for s in range(steps):
        r1 = g1.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        r2 = g2.step(torch.tensor(sc["u"][s], device=dev), dt, torch.tensor(sc["z"][s], device=dev), slot, R)
        torch.cuda.synchronize()
        assert (r2.cpu().numpy() == r1).all()
    assert np.array_equal(g1.get_state(), g2.get_state())
    assert np.array_equal(g1.get_covariance(), g2.get_covariance())


@pytest.mark.parametrize("N,M,prop", [(64, 4, 1), (150, 3, 1), (51, 51, 1), (150, 40, 1), (160, 35, 1), (160, 160, 1), (150, 40, 2), (64, 4, 2),
                                      (150, 40, -1), (160, 160, -1), (51, 51, -1)])
def test_wide_p_streaming_family(N, M, prop):
    """wide covariance (BASELINE config 5: N=150, n=466): the streaming family -- a few updates per step, and enough of them
    for full 16-measurement groups plus a partial one (N=160 is the ABI limit: whole 48-row super-tiles in the propagate,
    N=150 and N=51 leave partial ones).  prop = 1: the propagate in the K = 24 record form (k_propagate_wide, the default), 2: r02's
    form with the operand sets in global scratch (VIEKF_TUNE_STREAM_MFMA = 2, kept for A/B runs)"""
    B, steps = 2, 2
    sc = scene.make_scene(B, N, steps, seed=300 + N)
    z = np.ascontiguousarray(sc["z"][:, :, :M, :])
    slot = np.ascontiguousarray(sc["slot"][:, :M])
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    g = make_gpu(sc, B, N, kernel=1)     # (N <= 77 would otherwise run on chip)
    from vi_ekf_amd import capi
    if prop < 0:   # r03's grouped update, the measurement chain on every thread (VIEKF_TUNE_PANEL_SERVICE = 0, kept for A/B runs)
        g.set_tuning(capi.TUNE_PANEL_SERVICE, 0)
        prop = 1
        assert "k_update_feat_blocked" in g.describe(), g.describe()
    else:   # (the look-ahead kernel where its LDS layout fits: N <= 154)
        assert ("k_update_feat_panelsvc" if N <= 154 else "k_update_feat_blocked") in g.describe(), g.describe()
    g.set_tuning(capi.TUNE_STREAM_MFMA, prop)
    assert ("k_propagate_wide" in g.describe()) == (prop == 1), g.describe()
    for s in range(steps):
        res = g.step(sc["u"][s], sc["dt"], z[s], slot, sc["R"])
        for b in range(B):
            ref = fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], z[s, b][None], slot[b], sc["R"])[0]
            assert (res[b] == ref).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.gpu
@pytest.mark.parametrize("N,nfeat", [(100, 70), (150, 101), (150, 37)])
def test_wide_p_partially_filled(N, nfeat):
    """fewer tracked features than slots at wide P (len_features < NUM_FEATURES, include/vi_ekf.h:205-214: everything past the active
    16 + 3 len block stays as it was set up): the look-ahead kernel's pass runs the ragged last tile row without clamps or predicates on padded
    columns -- its rows past the active block must come back as they were.  x and the WHOLE n x n covariance against the oracle"""
    B, steps = 2, 2
    sc = scene.make_scene(B, N, steps, seed=700 + N + nfeat)
    keep = [np.flatnonzero(sc["slot"][b] < nfeat) for b in range(B)]
    M = min(len(k) for k in keep)
    slot = np.stack([sc["slot"][b][keep[b][:M]] for b in range(B)]).astype(np.int32)
    z = np.stack([np.stack([sc["z"][s][b][keep[b][:M]] for b in range(B)]) for s in range(steps)])
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(nfeat):
            f.init_feature(sc["pix"][b, i], i)
        fs.append(f)
    g = make_gpu(sc, B, N, nfeat=nfeat, kernel=1)
    assert "k_update_feat_panelsvc" in g.describe(), g.describe()
    for s in range(steps):
        res = g.step(sc["u"][s], sc["dt"], z[s], slot, sc["R"])
        for b in range(B):
            ref = fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], z[s, b][None], slot[b], sc["R"])[0]
            assert (res[b] == ref).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    P, Pr = g.get_covariance(), np.stack([f.P for f in fs])
    assert_close(P, Pr, "P")
    na = 16 + 3 * nfeat
    assert (P[:, na:, :] == Pr[:, na:, :]).all() and (P[:, :, na:] == Pr[:, :, na:]).all()    # (untouched: bit for bit)


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,kernel", [(12, 70, 2), (12, 70, 1), (50, 130, 2), (66, 90, 2), (50, 130, 3), (50, 130, 5)])
def test_more_measurements_than_one_launch_holds(N, M, kernel):
    """M > 64 measurements per step (the fused kernel takes 64 per launch: vi_ekf_amd chunks, P makes one extra HBM round
    trip per chunk), with repeated and skipped (-1) slots, against the oracle applying them one by one"""
    B = 2
    sc = scene.make_scene(B, N, 1, seed=300 + N)
    rng = np.random.default_rng(N + M)
    slot = rng.integers(-1, N, size=(B, M)).astype(np.int32)
    z = np.zeros((B, M, 2))
    for b in range(B):
        for k in range(M):
            f = max(int(slot[b, k]), 0)
            z[b, k] = sc["z"][0][b, list(sc["slot"][b]).index(f)] + rng.normal(0, 0.3, 2)
    g = make_gpu(sc, B, N, kernel=kernel)
    res = g.step(sc["u"][0], sc["dt"], z, slot, sc["R"])
    fs = [orc.OracleFilter(N).init(**oracle_params(sc["params"])) for _ in range(B)]
    R = np.asarray(sc["R"]).reshape(2, 2)
    for b in range(B):
        for i in range(N):
            fs[b].init_feature(sc["pix"][b, i], i)
        fs[b].propagate(sc["u"][0][b], float(sc["dt"][b]))
        for k in range(M):
            if slot[b, k] >= 0:
                r = fs[b].update(orc.FEAT, z[b, k], R, True, int(slot[b, k]))
                assert res[b, k] == r
            else:
                assert res[b, k] == -1
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.parametrize("N,steps,kernel", [(12, 400, 1), (12, 400, 2), (50, 120, 2), (70, 60, 2), (50, 120, 3), (50, 120, 5)])
def test_long_run_stays_at_parity_and_symmetric(N, steps, kernel):
    """hundreds of steps (20 k updates at N=50..12): the covariance stays symmetric bit for bit and the distance to the oracle
    does not grow beyond the rounding level (the rank-2 update form amplifies any asymmetry of P -- DESIGN section 3 -- which a
    handful of steps does not show)"""
    B = 2
    sc = scene.make_scene(B, N, steps, seed=11 + N)
    g = make_gpu(sc, B, N, kernel=kernel)
    f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
    for i in range(N):
        f.init_feature(sc["pix"][0, i], i)
    for s in range(steps):
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        f.run_steps(sc["u"][s, 0][None], sc["dt"][0], sc["z"][s, 0][None], sc["slot"][0], sc["R"])
    x, P = g.get_state(), g.get_covariance()
    assert np.array_equal(P, P.transpose(0, 2, 1))
    assert np.abs(x[0] - f.x).max() <= 1e-8 * np.abs(f.x).max()
    assert np.abs(P[0] - f.P).max() <= 1e-8 * np.abs(f.P).max()


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("off", [60.0, 200.0, 420.0])
def test_large_corrections(kernel, off):
    """a wide prior and a far-off (but not gated) pixel: rotation-vector corrections of ~0.1, ~0.3 and ~0.7 rad, i.e. every
    branch of the small-angle / series / library quaternion exponential of the fused kernel, against the oracle"""
    B, N = 3, 6
    sc = scene.make_scene(B, N, 1, seed=41)
    g = make_gpu(sc, B, N, kernel=kernel)
    x, P = g.get_state(), g.get_covariance()
    for f in range(N):
        d = 16 + 3 * f
        P[:, d, d] = 0.2
        P[:, d + 1, d + 1] = 0.2
    for k in (6, 7, 8):
        P[:, k, k] = 0.3
    g.set_state(x=x, P=P)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i)
        f.x[:] = x[b]
        f.P[:] = P[b]
        fs.append(f)
    z = sc["z"][0].copy()
    z[:, 0, 0] += off
    z[:, 3, 1] -= 0.8 * off
    res = g.update_feat(z, sc["slot"], sc["R"])
    x0 = x.copy()
    for b in range(B):
        for m in range(N):
            r = fs[b].update(orc.FEAT, z[b, m], sc["R"], True, int(sc["slot"][b, m]))
            assert r == res[b, m]
    assert (res[:, 0] == 0).all()                      # (not gated: the prior is wide enough)
    moved = np.abs(np.stack([f.x for f in fs]) - x0).max()
    assert moved > 0.02 * off / 60.0                   # the correction really is large
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")


@pytest.mark.parametrize("N,f_bad", [(40, 7), (150, 20), (150, 3)])
def test_wide_p_nan_guard_inside_a_group(N, f_bad):
    """the NaN guard of an update (vi_ekf_meas.cpp:247) INSIDE a group of the look-ahead kernel: a NaN in the zeta column of one
    feature makes that feature's K NaN on one row -- its update is skipped (fix_depth still runs, :271), the others run.  The service
    wave cannot know (the verdict needs every row): the row phase finds out, nothing has been committed, the group is redone one
    measurement at a time.  Compared with r03's kernel, which decides measurement by measurement (same codes, same NaN pattern, same
    numbers to rounding); the dense oracle is no guide here -- a dense P H^T turns ANY NaN in a row of P into a NaN gain for every
    measurement (0 x NaN), the block-sparse product of both kernels only when the measured feature's own column holds it."""
    from vi_ekf_amd import capi
    B = 3
    sc = scene.make_scene(B, N, 1, seed=900 + N)
    out = []
    for svc in (1, 0):
        g = make_gpu(sc, B, N, kernel=1)
        g.set_tuning(capi.TUNE_PANEL_SERVICE, svc)
        assert ("k_update_feat_panelsvc" if svc else "k_update_feat_blocked") in g.describe(), g.describe()
        g.propagate(sc["u"][0], sc["dt"])
        P = g.get_covariance()
        c = 16 + 3 * f_bad
        P[1, 5, c] = P[1, c, 5] = np.nan                  # filter 1 only: body row 5 x the first bearing column of feature f_bad
        g.set_state(P=P)
        res = g.update_feat(sc["z"][0], sc["slot"], sc["R"])
        out.append((res.copy(), g.get_state(), g.get_covariance(), g.get_status()))
    (ra, xa, Pa, sa), (rb, xb, Pb, sb) = out
    assert (ra == rb).all() and (ra == 0).all()            # (a NaN-guarded update reports success, like the reference)
    assert (sa == sb).all()
    assert (np.isnan(Pa) == np.isnan(Pb)).all() and (np.isnan(xa) == np.isnan(xb)).all()
    assert np.isnan(Pa[1]).any() and not np.isnan(Pa[0]).any() and not np.isnan(Pa[2]).any()
    fin = ~np.isnan(Pb)
    assert np.abs(Pa[fin] - Pb[fin]).max() <= 1e-9 * np.abs(Pb[fin]).max()
    dx_ = np.abs(xa - xb)
    assert dx_.max() <= 1e-9 * np.abs(xb).max(), (np.argwhere(dx_ > 1e-10)[:12].tolist(), dx_.max())
