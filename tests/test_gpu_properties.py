"""The reference's five jac_test properties and the symbolic Jacobians against the DEVICE.

The hot kernels hold A, G, H and the reset Jacobian N only in structured form (blocks, low-rank factors); the C ABI's test hooks
-- viekf_batch_eval_jacobians / _eval_h_jacobian / _boxplus / _boxminus / _eval_reset_jacobian, the counterparts of the reference's
public dynamics(x,u,xdot,dfdx,dfdu), h_*(x,h,H,id), boxplus, boxminus and keyframe_reset(xm,xp,N) -- evaluate them with the SAME
device functions and hand them out dense.  Here:
  * tests/properties.py (test/jac_test.cpp:245-487 restated, fixed seeds, the reference's tolerances) runs on those outputs;
  * the device's A, G and the pixel model's H are compared at 1e-9 with tests/golden/jac_sym_N*.npz (sympy from the MODEL, confirmed
    in 120-digit arithmetic) and, for every measurement model and both drag settings, with the oracle's blocks;
  * the run-time drag switch (VIEKF::set_drag_term, include/vi_ekf.h:290) changes the propagate as the oracle's does.
"""
import ctypes as C

import numpy as np
import pytest

import vi_ekf_amd as v
from oracle import oracle as orc
from tests import properties as prop
from tests.helpers import jac_fixture, make_oracle
from tests.test_gpu_parity import assert_close
from tests.test_jacobians_symbolic import load as load_sym, check as check_sym
from vi_ekf_amd import capi

pytestmark = pytest.mark.gpu


def _p(a):
    return C.c_void_p(a.ctypes.data)


def gpu_params(p):
    gp = {k: p[k] for k in ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c", "p_b_c",
                            "q_b_u", "min_depth")}
    gp.update(use_drag_term=int(bool(p["use_drag_term"])), use_partial_update=int(bool(p["use_partial_update"])),
              use_keyframe_reset=int(bool(p["use_keyframe_reset"])))
    return gp


class DeviceAdapter:
    """the filter interface of tests/properties.py over the C ABI's test hooks (a batch of one)"""

    def __init__(self, N, params, pix, depth):
        self.g = v.BatchVIEKF(1, N, gpu_params(params))
        for i in range(len(pix)):
            ok = self.g.init_feature(np.asarray(pix[i], dtype=float)[None], np.array([depth[i] if depth is not None else np.nan]))
            assert ok[0] == 1
        self.N, self.n, self.nx = N, 16 + 3 * N, 17 + 5 * N
        self.x = self.g.get_state()[0].copy()
        self.len_features = int(self.g.get_len_features()[0])
        self.L = capi.lib()

    def boxplus(self, x, dx):
        x = np.ascontiguousarray(x, dtype=float)[None]; dx = np.ascontiguousarray(dx, dtype=float)[None]
        out = np.empty((1, self.nx))
        capi.check(self.L.viekf_batch_boxplus(self.g._h, _p(x), _p(dx), _p(out), capi.HOST))
        return out[0]

    def boxminus(self, x1, x2):
        x1 = np.ascontiguousarray(x1, dtype=float)[None]; x2 = np.ascontiguousarray(x2, dtype=float)[None]
        out = np.empty((1, self.n))
        capi.check(self.L.viekf_batch_boxminus(self.g._h, _p(x1), _p(x2), _p(out), capi.HOST))
        return out[0]

    def dynamics(self, x, u):
        x = np.ascontiguousarray(x, dtype=float)[None]; u = np.ascontiguousarray(u, dtype=float)[None]
        xd, A, G = np.empty((1, self.n)), np.empty((1, self.n, self.n)), np.empty((1, 6, self.n))
        capi.check(self.L.viekf_batch_eval_jacobians(self.g._h, _p(x), _p(u), _p(xd), _p(A), _p(G), capi.HOST))
        return xd[0], A[0].T.copy(), G[0].T.copy()          # column-major on the wire -> [row, col]

    def h(self, mtype, x, id):
        x = np.ascontiguousarray(x, dtype=float)[None]
        slot = np.array([id], dtype=np.int32)
        zh, H = np.empty((1, 4)), np.empty((1, self.n, 3))
        capi.check(self.L.viekf_batch_eval_h_jacobian(self.g._h, _p(x), int(mtype), _p(slot), _p(zh), _p(H), capi.HOST))
        return zh[0], H[0].T.copy()                          # [3, n]

    def set_drag_term(self, on):
        capi.check(self.L.viekf_batch_set_drag_term(self.g._h, 1 if on else 0))
        got = C.c_int32(-1)
        capi.check(self.L.viekf_batch_get_drag_term(self.g._h, C.byref(got)))
        assert got.value == (1 if on else 0)

    def reset_map(self, xm):
        xm = np.ascontiguousarray(xm, dtype=float)[None]
        xp, Nm = np.empty((1, self.nx)), np.empty((1, self.n, self.n))
        capi.check(self.L.viekf_batch_eval_reset_jacobian(self.g._h, _p(xm), _p(xp), _p(Nm), capi.HOST))
        return xp[0], Nm[0].T.copy()


@pytest.mark.parametrize("N", prop.NS)
def test_manifold_on_the_device(N):
    prop.check_manifold(DeviceAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_dfdx_on_the_device(N):
    prop.check_dfdx(DeviceAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_dfdu_on_the_device(N):
    prop.check_dfdu(DeviceAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_h_on_the_device(N):
    prop.check_h(DeviceAdapter, N)


def test_kf_reset_on_the_device():
    prop.check_kf_reset(DeviceAdapter, 3)


@pytest.mark.parametrize("N", [3, 2])
def test_device_jacobians_equal_the_symbolic_derivation(N):
    """A, G (whole matrices: what the reference leaves zero is zero) and the pixel model's H from the DEVICE against sympy's, 1e-9"""
    z, params = load_sym(N)
    d = DeviceAdapter(N, params, z["pix"], z["depth"])
    assert_close(d.x, z["x"], "evaluation point")
    _, A, G = d.dynamics(z["x"], z["u"])
    check_sym(A, z["A"], "device A")
    check_sym(G, z["G"], "device G")
    for i in range(N):
        _, H = d.h(orc.FEAT, z["x"], i)
        check_sym(H[0:2, :], z["H"][i], "device H_feat[%d]" % i)
        assert (H[2] == 0).all()


@pytest.mark.parametrize("drag", [True, False])
def test_device_hooks_equal_the_oracle_on_every_model(drag):
    """xdot, A, G, every h / H, boxplus, boxminus and the reset map: device vs oracle at the parity tolerance, both drag settings"""
    N = 7
    for seed in (11, 12, 13):
        p, pix, dep, u = jac_fixture(N, seed)
        p["use_drag_term"] = drag
        f = make_oracle(N, p, pix, dep)
        d = DeviceAdapter(N, p, pix, dep)
        assert_close(d.x, f.x, "x after init_feature")
        xd, A, G = d.dynamics(f.x, u)
        xr, Ar, Gr = f.dynamics(f.x, u)
        assert_close(xd, xr, "xdot"); assert_close(A, Ar, "A"); assert_close(G, Gr, "G")
        for mtype, ids in ((orc.ACC, [0]), (orc.ALT, [0]), (orc.ATT, [0]), (orc.POS, [0]), (orc.VEL, [0]), (orc.QZETA, range(N)),
                           (orc.FEAT, range(N)), (orc.DEPTH, range(N)), (orc.INV_DEPTH, range(N))):
            for i in ids:
                zh, H = d.h(mtype, f.x, i)
                zr, Hr = f.h(mtype, f.x, i)
                k = len(zr)
                assert_close(zh[:k], zr, "zhat type %d" % mtype)
                assert_close(H, Hr[:3], "H type %d" % mtype)
        r = np.random.default_rng(seed)
        dx = r.uniform(-0.5, 0.5, f.n)
        xb = f.boxplus(f.x, dx)
        assert_close(d.boxplus(f.x, dx), xb, "boxplus")
        assert_close(d.boxminus(xb, f.x), f.boxminus(xb, f.x), "boxminus")
        g = f.clone(); g.keyframe_reset()
        xp, Nm = d.reset_map(f.x)
        assert_close(xp, g.x, "reset state"); assert_close(Nm, g.A, "reset Jacobian")


@pytest.mark.parametrize("N,kernel", [(6, 1), (12, 0), (50, 0), (50, 3), (50, 5)])
def test_drag_term_switch_at_run_time(N, kernel):
    """vi_ekf_ros starts with the drag term OFF and turns it ON after take-off (src/vi_ekf_ros.cpp:86,428-429): steps before and
    after the switch against the oracle doing the same"""
    from tests.helpers import apply_kernel
    from tests.test_gpu_parity import oracle_params
    from vi_ekf_amd import scene
    B, steps = 3, 4
    sc = scene.make_scene(B, N, steps, seed=60 + N, params=dict(use_drag_term=0))
    g = apply_kernel(v.BatchVIEKF(B, N, sc["params"]), kernel)
    fs = []
    for b in range(B):
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i, float("nan"))
        fs.append(f)
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    L = capi.lib()
    for s in range(steps):
        if s == 2:
            capi.check(L.viekf_batch_set_drag_term(g._h, 1))
            for f in fs:
                f.set_drag_term(True)
        res = g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        for b in range(B):
            ref = fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], sc["z"][s, b][None], sc["slot"][b], sc["R"])[0]
            assert (res[b] == ref).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P")
    # ... and the switch matters: without it the result differs
    assert np.abs(fs[0].x[3:6]).max() > 0
