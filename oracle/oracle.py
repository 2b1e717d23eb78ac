"""ctypes binding of the CPU ORACLE (oracle/viekf_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package (vi_ekf_amd).

Parity status: "parity unpinned" (see oracle/viekf_oracle.h): the reference has
no golden vectors and is unbuildable here; this restatement is pinned by the
restated jac_test properties, the numpy twin (oracle/np_twin.py) and the
committed fixtures under tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libviekf_oracle.so")

# enums (include/vi_ekf.h:87-138 of the reference)
xPOS, xVEL, xATT, xB_A, xB_G, xMU, xZ = 0, 3, 6, 10, 13, 16, 17
uA, uG = 0, 3
dxPOS, dxVEL, dxATT, dxB_A, dxB_G, dxMU, dxZ = 0, 3, 6, 9, 12, 15, 16
ACC, ALT, ATT, POS, VEL, QZETA, FEAT, PIXEL_VEL, DEPTH, INV_DEPTH, TOTAL_MEAS = range(11)
MEAS_SUCCESS, MEAS_GATED, MEAS_NAN, MEAS_INVALID, MEAS_NEW_FEATURE = range(5)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _VoFilter(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("nx", C.c_int), ("n", C.c_int),
        ("len_features", C.c_int), ("next_feature_id", C.c_int),
        ("feature_ids", _ip),
        ("x", _dp), ("P", _dp), ("Qx", _dp),
        ("Qu", C.c_double * 36),
        ("lam", _dp), ("Lambda", _dp),
        ("P0_feat", C.c_double * 9),
        ("use_drag_term", C.c_int), ("use_partial_update", C.c_int), ("use_keyframe_reset", C.c_int),
        ("min_depth", C.c_double),
        ("cam_center", C.c_double * 2),
        ("cam_F", C.c_double * 6),
        ("q_b_c", C.c_double * 4), ("p_b_c", C.c_double * 3), ("q_b_u", C.c_double * 4),
        ("A", _dp), ("G", _dp), ("dx", _dp), ("K", _dp), ("H", _dp), ("xp", _dp),
        ("zhat", C.c_double * 4),
        ("T1", _dp), ("T2", _dp), ("T3", _dp),
    ]


_fp = C.POINTER(_VoFilter)
_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("viekf_oracle.c", "viekf_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.vo_create.restype = _fp
        L.vo_create.argtypes = [C.c_int]
        L.vo_clone.restype = _fp
        L.vo_clone.argtypes = [_fp]
        L.vo_destroy.argtypes = [_fp]
        L.vo_init.argtypes = [_fp] + [_dp] * 13 + [C.c_double, C.c_int, C.c_int, C.c_int]
        L.vo_boxplus.argtypes = [_fp, _dp, _dp, _dp]
        L.vo_boxminus.argtypes = [_fp, _dp, _dp, _dp]
        L.vo_dynamics.argtypes = [_fp, _dp, _dp, C.c_int, C.c_int]
        L.vo_propagate.argtypes = [_fp, _dp, C.c_double]
        L.vo_fix_depth.argtypes = [_fp]
        L.vo_init_feature.argtypes = [_fp, _dp, C.c_int, C.c_double]
        L.vo_init_feature.restype = C.c_int
        L.vo_clear_feature.argtypes = [_fp, C.c_int]
        L.vo_h.argtypes = [_fp, C.c_int, _dp, _dp, _dp, C.c_int]
        L.vo_update.argtypes = [_fp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int]
        L.vo_update.restype = C.c_int
        L.vo_global_to_local_feature_id.argtypes = [_fp, C.c_int]
        L.vo_global_to_local_feature_id.restype = C.c_int
        L.vo_keyframe_reset.argtypes = [_fp]
        L.vo_keyframe_reset_edge.argtypes = [_fp, _dp]
        for nm in ("vo_nans_in_the_house", "vo_blowing_up", "vo_negative_depth"):
            getattr(L, nm).argtypes = [_fp]
            getattr(L, nm).restype = C.c_int
        L.vo_run_steps.argtypes = [_fp, C.c_int, _dp, C.c_double, _dp, _ip, C.c_int, _dp, _ip]
        L.vo_run_steps_mt.argtypes = [C.POINTER(_fp), C.c_int, C.c_int, C.c_int, _dp, C.c_double, _dp, _ip,
                                      C.c_int, _dp, _ip]
        L.vo_run_steps_mt_flavour.argtypes = L.vo_run_steps_mt.argtypes + [C.c_int]
        for nm, na in (("vo_q_otimes", 3), ("vo_q_exp", 2), ("vo_q_log", 2), ("vo_q_boxplus", 3),
                       ("vo_q_boxminus", 3), ("vo_q_R", 2), ("vo_q_rota", 3), ("vo_q_rotp", 3),
                       ("vo_q_from_two_unit_vectors", 3), ("vo_T_zeta", 2), ("vo_q_feat_boxplus", 3),
                       ("vo_q_feat_boxminus", 3)):
            getattr(L, nm).argtypes = [_dp] * na
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _vec(a, n=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


# ---- free quaternion helpers -------------------------------------------------
def _q_call(name, out_n, *ins):
    ins = [_vec(a) for a in ins]
    out = np.zeros(out_n)
    getattr(lib(), name)(*[_d(a) for a in ins], _d(out))
    return out


def q_otimes(a, b): return _q_call("vo_q_otimes", 4, a, b)
def q_exp(v): return _q_call("vo_q_exp", 4, v)
def q_log(q): return _q_call("vo_q_log", 3, q)
def q_boxplus(q, d): return _q_call("vo_q_boxplus", 4, q, d)
def q_boxminus(q1, q2): return _q_call("vo_q_boxminus", 3, q1, q2)
def q_R(q): return _q_call("vo_q_R", 9, q).reshape(3, 3, order="F")
def q_rota(q, v): return _q_call("vo_q_rota", 3, q, v)
def q_rotp(q, v): return _q_call("vo_q_rotp", 3, q, v)
def q_from_two_unit_vectors(u, v): return _q_call("vo_q_from_two_unit_vectors", 4, u, v)
def T_zeta(q): return _q_call("vo_T_zeta", 6, q).reshape(3, 2, order="F")
def q_feat_boxplus(q, dq): return _q_call("vo_q_feat_boxplus", 4, q, dq)
def q_feat_boxminus(qj, qi): return _q_call("vo_q_feat_boxminus", 2, qj, qi)


# reference params/ekf.yaml values (the bench/parity scene uses these)
EKF_YAML = dict(
    x0=[0, 0, -2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.1],
    P0=[1e-3] * 9 + [2e-1] * 3 + [1e-1] * 3 + [1e-6],
    Qx=[0.0] * 16,
    lam=[1.0] * 9 + [0.1] * 6 + [0.01],
    Qu=[1.0, 1.0, 1.0, 4e-4, 4e-4, 4e-4],
    P0_feat=[0.01, 0.01, 0.3],
    Qx_feat=[0.0, 0.0, 0.0],
    lam_feat=[1.0, 1.0, 0.4],
    cam_center=[315.83184814453125, 242.1165771484375],
    focal_len=[611.1864013671875, 611.5557861328125],
    q_b_c=[0.9974335273839892, 0.019768487288642146, -0.03564306227555538, 0.05886541830542371],
    p_b_c=[0.17363129, -0.02205945, 0.05490228],
    q_b_u=[0.993760669165504, 0.0497294816014604, 0.0997086508721388, 0.00498959122946198],
    min_depth=1.5, use_drag_term=True, use_partial_update=True, use_keyframe_reset=True,
)


class OracleFilter:
    """One reference-order dense fp64 filter (mirror of vi_ekf::VIEKF's numeric core)."""

    def __init__(self, num_features, _ptr=None):
        self._L = lib()
        self._p = _ptr if _ptr is not None else self._L.vo_create(int(num_features))
        s = self._p.contents
        self.N, self.nx, self.n = s.N, s.nx, s.n

    def __del__(self):
        try:
            if self._p:
                self._L.vo_destroy(self._p)
                self._p = None
        except Exception:
            pass

    def clone(self):
        return OracleFilter(self.N, _ptr=self._L.vo_clone(self._p))

    def init(self, x0, P0, Qx, lam, Qu, P0_feat, Qx_feat, lam_feat, cam_center, focal_len, q_b_c, p_b_c,
             q_b_u=(1.0, 0.0, 0.0, 0.0), min_depth=1.5, use_drag_term=True, use_partial_update=True,
             use_keyframe_reset=True):
        args = [_vec(x0, 17), _vec(P0, 16), _vec(Qx, 16), _vec(lam, 16), _vec(Qu, 6), _vec(P0_feat, 3),
                _vec(Qx_feat, 3), _vec(lam_feat, 3), _vec(cam_center, 2), _vec(focal_len, 2), _vec(q_b_c, 4),
                _vec(p_b_c, 3), _vec(q_b_u, 4)]
        self._L.vo_init(self._p, *[_d(a) for a in args], float(min_depth), int(use_drag_term),
                        int(use_partial_update), int(use_keyframe_reset))
        return self

    def init_from_yaml_defaults(self, **over):
        p = dict(EKF_YAML)
        p.update(over)
        return self.init(**p)

    # -- views into the C struct -------------------------------------------
    def _arr(self, field, shape):
        ptr = getattr(self._p.contents, field)
        a = np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),))
        return a.reshape(shape, order="F")

    @property
    def x(self): return self._arr("x", (self.nx,))
    @property
    def P(self): return self._arr("P", (self.n, self.n))
    @property
    def Qx(self): return self._arr("Qx", (self.n, self.n))
    @property
    def Qu(self): return np.array(self._p.contents.Qu).reshape(6, 6, order="F")
    @property
    def lam(self): return self._arr("lam", (self.n,))
    @property
    def Lambda(self): return self._arr("Lambda", (self.n, self.n))
    @property
    def A(self): return self._arr("A", (self.n, self.n))
    @property
    def G(self): return self._arr("G", (self.n, 6))
    @property
    def dx(self): return self._arr("dx", (self.n,))
    @property
    def K(self): return self._arr("K", (self.n, 3))
    @property
    def H(self): return self._arr("H", (3, self.n))
    @property
    def zhat(self): return np.array(self._p.contents.zhat)
    @property
    def len_features(self): return self._p.contents.len_features
    @property
    def feature_ids(self):
        return [self._p.contents.feature_ids[i] for i in range(self.len_features)]

    @property
    def q_b_u(self): return np.array(self._p.contents.q_b_u)
    @property
    def use_keyframe_reset(self): return bool(self._p.contents.use_keyframe_reset)

    def global_to_local_feature_id(self, gid):
        return int(self._L.vo_global_to_local_feature_id(self._p, int(gid)))

    def set_drag_term(self, v): self._p.contents.use_drag_term = int(v)
    def set_partial_update(self, v): self._p.contents.use_partial_update = int(v)

    # -- reference API mirror ----------------------------------------------
    def boxplus(self, x, dx):
        x, dx = _vec(x, self.nx), _vec(dx, self.n)
        out = x.copy()
        self._L.vo_boxplus(self._p, _d(x), _d(dx), _d(out))
        return out

    def boxminus(self, x1, x2):
        x1, x2 = _vec(x1, self.nx), _vec(x2, self.nx)
        out = np.zeros(self.n)
        self._L.vo_boxminus(self._p, _d(x1), _d(x2), _d(out))
        return out

    def dynamics(self, x, u):
        x, u = _vec(x, self.nx), _vec(u, 6)
        self._L.vo_dynamics(self._p, _d(x), _d(u), 1, 1)
        return self.dx.copy(), self.A.copy(), self.G.copy()

    def propagate(self, u_imu, dt):
        u = _vec(u_imu, 6)
        self._L.vo_propagate(self._p, _d(u), float(dt))

    def fix_depth(self): self._L.vo_fix_depth(self._p)

    def init_feature(self, l, id=-1, depth=float("nan")):
        l = _vec(l, 2)
        return bool(self._L.vo_init_feature(self._p, _d(l), int(id), float(depth)))

    def clear_feature(self, id): self._L.vo_clear_feature(self._p, int(id))

    def h(self, mtype, x=None, id=0):
        x = self.x.copy() if x is None else _vec(x, self.nx)
        h = np.zeros(4)
        H = np.zeros((3, self.n), order="F")
        self._L.vo_h(self._p, int(mtype), _d(x), _d(h), _d(H), int(id))
        return h, H

    def update(self, mtype, z, R, active=True, id=-1):
        z = _vec(z)
        R = np.asfortranarray(np.atleast_2d(np.asarray(R, dtype=np.float64)))
        Rf = np.ascontiguousarray(R.ravel(order="F"))
        return int(self._L.vo_update(self._p, int(mtype), _d(z), int(z.size), _d(Rf), int(R.shape[0]),
                                     int(active), int(id)))

    def keyframe_reset(self): self._L.vo_keyframe_reset(self._p)

    def keyframe_reset_edge(self):
        """keyframe reset that also returns the edge {t(3), q_yaw(4), cov_pos(9, column-major), cov_yaw}"""
        e = np.zeros(17)
        self._L.vo_keyframe_reset_edge(self._p, _d(e))
        return e
    def nans_in_the_house(self): return bool(self._L.vo_nans_in_the_house(self._p))
    def blowing_up(self): return bool(self._L.vo_blowing_up(self._p))
    def negative_depth(self): return bool(self._L.vo_negative_depth(self._p))

    def run_steps(self, u, dt, z, slot, R):
        """u [steps][6], z [steps][M][2], slot [M], R 2x2 -> results [steps][M]"""
        u = np.ascontiguousarray(u, dtype=np.float64)
        z = np.ascontiguousarray(z, dtype=np.float64)
        slot = np.ascontiguousarray(slot, dtype=np.int32)
        steps, M = u.shape[0], slot.shape[0]
        Rf = np.ascontiguousarray(np.asarray(R, dtype=np.float64).ravel(order="F"))
        res = np.zeros((steps, M), dtype=np.int32)
        self._L.vo_run_steps(self._p, steps, _d(u), float(dt), _d(z), slot.ctypes.data_as(_ip), M, _d(Rf),
                             res.ctypes.data_as(_ip))
        return res


def run_steps_mt(filters, threads, u, dt, z, slot, R, structured=False):
    """filters: list[OracleFilter]; u [nf][steps][6]; z [nf][steps][M][2]; slot [nf][M].
    structured=True: the block-sparse / rank-2 CPU flavour (second cpu_baseline figure)."""
    L = lib()
    nf = len(filters)
    u = np.ascontiguousarray(u, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    slot = np.ascontiguousarray(slot, dtype=np.int32)
    steps, M = u.shape[1], slot.shape[1]
    Rf = np.ascontiguousarray(np.asarray(R, dtype=np.float64).ravel(order="F"))
    res = np.zeros((nf, steps, M), dtype=np.int32)
    arr = (_fp * nf)(*[f._p for f in filters])
    L.vo_run_steps_mt_flavour(arr, nf, int(threads), steps, _d(u), float(dt), _d(z), slot.ctypes.data_as(_ip), M, _d(Rf),
                              res.ctypes.data_as(_ip), 1 if structured else 0)
    return res
