"""Python face of the host sequencer (include/viekf.h, viekf_seq_*): the reference's add_measurement /
handle_measurements / propagate_state / keep_only_features plumbing for a batch of filters that share one clock.
Plumbing only -- the logic lives in libviekf_hip.so (csrc/viekf_seq.cpp)."""
import ctypes as C

import numpy as np

from . import capi


def _bind():
    L = capi.lib()
    if getattr(L, "_seq_bound", False):
        return L
    vp = C.c_void_p
    L.viekf_seq_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.viekf_seq_destroy.argtypes = [vp]
    L.viekf_seq_propagate.argtypes = [vp, vp, C.c_double]
    L.viekf_seq_add_measurement.argtypes = [vp, C.c_double, C.c_int32, vp, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.viekf_seq_handle_measurements.argtypes = [vp, vp, C.c_int32, vp]
    L.viekf_seq_keep_only_features.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.viekf_seq_tracked_features.argtypes = [vp, vp, vp]
    L.viekf_seq_status.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.viekf_seq_init_logger.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int32]
    L.viekf_seq_disable_logger.argtypes = [vp]
    L.viekf_seq_create_independent.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.viekf_seq_propagate_t.argtypes = [vp, vp, vp, vp]
    L.viekf_seq_add_measurement_t.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.viekf_seq_get_global_pose.argtypes = [vp, vp, vp]
    L.viekf_seq_get_global_cov.argtypes = [vp, vp]
    L.viekf_seq_add_frame.argtypes = [vp, C.c_double, vp, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp]
    L.viekf_seq_propagate_state.argtypes = [vp, vp, C.c_double, C.c_int32]
    L.viekf_seq_set_x0.argtypes = [vp, vp]
    L.viekf_seq_set_imu_bias.argtypes = [vp, vp, vp]
    L.viekf_seq_keyframe_reset.argtypes = [vp, vp, vp]
    L.viekf_seq_get_features.argtypes = [vp, vp, vp, vp]
    L.viekf_seq_get_feat.argtypes = [vp, vp, vp, vp]
    L.viekf_seq_drop_features.argtypes = [vp, vp, C.c_int32]
    L._seq_bound = True
    return L


def _p(a):
    return C.c_void_p(a.ctypes.data)


class SeqVIEKF:
    """mirrors vi_ekf::VIEKF's measurement plumbing (reference include/vi_ekf.h:302-308) over a BatchVIEKF"""

    def __init__(self, batch, state_hist=250, meas_hist=200, independent=False):
        """independent=True: every filter on its own clock (viekf_seq_create_independent); propagate_state / add_measurement
        then also take an array of time stamps [B] and an optional mask [B]"""
        self.core = batch
        self.B, self.N = batch.B, batch.N
        self._L = _bind()
        self.independent = bool(independent)
        h = C.c_void_p()
        create = self._L.viekf_seq_create_independent if independent else self._L.viekf_seq_create
        capi.check(create(batch._h, int(state_hist), int(meas_hist), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                self._L.viekf_seq_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def propagate_state(self, u, t, mask=None):
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(self.B, 6)
        if np.ndim(t) == 0 and mask is None:
            capi.check(self._L.viekf_seq_propagate(self._h, _p(u), float(t)))
            return
        tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        capi.check(self._L.viekf_seq_propagate_t(self._h, _p(u), _p(tt), None if mk is None else _p(mk)))

    def add_measurement(self, t, z, mtype, R, active=False, id=None, depth=None, mask=None):
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(self.B, -1)
        R = np.asfortranarray(np.atleast_2d(np.asarray(R, dtype=np.float64)))
        Rf = np.ascontiguousarray(R.ravel(order="F"))
        idp = dp = None
        if id is not None:
            ida = np.ascontiguousarray(np.broadcast_to(np.asarray(id, dtype=np.int32), (self.B,)))
            idp = _p(ida)
        if depth is not None:
            da = np.ascontiguousarray(np.broadcast_to(np.asarray(depth, dtype=np.float64), (self.B,)))
            dp = _p(da)
        res = np.zeros(self.B, dtype=np.int32)
        if np.ndim(t) == 0 and mask is None:
            capi.check(self._L.viekf_seq_add_measurement(self._h, float(t), int(mtype), _p(z), z.shape[1], _p(Rf), R.shape[0],
                                                         int(bool(active)), idp, dp, _p(res)))
            return res
        tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        capi.check(self._L.viekf_seq_add_measurement_t(self._h, _p(tt), int(mtype), _p(z), z.shape[1], _p(Rf), R.shape[0],
                                                       int(bool(active)), idp, dp, None if mk is None else _p(mk), _p(res)))
        return res

    def add_frame(self, t, z, R, ids, active=True, depth=None, mask=None):
        """a whole camera frame in one call (viekf_seq_add_frame): z [B][count][2], ids [B][count] (or [count]: the same for every
        filter), depth [B][count] or None -> results [B][count]; what count add_measurement(FEAT) calls in order do"""
        z = np.ascontiguousarray(z, dtype=np.float64)
        count = z.shape[1]
        assert z.shape == (self.B, count, 2)
        ida = np.ascontiguousarray(np.broadcast_to(np.asarray(ids, dtype=np.int32), (self.B, count)))
        R = np.asfortranarray(np.atleast_2d(np.asarray(R, dtype=np.float64)))
        Rf = np.ascontiguousarray(R.ravel(order="F"))
        da = None if depth is None else np.ascontiguousarray(np.broadcast_to(np.asarray(depth, dtype=np.float64), (self.B, count)))
        res = np.zeros((self.B, count), dtype=np.int32)
        tt = None
        t0 = float(t) if np.ndim(t) == 0 else 0.0
        if np.ndim(t) != 0:
            tt = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.B,)))
        mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        capi.check(self._L.viekf_seq_add_frame(self._h, t0, None if tt is None else _p(tt), count, _p(z), _p(Rf), int(bool(active)), _p(ida),
                                               None if da is None else _p(da), None if mk is None else _p(mk), _p(res)))
        return res

    def handle_measurements(self, cap=64, want_gated=True):
        """want_gated=False = the reference's handle_measurements() without the optional list (test/vi_ekf_test.cpp:32): the frame's
        launch is then queued, not waited for"""
        if not want_gated:
            capi.check(self._L.viekf_seq_handle_measurements(self._h, None, 0, None))
            return None
        ids = np.full((self.B, cap), -1, dtype=np.int32)
        cnt = np.zeros(self.B, dtype=np.int32)
        capi.check(self._L.viekf_seq_handle_measurements(self._h, _p(ids), cap, _p(cnt)))
        return [ids[b, :cnt[b]].tolist() for b in range(self.B)]

    def keep_only_features(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(self.B, -1)
        did = np.zeros(self.B, dtype=np.uint8)
        edges = np.zeros((self.B, 17), dtype=np.float64)
        capi.check(self._L.viekf_seq_keep_only_features(self._h, _p(ids), ids.shape[1], _p(did), _p(edges)))
        return did, edges

    def init_logger(self, root_filename, ekf_name="", filter=0):
        """binary logs of one filter of the batch in the reference's formats (vi_ekf_log.cpp:79-117; matlab/plot_ekf.m)"""
        capi.check(self._L.viekf_seq_init_logger(self._h, str(root_filename).encode(), str(ekf_name).encode(), int(filter)))

    def disable_logger(self):
        capi.check(self._L.viekf_seq_disable_logger(self._h))

    def get_global_pose(self):
        """-> (pose [B][7], node [B][7]) = VIEKF::get_global_pose / get_current_node_global_pose, {t(3), q(4)}
        (reference vi_ekf_kfr.cpp:14-21; SE(3) convention in include/viekf.h)"""
        pose = np.zeros((self.B, 7))
        node = np.zeros((self.B, 7))
        capi.check(self._L.viekf_seq_get_global_pose(self._h, _p(pose), _p(node)))
        return pose, node

    def get_global_cov(self):
        """-> [B][6][6] = VIEKF::get_global_cov (reference vi_ekf_kfr.cpp:23-53), indexed [b, row, col]"""
        cov = np.zeros((self.B, 6, 6))
        capi.check(self._L.viekf_seq_get_global_cov(self._h, _p(cov)))
        return np.ascontiguousarray(cov.transpose(0, 2, 1))

    def tracked_features(self):
        ids = np.zeros((self.B, self.N), dtype=np.int32)
        ln = np.zeros(self.B, dtype=np.int32)
        capi.check(self._L.viekf_seq_tracked_features(self._h, _p(ids), _p(ln)))
        return [ids[b, :ln[b]].tolist() for b in range(self.B)]

    def status(self):
        t, i, q, u = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(self._L.viekf_seq_status(self._h, C.byref(t), C.byref(i), C.byref(q), C.byref(u)))
        return dict(t=t.value, ring_index=i.value, queued=q.value, inputs=u.value)
