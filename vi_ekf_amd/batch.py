"""BatchVIEKF: a batch of independent filters on one MI355X, driven through the C ABI.

Method names follow the reference class vi_ekf::VIEKF (reference include/vi_ekf.h:82-338):
propagate_state -> propagate, update(FEAT) -> update_feat, init_feature, get_state,
get_covariance.  numpy arrays are passed as host pointers, torch CUDA tensors as device
pointers; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import capi


def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


class BatchVIEKF:
    def __init__(self, batch, num_features, params, device=0):
        L = capi.lib()
        if isinstance(params, dict):
            params = capi.Params.from_dict(params)
        self.params = params
        self._h = C.c_void_p()
        capi.check(L.viekf_batch_create(int(batch), int(num_features), C.byref(params), int(device), C.byref(self._h)))
        self.B, self.N = int(batch), int(num_features)
        self.nx, self.n = 17 + 5 * self.N, 16 + 3 * self.N
        self.device = int(device)
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            capi.lib().viekf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- argument marshalling -------------------------------------------------------------
    def _arg(self, a, dtype, shape, where):
        """-> (pointer, where).  All array arguments of one call must live on the same side."""
        if a is None:
            return None, where
        if _is_torch(a):
            import torch
            want = {np.float64: torch.float64, np.int32: torch.int32, np.uint8: torch.uint8, np.uint32: torch.int32}[dtype]
            if not a.is_cuda or a.dtype != want or not a.is_contiguous():
                raise ValueError("device arguments must be contiguous CUDA tensors of dtype %s" % want)
            if tuple(a.shape) != tuple(shape):
                raise ValueError("expected shape %s, got %s" % (shape, tuple(a.shape)))
            if where not in (None, capi.DEVICE):
                raise ValueError("cannot mix host and device arguments in one call")
            return C.c_void_p(a.data_ptr()), capi.DEVICE
        arr = np.ascontiguousarray(a, dtype=dtype)
        if arr.shape != tuple(shape):
            raise ValueError("expected shape %s, got %s" % (shape, arr.shape))
        if where not in (None, capi.HOST):
            raise ValueError("cannot mix host and device arguments in one call")
        self._keep.append(arr)
        return C.c_void_p(arr.ctypes.data), capi.HOST

    def _out(self, a, dtype, shape, where):
        if _is_torch(a):
            return self._arg(a, dtype, shape, where)
        if a.dtype != dtype or not a.flags.c_contiguous or a.shape != tuple(shape):
            raise ValueError("output array must be C-contiguous %s of shape %s" % (dtype, shape))
        if where not in (None, capi.HOST):
            raise ValueError("cannot mix host and device arguments in one call")
        return C.c_void_p(a.ctypes.data), capi.HOST

    # ---- C ABI calls -----------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        capi.check(capi.lib().viekf_batch_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self):
        capi.check(capi.lib().viekf_batch_sync(self._h))

    def set_kernel(self, family):
        capi.check(capi.lib().viekf_batch_set_kernel(self._h, int(family)))

    def set_tuning(self, key, value):
        """kernel selection knobs for tests and experiments (viekf_batch_set_tuning; capi.TUNE_*)"""
        capi.check(capi.lib().viekf_batch_set_tuning(self._h, int(key), int(value)))

    def describe(self):
        """which kernels a step of this batch launches (viekf_batch_describe)"""
        buf = C.create_string_buffer(256)
        capi.check(capi.lib().viekf_batch_describe(self._h, buf, 256))
        return buf.value.decode()

    def reset(self):
        capi.check(capi.lib().viekf_batch_reset(self._h))

    def get_state(self):
        """-> x [B][nx] (reference get_state(), include/vi_ekf.h:277)"""
        x = np.empty((self.B, self.nx))
        capi.check(capi.lib().viekf_batch_get_state(self._h, C.c_void_p(x.ctypes.data), None, None, capi.HOST))
        return x

    def get_covariance(self):
        """-> P [B][n][n] as numpy [b, i, j] (reference get_covariance(), include/vi_ekf.h:278)"""
        P = np.empty((self.B, self.n, self.n))
        capi.check(capi.lib().viekf_batch_get_state(self._h, None, C.c_void_p(P.ctypes.data), None, capi.HOST))
        return np.ascontiguousarray(P.transpose(0, 2, 1))  # column-major per filter -> [b, row, col]

    def get_len_features(self):
        ln = np.empty(self.B, dtype=np.int32)
        capi.check(capi.lib().viekf_batch_get_state(self._h, None, None, C.c_void_p(ln.ctypes.data), capi.HOST))
        return ln

    def set_state(self, x=None, P=None, len_features=None):
        """x [B][nx]; P [B][n][n] indexed [b, row, col]; len_features [B]"""
        self._keep = []
        px = pP = pl = None
        if x is not None:
            px, _ = self._arg(x, np.float64, (self.B, self.nx), None)
        if P is not None:
            Pc = np.ascontiguousarray(np.asarray(P, dtype=np.float64).transpose(0, 2, 1))
            pP, _ = self._arg(Pc, np.float64, (self.B, self.n, self.n), None)
        if len_features is not None:
            pl, _ = self._arg(len_features, np.int32, (self.B,), None)
        capi.check(capi.lib().viekf_batch_set_state(self._h, px, pP, pl, capi.HOST))
        self._keep = []

    def get_status(self):
        f = np.empty(self.B, dtype=np.uint32)
        capi.check(capi.lib().viekf_batch_get_status(self._h, C.c_void_p(f.ctypes.data), capi.HOST))
        return f

    def propagate(self, u, dt):
        """numeric core of propagate_state (reference vi_ekf.cpp:262-318); u [B][6] raw IMU, dt [B]"""
        self._keep = []
        pu, w = self._arg(u, np.float64, (self.B, 6), None)
        pdt, w = self._arg(dt, np.float64, (self.B,), w)
        capi.check(capi.lib().viekf_batch_propagate(self._h, pu, pdt, w))
        self._keep = []

    def init_feature(self, pix, depth=None, mask=None):
        """reference vi_ekf_feat.cpp:6-47 for every filter (or those with mask != 0) -> ok [B]"""
        self._keep = []
        ppix, w = self._arg(pix, np.float64, (self.B, 2), None)
        pdep, w = self._arg(depth, np.float64, (self.B,), w)
        pm, w = self._arg(mask, np.uint8, (self.B,), w)
        if w == capi.DEVICE:
            import torch
            ok = torch.empty(self.B, dtype=torch.int32, device=pix.device)
        else:
            ok = np.empty(self.B, dtype=np.int32)
        pok, w = self._out(ok, np.int32, (self.B,), w)
        capi.check(capi.lib().viekf_batch_init_feature(self._h, ppix, pdep, pm, pok, w))
        self._keep = []
        return ok

    def _meas_args(self, z, slot, R, result):
        M = int(slot.shape[1])
        pz, w = self._arg(z, np.float64, (self.B, M, 2), None)
        ps, w = self._arg(slot, np.int32, (self.B, M), w)
        Rshape = tuple(R.shape)
        if Rshape == (2, 2):
            r_mode = 0
            # the ABI takes column-major 2x2; a symmetric R is the same either way, transpose to be exact
            R = R.t().contiguous() if _is_torch(R) else np.ascontiguousarray(np.asarray(R, dtype=np.float64).T)
        elif Rshape == (self.B, 2, 2):
            r_mode = 1
            R = R.transpose(1, 2).contiguous() if _is_torch(R) else np.ascontiguousarray(np.asarray(R).transpose(0, 2, 1))
        elif Rshape == (self.B, M, 2, 2):
            r_mode = 2
            R = R.transpose(2, 3).contiguous() if _is_torch(R) else np.ascontiguousarray(np.asarray(R).transpose(0, 1, 3, 2))
        else:
            raise ValueError("R must be (2,2), (B,2,2) or (B,M,2,2)")
        pR, w = self._arg(R, np.float64, tuple(R.shape), w)
        if _is_torch(R):
            self._keep.append(R)
        if result is None:
            if w == capi.DEVICE:
                import torch
                result = torch.empty((self.B, M), dtype=torch.int32, device=z.device)
            else:
                result = np.empty((self.B, M), dtype=np.int32)
        pres, w = self._out(result, np.int32, (self.B, M), w)
        return M, pz, ps, pR, r_mode, pres, result, w

    def update_feat(self, z, slot, R, result=None):
        """M sequential active FEAT updates (reference vi_ekf_meas.cpp:196-278) -> result [B][M]"""
        self._keep = []
        M, pz, ps, pR, r_mode, pres, result, w = self._meas_args(z, slot, R, result)
        capi.check(capi.lib().viekf_batch_update_feat(self._h, pz, ps, M, pR, r_mode, pres, w))
        self._keep = []
        return result

    def update(self, mtype, z, R, slot=None, active=None):
        """ONE measurement of any reference model per filter (reference vi_ekf_meas.cpp:196-386); host arrays.
        z [B][zdim]; R (rdim,rdim) shared or (B,rdim,rdim); slot [B] for feature models -> result [B]"""
        self._keep = []
        z = np.ascontiguousarray(z, dtype=np.float64)
        zdim = z.shape[1]
        R = np.asarray(R, dtype=np.float64)
        rdim = R.shape[-1]
        if R.ndim == 2:
            r_mode, Rc = 0, np.ascontiguousarray(R.T)
        else:
            r_mode, Rc = 1, np.ascontiguousarray(R.transpose(0, 2, 1))
        pz, w = self._arg(z, np.float64, (self.B, zdim), None)
        pR, w = self._arg(Rc, np.float64, Rc.shape, w)
        ps, w = self._arg(slot, np.int32, (self.B,), w)
        pa, w = self._arg(active, np.uint8, (self.B,), w)
        res = np.empty(self.B, dtype=np.int32)
        capi.check(capi.lib().viekf_batch_update(self._h, int(mtype), pz, zdim, pR, rdim, r_mode, ps, pa,
                                                 C.c_void_p(res.ctypes.data), capi.HOST))
        self._keep = []
        return res

    def keep_features(self, keep):
        """drop the features with keep[b, f] == 0 and compact (reference vi_ekf_feat.cpp:50-117) -> new len [B]"""
        self._keep = []
        pk, w = self._arg(keep, np.uint8, (self.B, self.N), None)
        nl = np.empty(self.B, dtype=np.int32)
        capi.check(capi.lib().viekf_batch_keep_features(self._h, pk, C.c_void_p(nl.ctypes.data), capi.HOST))
        self._keep = []
        return nl

    def keyframe_reset(self, mask=None):
        """keyframe reset of the filters with mask[b] != 0 (None = all), reference vi_ekf_kfr.cpp:56-157.
        Returns the edges [B][17] = {t(3), q_yaw(4), cov_pos(9, column-major), cov_yaw} (zeros outside the mask)."""
        self._keep = []
        pm = None
        if mask is not None:
            pm, _ = self._arg(mask, np.uint8, (self.B,), None)
        edge = np.zeros((self.B, 17), dtype=np.float64)
        capi.check(capi.lib().viekf_batch_keyframe_reset(self._h, pm, C.c_void_p(edge.ctypes.data), capi.HOST))
        self._keep = []
        return edge

    def eval_xdot(self, u):
        """dx_ of VIEKF::dynamics at the current state for the input u [B][6] (reference vi_ekf_dyn.cpp:6-134): [B][n]"""
        self._keep = []
        pu, _ = self._arg(u, np.float64, (self.B, 6), None)
        out = np.zeros((self.B, self.n), dtype=np.float64)
        capi.check(capi.lib().viekf_batch_eval_xdot(self._h, pu, C.c_void_p(out.ctypes.data), capi.HOST))
        self._keep = []
        return out

    def eval_h(self, mtype, slot=None):
        """zhat = h(x) of a measurement model at the current state (reference vi_ekf_meas.cpp:281-386): [B][4], NaN padded"""
        self._keep = []
        ps = None
        if slot is not None:
            ps, _ = self._arg(slot, np.int32, (self.B,), None)
        out = np.zeros((self.B, 4), dtype=np.float64)
        capi.check(capi.lib().viekf_batch_eval_h(self._h, int(mtype), ps, C.c_void_p(out.ctypes.data), capi.HOST))
        self._keep = []
        return out

    def get_cov_diag(self):
        out = np.zeros((self.B, self.n), dtype=np.float64)
        capi.check(capi.lib().viekf_batch_get_cov_diag(self._h, C.c_void_p(out.ctypes.data), capi.HOST))
        return out

    def history_resize(self, depth):
        capi.check(capi.lib().viekf_batch_history_resize(self._h, int(depth)))

    def snapshot(self, slot):
        capi.check(capi.lib().viekf_batch_snapshot(self._h, int(slot)))

    def restore(self, slot):
        capi.check(capi.lib().viekf_batch_restore(self._h, int(slot)))

    def step_n(self, u, dt, z, slot, R, result=None):
        """K propagates (u [K,B,6], dt [K,B]: the IMU samples since the last frame) + M feature updates in one launch"""
        self._keep = []
        K = int(u.shape[0])
        pu, w0 = self._arg(u, np.float64, (K, self.B, 6), None)
        pdt, w0 = self._arg(dt, np.float64, (K, self.B), w0)
        if z is None:     # K propagates, no measurements
            M, pz, ps, pR, r_mode, pres, result, w = 0, None, None, None, 0, None, None, w0
        else:
            M, pz, ps, pR, r_mode, pres, result, w = self._meas_args(z, slot, R, result)
        if w != w0:
            raise ValueError("cannot mix host and device arguments in one call")
        capi.check(capi.lib().viekf_batch_step_n(self._h, K, pu, pdt, pz, ps, M, pR, r_mode, pres, w))
        self._keep = []
        return result

    def step(self, u, dt, z, slot, R, result=None):
        """one hot-path step: propagate + M feature updates"""
        self._keep = []
        pu, w0 = self._arg(u, np.float64, (self.B, 6), None)
        pdt, w0 = self._arg(dt, np.float64, (self.B,), w0)
        M, pz, ps, pR, r_mode, pres, result, w = self._meas_args(z, slot, R, result)
        if w != w0:
            raise ValueError("cannot mix host and device arguments in one call")
        capi.check(capi.lib().viekf_batch_step(self._h, pu, pdt, pz, ps, M, pR, r_mode, pres, w))
        self._keep = []
        return result
