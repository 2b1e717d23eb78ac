"""ctypes binding of the C ABI in include/viekf.h (libviekf_hip.so).

This is plumbing only: every numeric operation runs in the HIP library.  There is no
Python/numpy fallback -- if the library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import _build

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_YAML, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
HOST, DEVICE = 0, 1
MEAS_SKIPPED, MEAS_SUCCESS, MEAS_GATED, MEAS_NAN, MEAS_INVALID, MEAS_NEW_FEATURE = -1, 0, 1, 2, 3, 4
FLAG_NAN, FLAG_BLOWING_UP, FLAG_NEGATIVE_DEPTH, FLAG_INTERNAL = 1, 2, 4, 8
TUNE_RES_INSTANCE, TUNE_UNIT_LAMBDA, TUNE_BLOCK_GROUP, TUNE_STREAM_MFMA, TUNE_TILES, TUNE_PANEL_SERVICE = 1, 2, 3, 4, 5, 6

# every symbol include/viekf.h declares (tests check the library exports exactly these)
SYMBOLS = [
    "viekf_abi_version", "viekf_last_error", "viekf_device_count", "viekf_params_default", "viekf_params_load_yaml",
    "viekf_batch_create", "viekf_batch_destroy", "viekf_batch_reset", "viekf_batch_dims", "viekf_batch_describe", "viekf_batch_set_stream",
    "viekf_batch_sync", "viekf_batch_set_kernel", "viekf_batch_get_state", "viekf_batch_set_state",
    "viekf_batch_get_status", "viekf_batch_propagate", "viekf_batch_init_feature", "viekf_batch_update_feat",
    "viekf_batch_step", "viekf_batch_update", "viekf_batch_keep_features", "viekf_batch_history_resize",
    "viekf_batch_snapshot", "viekf_batch_restore", "viekf_batch_keyframe_reset", "viekf_batch_get_params", "viekf_batch_select", "viekf_batch_propagate_to", "viekf_batch_propagate_n_to",
    "viekf_seq_create", "viekf_seq_destroy", "viekf_seq_propagate", "viekf_seq_add_measurement",
    "viekf_seq_handle_measurements", "viekf_seq_keep_only_features", "viekf_seq_tracked_features", "viekf_seq_status",
    "viekf_batch_eval_xdot", "viekf_batch_eval_h", "viekf_batch_get_cov_diag", "viekf_seq_init_logger",
    "viekf_seq_disable_logger", "viekf_batch_step_n", "viekf_batch_get_cov_block", "viekf_seq_get_global_pose",
    "viekf_seq_get_global_cov", "viekf_seq_init_feature", "viekf_batch_set_active", "viekf_batch_snapshot_filters",
    "viekf_batch_restore_filters", "viekf_seq_create_independent", "viekf_seq_propagate_t", "viekf_seq_add_measurement_t",
    "viekf_batch_set_tuning", "viekf_batch_set_drag_term", "viekf_batch_get_drag_term", "viekf_batch_eval_jacobians",
    "viekf_batch_eval_h_jacobian", "viekf_batch_boxplus", "viekf_batch_boxminus", "viekf_batch_eval_reset_jacobian",
    "viekf_seq_propagate_state", "viekf_seq_set_x0", "viekf_seq_set_imu_bias", "viekf_seq_keyframe_reset",
    "viekf_seq_get_features", "viekf_seq_get_feat", "viekf_seq_drop_features", "viekf_seq_add_frame", "viekf_batch_set_async",
    "viekf_batch_select_filters", "viekf_batch_propagate_filters_to",
]


class Params(C.Structure):
    """struct viekf_params (include/viekf.h) == the keys VIEKF::load reads (reference vi_ekf.cpp:114-131)."""
    _fields_ = [
        ("x0", C.c_double * 17), ("P0", C.c_double * 16), ("Qx", C.c_double * 16), ("lam", C.c_double * 16),
        ("Qu", C.c_double * 6), ("P0_feat", C.c_double * 3), ("Qx_feat", C.c_double * 3),
        ("lam_feat", C.c_double * 3), ("cam_center", C.c_double * 2), ("focal_len", C.c_double * 2),
        ("q_b_c", C.c_double * 4), ("p_b_c", C.c_double * 3), ("q_b_u", C.c_double * 4),
        ("min_depth", C.c_double), ("keyframe_overlap_threshold", C.c_double),
        ("use_drag_term", C.c_int32), ("use_partial_update", C.c_int32), ("use_keyframe_reset", C.c_int32),
        ("name", C.c_char * 64),
    ]

    ARRAYS = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c",
              "p_b_c", "q_b_u")
    SCALARS = ("min_depth", "keyframe_overlap_threshold", "use_drag_term", "use_partial_update",
               "use_keyframe_reset")

    def to_dict(self):
        d = {k: np.array(getattr(self, k)) for k in self.ARRAYS}
        d.update({k: getattr(self, k) for k in self.SCALARS})
        d["name"] = self.name.decode()
        return d

    @classmethod
    def from_dict(cls, d):
        p = cls()
        lib().viekf_params_default(C.byref(p))
        for k in cls.ARRAYS:
            if k in d:
                v = np.asarray(d[k], dtype=np.float64).ravel()
                arr = getattr(p, k)
                if v.size != len(arr):
                    raise ValueError("param %s needs %d values, got %d" % (k, len(arr), v.size))
                for i in range(v.size):
                    arr[i] = float(v[i])
        for k in cls.SCALARS:
            if k in d:
                setattr(p, k, type(getattr(p, k))(d[k]))
        if "name" in d:
            p.name = str(d["name"]).encode()[:63]
        return p


_lib = None
_vp = C.c_void_p


class ViekfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("viekf error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("VIEKF_LIB") or _build.LIB
        # One HIP runtime per process: torch ships its own libamdhip64.so.7.  If it is loaded first,
        # our NEEDED libamdhip64.so.7 binds to that copy (same SONAME) and device pointers / streams
        # can be shared with torch; loaded the other way round the process ends up with two runtimes
        # and torch reports "No HIP GPUs are available".
        if os.environ.get("VIEKF_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(path):
            raise ImportError("libviekf_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                              "vi_ekf_amd has no CPU fallback")
        L = C.CDLL(path)
        L.viekf_abi_version.restype = C.c_int
        L.viekf_last_error.restype = C.c_char_p
        L.viekf_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.viekf_params_default.argtypes = [C.POINTER(Params)]
        L.viekf_params_load_yaml.argtypes = [C.c_char_p, C.POINTER(Params)]
        L.viekf_batch_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32, C.POINTER(_vp)]
        L.viekf_batch_destroy.argtypes = [_vp]
        L.viekf_batch_reset.argtypes = [_vp]
        L.viekf_batch_dims.argtypes = [_vp] + [C.POINTER(C.c_int32)] * 4
        L.viekf_batch_describe.argtypes = [_vp, C.c_char_p, C.c_int32]
        L.viekf_batch_set_stream.argtypes = [_vp, _vp]
        L.viekf_batch_sync.argtypes = [_vp]
        L.viekf_batch_set_kernel.argtypes = [_vp, C.c_int32]
        L.viekf_batch_set_tuning.argtypes = [_vp, C.c_int32, C.c_int32]
        L.viekf_batch_set_drag_term.argtypes = [_vp, C.c_int32]
        L.viekf_batch_get_drag_term.argtypes = [_vp, C.POINTER(C.c_int32)]
        L.viekf_batch_eval_jacobians.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_eval_h_jacobian.argtypes = [_vp, _vp, C.c_int32, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_boxplus.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_boxminus.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_eval_reset_jacobian.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_get_state.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_set_state.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_get_status.argtypes = [_vp, _vp, C.c_int]
        L.viekf_batch_propagate.argtypes = [_vp, _vp, _vp, C.c_int]
        L.viekf_batch_init_feature.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_update_feat.argtypes = [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int]
        L.viekf_batch_update.argtypes = [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_keep_features.argtypes = [_vp, _vp, _vp, C.c_int]
        L.viekf_batch_keyframe_reset.argtypes = [_vp, _vp, _vp, C.c_int]
        L.viekf_batch_select.argtypes = [_vp, C.c_int32]
        L.viekf_batch_propagate_to.argtypes = [_vp, _vp, _vp, C.c_int32, C.c_int]
        L.viekf_batch_propagate_n_to.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int]
        L.viekf_batch_eval_xdot.argtypes = [_vp, _vp, _vp, C.c_int]
        L.viekf_batch_eval_h.argtypes = [_vp, C.c_int32, _vp, _vp, C.c_int]
        L.viekf_batch_get_cov_diag.argtypes = [_vp, _vp, C.c_int]
        L.viekf_batch_history_resize.argtypes = [_vp, C.c_int32]
        L.viekf_batch_snapshot.argtypes = [_vp, C.c_int32]
        L.viekf_batch_restore.argtypes = [_vp, C.c_int32]
        L.viekf_batch_step.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int]
        L.viekf_batch_step_n.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise ViekfError(rc, lib().viekf_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int32(0)
    check(lib().viekf_device_count(C.byref(n)))
    return n.value


def load_yaml(path):
    p = Params()
    check(lib().viekf_params_load_yaml(os.fsencode(path), C.byref(p)))
    return p
