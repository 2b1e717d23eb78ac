"""vi_ekf_amd -- MI355X-native batched VI-EKF predict/update core (HIP, gfx950).

The numeric path lives in libviekf_hip.so (C ABI: include/viekf.h).  Importing the package
does not need a GPU; using it does, and there is no CPU fallback.
"""
from . import capi, scene  # noqa: F401
from .batch import BatchVIEKF  # noqa: F401
from .capi import Params, ViekfError, device_count, load_yaml  # noqa: F401
from .seq import SeqVIEKF  # noqa: E402,F401
