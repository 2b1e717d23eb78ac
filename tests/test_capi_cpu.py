"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/viekf.h
declares, reads the parameter YAML like VIEKF::load, and fails loudly (no CPU fallback) without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vi_ekf_amd as v
from vi_ekf_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YAML = os.path.join(ROOT, "vi_ekf_amd", "params", "ekf.yaml")


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "viekf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(viekf_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_library_agree():
    syms = declared_symbols()
    assert sorted(capi.SYMBOLS) == syms
    L = capi.lib()
    for s in syms:
        assert hasattr(L, s), "libviekf_hip.so does not export %s" % s
    assert L.viekf_abi_version() == 1


def test_yaml_loader_reads_every_key():
    p = v.load_yaml(YAML).to_dict()
    from vi_ekf_amd.scene import EKF_YAML
    for k in capi.Params.ARRAYS:
        np.testing.assert_allclose(p[k], np.asarray(EKF_YAML[k], dtype=float), rtol=0, atol=0, err_msg=k)
    assert p["name"] == "ekf1" and p["min_depth"] == 1.5 and p["keyframe_overlap_threshold"] == 0.8
    assert p["use_drag_term"] == 1 and p["use_partial_update"] == 1 and p["use_keyframe_reset"] == 1


def test_yaml_errors(tmp_path):
    with pytest.raises(v.ViekfError) as e:
        v.load_yaml(str(tmp_path / "missing.yaml"))
    assert e.value.code == capi.ERR_YAML
    txt = open(YAML).read()
    bad = tmp_path / "short.yaml"
    bad.write_text(txt.replace("lambda_feat: [1.0, 1.0, 0.4]", "lambda_feat: [1.0, 1.0]"))
    with pytest.raises(v.ViekfError) as e:
        v.load_yaml(str(bad))
    assert e.value.code == capi.ERR_YAML and "lambda_feat" in str(e.value)
    nokey = tmp_path / "nokey.yaml"
    nokey.write_text("\n".join(l for l in txt.splitlines() if not l.startswith("min_depth")))
    with pytest.raises(v.ViekfError) as e:
        v.load_yaml(str(nokey))
    assert "min_depth" in str(e.value)


def test_params_roundtrip_and_validation():
    from vi_ekf_amd.scene import EKF_YAML
    p = capi.Params.from_dict(EKF_YAML)
    d = p.to_dict()
    np.testing.assert_array_equal(d["q_b_u"], EKF_YAML["q_b_u"])
    with pytest.raises(ValueError):
        capi.Params.from_dict(dict(EKF_YAML, x0=[0.0] * 5))


@pytest.mark.skipif(v.device_count() > 0, reason="needs a machine WITHOUT a GPU")
def test_no_cpu_fallback():
    from vi_ekf_amd.scene import EKF_YAML
    with pytest.raises(v.ViekfError) as e:
        v.BatchVIEKF(4, 3, EKF_YAML)
    assert e.value.code == capi.ERR_NO_DEVICE


def test_argument_validation_without_device():
    L = capi.lib()
    assert L.viekf_params_default(None) == capi.ERR_INVALID
    assert L.viekf_batch_sync(None) == capi.ERR_INVALID
    out = C.c_void_p()
    p = capi.Params.from_dict({})
    assert L.viekf_batch_create(0, 3, C.byref(p), 0, C.byref(out)) == capi.ERR_INVALID
    assert b"batch" in L.viekf_last_error()
