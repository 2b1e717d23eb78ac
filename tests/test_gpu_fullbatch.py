"""GPU parity at the BASELINE batch sizes: filters of EVERY dispatch round against the oracle.

The small parity tests (tests/test_gpu_parity.py, B <= 12) only ever exercise the first workgroup
that lands on a CU.  The headline runs 1024 workgroups over 256 CUs (four rounds; a later workgroup
starts on LDS / registers that still hold a previous filter's data), and the two-per-CU instance
<3,2> is only picked when the batch exceeds the CU count.  Here the HIP path runs the full batch
and a strided sample of filters -- first, last and the ones either side of a round boundary -- is
compared with the CPU oracle on the same seeded inputs, with the same bar as test_gpu_parity.py.

Reference behaviour compared: propagate (vi_ekf.cpp:262-318) + N sequential FEAT updates
(vi_ekf_meas.cpp:196-278) per step.
"""
import os

import numpy as np
import pytest

import vi_ekf_amd as v
from oracle import oracle as orc
from vi_ekf_amd import scene
from tests.test_gpu_parity import assert_close, oracle_params

pytestmark = pytest.mark.gpu


def oracle_subset(sc, N, steps, which):
    """the oracle on filters `which` of the scene, all host threads -> x, P, res [len(which)][steps][M]"""
    fs = []
    for b in which:
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i, float("nan"))
        fs.append(f)
    u = np.ascontiguousarray(sc["u"][:steps, which].transpose(1, 0, 2))
    z = np.ascontiguousarray(sc["z"][:steps, which].transpose(1, 0, 2, 3))
    threads = max(1, min(len(which), os.cpu_count() or 1, 16))
    res = orc.run_steps_mt(fs, threads, u, float(sc["dt"][0]), z, sc["slot"][which], sc["R"])
    return np.stack([f.x.copy() for f in fs]), np.stack([f.P.copy() for f in fs]), res


def run_full(B, N, steps, which, kernel=0, seed=None, tune=()):
    sc = scene.make_scene(B, N, steps, seed=4000 + N if seed is None else seed)
    g = v.BatchVIEKF(B, N, sc["params"])
    if kernel:
        g.set_kernel(kernel)
    for key, value in tune:
        g.set_tuning(key, value)
    for i in range(N):
        ok = g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        assert (ok == 1).all()
    res = np.zeros((steps, B, N), dtype=np.int32)
    for s in range(steps):
        res[s] = g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    which = np.asarray(which)
    x_ref, P_ref, res_ref = oracle_subset(sc, N, steps, which)
    assert (res[:, which].transpose(1, 0, 2) == res_ref).all(), "meas_result codes differ"
    x = g.get_state()
    P = g.get_covariance()
    assert_close(x[which], x_ref, "x of filters %s" % list(which))
    assert_close(P[which], P_ref, "P of filters %s" % list(which))
    st = g.get_status()
    assert (st & (1 | 2 | 8) == 0).all(), "NaN / blow-up / internal flags raised: %s" % np.unique(st)
    # every filter ran the same kind of step: P symmetric and finite everywhere, not only in the sample
    assert np.isfinite(x).all() and np.isfinite(P).all()
    assert (P == P.transpose(0, 2, 1)).all()
    return g


def test_headline_batch_every_dispatch_round():
    """B=1024, N=50 (BASELINE configs[2]): the headline instance, two workgroups per CU, two rounds over 256 CUs"""
    run_full(1024, 50, 3, [0, 255, 256, 511, 700, 1023])


@pytest.mark.parametrize("N", [12, 20])
def test_two_per_cu_instance_beyond_one_round(N):
    """B=600 > 256 CUs: the <3,2> instance (192 threads, two workgroups per CU) is the one picked"""
    run_full(600, N, 4, [0, 1, 255, 256, 511, 512, 598, 599])


def test_config1_batch256_n25():
    """BASELINE configs[1]: B=256, N=25"""
    run_full(256, 25, 4, [0, 1, 127, 128, 254, 255])


def test_wide_p_full_batch():
    """BASELINE configs[4]: B=1024, N=150, P in HBM (MFMA propagate + grouped update), one step"""
    run_full(1024, 150, 1, [0, 1023])


def test_two_service_waves_full_batch():
    """B=1024, N=64: the <6,6> instance with the body lanes on a second service wave (N + 14 > 64)"""
    run_full(1024, 64, 2, [0, 255, 256, 1023])


def test_two_filters_per_workgroup_sizes_between():
    """B=1024 at feature counts either side of the instance boundaries"""
    run_full(1024, 26, 2, [0, 300, 1023])     # <4,3>
    run_full(1024, 38, 2, [0, 511, 1023])     # <4,3>, its last size
    run_full(1024, 39, 2, [0, 256, 1023])     # <5,3>
    run_full(1024, 43, 2, [0, 255, 1023])     # <5,3>, its last size
    run_full(1024, 44, 2, [0, 700, 1023])     # <6,3>
    run_full(1024, 47, 2, [0, 256, 1023])     # <6,3>, its last size
    run_full(1024, 48, 2, [0, 256, 1023])     # <7,3>, its first
    run_full(1024, 49, 2, [0, 512, 1023])


@pytest.mark.parametrize("N", [1, 3, 12, 15])
def test_four_per_cu_instance_small_filters(N):
    """B=1024 > 2 x 256 CUs, N <= 15: the <2,1> instance (ONE worker wave + the service wave, four 128-thread workgroups per CU;
    the reference's own sizes: NUM_FEATURES 12, include/vi_ekf.h:39-45)"""
    run_full(1024, N, 3, [0, 1, 255, 256, 511, 512, 767, 1022, 1023])


def test_features_on_both_service_waves_full_batch():
    """B=1024, N=70: the <7,6> instance (features 64.. on the body wave's lanes; the measurement list crosses the 64 per launch)"""
    run_full(1024, 70, 2, [0, 255, 256, 1023])


@pytest.mark.parametrize("N", [12, 50, 70])
def test_general_lambda_instances_full_batch(N):
    """the same batches on the general-Lambda instances (what a parameter file with lambda_feat[0:2] != 1 selects)"""
    from vi_ekf_amd import capi
    g = run_full(1024, N, 2, [0, 255, 256, 1023], tune=[(capi.TUNE_UNIT_LAMBDA, 0)])
    assert " ZU" not in g.describe()


@pytest.mark.parametrize("N,group", [(85, 0), (85, 16), (85, 24), (100, 0), (100, 16), (90, 24), (85, -1), (100, -1), (160, 0)])
def test_wide_p_group_sizes(N, group):
    """the grouped update at the sizes between the families: the look-ahead kernel (groups of 16, the default where its LDS layout
    fits: N <= 154), r03's kernel with forced group sizes 24 / 32 and with its own choice (group = -1: VIEKF_TUNE_PANEL_SERVICE = 0
    -> 32 at N = 85, 24 at N = 100), N = 160 where only r03's kernel fits; more measurements than one group: x, P (whole, mirrored
    from the lower triangle) and codes against the oracle, P == P^T bit for bit"""
    from vi_ekf_amd import capi
    tune = []
    if group > 0:
        tune.append((capi.TUNE_BLOCK_GROUP, group))
    if group < 0:
        tune.append((capi.TUNE_PANEL_SERVICE, 0))
    g = run_full(64, N, 2, [0, 31, 63], tune=tune)
    d = g.describe()
    if group in (0, 16) and N <= 154:
        assert "k_update_feat_panelsvc<512,16>" in d, d
    elif group > 0:
        assert "k_update_feat_blocked<512,%d>" % group in d, d
    elif N == 160:
        assert "k_update_feat_blocked<512,16>" in d, d
    else:
        assert "k_update_feat_blocked<512,%d>" % (32 if N == 85 else 24) in d, d
