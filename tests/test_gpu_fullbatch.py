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


@pytest.mark.parametrize("N,group", [(85, 0), (85, 16), (85, 24), (100, 0), (100, 16), (90, 24)])
def test_wide_p_group_sizes(N, group):
    """the grouped update's 24- and 32-measurement instances (chosen on their own for n > 256 where the panel fits: N = 85 ->
    32, N = 100 -> 24) and forced group sizes, more measurements than one group: x, P (whole, mirrored from the lower
    triangle) and codes against the oracle, P == P^T bit for bit (those instances had no parity case)"""
    from vi_ekf_amd import capi
    g = run_full(64, N, 2, [0, 31, 63], tune=[(capi.TUNE_BLOCK_GROUP, group)] if group else ())
    d = g.describe()
    assert "k_update_feat_blocked<512,%d>" % (group if group else (32 if N == 85 else 24)) in d, d


# ---- the OTHER routes bench.py times, at the size it times them (VERDICT r03 "weak" #2): the multi-propagate instance of the
# headline kernel (viekf_batch_step_n / _propagate_n_to) and the out-of-place ring store (P_out != P), B = 1024, N = 50, filters
# of every dispatch round against the ORACLE (K vo_propagate calls + the updates; reference vi_ekf.cpp:262-318,
# vi_ekf_meas.cpp:196-278), not against another HIP route.
HEADLINE_SAMPLE = [0, 255, 256, 511, 700, 1023]


def _headline_batch(steps, seed):
    B, N = 1024, 50
    sc = scene.make_scene(B, N, steps, seed=seed)
    g = v.BatchVIEKF(B, N, sc["params"])
    for i in range(N):
        assert (g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan)) == 1).all()
    assert g.describe().startswith("k_step_resident<7,3>"), g.describe()
    return sc, g


def _oracle_filters(sc, N, which):
    fs = []
    for b in which:
        f = orc.OracleFilter(N).init(**oracle_params(sc["params"]))
        for i in range(N):
            f.init_feature(sc["pix"][b, i], i, float("nan"))
        fs.append(f)
    return fs


def _check_sample(g, fs, which, what):
    x, P = g.get_state(), g.get_covariance()
    assert_close(x[which], np.stack([f.x for f in fs]), "x, " + what)
    assert_close(P[which], np.stack([f.P for f in fs]), "P, " + what)
    assert (g.get_status() & (1 | 2 | 8) == 0).all()
    assert np.isfinite(x).all() and np.isfinite(P).all()
    assert (P == P.transpose(0, 2, 1)).all(), "P != P^T somewhere in the batch, " + what


@pytest.mark.parametrize("K", [8, 9])
def test_step_n_headline_batch_vs_oracle(K):
    """viekf_batch_step_n (cadence_250_30.fused in bench.py): K IMU samples + the frame's 50 updates in ONE launch of the
    multi-propagate instance k_step_resident<7,3,MP>, two frames in a row, with uneven dt"""
    which = np.asarray(HEADLINE_SAMPLE)
    sc, g = _headline_batch(2 * K, 7100 + K)
    fs = _oracle_filters(sc, 50, which)
    dt = np.tile(sc["dt"], (K, 1)) * np.linspace(0.8, 1.2, K)[:, None]
    for fr in range(2):
        u = np.ascontiguousarray(sc["u"][fr * K:(fr + 1) * K])
        res = g.step_n(u, dt, sc["z"][fr], sc["slot"], sc["R"])
        for j, b in enumerate(which):
            for k in range(K - 1):
                fs[j].propagate(u[k, b], dt[k, b])
            ref = fs[j].run_steps(u[K - 1, b][None], dt[K - 1, b], sc["z"][fr, b][None], sc["slot"][b], sc["R"])[0]
            assert (res[b] == ref).all(), "meas_result codes differ (filter %d, frame %d)" % (b, fr)
    _check_sample(g, fs, which, "step_n K=%d" % K)


def test_propagate_n_to_headline_batch_vs_oracle():
    """viekf_batch_propagate_n_to (the sequencer's closing replay): K = 8 propagates into ring slots 1..8 in one launch of the
    same instance, only the last slot written; then select + get_state, then the frame's updates IN that slot"""
    import ctypes as C
    from vi_ekf_amd import capi
    K = 8
    which = np.asarray(HEADLINE_SAMPLE)
    sc, g = _headline_batch(K, 7200)
    fs = _oracle_filters(sc, 50, which)
    L = capi.lib()
    g.history_resize(K + 2)
    g.snapshot(0)
    capi.check(L.viekf_batch_select(g._h, 0))
    u = np.ascontiguousarray(sc["u"][:K])
    dt = np.ascontiguousarray(np.tile(sc["dt"], (K, 1)) * (1.0 + 0.05 * np.arange(K))[:, None])
    slots = np.arange(1, K + 1, dtype=np.int32)
    written = C.c_int32(-1)
    p = lambda a: C.c_void_p(a.ctypes.data)
    capi.check(L.viekf_batch_propagate_n_to(g._h, K, p(u), p(dt), p(slots), C.byref(written), capi.HOST))
    assert written.value == 0
    for j, b in enumerate(which):
        for k in range(K):
            fs[j].propagate(u[k, b], dt[k, b])
    _check_sample(g, fs, which, "propagate_n_to K=8, slot 8")
    res = g.update_feat(sc["z"][0], sc["slot"], sc["R"])
    for j, b in enumerate(which):
        for m in range(50):
            r = fs[j].update(orc.FEAT, sc["z"][0, b, m], sc["R"], True, int(sc["slot"][b, m]))
            assert r == res[b, m]
    _check_sample(g, fs, which, "updates in ring slot 8")
    g.history_resize(0)
    _check_sample(g, fs, which, "after leaving the ring")


def test_propagate_to_ring_slot_headline_batch_vs_oracle():
    """viekf_batch_propagate_to: the out-of-place store of the fused kernel (P read from slot i, written to slot i + 1,
    viekf_resident_worker.hpp: P_out != P) through three slots at B = 1024, N = 50; the slot left behind still holds its state"""
    import ctypes as C
    from vi_ekf_amd import capi
    which = np.asarray(HEADLINE_SAMPLE)
    sc, g = _headline_batch(3, 7300)
    fs = _oracle_filters(sc, 50, which)
    L = capi.lib()
    g.history_resize(4)
    g.snapshot(0)
    capi.check(L.viekf_batch_select(g._h, 0))
    p = lambda a: C.c_void_p(a.ctypes.data)
    dt = np.ascontiguousarray(sc["dt"])
    kept = None
    for k in range(3):
        u = np.ascontiguousarray(sc["u"][k])
        capi.check(L.viekf_batch_propagate_to(g._h, p(u), p(dt), k + 1, capi.HOST))
        for j, b in enumerate(which):
            fs[j].propagate(u[b], dt[b])
        _check_sample(g, fs, which, "propagate_to slot %d" % (k + 1))
        if k == 0:
            kept = (np.stack([f.x.copy() for f in fs]), np.stack([f.P.copy() for f in fs]))
    capi.check(L.viekf_batch_select(g._h, 1))       # rewind: slot 1 was read by the second call, never written again
    assert_close(g.get_state()[which], kept[0], "x of slot 1 after two more propagates")
    assert_close(g.get_covariance()[which], kept[1], "P of slot 1 after two more propagates")
