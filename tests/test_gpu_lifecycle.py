"""Feature lifecycle and history on the device: keep_only_features / clear_feature compaction
(reference src/vi_ekf/vi_ekf_feat.cpp:50-117) against the oracle, and the snapshot ring used for rewinds."""
import numpy as np
import pytest

import vi_ekf_amd as v
from oracle import oracle as orc
from vi_ekf_amd import scene
from tests.test_gpu_parity import assert_close, oracle_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6, 12, 80])
def test_keep_features_matches_clear_feature(N):
    B = 3
    sc = scene.make_scene(B, N, 3, seed=80 + N)
    g = v.BatchVIEKF(B, N, sc["params"])
    fs = [orc.OracleFilter(N).init(**oracle_params(sc["params"])) for _ in range(B)]
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        for b in range(B):
            fs[b].init_feature(sc["pix"][b, i], i)
    for s in range(2):
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        for b in range(B):
            fs[b].run_steps(sc["u"][s, b][None], sc["dt"][b], sc["z"][s, b][None], sc["slot"][b], sc["R"])
    keep = np.ones((B, N), dtype=np.uint8)
    keep[0, [1, 4]] = 0            # drop two in the middle
    keep[1, N - 1] = 0             # drop the last
    # filter 2 keeps everything
    nl = g.keep_features(keep)
    for b in range(B):
        for fid in [i for i in range(N) if not keep[b, i]]:
            fs[b].clear_feature(fid)        # ids == initial slot numbers (init_feature counts from 0)
    assert (nl == [N - 2, N - 1, N]).all() and (g.get_len_features() == nl).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x after keep_features")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P after keep_features")
    # the compacted filters keep running: re-initialise one feature and do another step on the survivors
    ok = g.init_feature(sc["pix"][:, 0, :].copy() + 3.0, np.full(B, 4.0))
    for b in range(B):
        assert bool(ok[b]) == fs[b].init_feature(sc["pix"][b, 0] + 3.0, 99, 4.0)
    M = N - 2
    z = np.ascontiguousarray(sc["z"][2][:, :M, :])
    slot = np.tile(np.arange(M - 1, -1, -1, dtype=np.int32), (B, 1))
    res = g.step(sc["u"][2], sc["dt"], z, slot, sc["R"])
    for b in range(B):
        ref = fs[b].run_steps(sc["u"][2, b][None], sc["dt"][b], z[b][None], slot[b], sc["R"])[0]
        assert (res[b] == ref).all()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x after step on compacted filters")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P after step on compacted filters")


def test_snapshot_restore_replays_identically():
    B, N = 4, 12
    sc = scene.make_scene(B, N, 4, seed=91)
    g = v.BatchVIEKF(B, N, sc["params"])
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    g.history_resize(2)
    g.step(sc["u"][0], sc["dt"], sc["z"][0], sc["slot"], sc["R"])
    g.snapshot(1)
    x1, P1 = g.get_state(), g.get_covariance()
    for s in (1, 2):
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    x3, P3 = g.get_state(), g.get_covariance()
    g.restore(1)                                   # rewind (reference vi_ekf_meas.cpp:45-63) ...
    assert np.array_equal(g.get_state(), x1) and np.array_equal(g.get_covariance(), P1)
    for s in (1, 2):                               # ... and replay: bit-identical
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    assert np.array_equal(g.get_state(), x3) and np.array_equal(g.get_covariance(), P3)
    with pytest.raises(v.ViekfError):
        g.snapshot(2)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [5, 50, 80])
def test_keyframe_reset_matches_oracle(N):
    """device keyframe reset (state, N P N^T, edge) vs the oracle's restatement of vi_ekf_kfr.cpp:56-157"""
    import vi_ekf_amd as v
    from vi_ekf_amd import scene
    from oracle import oracle as orc
    from tests.test_gpu_parity import assert_close
    B = 3
    sc = scene.make_scene(B, N, 3, seed=11)
    prm = {k: sc["params"][k] for k in sc["params"] if k not in ("name", "keyframe_overlap_threshold")}
    g = v.BatchVIEKF(B, N, sc["params"])
    fs = [orc.OracleFilter(N).init(**prm) for _ in range(B)]
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        for b in range(B):
            fs[b].init_feature(sc["pix"][b, i], i, float("nan"))
    R = np.asarray(sc["R"]).reshape(2, 2)
    for s in range(3):   # a few steps so that attitude, position and P are generic
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        for b in range(B):
            fs[b].propagate(sc["u"][s][b], float(sc["dt"][b]))
            for k in range(N):
                fs[b].update(orc.FEAT, sc["z"][s][b, k], R, True, int(sc["slot"][b, k]))
    mask = np.array([1, 0, 1], dtype=np.uint8)
    edge = g.keyframe_reset(mask)
    ref_edge = np.zeros((B, 17))
    for b in range(B):
        if mask[b]:
            ref_edge[b] = fs[b].keyframe_reset_edge()
    assert_close(g.get_state(), np.stack([f.x for f in fs]), "x after keyframe reset")
    assert_close(g.get_covariance(), np.stack([f.P for f in fs]), "P after keyframe reset")
    assert_close(edge, ref_edge, "edge")
    assert np.all(g.get_state()[mask == 1, 0:3] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("N,kernel", [(12, 0), (12, 1), (50, 0), (85, 0), (50, 3), (50, 5)])
def test_propagate_to_ring_slot_equals_in_place(N, kernel):
    """zero-copy history: propagate_to writes the next ring slot and selects it; the old slot keeps the old state"""
    import ctypes as C
    from vi_ekf_amd import capi
    B = 3
    sc = scene.make_scene(B, N, 2, seed=13)
    ga, gb = v.BatchVIEKF(B, N, sc["params"]), v.BatchVIEKF(B, N, sc["params"])
    for g in (ga, gb):
        from tests.helpers import apply_kernel
        apply_kernel(g, kernel)
        for i in range(N):
            g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    L = capi.lib()
    gb.history_resize(3)
    gb.snapshot(0)
    capi.check(L.viekf_batch_select(gb._h, 0))
    x0, P0 = gb.get_state(), gb.get_covariance()
    u = np.ascontiguousarray(sc["u"][0]); dt = np.ascontiguousarray(sc["dt"])
    ga.propagate(u, dt)
    capi.check(L.viekf_batch_propagate_to(gb._h, C.c_void_p(u.ctypes.data), C.c_void_p(dt.ctypes.data), 1, capi.HOST))
    assert np.array_equal(ga.get_state(), gb.get_state()) and np.array_equal(ga.get_covariance(), gb.get_covariance())
    ga.update_feat(sc["z"][0], sc["slot"], sc["R"]); gb.update_feat(sc["z"][0], sc["slot"], sc["R"])   # in the new slot
    assert np.array_equal(ga.get_covariance(), gb.get_covariance())
    capi.check(L.viekf_batch_select(gb._h, 0))       # rewind without a copy: slot 0 still holds the state before
    assert np.array_equal(gb.get_state(), x0) and np.array_equal(gb.get_covariance(), P0)
    capi.check(L.viekf_batch_select(gb._h, 1))
    assert np.array_equal(ga.get_covariance(), gb.get_covariance())
    gb.history_resize(0)                            # leaving the ring brings the live state home
    assert np.array_equal(ga.get_state(), gb.get_state()) and np.array_equal(ga.get_covariance(), gb.get_covariance())


@pytest.mark.gpu
@pytest.mark.parametrize("N,kernel,K", [(12, 0, 4), (50, 0, 8), (12, 1, 3), (90, 0, 3), (50, 3, 5)])
def test_propagate_n_to_equals_k_propagate_to_calls(N, kernel, K):
    """viekf_batch_propagate_n_to (the closing replay of handle_measurements): K propagates into K ring slots.  The fused family
    does it in ONE launch, writes only the last slot (intermediates_written = 0, the others keep their old contents) and ends
    bit for bit where K viekf_batch_propagate_to calls end; the HBM-path family runs the steps one by one and writes every slot.
    Bad arguments (a repeated slot, the live slot, a slot out of range, K out of range) are refused before anything is touched."""
    import ctypes as C
    from vi_ekf_amd import capi
    from tests.helpers import apply_kernel
    B = 3
    sc = scene.make_scene(B, N, K, seed=31 + N)
    ga, gb = v.BatchVIEKF(B, N, sc["params"]), v.BatchVIEKF(B, N, sc["params"])
    L = capi.lib()
    for g in (ga, gb):
        apply_kernel(g, kernel)
        for i in range(N):
            g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        g.history_resize(K + 2)
        g.snapshot(0)
        capi.check(L.viekf_batch_select(g._h, 0))
    u = np.ascontiguousarray(sc["u"][:K]); dt = np.ascontiguousarray(np.tile(sc["dt"], (K, 1)) * (1.0 + 0.1 * np.arange(K))[:, None])
    p = lambda a: C.c_void_p(a.ctypes.data)
    for k in range(K):
        capi.check(L.viekf_batch_propagate_to(ga._h, p(u[k]), p(dt[k]), k + 1, capi.HOST))
    slots = np.arange(1, K + 1, dtype=np.int32)
    written = C.c_int32(-1)
    x0, P0 = gb.get_state(), gb.get_covariance()
    for bad in (np.array([1, 1] + list(range(2, K)), dtype=np.int32)[:K], np.array([0] + list(range(2, K + 1)), dtype=np.int32)[:K],
                np.array(list(range(1, K)) + [K + 2], dtype=np.int32)):
        assert L.viekf_batch_propagate_n_to(gb._h, K, p(u), p(dt), p(bad), C.byref(written), capi.HOST) == capi.ERR_INVALID
    assert L.viekf_batch_propagate_n_to(gb._h, 65, p(u), p(dt), p(slots), C.byref(written), capi.HOST) == capi.ERR_INVALID
    assert np.array_equal(gb.get_state(), x0) and np.array_equal(gb.get_covariance(), P0)
    capi.check(L.viekf_batch_propagate_n_to(gb._h, K, p(u), p(dt), p(slots), C.byref(written), capi.HOST))
    fused = "k_step_" in gb.describe()
    assert written.value == (0 if fused else 1)
    assert np.array_equal(ga.get_state(), gb.get_state()) and np.array_equal(ga.get_covariance(), gb.get_covariance())
    capi.check(L.viekf_batch_select(gb._h, 0))       # the slot the replay started from is untouched
    assert np.array_equal(gb.get_state(), x0) and np.array_equal(gb.get_covariance(), P0)
    if not fused:                                    # every slot written: the same as the step-by-step batch, slot by slot
        for k in range(1, K + 1):
            capi.check(L.viekf_batch_select(ga._h, k)); capi.check(L.viekf_batch_select(gb._h, k))
            assert np.array_equal(ga.get_state(), gb.get_state()) and np.array_equal(ga.get_covariance(), gb.get_covariance())


@pytest.mark.gpu
@pytest.mark.parametrize("N,kernel,K", [(12, 2, 5), (50, 2, 9), (20, 2, 3), (12, 1, 4), (55, 2, 4), (66, 2, 3), (50, 3, 9), (47, 3, 2), (50, 5, 9), (48, 5, 3)])
def test_step_n_equals_k_propagates_and_a_step(N, kernel, K):
    """viekf_batch_step_n: K IMU samples and the frame's updates in one launch (P stays on chip in the fused kernel) --
    bit for bit the K - 1 propagate calls + one step they replace; a forced negative depth makes a propagate's fix_depth
    edit P(rho,rho) in the middle of the sequence"""
    import vi_ekf_amd as v
    from vi_ekf_amd import scene
    B = 300 if N == 20 else 3        # (N = 20 with a batch above the CU count: the two-per-CU instance)
    sc = scene.make_scene(B, N, K, seed=17 + N)
    gs = []
    for _ in range(2):
        g = v.BatchVIEKF(B, N, sc["params"])
        from tests.helpers import apply_kernel
        apply_kernel(g, kernel)
        for i in range(N):
            g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        x = g.get_state()
        x[0, 17 + 4] = -0.2           # rho < 0: fixed (and flagged) by the first propagate
        x[1 % B, 17 + 5 + 4] = 300.0  # rho > 100: reset by the first propagate
        g.set_state(x=x)
        gs.append(g)
    dt = np.tile(sc["dt"], (K, 1)) * np.linspace(0.8, 1.2, K)[:, None]
    ra = gs[0].step_n(sc["u"][:K], dt, sc["z"][0], sc["slot"], sc["R"])
    for k in range(K - 1):
        gs[1].propagate(sc["u"][k], dt[k])
    rb = gs[1].step(sc["u"][K - 1], dt[K - 1], sc["z"][0], sc["slot"], sc["R"])
    assert (ra == rb).all()
    assert np.array_equal(gs[0].get_state(), gs[1].get_state())
    assert np.array_equal(gs[0].get_covariance(), gs[1].get_covariance())
    assert (gs[0].get_status() == gs[1].get_status()).all() and (gs[0].get_status()[0] & 4)
    # K propagates without measurements
    gs[0].step_n(sc["u"][:K], dt, None, None, sc["R"])
    for k in range(K):
        gs[1].propagate(sc["u"][k], dt[k])
    assert np.array_equal(gs[0].get_covariance(), gs[1].get_covariance())


def test_two_batches_of_one_instance_with_different_lds():
    """a second batch with fewer features on the same kernel instance must not shrink the first one's dynamic-LDS allowance
    (the MaxDynamicSharedMemorySize attribute is per kernel, not per batch)"""
    import vi_ekf_amd as v
    from vi_ekf_amd import scene
    B = 600
    sc_a = scene.make_scene(B, 50, 1, seed=5)
    sc_b = scene.make_scene(B, 48, 1, seed=6)
    ga = v.BatchVIEKF(B, 50, sc_a["params"])
    gb = v.BatchVIEKF(B, 48, sc_b["params"])     # same <7,3> instance, smaller LDS image
    for g, sc, N in ((ga, sc_a, 50), (gb, sc_b, 48)):
        for i in range(N):
            g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    ra = ga.step(sc_a["u"][0], sc_a["dt"], sc_a["z"][0], sc_a["slot"], sc_a["R"])
    rb = gb.step(sc_b["u"][0], sc_b["dt"], sc_b["z"][0], sc_b["slot"], sc_b["R"])
    assert (ra == 0).all() and (rb == 0).all()
    assert np.isfinite(ga.get_state()).all() and np.isfinite(gb.get_state()).all()
