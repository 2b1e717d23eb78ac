"""Participation mask and per-filter ring copies of the C ABI (viekf_batch_set_active, _snapshot_filters / _restore_filters:
what lets filters on independent clocks share a batch) and the covariance block read (viekf_batch_get_cov_block)."""
import ctypes as C

import numpy as np
import pytest

import vi_ekf_amd as v
from vi_ekf_amd import capi, scene

pytestmark = pytest.mark.gpu


def _p(a):
    return C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("N,kernel", [(6, 0), (50, 2), (30, 1), (60, 0), (70, 0), (50, 3), (50, 5)])
def test_masked_filters_are_untouched_and_the_others_unchanged(N, kernel):
    B, steps = 6, 2
    sc = scene.make_scene(B, N, steps, seed=9)
    L = capi.lib()

    def make():
        g = v.BatchVIEKF(B, N, sc["params"])
        from tests.helpers import apply_kernel
        apply_kernel(g, kernel)
        for i in range(N):
            g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
        return g

    ga, gb = make(), make()
    x0, P0 = ga.get_state(), ga.get_covariance()
    mask = np.array([1, 0, 1, 1, 0, 1], dtype=np.uint8)
    capi.check(L.viekf_batch_set_active(ga._h, _p(mask), capi.HOST))
    for s in range(steps):
        ga.propagate(sc["u"][s], sc["dt"])
        ga.update_feat(sc["z"][s], sc["slot"], sc["R"])
        ga.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        gb.propagate(sc["u"][s], sc["dt"])
        gb.update_feat(sc["z"][s], sc["slot"], sc["R"])
        gb.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
    capi.check(L.viekf_batch_set_active(ga._h, None, capi.HOST))
    xa, Pa, xb, Pb = ga.get_state(), ga.get_covariance(), gb.get_state(), gb.get_covariance()
    on, off = mask.astype(bool), ~mask.astype(bool)
    assert np.array_equal(xa[off], x0[off]) and np.array_equal(Pa[off], P0[off])          # masked out: bit for bit untouched
    assert np.array_equal(xa[on], xb[on]) and np.array_equal(Pa[on], Pb[on])              # the others: as without a mask
    assert not np.array_equal(xa[on], x0[on])


def test_per_filter_ring_copies_and_cov_block():
    B, N = 4, 8
    sc = scene.make_scene(B, N, 3, seed=4)
    g = v.BatchVIEKF(B, N, sc["params"])
    L = capi.lib()
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    g.history_resize(5)
    states = []
    for s in range(3):
        g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"])
        states.append((g.get_state(), g.get_covariance()))
        slot = np.array([s, 4 - s, -1, s], dtype=np.int32)            # every filter its own slot; filter 2 records nothing
        capi.check(L.viekf_batch_snapshot_filters(g._h, _p(slot), capi.HOST))
    # rewind: filter 0 to step 0, filter 1 to step 2 (slot 2), filter 2 stays, filter 3 to step 1
    slot = np.array([0, 2, -1, 1], dtype=np.int32)
    capi.check(L.viekf_batch_restore_filters(g._h, _p(slot), capi.HOST))
    x, P = g.get_state(), g.get_covariance()
    for b, s in ((0, 0), (1, 2), (2, 2), (3, 1)):
        assert np.array_equal(x[b], states[s][0][b]) and np.array_equal(P[b], states[s][1][b])
    bad = np.array([0, 9, 0, 0], dtype=np.int32)
    assert L.viekf_batch_restore_filters(g._h, _p(bad), capi.HOST) == capi.ERR_INVALID
    # a block of P: rows 6..8 (attitude) x columns 0..2 (position), column-major per filter
    blk = np.zeros((B, 3, 3))
    capi.check(L.viekf_batch_get_cov_block(g._h, 6, 0, 3, 3, _p(blk), capi.HOST))
    assert np.array_equal(blk.transpose(0, 2, 1), P[:, 6:9, 0:3])
    assert L.viekf_batch_get_cov_block(g._h, 0, 0, 3, 16 + 3 * N + 1, _p(blk), capi.HOST) == capi.ERR_INVALID


@pytest.mark.parametrize("N,kernel", [(30, 1), (90, 0)])
def test_masked_propagate_keeps_the_other_filters_staleness(N, kernel):
    """Streaming family: a grouped update of filters A leaves their upper triangle stale; a masked propagate of the DISJOINT
    filters B must not clear that (it did, and every whole-P reader then took A's stale upper triangle as valid).
    Compared with the same calls made on two separate batches of the filters A and B."""
    B, M = 6, 5
    sc = scene.make_scene(B, N, 2, seed=21)
    L = capi.lib()

    def make(sel):
        p = sc["params"]
        g = v.BatchVIEKF(len(sel), N, p)
        if kernel:
            g.set_kernel(kernel)
        for i in range(N):
            g.init_feature(sc["pix"][sel, i, :].copy(), np.full(len(sel), np.nan))
        return g

    allf = np.arange(B)
    A, Bm = np.array([0, 2, 3]), np.array([1, 4, 5])
    mA = np.zeros(B, dtype=np.uint8); mA[A] = 1
    mB = np.zeros(B, dtype=np.uint8); mB[Bm] = 1
    g = make(allf)
    g.propagate(sc["u"][0], sc["dt"])                                    # everybody: P no longer diagonal
    capi.check(L.viekf_batch_set_active(g._h, _p(mA), capi.HOST))
    g.update_feat(sc["z"][0][:, :M], sc["slot"][:, :M], sc["R"])          # A: grouped update (M >= 2)
    capi.check(L.viekf_batch_set_active(g._h, _p(mB), capi.HOST))
    g.propagate(sc["u"][1], sc["dt"])                                    # B only
    capi.check(L.viekf_batch_set_active(g._h, None, capi.HOST))
    P = g.get_covariance()
    x = g.get_state()
    assert (P == P.transpose(0, 2, 1)).all(), "a stale upper triangle was handed out"
    # the same history on separate batches
    ga, gb = make(A), make(Bm)
    ga.propagate(sc["u"][0][A], sc["dt"][A]); gb.propagate(sc["u"][0][Bm], sc["dt"][Bm])
    ga.update_feat(sc["z"][0][A][:, :M], sc["slot"][A][:, :M], sc["R"])
    gb.propagate(sc["u"][1][Bm], sc["dt"][Bm])
    assert np.array_equal(P[A], ga.get_covariance()) and np.array_equal(x[A], ga.get_state())
    assert np.array_equal(P[Bm], gb.get_covariance()) and np.array_equal(x[Bm], gb.get_state())
    # ... and a generic update (reads all of P) after the same sequence agrees as well
    z = np.tile(np.array([0.1, -0.2, 0.3]), (B, 1))
    Rp = np.eye(3) * 0.01
    g.update(capi_type("POS"), z, Rp)
    ga.update(capi_type("POS"), z[A], Rp)
    assert np.array_equal(g.get_covariance()[A], ga.get_covariance())


def capi_type(name):
    return {"ACC": 0, "ALT": 1, "ATT": 2, "POS": 3, "VEL": 4, "QZETA": 5, "FEAT": 6, "DEPTH": 8, "INV_DEPTH": 9}[name]


def test_propagate_to_refuses_a_participation_mask_and_device_slots_are_checked():
    import torch
    B, N = 4, 8
    sc = scene.make_scene(B, N, 1, seed=5)
    g = v.BatchVIEKF(B, N, sc["params"])
    L = capi.lib()
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    g.history_resize(3)
    capi.check(L.viekf_batch_select(g._h, 0))
    mask = np.array([1, 0, 1, 1], dtype=np.uint8)
    capi.check(L.viekf_batch_set_active(g._h, _p(mask), capi.HOST))
    u, dt = np.ascontiguousarray(sc["u"][0]), np.ascontiguousarray(sc["dt"])
    assert L.viekf_batch_propagate_to(g._h, _p(u), _p(dt), 1, capi.HOST) == capi.ERR_INVALID
    capi.check(L.viekf_batch_set_active(g._h, None, capi.HOST))
    capi.check(L.viekf_batch_propagate_to(g._h, _p(u), _p(dt), 1, capi.HOST))
    capi.check(L.viekf_batch_select(g._h, -1))
    # device-resident slot list with an out-of-range entry: nothing is copied for that filter, its INTERNAL flag is raised
    x0, P0 = g.get_state(), g.get_covariance()
    slots = torch.tensor([0, 7, -1, 1], dtype=torch.int32, device="cuda:0")
    capi.check(L.viekf_batch_restore_filters(g._h, C.c_void_p(slots.data_ptr()), capi.DEVICE))
    g.sync()
    st = g.get_status()
    assert st[1] & capi.FLAG_INTERNAL and not (st[[0, 2, 3]] & capi.FLAG_INTERNAL).any()
    x1, P1 = g.get_state(), g.get_covariance()
    assert np.array_equal(x1[1], x0[1]) and np.array_equal(P1[1], P0[1]) and np.array_equal(x1[2], x0[2])
