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


@pytest.mark.parametrize("N,kernel", [(6, 0), (50, 2), (30, 1), (60, 0), (70, 0)])
def test_masked_filters_are_untouched_and_the_others_unchanged(N, kernel):
    B, steps = 6, 2
    sc = scene.make_scene(B, N, steps, seed=9)
    L = capi.lib()

    def make():
        g = v.BatchVIEKF(B, N, sc["params"])
        if kernel:
            g.set_kernel(kernel)
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
