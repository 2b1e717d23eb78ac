"""Thread contract of include/viekf.h: calls on ONE batch are serialised by the caller, DIFFERENT batches are independent and
may be driven from different host threads (the reference's contract is one mutex per filter object, include/vi_ekf_ros.h:65).

Two host threads create and step two batches that resolve to the SAME kernel instance with different feature counts (so both
raise that instance's dynamic-LDS attribute -- the per-device high-water marks of viekf_capi.hip), 50 steps each, at the same
time; both must equal their single-threaded runs bit for bit.  ctypes releases the GIL for the duration of every C call.
"""
import threading

import numpy as np
import pytest

import vi_ekf_amd as v
from vi_ekf_amd import scene

pytestmark = pytest.mark.gpu

STEPS = 50


def fly(B, N, seed, barrier=None, out=None, key=None):
    sc = scene.make_scene(B, N, STEPS, seed=seed)
    if barrier is not None:
        barrier.wait()                     # both threads create their batch (attribute set-up) at the same moment
    g = v.BatchVIEKF(B, N, sc["params"])
    for i in range(N):
        g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
    res = []
    for s in range(STEPS):
        res.append(g.step(sc["u"][s], sc["dt"], sc["z"][s], sc["slot"], sc["R"]))
    r = (g.get_state(), g.get_covariance(), np.stack(res), g.get_status(), g.describe())
    g.close()
    if out is not None:
        out[key] = r
    return r


@pytest.mark.parametrize("cfg", [((8, 30, 11), (8, 41, 12)),        # <2,7> and <3,7>: one-per-CU instances, small batches
                                 ((600, 16, 13), (600, 22, 14)),    # both on <2,2>, two workgroups per CU
                                 ((8, 90, 15), (8, 120, 16))])      # grouped wide-P update: groups of 32 and of 24
def test_two_threads_two_batches_equal_their_single_threaded_runs(cfg):
    a, b = cfg
    out = {}
    bar = threading.Barrier(2)
    ts = [threading.Thread(target=fly, args=a + (bar, out, "a")), threading.Thread(target=fly, args=b + (bar, out, "b"))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
        assert not t.is_alive()
    assert set(out) == {"a", "b"}, "a thread died: %s" % list(out)
    for key, c in (("a", a), ("b", b)):
        ref = fly(*c)
        got = out[key]
        assert got[4] == ref[4]
        for k in range(4):
            assert np.array_equal(got[k], ref[k]), "batch %s (B, N, seed = %s): output %d differs from its single-threaded run" % (key, c, k)
