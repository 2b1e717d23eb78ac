"""The reference's five jac_test properties, restated with FIXED seeds against the CPU oracle.

Mirrors /root/reference/test/jac_test.cpp (manifold :245-280, dfdx :306-365, dfdu :367-415,
h_test :417-443, KF_reset :446-487) with the same tolerances.  These properties are what pins
the oracle's conventions (active/passive rotation, right-[+] attitude, left-[+] bearing, T_zeta):
the reference has no stored vectors ("parity unpinned").  The checks themselves live in tests/properties.py
and also run against the device (tests/test_gpu_properties.py).
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import properties as prop
from tests.helpers import make_oracle


class OracleAdapter:
    """the filter interface of tests/properties.py over oracle.OracleFilter"""

    def __init__(self, N, params, pix, depth):
        self.f = make_oracle(N, params, pix, depth)
        self.x, self.n, self.len_features = self.f.x, self.f.n, self.f.len_features

    def boxplus(self, x, dx): return self.f.boxplus(x, dx)
    def boxminus(self, x1, x2): return self.f.boxminus(x1, x2)
    def dynamics(self, x, u): return self.f.dynamics(x, u)
    def h(self, mtype, x, id): return self.f.h(mtype, x, id)
    def set_drag_term(self, on): self.f.set_drag_term(on)

    def reset_map(self, xm):
        g = self.f.clone()
        g.x[:] = xm
        g.keyframe_reset()
        return g.x.copy(), g.A.copy()


@pytest.mark.parametrize("N", prop.NS)
def test_manifold(N):
    prop.check_manifold(OracleAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_dfdx(N):
    prop.check_dfdx(OracleAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_dfdu(N):
    prop.check_dfdu(OracleAdapter, N)


@pytest.mark.parametrize("N", prop.NS)
def test_h(N):
    prop.check_h(OracleAdapter, N)


@pytest.mark.parametrize("N", [3])
def test_kf_reset(N):
    prop.check_kf_reset(OracleAdapter, N)


def test_seq_oracle_delayed_measurement_equals_in_order_when_input_is_unrotated():
    """host-plumbing restatement (oracle/seq_oracle.py): with q_b_u = identity the rewind/replay of a delayed measurement
    reproduces the filter that received the same measurement in order (the reference's replay rotates the stored, already
    rotated input a second time -- vi_ekf.cpp:265-271 -- so the two differ for a non-trivial q_b_u)."""
    from oracle import seq_oracle as so
    p = dict(orc.EKF_YAML)
    p["q_b_u"] = [1.0, 0.0, 0.0, 0.0]
    N = 3
    rng = np.random.default_rng(2)
    pix = rng.uniform(150, 450, (N, 2))
    us = [np.array([0, 0, -9.80665, 0, 0, 0.0]) + rng.normal(0, 0.2, 6) for _ in range(30)]
    zs = {k: pix + rng.normal(0, 0.5, (N, 2)) for k in range(40)}
    R = np.eye(2) * 10.0

    def run(delay_steps):
        s = so.SeqOracle(orc.OracleFilter(N).init(**p), 0.8, state_hist=32)
        for k in range(33):
            s.propagate_state(us[k % 30], 0.004 * k)
            if k == 0:                        # features are initialised at the state of their first arrival: same in both runs
                for i in range(N):
                    assert s.add_measurement(0.0, pix[i], orc.FEAT, R, True, i, float("nan")) == orc.MEAS_NEW_FEATURE
            kk = k - delay_steps
            if kk >= 1 and kk % 5 == 2:      # the frame taken at step kk arrives delay_steps later
                for i in range(N):
                    s.add_measurement(0.004 * kk, zs[kk][i], orc.FEAT, R, True, i, float("nan"))
                s.handle_measurements()
        assert not s.log, s.log
        return s

    a, b = run(0), run(3)
    assert a.f.len_features == b.f.len_features == N
    assert np.abs(a.f.x - b.f.x).max() < 1e-9 and np.abs(a.f.P - b.f.P).max() < 1e-9
