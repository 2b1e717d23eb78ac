"""The oracle's ANALYTIC Jacobians against the symbolic derivation of tests/golden/derive_jacobians.py.

tests/golden/jac_sym_N*.npz hold A = d x~'/d x~, G = d x~'/d eta and the pixel-model H obtained by
sympy differentiation of the error-state dynamics built from the MODEL (state rates, [+], [-], h) --
not from the reference's analytic blocks -- and confirmed against the exact f_tilde of
test/jac_test.cpp:283-304 in 120-digit arithmetic.  The C oracle and the numpy twin restate the
reference's analytic blocks (vi_ekf_dyn.cpp:55-79,121-132, vi_ekf_meas.cpp:366); the reference itself
only checks them by finite differences at 1e-2 ... 5e-1 (test/jac_test.cpp:306-443).  Here every block
must agree at 1e-9 (relative to the block scale): a transcription slip -- or a first-order-only block in
the reference -- would show as a gap.  Measured gap: <= 4e-15 on every block (nothing to record).
"""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import make_oracle, make_twin

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-9


def load(N):
    z = np.load(os.path.join(HERE, "golden", "jac_sym_N%d.npz" % N))
    params = {k[2:]: (z[k] if z[k].ndim else z[k].item()) for k in z.files if k.startswith("p_")}
    for k in ("use_drag_term", "use_partial_update", "use_keyframe_reset"):
        params[k] = bool(params[k])
    return z, params


# (row block, column block) pairs the reference fills: body vi_ekf_dyn.cpp:55-79, per feature :121-132
BODY_A = [("POS", "VEL"), ("POS", "ATT"), ("VEL", "VEL"), ("VEL", "ATT"), ("VEL", "B_A"), ("VEL", "B_G"), ("VEL", "MU"),
          ("ATT", "ATT"), ("ATT", "B_G")]
BLK = dict(POS=(0, 3), VEL=(3, 3), ATT=(6, 3), B_A=(9, 3), B_G=(12, 3), MU=(15, 1))   # test/jac_test.cpp:62-78


def blk(name, i=None):
    if name == "ZETA":
        return 16 + 3 * i, 2
    if name == "RHO":
        return 18 + 3 * i, 1
    return BLK[name]


def check(got, ref, what):
    scale = max(np.abs(ref).max(), 1.0)
    err = np.abs(got - ref).max()
    assert err <= TOL * scale, "%s: analytic vs symbolic differ by %.3e (scale %.3e)" % (what, err, scale)
    return err / scale


@pytest.mark.parametrize("N", [3, 2])
def test_oracle_blocks_equal_symbolic_derivation(N):
    z, params = load(N)
    f = make_oracle(N, params, z["pix"], z["depth"])
    assert (f.x == z["x"]).all(), "the evaluation point is regenerated bit for bit"
    _, A, G = f.dynamics(z["x"], z["u"])
    worst = 0.0
    for rn, cn in BODY_A:
        (r0, rl), (c0, cl) = blk(rn), blk(cn)
        worst = max(worst, check(A[r0:r0 + rl, c0:c0 + cl], z["A"][r0:r0 + rl, c0:c0 + cl], "A[%s,%s]" % (rn, cn)))
    for i in range(N):
        for rn in ("ZETA", "RHO"):
            r0, rl = blk(rn, i)
            for cn in ("VEL", "B_G", "ZETA", "RHO"):
                c0, cl = blk(cn, i)
                worst = max(worst, check(A[r0:r0 + rl, c0:c0 + cl], z["A"][r0:r0 + rl, c0:c0 + cl], "A[%s%d,%s]" % (rn, i, cn)))
            worst = max(worst, check(G[r0:r0 + rl, 3:6], z["G"][r0:r0 + rl, 3:6], "G[%s%d,uG]" % (rn, i)))
    for rn, c0 in (("VEL", 0), ("VEL", 3), ("ATT", 3)):
        r0, rl = blk(rn)
        worst = max(worst, check(G[r0:r0 + rl, c0:c0 + 3], z["G"][r0:r0 + rl, c0:c0 + 3], "G[%s,%d]" % (rn, c0)))
    # ... and nothing outside those blocks: the whole matrices agree (the reference leaves the rest zero)
    check(A, z["A"], "A (whole)")
    check(G, z["G"], "G (whole)")
    for i in range(N):
        _, H = f.h(orc.FEAT, z["x"], i)
        check(H[0:2, :], z["H"][i], "H_feat[%d]" % i)
    assert worst < 1e-12


@pytest.mark.parametrize("N", [3, 2])
def test_twin_blocks_equal_symbolic_derivation(N):
    z, params = load(N)
    t = make_twin(N, params, z["pix"], z["depth"])
    _, A, G = t.dynamics(z["x"], z["u"])
    check(A, z["A"], "twin A")
    check(G, z["G"], "twin G")
    for i in range(N):
        check(t.h_feat(z["x"], i)[1], z["H"][i], "twin H_feat[%d]" % i)


def test_fixture_records_its_own_confirmation():
    for N in (3, 2):
        z, _ = load(N)
        assert float(z["mp_gap"]) < 1e-12   # symbolic vs exact 120-digit f_tilde, written by the deriving script
