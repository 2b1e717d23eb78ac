"""Host-side invariants of the fused-step kernel's block ownership map (vi_ekf_amd/csrc/viekf_capi.hip: build_resmap).

The kernel keeps one 3x3 block of each symmetric pair {I, J} of feature blocks of P (vi_ekf.cpp:302-304 acts on all of P; the
mirror is implied) in a register slot of one worker thread; which one is a table the host builds.  A wrong table would drop
or duplicate a block of the covariance, so its invariants are checked here for every instance and feature count the dispatch
table of the library uses -- pure host arithmetic, no GPU.
"""
import ctypes as C

import numpy as np
import pytest

from vi_ekf_amd import capi

# (RB, NW, n_min, n_max) of kResInst in viekf_capi.hip
INSTANCES = [(2, 1, 1, 15), (2, 2, 1, 22), (3, 2, 1, 25), (1, 7, 1, 29), (2, 7, 30, 41), (4, 3, 26, 38), (5, 3, 39, 43), (6, 3, 44, 47), (7, 3, 26, 50), (3, 7, 1, 50), (5, 6, 51, 57), (6, 6, 51, 67), (7, 6, 65, 72), (8, 6, 73, 77)]


def build(n, rb, nw):
    lib = capi.lib()
    fn = lib.viekf_debug_build_resmap
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    fn.restype = C.c_int
    out = np.full(rb * 64 * nw, -1, dtype=np.int32)
    rc = fn(n, rb, nw, C.c_void_p(out.ctypes.data))
    return rc, out.reshape(rb, 64 * nw)


@pytest.mark.parametrize("rb,nw,n_min,n_max", INSTANCES)
def test_every_pair_once_diagonal_in_slot_zero(rb, nw, n_min, n_max):
    for n in range(n_min, n_max + 1):
        rc, m = build(n, rb, nw)
        assert rc > 0, "N=%d does not fit <%d,%d>" % (n, rb, nw)
        assert not ((m[rc:] >> 16) != 0).any()               # nothing owned past the slots reported in use
        owned = (m >> 16) != 0
        I, J = m & 0xFF, (m >> 8) & 0xFF
        assert (m[~owned] == 0).all()                       # unowned places read block (0, 0) and never store
        assert owned.sum() == n * (n + 1) // 2
        assert (I[owned] >= J[owned]).all() and (I[owned] < n).all()
        pairs = set(zip(I[owned].tolist(), J[owned].tolist()))
        assert len(pairs) == n * (n + 1) // 2                # every unordered pair exactly once
        t = np.arange(n)
        assert owned[0, :n].all() and (I[0, :n] == t).all() and (J[0, :n] == t).all()   # the kernel's own_diag convention
        offdiag = owned.copy()
        offdiag[0, :n] = False
        assert (I[offdiag] > J[offdiag]).all()


def test_headline_map_publishes_from_few_groups():
    """N = 50 on three worker waves: no wave holds blocks of one feature's column in more than 3 of its 7 slots"""
    rc, m = build(50, 7, 3)
    assert rc == 7
    owned = (m >> 16) != 0
    I, J = m & 0xFF, (m >> 8) & 0xFF
    worst = 0
    for f in range(50):
        for w in range(3):
            sl = slice(64 * w, 64 * w + 64)
            hit = (owned[:, sl] & ((I[:, sl] == f) | (J[:, sl] == f))).any(axis=1).sum()
            worst = max(worst, int(hit))
    assert worst <= 3


def test_too_many_blocks_is_refused():
    rc, _ = build(52, 7, 3)
    assert rc < 0
