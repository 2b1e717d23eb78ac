"""The SE(3) algebra of the global node frame as restated in oracle/seq_oracle.py (reference vi_ekf_kfr.cpp:14-53,147-150;
convention stated in include/viekf.h because the reference's Xformd lives in its absent `geometry` submodule)."""
import numpy as np

from oracle import np_twin as tw
from oracle import oracle as orc
from oracle import seq_oracle as so


def _rand_T(r):
    q = r.normal(size=4)
    return np.concatenate([r.normal(size=3), q / np.linalg.norm(q)])


def test_composition_is_a_group_action_consistent_with_rota():
    r = np.random.default_rng(0)
    I = np.array([0.0, 0, 0, 1, 0, 0, 0])
    for _ in range(20):
        A, B, C = _rand_T(r), _rand_T(r), _rand_T(r)
        np.testing.assert_allclose(so.xform_mul(so.xform_mul(A, B), C), so.xform_mul(A, so.xform_mul(B, C)), atol=1e-12)
        np.testing.assert_allclose(so.xform_mul(I, A), A, atol=0)
        np.testing.assert_allclose(so.xform_mul(A, I), A, atol=0)
        # a point p given in the child frame maps to  t + q.rota(p)  in the parent frame: composing must chain that map
        p = r.normal(size=3)
        def act(T, v): return T[:3] + tw.rota(T[3:], v)
        np.testing.assert_allclose(act(so.xform_mul(A, B), p), act(A, act(B, p)), atol=1e-12)


def test_adjoint_shape_and_covariance_accumulation():
    r = np.random.default_rng(1)
    T = _rand_T(r)
    A = so.xform_adj(T)
    R = tw.Rmat(T[3:])
    np.testing.assert_allclose(A[:3, :3], R)
    np.testing.assert_allclose(A[3:, 3:], R)
    np.testing.assert_allclose(A[:3, 3:], tw.skew(T[:3]) @ R)
    assert (A[3:, :3] == 0).all()
    # global covariance of a filter with an identity node = its own [POS, ATT] blocks (vi_ekf_kfr.cpp:23-35,47-53)
    f = orc.OracleFilter(3).init(**orc.EKF_YAML)
    s = so.SeqOracle(f)
    idx = [0, 1, 2, 6, 7, 8]
    np.testing.assert_allclose(s.get_global_cov(), f.P[np.ix_(idx, idx)], atol=0)
    np.testing.assert_allclose(s.get_global_pose(), np.concatenate([f.x[0:3], f.x[6:10]]), atol=0)
    # after a node update the accumulated covariance is symmetric positive semi-definite and the node has moved by the edge
    edge = np.zeros(17)
    edge[0:3] = [1.0, -2.0, 0.5]
    edge[3:7] = tw.qexp(np.array([0, 0, 0.7]))
    edge[7:16] = (np.diag([0.1, 0.2, 0.3])).ravel(order="F")
    edge[16] = 0.05
    s._node_update(edge)
    np.testing.assert_allclose(s.node[:3], edge[:3])
    s._node_update(edge)
    np.testing.assert_allclose(s.node[:3], edge[:3] + tw.rota(edge[3:7], edge[:3]), atol=1e-14)
    assert np.allclose(s.node_cov, s.node_cov.T) and np.linalg.eigvalsh(s.node_cov).min() > -1e-12
