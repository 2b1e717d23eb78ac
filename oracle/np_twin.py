"""Independent numpy restatement of the VI-EKF hot path (second opinion for the C oracle).

TEST INFRASTRUCTURE ONLY.  Written from the reference's equations with rotation
MATRICES and vectorised numpy (not a transliteration of oracle/viekf_oracle.c),
so that an indexing/transposition slip in either restatement shows up as a
disagreement.  "Parity unpinned": the reference offers no golden vectors.

Reference equations followed (paths relative to /root/reference):
  dynamics      src/vi_ekf/vi_ekf_dyn.cpp:14-135
  propagate     src/vi_ekf/vi_ekf.cpp:262-318
  update/h_feat src/vi_ekf/vi_ekf_meas.cpp:196-278, 354-367
  boxplus/minus src/vi_ekf/vi_ekf_helper.cpp:88-111, include/math_helper.h:14-48
  init_feature  src/vi_ekf/vi_ekf_feat.cpp:6-47
  fix_depth     src/vi_ekf/vi_ekf_helper.cpp:128-156
  quaternion conventions: src/quat.cpp (Hamilton [w,x,y,z], R() passive, rota = R^T v, rotp = R v)
"""
import numpy as np

E_Z = np.array([0.0, 0.0, 1.0])
GRAV = np.array([0.0, 0.0, 9.80665])
I23 = np.array([[1.0, 0, 0], [0, 1.0, 0]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def Rmat(q):
    """Passive rotation matrix R_I^b of src/quat.cpp:226-242, via I - 2w[v]x + 2[v]x^2."""
    w, v = q[0], np.asarray(q[1:4])
    S = skew(v)
    return np.eye(3) - 2.0 * w * S + 2.0 * S @ S


def qmul(a, b):
    aw, av = a[0], np.asarray(a[1:4])
    bw, bv = b[0], np.asarray(b[1:4])
    return np.concatenate([[aw * bw - av @ bv], aw * bv + bw * av + np.cross(av, bv)])


def qexp(v):
    th = np.linalg.norm(v)
    if th > 1e-4:  # src/quat.cpp:69
        return np.concatenate([[np.cos(th / 2)], np.sin(th / 2) / th * np.asarray(v)])
    q = np.concatenate([[1.0], np.asarray(v) / 2.0])
    return q / np.linalg.norm(q)


def qlog(q):
    nv = np.linalg.norm(q[1:4])
    if nv < 1e-8:
        return np.zeros(3)
    return 2.0 * np.arctan2(nv, q[0]) * np.asarray(q[1:4]) / nv


def q_boxplus(q, d):
    return qmul(q, qexp(d))


def q_boxminus(q1, q2):
    dq = qmul(np.concatenate([[q2[0]], -np.asarray(q2[1:4])]), q1)
    if dq[0] < 0:
        dq = -dq
    return qlog(dq)


def rota(q, v):
    return Rmat(q).T @ v


def rotp(q, v):
    return Rmat(q) @ v


def T_zeta(q):
    return Rmat(q).T @ I23.T  # 3x2


def q_feat_boxplus(q, dq):
    return qmul(qexp(T_zeta(q) @ dq), q)


def q_feat_boxminus(qj, qi):
    zi, zj = rota(qi, E_Z), rota(qj, E_Z)
    if np.linalg.norm(zi - zj) > 1e-8:
        s = np.cross(zi, zj)
        s = s / np.linalg.norm(s)
        th = np.arccos(zi @ zj)
        return T_zeta(qi).T @ (th * s)
    return np.zeros(2)


def from_two_unit_vectors(u, v):
    d = u @ v
    if d < 1.0:
        s = np.sqrt(2.0 * (1.0 + d))
        q = np.concatenate([[0.5 * s], np.cross(u, v) / s])
        return q / np.linalg.norm(q)
    return np.array([1.0, 0, 0, 0])


class TwinFilter:
    def __init__(self, N, x0, P0, Qx, lam, Qu, P0_feat, Qx_feat, lam_feat, cam_center, focal_len, q_b_c, p_b_c,
                 q_b_u=(1.0, 0, 0, 0), min_depth=1.5, use_drag_term=True, use_partial_update=True, **_):
        self.N, self.nx, self.n = N, 17 + 5 * N, 16 + 3 * N
        n = self.n
        self.x = np.zeros(self.nx)
        self.x[:17] = x0
        d = np.concatenate([P0, np.tile(P0_feat, N)])
        self.P = np.diag(d)
        self.Qx = np.diag(np.concatenate([Qx, np.tile(Qx_feat, N)]))
        self.Qu = np.diag(Qu)
        self.lam = np.concatenate([lam, np.tile(lam_feat, N)])
        one = np.ones(n)
        self.Lambda = np.outer(one, self.lam) + np.outer(self.lam, one) - np.outer(self.lam, self.lam)
        self.P0_feat = np.diag(P0_feat)
        self.c = np.asarray(cam_center, float)
        self.F = np.array([[focal_len[0], 0, 0], [0, focal_len[1], 0.0]])
        self.q_b_c, self.p_b_c, self.q_b_u = np.asarray(q_b_c, float), np.asarray(p_b_c, float), np.asarray(q_b_u, float)
        self.min_depth, self.drag, self.partial = min_depth, use_drag_term, use_partial_update
        self.len = 0

    # ---- manifold
    def boxplus(self, x, dx):
        out = x.copy()
        out[0:6] = x[0:6] + dx[0:6]
        out[6:10] = q_boxplus(x[6:10], dx[6:9])
        out[10:17] = x[10:17] + dx[9:16]
        for i in range(self.len):
            out[17 + 5 * i:21 + 5 * i] = q_feat_boxplus(x[17 + 5 * i:21 + 5 * i], dx[16 + 3 * i:18 + 3 * i])
            out[21 + 5 * i] = x[21 + 5 * i] + dx[18 + 3 * i]
        return out

    def boxminus(self, x1, x2):
        out = np.zeros(self.n)
        out[0:6] = x1[0:6] - x2[0:6]
        out[6:9] = q_boxminus(x1[6:10], x2[6:10])
        out[9:16] = x1[10:17] - x2[10:17]
        for i in range(self.len):
            out[16 + 3 * i:18 + 3 * i] = q_feat_boxminus(x1[17 + 5 * i:21 + 5 * i], x2[17 + 5 * i:21 + 5 * i])
            out[18 + 3 * i] = x1[21 + 5 * i] - x2[21 + 5 * i]
        return out

    # ---- dynamics
    def dynamics(self, x, u):
        n = self.n
        A, G, xd = np.zeros((n, n)), np.zeros((n, 6)), np.zeros(n)
        v, q, mu = x[3:6], x[6:10], x[16]
        acc, om = u[0:3] - x[10:13], u[3:6] - x[13:16]
        R = Rmat(q)
        gB = R @ GRAV
        vxy = np.array([v[0], v[1], 0.0])
        xd[0:3] = R.T @ v
        if self.drag:
            xd[3:6] = np.array([0, 0, acc[2]]) + gB - np.cross(om, v) - mu * vxy
        else:
            xd[3:6] = acc + gB - np.cross(om, v)
        xd[6:9] = om
        A[0:3, 3:6] = R.T
        A[0:3, 6:9] = -R.T @ skew(v)
        if self.drag:
            A[3:6, 3:6] = -mu * np.diag([1.0, 1.0, 0.0]) - skew(om)
            A[5, 11] = -1.0
            A[3:6, 15] = -vxy
            G[5, 2] = -1.0
        else:
            A[3:6, 3:6] = -skew(om)
            A[3:6, 9:12] = -np.eye(3)
            G[3:6, 0:3] = -np.eye(3)
        A[3:6, 6:9] = skew(gB)
        A[3:6, 12:15] = -skew(v)
        A[6:9, 6:9] = -skew(om)
        A[6:9, 12:15] = -np.eye(3)
        G[3:6, 3:6] = -skew(v)
        G[6:9, 3:6] = -np.eye(3)
        Rbc = Rmat(self.q_b_c)
        vc = Rbc @ (v + np.cross(om, self.p_b_c))
        wc = Rbc @ om
        Sp = skew(self.p_b_c)
        for i in range(self.len):
            qz, rho = x[17 + 5 * i:21 + 5 * i], x[21 + 5 * i]
            z = rota(qz, E_Z)
            Tz = T_zeta(qz)
            Sz = skew(z)
            r0, rr = 16 + 3 * i, 18 + 3 * i
            zxv = np.cross(z, vc)
            xd[r0:r0 + 2] = -Tz.T @ (wc + rho * zxv)
            xd[rr] = rho * rho * (z @ vc)
            A[r0:r0 + 2, 3:6] = -rho * Tz.T @ Sz @ Rbc
            Bg = -Tz.T @ (rho * Sz @ Rbc @ Sp - Rbc)
            A[r0:r0 + 2, 12:15] = Bg
            A[r0:r0 + 2, r0:r0 + 2] = -Tz.T @ (skew(wc + rho * zxv) + rho * skew(vc) @ Sz) @ Tz
            A[r0:r0 + 2, rr] = -Tz.T @ zxv
            A[rr, 3:6] = rho * rho * z @ Rbc
            A[rr, 12:15] = rho * rho * z @ Rbc @ Sp
            A[rr, r0:r0 + 2] = rho * rho * z @ skew(vc) @ Tz
            A[rr, rr] = 2 * rho * (z @ vc)
            G[r0:r0 + 2, 3:6] = Bg
            G[rr, 3:6] = rho * rho * z @ Rbc @ Sp
        return xd, A, G

    def fix_depth(self):
        for i in range(self.len):
            xr, dr = 21 + 5 * i, 18 + 3 * i
            if np.isnan(self.x[xr]):
                self.x[xr] = 1.0 / (2.0 * self.min_depth)
            if self.x[xr] < 0.0:
                err = 1.0 / (2.0 * self.min_depth) - self.x[xr]
                self.P[dr, dr] += err * err
                self.x[xr] = 1.0 / (2.0 * self.min_depth)
            elif self.x[xr] > 1e2:
                self.P[dr, dr] = self.P0_feat[2, 2]
                self.x[xr] = 1.0 / (2.0 * self.min_depth)

    def propagate(self, u_imu, dt):
        ub = np.concatenate([rota(self.q_b_u, u_imu[0:3]), rota(self.q_b_u, u_imu[3:6])])
        xd, A, G = self.dynamics(self.x, ub)
        self.x = self.boxplus(self.x, xd * dt)
        I = np.eye(self.n)
        A2 = A @ A
        Gd = (I + A * dt / 2.0 + A2 * dt * dt / 6.0) @ G * dt
        Phi = I + A * dt + A2 * dt * dt / 2.0
        self.P = Phi @ self.P @ Phi.T + Gd @ self.Qu @ Gd.T + self.Qx
        self.fix_depth()

    def init_feature(self, l, depth=np.nan):
        if self.len >= self.N:
            return False
        lc = np.asarray(l, float) - self.c
        z = np.array([lc[0], lc[1] * (self.F[1, 1] / self.F[0, 0]), self.F[0, 0]])
        z = z / np.linalg.norm(z)
        qz = from_two_unit_vectors(E_Z, z)
        d = 2.0 * self.min_depth if np.isnan(depth) else depth
        k = self.len
        self.len += 1
        self.x[17 + 5 * k:21 + 5 * k] = qz
        self.x[21 + 5 * k] = 1.0 / d
        a = 16 + 3 * k
        self.P[a:a + 3, :a] = 0
        self.P[:a, a:a + 3] = 0
        self.P[a:a + 3, a:a + 3] = self.P0_feat
        return True

    def h_feat(self, x, i):
        qz = x[17 + 5 * i:21 + 5 * i]
        z = rota(qz, E_Z)
        ez = z[2]
        h = self.F @ z / ez + self.c
        H = np.zeros((2, self.n))
        H[:, 16 + 3 * i:18 + 3 * i] = (1.0 / ez) * self.F @ (np.outer(z, E_Z) / ez - np.eye(3)) @ skew(z) @ T_zeta(qz)
        return h, H

    def update_feat(self, z, R, slot, active=True):
        h, H = self.h_feat(self.x, slot)
        res = np.asarray(z, float) - h
        if active:
            Sinv = np.linalg.inv(H @ self.P @ H.T + R)
            if res @ Sinv @ res > 9.0:
                return 1
            K = self.P @ H.T @ Sinv
            I = np.eye(self.n)
            A = I - K @ H
            if self.partial:
                self.x = self.boxplus(self.x, self.lam * (K @ res))
                self.P = self.P + self.Lambda * (A @ self.P @ A.T + K @ R @ K.T - self.P)
            else:
                self.x = self.boxplus(self.x, K @ res)
                self.P = A @ self.P @ A.T + K @ R @ K.T
        self.fix_depth()
        return 0
