#!/usr/bin/env python3
"""Third, SYMBOLIC derivation of the VI-EKF Jacobians (build-container script; needs sympy + mpmath).

Why: the C oracle (oracle/viekf_oracle.c) and the numpy twin (oracle/np_twin.py) both restate the
reference's ANALYTIC Jacobian blocks (src/vi_ekf/vi_ekf_dyn.cpp:55-79,121-132; vi_ekf_meas.cpp:354-367)
from one reading of the same lines.  This script does not read those blocks at all.  It starts from
the MODEL -- the state dynamics f (vi_ekf_dyn.cpp:44-49,116-117), the manifold operators
(vi_ekf_helper.cpp:88-111, include/math_helper.h:14-48, conventions of src/quat.cpp) and the pixel
model h (vi_ekf_meas.cpp:357-360) -- builds the ERROR-STATE dynamics that test/jac_test.cpp:283-304
(`f_tilde`) differentiates numerically,

    x~'(x~) = d/dt [ x(t) [-] x^(t) ]_{t=0},   x(0) = x^ [+] x~,   x(t+dt) = x(t) [+] f(x(t), u) dt,

and differentiates it SYMBOLICALLY (sympy.diff) with respect to x~ and the input noise at x~ = 0:
A = d x~' / d x~, G = d x~' / d eta, H = d h(x^ [+] x~) / d x~.  Evaluation point: the fixed-seed
restatement of init_jacobians_test (test/jac_test.cpp:118-170; tests/helpers.py::jac_fixture).

Two remarks on rigour:
 * A Jacobian at x~ = 0 only sees the first-order behaviour of the retraction [+] and of its inverse
   [-].  exp() and log() / acos() are 0/0 at the origin, so the symbolic part uses retractions that
   agree with them to first order and are rational:  delta_q(theta) = normalise([1, theta/2])  for
   exp(theta),  2 vec(dq)/w(dq)  for log(dq),  T^T (zeta^ x zeta)  for the bearing difference
   theta * s of math_helper.h:25-43.
 * That substitution is then CHECKED, not assumed: part 2 evaluates the exact f_tilde of jac_test.cpp
   (true exp / log / acos) in 120-digit mpmath arithmetic and differentiates it by central differences
   (steps 1e-25 in t, 1e-15 in x~): both derivations must agree to 1e-12.

Output: tests/golden/jac_sym_N<k>.npz  with the evaluation point (params, x, u) and A, G, H.
The CPU test tests/test_jacobians_symbolic.py compares the oracle's analytic blocks with them.

    python tests/golden/derive_jacobians.py          # writes jac_sym_N3.npz (seed 11) and jac_sym_N2.npz (seed 5)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import mpmath as mp   # noqa: E402
import sympy as sp    # noqa: E402

GRAV = (0, 0, sp.Rational(980665, 100000))   # include/vi_ekf.h:70-74


# ------------------------------------------------------------------ conventions (src/quat.cpp), symbolic
def skew(v):
    return sp.Matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def Rmat(q):
    """passive rotation matrix R_I^b (src/quat.cpp:226-242) = I - 2 w [v]x + 2 [v]x^2"""
    w, v = q[0], sp.Matrix(q[1:4])
    S = skew(v)
    return sp.eye(3) - 2 * w * S + 2 * S * S


def qmul(a, b):   # Hamilton product, [w, x, y, z] (src/quat.cpp:304-312)
    aw, av = a[0], sp.Matrix(a[1:4])
    bw, bv = b[0], sp.Matrix(b[1:4])
    w = aw * bw - av.dot(bv)
    v = aw * bv + bw * av + av.cross(bv)
    return sp.Matrix([w, v[0], v[1], v[2]])


def qinv(q):
    return sp.Matrix([q[0], -q[1], -q[2], -q[3]])


def small_q(theta):
    """first-order-exact stand-in for exp(theta) (src/quat.cpp:64-80): unit quaternion [1, theta/2] / |.|"""
    t = sp.Matrix(theta)
    nrm = sp.sqrt(1 + t.dot(t) / 4)
    return sp.Matrix([1 / nrm, t[0] / (2 * nrm), t[1] / (2 * nrm), t[2] / (2 * nrm)])


def T_zeta(q):    # include/math_helper.h:19-22: q.doublerota(I_2x3^T) = first two columns of R^T
    return Rmat(q).T[:, 0:2]


def zeta(q):      # include/math_helper.h:14-17: q.rota(e_z) = R^T e_z
    return Rmat(q).T[:, 2]


# ------------------------------------------------------------------ the model f (vi_ekf_dyn.cpp:27-49,83-117)
def dynamics(x, u, cam, drag):
    """x = dict(p, v, q, ba, bg, mu, feats=[(qz, rho), ...]) of sympy objects -> dict of rates (same keys; q -> omega,
    feature -> (zeta coordinates rate (2), rho rate))"""
    v, q = sp.Matrix(x["v"]), x["q"]
    acc = sp.Matrix(u[0:3]) - sp.Matrix(x["ba"])
    omega = sp.Matrix(u[3:6]) - sp.Matrix(x["bg"])
    R = Rmat(q)
    gB = R * sp.Matrix(GRAV)                       # q.rotp(gravity)
    pdot = R.T * v                                  # q.rota(vel)
    if drag:
        vdot = sp.Matrix([0, 0, acc[2]]) + gB - omega.cross(v) - x["mu"] * sp.Matrix([v[0], v[1], 0])
    else:
        vdot = acc + gB - omega.cross(v)
    Rbc = Rmat(cam["q_b_c"])
    pbc = sp.Matrix(cam["p_b_c"])
    vel_c = Rbc * (v + omega.cross(pbc))            # q_b_c.rotp(vel + omega x p_b_c)
    om_c = Rbc * omega
    feats = []
    for qz, rho in x["feats"]:
        z, T = zeta(qz), T_zeta(qz)
        zd = -T.T * (om_c + rho * z.cross(vel_c))   # :116
        rd = rho * rho * z.dot(vel_c)               # :117
        feats.append((zd, rd))
    return dict(p=pdot, v=vdot, q=omega, feats=feats)


def boxplus_first_order(xh, xt):
    """x^ [+] x~ (vi_ekf_helper.cpp:88-98) with the first-order-exact retractions; xt = dict(p,v,th,ba,bg,mu,feats=[(dz(2), drho)])"""
    out = dict(p=sp.Matrix(xh["p"]) + sp.Matrix(xt["p"]), v=sp.Matrix(xh["v"]) + sp.Matrix(xt["v"]),
               q=qmul(xh["q"], small_q(xt["th"])),                         # q (x) exp(theta)
               ba=sp.Matrix(xh["ba"]) + sp.Matrix(xt["ba"]), bg=sp.Matrix(xh["bg"]) + sp.Matrix(xt["bg"]),
               mu=xh["mu"] + xt["mu"], feats=[])
    for (qz, rho), (dz, dr) in zip(xh["feats"], xt["feats"]):
        a = T_zeta(qz) * sp.Matrix(dz)
        out["feats"].append((qmul(small_q(a), qz), rho + dr))              # exp(T_z dz) (x) q_zeta
    return out


def error_state_rate(xh, xt, u, eta, cam, drag):
    """x~' as a column (16 + 3 N), built from the model and the manifold operators only"""
    ut = sp.Matrix(u) - sp.Matrix(eta)       # acc = u_a - b_a - eta_a, omega = u_g - b_g - eta_g (jac_test.cpp:369: "same as biases")
    x = boxplus_first_order(xh, xt)
    f = dynamics(x, ut, cam, drag)
    fh = dynamics(xh, u, cam, drag)
    rows = list(f["p"] - fh["p"]) + list(f["v"] - fh["v"])
    # attitude: dq = q^-1 (x) q,  q' = 1/2 q (x) [0, omega]  =>  dq' = 1/2 (dq (x) w - w^ (x) dq);  x~_theta ~ 2 vec(dq)/w(dq)
    dq = qmul(qinv(xh["q"]), x["q"])
    wq = sp.Matrix([0, f["q"][0], f["q"][1], f["q"][2]])
    whq = sp.Matrix([0, fh["q"][0], fh["q"][1], fh["q"][2]])
    dqd = (qmul(dq, wq) - qmul(whq, dq)) / 2
    for k in range(3):
        rows.append(2 * (dqd[1 + k] * dq[0] - dq[1 + k] * dqd[0]) / dq[0] ** 2)
    rows += [0] * 7   # biases and mu: constant states (random walks enter through Qx, not through f)
    feat_rows = []
    for (qz, rho), (qzh, rhoh), (zd, rd), (zdh, rdh) in zip(x["feats"], xh["feats"], f["feats"], fh["feats"]):
        # q_zeta(t+dt) = exp(T zd dt) (x) q_zeta  =>  d/dt R^T = [s]x R^T,  s = T zd   (R(exp(s) (x) q) = R(q) R(exp(s)))
        T, Th = T_zeta(qz), T_zeta(qzh)
        z, zh = zeta(qz), zeta(qzh)
        s, sh = T * zd, Th * zdh
        zdot, zhdot = s.cross(z), sh.cross(zh)
        Thdot = skew(sh) * Th
        # x~_zeta ~ T^^T (zeta^ x zeta)   (first order of theta * s, math_helper.h:25-43)
        val = Thdot.T * zh.cross(z) + Th.T * (zhdot.cross(z) + zh.cross(zdot))
        feat_rows += [val[0], val[1], rd - rdh]
    return sp.Matrix(rows + feat_rows)


def pixel_model(x, cam, i):
    """h_feat (vi_ekf_meas.cpp:357-360): F zeta / zeta_z + c"""
    z = zeta(x["feats"][i][0])
    fx, fy = cam["focal_len"]
    return sp.Matrix([fx * z[0] / z[2] + cam["cam_center"][0], fy * z[1] / z[2] + cam["cam_center"][1]])


def F(v):
    return sp.Float(float(v), 40)   # the binary64 value, exactly, carried at 40 digits


def derive(params, xnum, unum, N):
    """-> A (n x n), G (n x 6), H (N x 2 x n) as float64 arrays, by symbolic differentiation"""
    n = 16 + 3 * N
    xh = dict(p=[F(v) for v in xnum[0:3]], v=[F(v) for v in xnum[3:6]], q=sp.Matrix([F(v) for v in xnum[6:10]]),
              ba=[F(v) for v in xnum[10:13]], bg=[F(v) for v in xnum[13:16]], mu=F(xnum[16]),
              feats=[(sp.Matrix([F(v) for v in xnum[17 + 5 * i:21 + 5 * i]]), F(xnum[21 + 5 * i])) for i in range(N)])
    cam = dict(q_b_c=sp.Matrix([F(v) for v in params["q_b_c"]]), p_b_c=[F(v) for v in params["p_b_c"]],
               focal_len=[F(v) for v in params["focal_len"]], cam_center=[F(v) for v in params["cam_center"]])
    u = [F(v) for v in unum]
    eps = sp.Symbol("eps")   # a partial derivative is taken along one coordinate axis at a time: x~ = eps e_c, d/d eps at 0
    drag = bool(params["use_drag_term"])

    def axis(c, nvars):
        return [eps if k == c else sp.Integer(0) for k in range(nvars)]

    def split(d):
        return dict(p=d[0:3], v=d[3:6], th=d[6:9], ba=d[9:12], bg=d[12:15], mu=d[15],
                    feats=[(d[16 + 3 * i:18 + 3 * i], d[18 + 3 * i]) for i in range(N)])

    def ddeps(ex):
        return float(sp.diff(ex, eps).subs(eps, 0)) if (ex != 0 and ex.has(eps)) else 0.0

    A = np.zeros((n, n))
    G = np.zeros((n, 6))
    H = np.zeros((N, 2, n))
    for c in range(n):
        xt = split(axis(c, n))
        rate = error_state_rate(xh, xt, u, [0] * 6, cam, drag)
        for r in range(n):
            A[r, c] = ddeps(rate[r])
        x = boxplus_first_order(xh, xt)
        for i in range(N):
            h = pixel_model(x, cam, i)
            for r in range(2):
                H[i, r, c] = ddeps(h[r])
    for c in range(6):
        rate = error_state_rate(xh, split([sp.Integer(0)] * n), u, axis(c, 6), cam, drag)
        for r in range(n):
            G[r, c] = ddeps(rate[r])
    return A, G, H


# ------------------------------------------------------------------ part 2: the exact f_tilde in 120-digit arithmetic
def mp_check(params, xnum, unum, N, A, G, H):
    mp.mp.dps = 120   # acos near 1 loses half the digits, and the time step is 1e-25
    M = mp.matrix

    def mskew(v):
        return M([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    def mR(q):
        S = mskew(q[1:4])
        return mp.eye(3) - 2 * q[0] * S + 2 * S * S

    def mqmul(a, b):
        av, bv = M(a[1:4]), M(b[1:4])
        w = a[0] * b[0] - (av.T * bv)[0]
        cr = mskew(av) * bv
        v = a[0] * bv + b[0] * av + cr
        return [w, v[0], v[1], v[2]]

    def mexp(v):   # src/quat.cpp:64-80 (the exact branch)
        th = mp.sqrt(sum(c * c for c in v))
        if th == 0:
            return [mp.mpf(1), mp.mpf(0), mp.mpf(0), mp.mpf(0)]
        s = mp.sin(th / 2) / th
        return [mp.cos(th / 2), s * v[0], s * v[1], s * v[2]]

    def mlog(q):   # src/quat.cpp:82-98
        nv = mp.sqrt(q[1] ** 2 + q[2] ** 2 + q[3] ** 2)
        if nv == 0:
            return [mp.mpf(0)] * 3
        k = 2 * mp.atan2(nv, q[0]) / nv
        return [k * q[1], k * q[2], k * q[3]]

    def boxplus(x, dx):   # vi_ekf_helper.cpp:88-98
        out = list(x)
        for k in range(6):
            out[k] = x[k] + dx[k]
        out[6:10] = mqmul(x[6:10], mexp(dx[6:9]))
        for k in range(7):
            out[10 + k] = x[10 + k] + dx[9 + k]
        for i in range(N):
            q = x[17 + 5 * i:21 + 5 * i]
            T = mR(q).T[:, 0:2]
            a = T * M(dx[16 + 3 * i:18 + 3 * i])
            out[17 + 5 * i:21 + 5 * i] = mqmul(mexp([a[0], a[1], a[2]]), q)   # math_helper.h:45-48
            out[21 + 5 * i] = x[21 + 5 * i] + dx[18 + 3 * i]
        return out

    def boxminus(x1, x2):   # vi_ekf_helper.cpp:100-111
        out = [mp.mpf(0)] * (16 + 3 * N)
        for k in range(6):
            out[k] = x1[k] - x2[k]
        dq = mqmul([x2[6], -x2[7], -x2[8], -x2[9]], x1[6:10])
        if dq[0] < 0:
            dq = [-c for c in dq]
        out[6:9] = mlog(dq)
        for k in range(7):
            out[9 + k] = x1[10 + k] - x2[10 + k]
        for i in range(N):
            qj, qi = x1[17 + 5 * i:21 + 5 * i], x2[17 + 5 * i:21 + 5 * i]
            zi, zj = mR(qi).T[:, 2], mR(qj).T[:, 2]
            s = mskew(zi) * zj
            ns = mp.sqrt((s.T * s)[0])
            if ns == 0:
                dz = [mp.mpf(0), mp.mpf(0)]
            else:
                th = mp.acos(min((zi.T * zj)[0], mp.mpf(1)))
                dzv = mR(qi).T[:, 0:2].T * (s * (th / ns))   # math_helper.h:25-43
                dz = [dzv[0], dzv[1]]
            out[16 + 3 * i:18 + 3 * i] = dz
            out[18 + 3 * i] = x1[21 + 5 * i] - x2[21 + 5 * i]
        return out

    Rbc = mR([mp.mpf(float(v)) for v in params["q_b_c"]])
    pbc = M([mp.mpf(float(v)) for v in params["p_b_c"]])
    g = M([0, 0, mp.mpf("9.80665")])
    drag = bool(params["use_drag_term"])

    def dyn(x, u):   # vi_ekf_dyn.cpp:27-49,83-117
        v = M(x[3:6])
        acc = M([u[k] - x[10 + k] for k in range(3)])
        om = M([u[3 + k] - x[13 + k] for k in range(3)])
        R = mR(x[6:10])
        pd = R.T * v
        if drag:
            vd = M([0, 0, acc[2]]) + R * g - mskew(om) * v - x[16] * M([v[0], v[1], 0])
        else:
            vd = acc + R * g - mskew(om) * v
        out = [pd[0], pd[1], pd[2], vd[0], vd[1], vd[2], om[0], om[1], om[2]] + [mp.mpf(0)] * 7
        vc = Rbc * (v + mskew(om) * pbc)
        oc = Rbc * om
        for i in range(N):
            q, rho = x[17 + 5 * i:21 + 5 * i], x[21 + 5 * i]
            Rt = mR(q).T
            z, T = Rt[:, 2], Rt[:, 0:2]
            zd = -T.T * (oc + rho * (mskew(z) * vc))
            out += [zd[0], zd[1], rho * rho * (z.T * vc)[0]]
        return out

    xh = [mp.mpf(float(v)) for v in xnum]
    # (the binary64 quaternions of the evaluation point are unit to 1e-16 only; zeta^ . zeta would exceed 1 at 120 digits and
    #  acos would leave the reals: renormalise them here -- a 1e-16 move of the evaluation point)
    for o in [6] + [17 + 5 * i for i in range(N)]:
        nq = mp.sqrt(sum(c * c for c in xh[o:o + 4]))
        xh[o:o + 4] = [c / nq for c in xh[o:o + 4]]
    u0 = [mp.mpf(float(v)) for v in unum]
    n = 16 + 3 * N
    dt = mp.mpf("1e-25")

    def f_tilde(xt, u):   # test/jac_test.cpp:283-304, verbatim in structure
        x = boxplus(xh, xt)
        dx, dxh = dyn(x, u), dyn(xh, u0)
        xp = boxplus(x, [c * dt for c in dx])
        xm = boxplus(x, [-c * dt for c in dx])
        xhp = boxplus(xh, [c * dt for c in dxh])
        xhm = boxplus(xh, [-c * dt for c in dxh])
        tp, tm = boxminus(xp, xhp), boxminus(xm, xhm)
        return [(a - b) / (2 * dt) for a, b in zip(tp, tm)]

    eps = mp.mpf("1e-15")
    worst = 0.0
    for c in range(n):
        ep = [mp.mpf(0)] * n
        em = [mp.mpf(0)] * n
        ep[c], em[c] = eps, -eps
        col = [(a - b) / (2 * eps) for a, b in zip(f_tilde(ep, u0), f_tilde(em, u0))]
        worst = max(worst, max(abs(float(col[r]) - A[r, c]) for r in range(n)))
    for c in range(6):   # noise enters as u - eta
        up, um = list(u0), list(u0)
        up[c] -= eps
        um[c] += eps
        z = [mp.mpf(0)] * n
        col = [(a - b) / (2 * eps) for a, b in zip(f_tilde(z, up), f_tilde(z, um))]
        worst = max(worst, max(abs(float(col[r]) - G[r, c]) for r in range(n)))
    fx, fy = [mp.mpf(float(v)) for v in params["focal_len"]]

    def pix(x, i):
        z = mR(x[17 + 5 * i:21 + 5 * i]).T[:, 2]
        return [fx * z[0] / z[2], fy * z[1] / z[2]]
    for i in range(N):
        for c in range(n):
            ep = [mp.mpf(0)] * n
            em = [mp.mpf(0)] * n
            ep[c], em[c] = eps, -eps
            hp, hm = pix(boxplus(xh, ep), i), pix(boxplus(xh, em), i)
            for r in range(2):
                worst = max(worst, abs(float((hp[r] - hm[r]) / (2 * eps)) - H[i, r, c]))
    return worst


def main():
    from tests.helpers import jac_fixture, make_oracle
    for N, seed in ((3, 11), (2, 5)):
        params, pix, depth, u0 = jac_fixture(N, seed)
        params = dict(params)
        if N == 2:
            params["use_drag_term"] = False      # the non-drag branch of vi_ekf_dyn.cpp:46-49,65-69
        f = make_oracle(N, params, pix, depth)   # only to place the features: the evaluation point x^ (state, no Jacobians)
        x = f.x.copy()
        A, G, H = derive(params, x, u0, N)
        gap = mp_check(params, x, u0, N, A, G, H)
        print("N=%d seed=%d: symbolic vs exact f_tilde (120 digits): max abs difference %.3e" % (N, seed, gap))
        assert gap < 1e-12, gap
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jac_sym_N%d.npz" % N)
        np.savez(out, N=N, seed=seed, x=x, u=u0, A=A, G=G, H=H, pix=pix, depth=depth, mp_gap=gap,
                 **{"p_" + k: np.asarray(v, dtype=float) for k, v in params.items()})
        print("wrote", out)


if __name__ == "__main__":
    main()
