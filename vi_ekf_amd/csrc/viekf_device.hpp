// viekf_device.hpp -- per-lane fp64 math of the VI-EKF hot path (gfx950 device code).
//
// Quaternions are Hamilton [w,x,y,z]; R(q) is the passive matrix R_I^b, rota(v) = R^T v,
// rotp(v) = R v (conventions documented by the reference's src/quat.cpp:226-290).
// Small matrices are ROW-major here (m[r*cols + c]); the covariance in HBM is column-major.
// Reference equations are cited as file:line relative to the reference tree.
#pragma once
#include <hip/hip_runtime.h>

namespace viekf {

constexpr int xPOS = 0, xVEL = 3, xATT = 6, xB_A = 10, xB_G = 13, xMU = 16, xZ = 17;  // include/vi_ekf.h:87-95
constexpr int dxPOS = 0, dxVEL = 3, dxATT = 6, dxB_A = 9, dxB_G = 12, dxMU = 15, dxZ = 16;  // :103-111
constexpr double kGravity = 9.80665;  // include/vi_ekf.h:70-74

struct DevParams {  // shared by every filter of a batch (kernel argument, lives in SGPRs/constant)
  double Qu[6];
  double P0_feat[3];
  double cam_center[2];
  double focal[2];
  double q_b_c[4];
  double p_b_c[3];
  double q_b_u[4];
  double min_depth;
  int use_drag_term;
  int use_partial_update;
  double sqrtQu[6];  // sqrt(Qu): the fused step carries Gd sqrt(Qu), so that Gd Qu Gd^T is one symmetric product
};

#define VD __device__ __forceinline__

VD void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
VD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

VD void skew3(const double* v, double* S) {  // row-major [v]x
  S[0] = 0.0;   S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2];  S[4] = 0.0;   S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0];  S[8] = 0.0;
}

// C(m x n) = A(m x k) B(k x n), row-major, fully unrolled for the tiny sizes used here
template <int M, int K, int N>
VD void mm(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < M; i++)
#pragma unroll
    for (int j = 0; j < N; j++) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < K; k++) s += A[i * K + k] * B[k * N + j];
      C[i * N + j] = s;
    }
}

VD void q_otimes(const double* a, const double* b, double* o) {  // src/quat.cpp:304-312
  const double r0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  const double r1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  const double r2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  const double r3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

VD void q_rota(const double* q, const double* v, double* o) {  // src/quat.cpp:279-283
  double t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
  cross3(q + 1, t, c);
  const double r0 = v[0] + q[0] * t[0] + c[0], r1 = v[1] + q[0] * t[1] + c[1], r2 = v[2] + q[0] * t[2] + c[2];
  o[0] = r0; o[1] = r1; o[2] = r2;
}

VD void q_rotp(const double* q, const double* v, double* o) {  // src/quat.cpp:286-290
  double t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= -2.0; t[1] *= -2.0; t[2] *= -2.0;
  cross3(q + 1, t, c);
  const double r0 = v[0] + q[0] * t[0] - c[0], r1 = v[1] + q[0] * t[1] - c[1], r2 = v[2] + q[0] * t[2] - c[2];
  o[0] = r0; o[1] = r1; o[2] = r2;
}

VD void q_R(const double* q, double* R) {  // src/quat.cpp:226-242, row-major
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double wx = w * x, wy = w * y, wz = w * z, xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z,
               zz = z * z;
  R[0] = 1. - 2. * yy - 2. * zz; R[1] = 2. * xy + 2. * wz;      R[2] = 2. * xz - 2. * wy;
  R[3] = 2. * xy - 2. * wz;      R[4] = 1. - 2. * xx - 2. * zz; R[5] = 2. * yz + 2. * wx;
  R[6] = 2. * xz + 2. * wy;      R[7] = 2. * yz - 2. * wx;      R[8] = 1. - 2. * xx - 2. * yy;
}

VD void q_exp(const double* v, double* o) {  // src/quat.cpp:64-80
  const double nv = sqrt(dot3(v, v));
  if (nv > 1e-4) {
    double s, c;
    sincos(nv / 2.0, &s, &c);
    s = s / nv;
    o[0] = c; o[1] = s * v[0]; o[2] = s * v[1]; o[3] = s * v[2];
  } else {
    const double q1 = v[0] / 2.0, q2 = v[1] / 2.0, q3 = v[2] / 2.0;
    const double nq = sqrt(1.0 + q1 * q1 + q2 * q2 + q3 * q3);
    o[0] = 1.0 / nq; o[1] = q1 / nq; o[2] = q2 / nq; o[3] = q3 / nq;
  }
}

// T_zeta(q) = [rota(e_x) rota(e_y)]  (include/math_helper.h:19-22); returns the columns t1,t2 and zeta = rota(e_z)
VD void bearing_frame(const double* q, double* t1, double* t2, double* zeta) {
  const double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
  q_rota(q, ex, t1);
  q_rota(q, ey, t2);
  q_rota(q, ez, zeta);
}

// q_feat_boxplus: exp(T_zeta(q) dq) (x) q   (include/math_helper.h:45-48)
VD void q_feat_boxplus(const double* q, double d0, double d1, double* o) {
  double t1[3], t2[3], z[3], v[3], e[4];
  bearing_frame(q, t1, t2, z);
  v[0] = t1[0] * d0 + t2[0] * d1;
  v[1] = t1[1] * d0 + t2[1] * d1;
  v[2] = t1[2] * d0 + t2[2] * d1;
  q_exp(v, e);
  q_otimes(e, q, o);
}

// body part of boxplus (vi_ekf_helper.cpp:90-92): x[0..16] (+) dx[0..15]
VD void body_boxplus(const double* x, const double* dx, double* o) {
#pragma unroll
  for (int i = 0; i < 6; i++) o[xPOS + i] = x[xPOS + i] + dx[dxPOS + i];
  double e[4], q[4];
  q_exp(dx + dxATT, e);
  q_otimes(x + xATT, e, q);  // q (x) exp(d)  (src/quat.cpp:314-317)
  o[xATT] = q[0]; o[xATT + 1] = q[1]; o[xATT + 2] = q[2]; o[xATT + 3] = q[3];
#pragma unroll
  for (int i = 0; i < 7; i++) o[xB_A + i] = x[xB_A + i] + dx[dxB_A + i];
}

// Quantities of the body state shared by the body and every feature block (vi_ekf_dyn.cpp:27-39, 83-94).
struct BodyCtx {
  double vel[3], omega[3], acc[3], mu;
  double R_I_b[9], gravity_B[3];
  double vel_c[3], omega_c[3];  // vel_c_i, omega_c_i
  double R_b_c[9], sk_p[9];     // q_b_c.R(), skew(p_b_c)
  double RS[9];                 // R_b_c * skew(p_b_c)
};

VD void body_ctx(const double* x, const double* ub, const DevParams& p, BodyCtx& c) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    c.vel[i] = x[xVEL + i];
    c.acc[i] = ub[i] - x[xB_A + i];
    c.omega[i] = ub[3 + i] - x[xB_G + i];
  }
  c.mu = x[xMU];
  q_R(x + xATT, c.R_I_b);
  const double g[3] = {0.0, 0.0, kGravity};
  q_rotp(x + xATT, g, c.gravity_B);
  double wxp[3], t[3];
  cross3(c.omega, p.p_b_c, wxp);
  t[0] = c.vel[0] + wxp[0]; t[1] = c.vel[1] + wxp[1]; t[2] = c.vel[2] + wxp[2];
  q_rotp(p.q_b_c, t, c.vel_c);
  q_rotp(p.q_b_c, c.omega, c.omega_c);
  q_R(p.q_b_c, c.R_b_c);
  skew3(p.p_b_c, c.sk_p);
  mm<3, 3, 3>(c.R_b_c, c.sk_p, c.RS);
}

// Body dynamics + Jacobians (vi_ekf_dyn.cpp:42-80).  A is 16x16 row-major, G 16x6 row-major, xdot 16.
// ZERO = false: the caller has already cleared xdot / A / G (e.g. cooperatively, one word per lane)
template <bool ZERO = true>
VD void body_dynamics(const BodyCtx& c, const DevParams& p, double* xdot, double* A, double* G) {
  if (ZERO) {
    for (int i = 0; i < 16; i++) xdot[i] = 0.0;
    for (int i = 0; i < 256; i++) A[i] = 0.0;
    for (int i = 0; i < 96; i++) G[i] = 0.0;
  }
  double wxv[3];
  cross3(c.omega, c.vel, wxv);
  // pdot = q.rota(vel) = R^T vel
  for (int i = 0; i < 3; i++)
    xdot[dxPOS + i] = c.R_I_b[0 * 3 + i] * c.vel[0] + c.R_I_b[1 * 3 + i] * c.vel[1] + c.R_I_b[2 * 3 + i] * c.vel[2];
  const double vxy[3] = {c.vel[0], c.vel[1], 0.0};
  for (int i = 0; i < 3; i++) {
    if (p.use_drag_term) xdot[dxVEL + i] = ((i == 2) ? c.acc[2] : 0.0) + c.gravity_B[i] - wxv[i] - c.mu * vxy[i];
    else xdot[dxVEL + i] = c.acc[i] + c.gravity_B[i] - wxv[i];
    xdot[dxATT + i] = c.omega[i];
  }
  double skv[9], sko[9], skg[9];
  skew3(c.vel, skv);
  skew3(c.omega, sko);
  skew3(c.gravity_B, skg);
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) {
      A[(dxPOS + r) * 16 + dxVEL + cc] = c.R_I_b[cc * 3 + r];  // R^T
      double s = 0.0;                                          // -R^T skew(vel)
      for (int k = 0; k < 3; k++) s += -c.R_I_b[k * 3 + r] * skv[k * 3 + cc];
      A[(dxPOS + r) * 16 + dxATT + cc] = s;
      A[(dxVEL + r) * 16 + dxVEL + cc] = -sko[r * 3 + cc];
      A[(dxVEL + r) * 16 + dxATT + cc] = skg[r * 3 + cc];
      A[(dxVEL + r) * 16 + dxB_G + cc] = -skv[r * 3 + cc];
      A[(dxATT + r) * 16 + dxATT + cc] = -sko[r * 3 + cc];
      G[(dxVEL + r) * 6 + 3 + cc] = -skv[r * 3 + cc];
    }
  for (int r = 0; r < 3; r++) {
    A[(dxATT + r) * 16 + dxB_G + r] = -1.0;
    G[(dxATT + r) * 6 + 3 + r] = -1.0;
  }
  if (p.use_drag_term) {
    A[(dxVEL + 0) * 16 + dxVEL + 0] += -c.mu;
    A[(dxVEL + 1) * 16 + dxVEL + 1] += -c.mu;
    A[(dxVEL + 2) * 16 + dxB_A + 2] = -1.0;
    A[(dxVEL + 0) * 16 + dxMU] = -vxy[0];
    A[(dxVEL + 1) * 16 + dxMU] = -vxy[1];
    A[(dxVEL + 2) * 16 + dxMU] = -vxy[2];
    G[(dxVEL + 2) * 6 + 2] = -1.0;
  } else {
    for (int r = 0; r < 3; r++) {
      A[(dxVEL + r) * 16 + dxB_A + r] = -1.0;
      G[(dxVEL + r) * 6 + r] = -1.0;
    }
  }
}

// One feature's dynamics and Jacobian blocks (vi_ekf_dyn.cpp:96-134).  All 3x3 row-major with rows
// (zeta0, zeta1, rho):  Afv = d/dVEL, Afg = d/dB_G ( = the uG block of G, :131-132), Aff = own block.
VD void feature_dynamics(const double* qz, double rho, const BodyCtx& c, double* xdot3, double* Afv, double* Afg,
                         double* Aff) {
  double t1[3], t2[3], z[3];
  bearing_frame(qz, t1, t2, z);
  const double rho2 = rho * rho;
  double zxv[3], wv[3];
  cross3(z, c.vel_c, zxv);
  wv[0] = c.omega_c[0] + rho * zxv[0];
  wv[1] = c.omega_c[1] + rho * zxv[1];
  wv[2] = c.omega_c[2] + rho * zxv[2];
  xdot3[0] = -dot3(t1, wv);                 // :114
  xdot3[1] = -dot3(t2, wv);
  xdot3[2] = rho2 * dot3(z, c.vel_c);       // :115
  double skz[9];
  skew3(z, skz);
  const double nT[6] = {-t1[0], -t1[1], -t1[2], -t2[0], -t2[1], -t2[2]};  // -T_z^T (2x3)
  // :121  -rho * T_z^T * skew_zeta * R_b_c
  double rT[6], m23[6], o23[6];
  for (int k = 0; k < 6; k++) rT[k] = rho * nT[k];
  mm<2, 3, 3>(rT, skz, m23);
  mm<2, 3, 3>(m23, c.R_b_c, o23);
  for (int k = 0; k < 6; k++) Afv[k] = o23[k];
  // :122  -T_z^T * (rho * skew_zeta * R_b_c * skew_p_b_c - R_b_c)
  double m33[9], n33[9];
  mm<3, 3, 3>(skz, c.RS, m33);
  for (int k = 0; k < 9; k++) n33[k] = rho * m33[k] - c.R_b_c[k];
  mm<2, 3, 3>(nT, n33, o23);
  for (int k = 0; k < 6; k++) Afg[k] = o23[k];
  // :123  -T_z^T * (skew(omega_c + rho zeta x v_c) + rho * skew_vel_c * skew_zeta) * T_z
  double skw[9], skvc[9];
  skew3(wv, skw);
  skew3(c.vel_c, skvc);
  mm<3, 3, 3>(skvc, skz, m33);
  for (int k = 0; k < 9; k++) n33[k] = skw[k] + rho * m33[k];
  mm<2, 3, 3>(nT, n33, o23);
  for (int r = 0; r < 2; r++) {
    Aff[r * 3 + 0] = o23[r * 3 + 0] * t1[0] + o23[r * 3 + 1] * t1[1] + o23[r * 3 + 2] * t1[2];
    Aff[r * 3 + 1] = o23[r * 3 + 0] * t2[0] + o23[r * 3 + 1] * t2[1] + o23[r * 3 + 2] * t2[2];
  }
  // :124  -T_z^T * zeta x v_c
  Aff[0 * 3 + 2] = -dot3(t1, zxv);
  Aff[1 * 3 + 2] = -dot3(t2, zxv);
  // :125-126  rho2 zeta^T R_b_c ;  rho2 zeta^T R_b_c skew_p
  const double rz[3] = {rho2 * z[0], rho2 * z[1], rho2 * z[2]};
  for (int cc = 0; cc < 3; cc++) {
    Afv[6 + cc] = rz[0] * c.R_b_c[0 * 3 + cc] + rz[1] * c.R_b_c[1 * 3 + cc] + rz[2] * c.R_b_c[2 * 3 + cc];
    Afg[6 + cc] = rz[0] * c.RS[0 * 3 + cc] + rz[1] * c.RS[1 * 3 + cc] + rz[2] * c.RS[2 * 3 + cc];
  }
  // :127  rho2 zeta^T skew_vel_c T_z
  double zs[3];
  for (int cc = 0; cc < 3; cc++) zs[cc] = rz[0] * skvc[0 * 3 + cc] + rz[1] * skvc[1 * 3 + cc] + rz[2] * skvc[2 * 3 + cc];
  Aff[6 + 0] = dot3(zs, t1);
  Aff[6 + 1] = dot3(zs, t2);
  // :128
  Aff[6 + 2] = 2.0 * rho * dot3(z, c.vel_c);
}

// h_feat (vi_ekf_meas.cpp:354-367): predicted pixel and the 2x2 non-zero block of H (row-major Hb).
VD void h_feat(const double* qz, const DevParams& p, double* zhat, double* Hb) {
  double t1[3], t2[3], z[3];
  bearing_frame(qz, t1, t2, z);
  const double ez = z[2];
  zhat[0] = p.focal[0] * z[0] / ez + p.cam_center[0];
  zhat[1] = p.focal[1] * z[1] / ez + p.cam_center[1];
  // (1/ez) * F * ((zeta e_z^T)/ez - I) * skew(zeta) * T_z ;  F = [[f0,0,0],[0,f1,0]]
  double skz[9];
  skew3(z, skz);
  double M[9];
  for (int k = 0; k < 9; k++) M[k] = 0.0;
  M[0 * 3 + 2] = z[0] / ez; M[1 * 3 + 2] = z[1] / ez; M[2 * 3 + 2] = z[2] / ez;
  M[0] -= 1.0; M[4] -= 1.0; M[8] -= 1.0;
  const double sF[6] = {(1.0 / ez) * p.focal[0], 0.0, 0.0, 0.0, (1.0 / ez) * p.focal[1], 0.0};
  double a[6], b[6];
  mm<2, 3, 3>(sF, M, a);
  mm<2, 3, 3>(a, skz, b);
  for (int r = 0; r < 2; r++) {
    Hb[r * 2 + 0] = b[r * 3 + 0] * t1[0] + b[r * 3 + 1] * t1[1] + b[r * 3 + 2] * t1[2];
    Hb[r * 2 + 1] = b[r * 3 + 0] * t2[0] + b[r * 3 + 1] * t2[1] + b[r * 3 + 2] * t2[2];
  }
}

// 2x2 inverse by LU with partial pivoting (what Eigen's dynamic-size inverse does, vi_ekf_meas.cpp:232).
// S, Si row-major.
VD void inv2(const double* S, double* Si) {
  double a = S[0], b = S[1], c = S[2], d = S[3];
  const bool swap = fabs(c) > fabs(a);
  if (swap) { double t = a; a = c; c = t; t = b; b = d; d = t; }
  const double l = c / a;
  const double u22 = d - l * b;
  // solve for the two columns of the (row-permuted) identity
  double e0[2] = {1.0, 0.0}, e1[2] = {0.0, 1.0};
  if (swap) { e0[0] = 0.0; e0[1] = 1.0; e1[0] = 1.0; e1[1] = 0.0; }
  const double y01 = e0[1] - l * e0[0], y11 = e1[1] - l * e1[0];
  const double x01 = y01 / u22, x11 = y11 / u22;
  const double x00 = (e0[0] - b * x01) / a, x10 = (e1[0] - b * x11) / a;
  Si[0] = x00; Si[1] = x10; Si[2] = x01; Si[3] = x11;
}

// init_feature's state part (vi_ekf_feat.cpp:13-36): pixel -> bearing quaternion, inverse depth
VD void init_feature_state(const double* pix, double depth, const DevParams& p, double* q, double* rho) {
  const double l0 = pix[0] - p.cam_center[0], l1 = pix[1] - p.cam_center[1];
  double z[3] = {l0, l1 * (p.focal[1] / p.focal[0]), p.focal[0]};
  const double nz = sqrt(dot3(z, z));
  z[0] /= nz; z[1] /= nz; z[2] /= nz;
  // from_two_unit_vectors(e_z, zeta)  (src/quat.cpp:167-185)
  const double d = z[2];
  if (d < 1.0) {
    const double invs = 1.0 / sqrt(2.0 * (1.0 + d));
    // e_z x zeta = (-z1, z0, 0)
    double qq[4] = {0.5 / invs, -z[1] * invs, z[0] * invs, 0.0};
    const double nq = sqrt(qq[0] * qq[0] + qq[1] * qq[1] + qq[2] * qq[2] + qq[3] * qq[3]);
    q[0] = qq[0] / nq; q[1] = qq[1] / nq; q[2] = qq[2] / nq; q[3] = qq[3] / nq;
  } else {
    q[0] = 1.0; q[1] = 0.0; q[2] = 0.0; q[3] = 0.0;
  }
  double dep = depth;
  if (depth != depth) dep = 2.0 * p.min_depth;
  *rho = 1.0 / dep;
}

#undef VD
}  // namespace viekf
