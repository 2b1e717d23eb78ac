// viekf_fastmath.hpp -- short-dependency-chain forms of the small geometric kernels of the filter (reciprocal, small-angle
// quaternion exp, bearing frame, h_feat, 2x2 inverse, boxplus).  Same arithmetic as the reference up to rounding; used where
// they sit on a per-measurement critical path (the fused-step service wave, the panel phase of the blocked update).
#pragma once
#include "viekf_device.hpp"

// Accounting build only (-DVIEKF_ISA_MARKS, tools/isa_regions.py): brackets around code a usual update never executes, so the
// listing's per-region counts can be split into the executed path and the rest.
#ifdef VIEKF_ISA_MARKS
#define VIEKF_COLD_BEGIN() do { __builtin_amdgcn_sched_barrier(0); asm volatile("; @@COLD begin"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VIEKF_COLD_END() do { __builtin_amdgcn_sched_barrier(0); asm volatile("; @@COLD end"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VIEKF_COLD_BEGIN() do {} while (0)
#define VIEKF_COLD_END() do {} while (0)
#endif

namespace viekf {

// 1/d from v_rcp_f64 and two Newton steps (5 instructions; the IEEE division expands to ~14 with a longer dependent chain)
__device__ __forceinline__ double rcp_fast(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}

// A double literal that is not an inline constant, materialised in SGPRs at the point of use.  Inside the per-update loops the
// compiler otherwise hoists every such literal into a VGPR pair for the whole loop (the 18 series coefficients below alone
// were 36 VGPRs of the service wave's live set -- spilled to scratch and re-loaded every update under a tighter register cap).
__device__ __forceinline__ double kconst(double v) {
  unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// exp() of a small rotation vector as a quaternion (src/quat.cpp:64-80).  For |v| < 0.5 both
// cos(h) and sin(h)/(2h), h = |v|/2, are even series in h^2 -- no sqrt, no range reduction; they
// agree with either branch of the reference (the 1e-4 small-angle branch differs from the exact
// one by h^4/3 < 3e-18) to 1 ulp.  Larger steps take the library path.
__device__ __forceinline__ void q_exp_fast(const double* v, double* o) {
  const double n2 = dot3(v, v);
  const double h2 = 0.25 * n2;
  if (h2 < 2.5e-3) {                              // |v| < 0.1: truncation < 3e-20, the usual size of a filter correction
    double c = kconst(1.0 / 40320.0);
    c = fma(c, h2, kconst(-1.0 / 720.0));
    c = fma(c, h2, kconst(1.0 / 24.0));
    c = fma(c, h2, -0.5);
    c = fma(c, h2, 1.0);                          // cos(h)
    double s = kconst(1.0 / 362880.0);
    s = fma(s, h2, kconst(-1.0 / 5040.0));
    s = fma(s, h2, kconst(1.0 / 120.0));
    s = fma(s, h2, kconst(-1.0 / 6.0));
    s = fma(s, h2, 1.0);                          // sin(h)/h
    s *= 0.5;
    o[0] = c; o[1] = s * v[0]; o[2] = s * v[1]; o[3] = s * v[2];
  } else if (h2 < 0.0625) {
    VIEKF_COLD_BEGIN();
    double c = kconst(-1.0 / 20922789888000.0);          // -1/16!
    c = fma(c, h2, kconst(1.0 / 87178291200.0));          // 1/14!
    c = fma(c, h2, kconst(-1.0 / 479001600.0));           // -1/12!
    c = fma(c, h2, kconst(1.0 / 3628800.0));              // 1/10!
    c = fma(c, h2, kconst(-1.0 / 40320.0));               // -1/8!
    c = fma(c, h2, kconst(1.0 / 720.0));
    c = fma(c, h2, kconst(-1.0 / 24.0));
    c = fma(c, h2, 0.5);
    c = fma(-c, h2, 1.0);                         // cos(h)
    double s = kconst(-1.0 / 1307674368000.0);            // -1/15!
    s = fma(s, h2, kconst(1.0 / 6227020800.0));           // 1/13!
    s = fma(s, h2, kconst(-1.0 / 39916800.0));            // -1/11!
    s = fma(s, h2, kconst(1.0 / 362880.0));               // 1/9!
    s = fma(s, h2, kconst(-1.0 / 5040.0));
    s = fma(s, h2, kconst(1.0 / 120.0));
    s = fma(s, h2, kconst(-1.0 / 6.0));
    s = fma(s, h2, 1.0);                          // sin(h)/h
    s *= 0.5;
    o[0] = c; o[1] = s * v[0]; o[2] = s * v[1]; o[3] = s * v[2];
    VIEKF_COLD_END();
  } else {
    VIEKF_COLD_BEGIN();
    q_exp(v, o);
    VIEKF_COLD_END();
  }
}

// [t1 t2 zeta] = columns of R(q)^T = rota(e_x), rota(e_y), rota(e_z) written out (the same polynomial in q as
// src/quat.cpp:279-283 applied to the unit vectors, ~30 flops instead of three generic rotations)
__device__ __forceinline__ void bearing_frame_fast(const double* q, double* t1, double* t2, double* z) {
  const double w = q[0], x = q[1], y = q[2], zz_ = q[3];
  const double xx = x * x, yy = y * y, zz = zz_ * zz_, xy = x * y, xz = x * zz_, yz = y * zz_, wx = w * x, wy = w * y,
               wz = w * zz_;
  t1[0] = 1.0 - 2.0 * (yy + zz); t1[1] = 2.0 * (xy + wz);       t1[2] = 2.0 * (xz - wy);
  t2[0] = 2.0 * (xy - wz);       t2[1] = 1.0 - 2.0 * (xx + zz); t2[2] = 2.0 * (yz + wx);
  z[0] = 2.0 * (xz + wy);        z[1] = 2.0 * (yz - wx);        z[2] = 1.0 - 2.0 * (xx + yy);
}

// h_feat (vi_ekf_meas.cpp:354-367) with the matrix chain multiplied out, from an already computed frame of a UNIT
// bearing quaternion: [zeta]x T_z = [zeta x t1, zeta x t2] and F ((zeta e_z^T)/ez - I) w = (f0 (zeta_x w_z/ez - w_x),
// f1 (zeta_y w_z/ez - w_y)); one reciprocal.
// The same from an already computed frame of a UNIT bearing quaternion: (t1, t2, zeta) is then orthonormal and right-handed,
// so zeta x t1 = t2 and zeta x t2 = -t1 (for |q|^2 = 1 + e the shortcut is off by O(e) ~ 1e-15, far inside the parity bar).
__device__ __forceinline__ void h_feat_frame(const double* t1, const double* t2, const double* z, const DevParams& p,
                                             double* zhat, double* Hb) {
  const double iez = rcp_fast(z[2]);
  const double zx = z[0] * iez, zy = z[1] * iez;
  zhat[0] = fma(p.focal[0], zx, p.cam_center[0]);
  zhat[1] = fma(p.focal[1], zy, p.cam_center[1]);
  const double f0 = p.focal[0] * iez, f1 = p.focal[1] * iez;
  Hb[0] = f0 * fma(zx, t2[2], -t2[0]);
  Hb[1] = f0 * fma(-zx, t1[2], t1[0]);
  Hb[2] = f1 * fma(zy, t2[2], -t2[1]);
  Hb[3] = f1 * fma(-zy, t1[2], t1[1]);
}

// 2x2 inverse through the adjugate and ONE reciprocal (the reference's LU form, vi_ekf_meas.cpp:232, differs by
// rounding only; three dependent fp64 divisions would sit on the per-update critical path)
__device__ __forceinline__ void inv2_fast(const double* S, double* Si) {
  const double det = S[0] * S[3] - S[1] * S[2];
  const double r = rcp_fast(det);
  Si[0] = S[3] * r; Si[1] = -S[1] * r; Si[2] = -S[2] * r; Si[3] = S[0] * r;
}

__device__ __forceinline__ void q_feat_boxplus_fast(const double* q, double d0, double d1, double* o) {
  double t1[3], t2[3], z[3], v[3], e[4];
  bearing_frame_fast(q, t1, t2, z);
  v[0] = t1[0] * d0 + t2[0] * d1;
  v[1] = t1[1] * d0 + t2[1] * d1;
  v[2] = t1[2] * d0 + t2[2] * d1;
  q_exp_fast(v, e);
  q_otimes(e, q, o);
}

__device__ __forceinline__ void body_boxplus_fast(const double* x, const double* dx, double* o) {
#pragma unroll
  for (int i = 0; i < 6; i++) o[xPOS + i] = x[xPOS + i] + dx[dxPOS + i];
  double e[4], q[4];
  q_exp_fast(dx + dxATT, e);
  q_otimes(x + xATT, e, q);
  o[xATT] = q[0]; o[xATT + 1] = q[1]; o[xATT + 2] = q[2]; o[xATT + 3] = q[3];
#pragma unroll
  for (int i = 0; i < 7; i++) o[xB_A + i] = x[xB_A + i] + dx[dxB_A + i];
}

}  // namespace viekf
