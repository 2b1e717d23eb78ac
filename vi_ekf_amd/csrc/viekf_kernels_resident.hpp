// viekf_kernels_resident.hpp -- "resident" kernel family: one workgroup per filter, the whole
// covariance lives in VGPRs for the complete step (propagate + M sequential feature updates), so
// P crosses HBM once per step instead of (M+1) times.
//
// Ownership (DESIGN.md "resident layout"): the 3N x 3N feature part of P is a grid of 3x3 blocks
// (I,J).  Threads form a TR x TC grid (tr = tid % TR, tc = tid / TR); thread (tr,tc) keeps the
// RB x CB blocks  I = tr + TR*a,  J = tc + TC*c  in registers.  The 16 body rows/columns are cut
// into 3-vectors: P[3I..3I+2, k] pieces go to the threads of block-row I's tr, P[k, 3J..3J+2]
// pieces to the threads of block-column J's tc, P_bb one element per thread.  With this cartesian
// ownership a rank-2 sweep needs K for RB block-rows and W for CB block-columns per thread, and
// the propagation Phi P Phi^T + Gd Qu Gd^T becomes a register-tiled contraction
//     P+[I,J] = X_I Y_J^T + Phi_ff[I] P[I,J] Phi_ff[J]^T,   X_I = [U_I | Phi_fb[I] | Gd_I Qu],
//                                                          Y_J = [Phi_fb[J] | V_J | Gd_J]   (K = 38)
// with U = (Phi P)[feat, body], V_J = (P[b,J] Phi_ff[J]^T)^T staged in LDS.
// lambda_feat is the same for every feature slot (vi_ekf.cpp:139-144), so the partial-update
// mask Lambda (vi_ekf.cpp:83,146) is ONE 3x3 constant for every feature/feature block.
#pragma once
#include "viekf_kernels_stream.hpp"

namespace viekf {

#ifndef RES_INLINE
#define RES_INLINE __forceinline__
#endif
constexpr int XK = 38;  // contraction depth of the propagate GEMM: 16 (U) + 16 (Phi_fb) + 6 (Gd)

struct ResLds {  // LDS carve-up in doubles, shared by host (size) and device (offsets)
  int xs, Kt, Wt, Praw, lam, sm, fixadd, fixset, X, Y, phiff, Abb, Gb, Phibb, Mbb, Gdb, Pbb, T16, xdb, ctx, Pbc, Pbr, total;
  __host__ __device__ ResLds(int N, int n, int nxs) {
    const int nf = 3 * N;
    int o = 0;
    auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
    xs = take(nxs);
    Kt = take(2 * n); Wt = take(2 * n); Praw = take(2 * n); lam = take(n);
    sm = take(32);
    fixadd = take(2 * (N > 0 ? N : 1)); fixset = take(2 * (N > 0 ? N : 1));
    X = take(nf * XK); Y = take(nf * XK);
    phiff = take(9 * (N > 0 ? N : 1));
    Abb = take(256); Gb = take(96); Phibb = take(256); Mbb = take(256); Gdb = take(96); Pbb = take(256);
    T16 = take(256); xdb = take(16);
    ctx = take((int)((sizeof(BodyCtx) + 7) / 8));
    Pbc = take(nf * 16); Pbr = take(16 * nf);
    total = o;
  }
};

// exp() of a small rotation vector as a quaternion (src/quat.cpp:64-80).  For |v| < 0.5 both
// cos(h) and sin(h)/(2h), h = |v|/2, are even series in h^2 -- no sqrt, no range reduction; they
// agree with either branch of the reference (the 1e-4 small-angle branch differs from the exact
// one by h^4/3 < 3e-18) to 1 ulp.  Larger steps take the library path.
__device__ __forceinline__ void q_exp_fast(const double* v, double* o) {
  const double n2 = dot3(v, v);
  const double h2 = 0.25 * n2;
  if (h2 < 0.0625) {
    double c = -1.0 / 20922789888000.0;          // -1/16!
    c = fma(c, h2, 1.0 / 87178291200.0);          // 1/14!
    c = fma(c, h2, -1.0 / 479001600.0);           // -1/12!
    c = fma(c, h2, 1.0 / 3628800.0);              // 1/10!
    c = fma(c, h2, -1.0 / 40320.0);               // -1/8!
    c = fma(c, h2, 1.0 / 720.0);
    c = fma(c, h2, -1.0 / 24.0);
    c = fma(c, h2, 0.5);
    c = fma(-c, h2, 1.0);                         // cos(h)
    double s = -1.0 / 1307674368000.0;            // -1/15!
    s = fma(s, h2, 1.0 / 6227020800.0);           // 1/13!
    s = fma(s, h2, -1.0 / 39916800.0);            // -1/11!
    s = fma(s, h2, 1.0 / 362880.0);               // 1/9!
    s = fma(s, h2, -1.0 / 5040.0);
    s = fma(s, h2, 1.0 / 120.0);
    s = fma(s, h2, -1.0 / 6.0);
    s = fma(s, h2, 1.0);                          // sin(h)/h
    s *= 0.5;
    o[0] = c; o[1] = s * v[0]; o[2] = s * v[1]; o[3] = s * v[2];
  } else {
    q_exp(v, o);
  }
}

__device__ __forceinline__ void q_feat_boxplus_fast(const double* q, double d0, double d1, double* o) {
  double t1[3], t2[3], z[3], v[3], e[4];
  bearing_frame(q, t1, t2, z);
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

// Rarely-executed, register-hungry pieces are kept out of line so that they do not inflate the
// register allocation of the sweep loops that hold P.
__device__ RES_INLINE void res_body_phase(const double* xs, const double* u, const DevParams* p, BodyCtx* ctx,
                                            double* xdb, double* Abb, double* Gb) {
  double ub[6];
  q_rota(p->q_b_u, u, ub);
  q_rota(p->q_b_u, u + 3, ub + 3);
  body_ctx(xs, ub, *p, *ctx);
  body_dynamics(*ctx, *p, xdb, Abb, Gb);
}

// one feature's share of the propagate set-up: dynamics -> rows of X / Y / phiff, state step
__device__ RES_INLINE void res_feature_phase(int f, int len, double dt, double* xs, const BodyCtx* ctx,
                                               const DevParams* p, const double* Abb, const double* Gb, double* X,
                                               double* Y, double* phiff) {
  double* Xr = X + (3 * f) * XK;
  double* Yr = Y + (3 * f) * XK;
  if (f < len) {
    double xd3[3], Afv[9], Afg[9], Aff[9];
    const double* qz = xs + xZ + 5 * f;
    const double rho = qz[4];
    feature_dynamics(qz, rho, *ctx, xd3, Afv, Afg, Aff);
    double Aff2[9], Mff[9];
    mm<3, 3, 3>(Aff, Aff, Aff2);
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
      Mff[e] = id + Aff[e] * dt / 2.0 + Aff2[e] * dt * dt / 6.0;
      phiff[9 * f + e] = id + Aff[e] * dt + Aff2[e] * dt * dt / 2.0;
    }
    double gacc[18];
#pragma unroll
    for (int e = 0; e < 18; e++) gacc[e] = 0.0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double afb = 0.0;
        if (c >= dxVEL && c < dxVEL + 3) afb = Afv[r * 3 + (c - dxVEL)];
        else if (c >= dxB_G && c < dxB_G + 3) afb = Afg[r * 3 + (c - dxB_G)];
        double a2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++)
          a2 += Afv[r * 3 + k] * Abb[(dxVEL + k) * 16 + c] + Afg[r * 3 + k] * Abb[(dxB_G + k) * 16 + c];
        if (c >= dxVEL && c < dxVEL + 3) {
#pragma unroll
          for (int k = 0; k < 3; k++) a2 += Aff[r * 3 + k] * Afv[k * 3 + (c - dxVEL)];
        } else if (c >= dxB_G && c < dxB_G + 3) {
#pragma unroll
          for (int k = 0; k < 3; k++) a2 += Aff[r * 3 + k] * Afg[k * 3 + (c - dxB_G)];
        }
        const double ph = afb * dt + a2 * dt * dt / 2.0;
        Xr[r * XK + 16 + c] = ph;
        Yr[r * XK + c] = ph;
        const double mfb = afb * dt / 2.0 + a2 * dt * dt / 6.0;
#pragma unroll
        for (int k = 0; k < 6; k++) gacc[r * 6 + k] += mfb * Gb[c * 6 + k];
      }
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 3; m++) s += Mff[r * 3 + m] * Afg[m * 3 + k];
        gacc[r * 6 + 3 + k] += s;
      }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const double g = gacc[r * 6 + k] * dt;
        Yr[r * XK + 32 + k] = g;
        Xr[r * XK + 32 + k] = g * p->Qu[k];
      }
    double qn[4];
    q_feat_boxplus_fast(qz, xd3[0] * dt, xd3[1] * dt, qn);
    double* xf = xs + xZ + 5 * f;
    xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
    xf[4] = rho + xd3[2] * dt;
  } else {  // inactive slot: Phi = I, G = 0
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 16; c++) { Xr[r * XK + 16 + c] = 0.0; Yr[r * XK + c] = 0.0; }
      for (int k = 0; k < 6; k++) { Xr[r * XK + 32 + k] = 0.0; Yr[r * XK + 32 + k] = 0.0; }
    }
    for (int e = 0; e < 9; e++) phiff[9 * f + e] = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
  }
}

// state correction of one feature + fix_depth + (optionally) the next measurement's prediction
__device__ RES_INLINE void res_feature_update(double* xf, bool do_corr, bool do_fix, double d0, double d1, double d2,
                                                const DevParams* p, double* fixadd_slot, double* fixset_slot,
                                                unsigned* flag, const double* z_next, double* smw) {
  if (do_corr) {
    double qn[4];
    q_feat_boxplus_fast(xf, d0, d1, qn);
    xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
    xf[4] += d2;
  }
  if (do_fix) {
    double rho = xf[4];
    const double reset = 1.0 / (2.0 * p->min_depth);
    if (rho != rho) { rho = reset; *flag |= FLAG_NAN; }
    if (rho < 0.0) {
      const double err = reset - rho;
      *fixadd_slot = err * err;
      rho = reset;
      *flag |= FLAG_NEGDEPTH;
    } else if (rho > 1e2) {
      *fixset_slot = 1.0;
      rho = reset;
    }
    xf[4] = rho;
  }
  if (z_next) {
    double zhat[2], Hb[4];
    h_feat(xf, *p, zhat, Hb);
    smw[2] = Hb[0]; smw[3] = Hb[1]; smw[4] = Hb[2]; smw[5] = Hb[3];
    smw[6] = z_next[0] - zhat[0]; smw[7] = z_next[1] - zhat[1];
  }
}

__device__ RES_INLINE void res_body_update(double* xs, const double* Kt, const double* lam, bool partial, double r0,
                                             double r1) {
  double dxb[16], xo[17];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const double l = partial ? lam[i] : 1.0;
    dxb[i] = (l * Kt[2 * i]) * r0 + (l * Kt[2 * i + 1]) * r1;
  }
  body_boxplus_fast(xs, dxb, xo);
#pragma unroll
  for (int i = 0; i < 17; i++) xs[i] = xo[i];
}

// ------------------------------------------------------------------------------------------------
// fused step: [propagate] + M feature updates with P resident in registers
// ------------------------------------------------------------------------------------------------
template <int RB, int CB, int T, int SI, int SJ>
__global__ __launch_bounds__(T) void k_step_resident(StreamArgs a, int TR, int TC, int do_prop,
                                                     const double* __restrict__ u_all,
                                                     const double* __restrict__ dt_all,
                                                     const double* __restrict__ z_all,
                                                     const int* __restrict__ slot_all, int M,
                                                     const double* __restrict__ R_all, long r_stride_b, long r_stride_m,
                                                     int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int BBE = 1;  // P_bb: one element per thread, threads 0..255
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int N = a.N, n = a.n, ld = a.ld, nf = 3 * N;
  const ResLds L(N, n, a.nxs);
  double* xs = smem + L.xs;
  double* Kt = smem + L.Kt;
  double* Wt = smem + L.Wt;
  double* Praw = smem + L.Praw;
  double* lam = smem + L.lam;
  double* sm = smem + L.sm;
  double* fixadd = smem + L.fixadd;
  double* fixset = smem + L.fixset;
  double* X = smem + L.X;
  double* Y = smem + L.Y;
  double* phiff = smem + L.phiff;
  double* Pbb = smem + L.Pbb;

  double* xg = a.x + (long)b * a.nxs;
  double* P = a.P + (long)b * n * ld;
  const int len = a.len[b];
  const int tr = tid % TR, tc = tid / TR;
  const bool own = tid < TR * TC;
  unsigned flag = 0;

  // ---------------- load x, lambda, P (registers) ----------------
  for (int i = tid; i < a.nxs; i += T) xs[i] = (i < xZ + 5 * len) ? xg[i] : 0.0;
  for (int i = tid; i < n; i += T) lam[i] = a.lambda[i];
  for (int i = tid; i < 2 * N; i += T) { fixadd[i] = 0.0; fixset[i] = 0.0; }

  double pb[RB][CB][9];   // pb[a][c][r*3+s] = P[16+3I+r][16+3J+s]
  double sI[SI][3];       // P[16+3I+r][k]
  double sJ[SJ][3];       // P[k][16+3J+s]
  double sbb[BBE];        // P[r][c], e = tid + T*w, r = e & 15, c = e >> 4
#pragma unroll
  for (int ia = 0; ia < RB; ia++)
#pragma unroll
    for (int ic = 0; ic < CB; ic++) {
      const int I = tr + TR * ia, J = tc + TC * ic;
      const bool v = own && I < N && J < N;
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int r = 0; r < 3; r++) pb[ia][ic][r * 3 + s] = v ? P[(16 + 3 * I + r) + (long)(16 + 3 * J + s) * ld] : 0.0;
    }
#pragma unroll
  for (int q = 0; q < SI; q++) {
    const int e = tc + TC * q, ia = e >> 4, k = e & 15, I = tr + TR * ia;
    const bool v = own && e < RB * 16 && I < N;
#pragma unroll
    for (int r = 0; r < 3; r++) sI[q][r] = v ? P[(16 + 3 * I + r) + (long)k * ld] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < SJ; q++) {
    const int e = tr + TR * q, ic = e >> 4, k = e & 15, J = tc + TC * ic;
    const bool v = own && e < CB * 16 && J < N;
#pragma unroll
    for (int s = 0; s < 3; s++) sJ[q][s] = v ? P[k + (long)(16 + 3 * J + s) * ld] : 0.0;
  }
#pragma unroll
  for (int w = 0; w < BBE; w++) {
    const int e = tid + T * w;
    sbb[w] = (e < 256) ? P[(e & 15) + (long)(e >> 4) * ld] : 0.0;
  }
  // Lambda for feature/feature blocks: one 3x3 constant (lambda_feat identical for all slots)
  double Lff[9];
  {
    const double l0 = a.lambda[16 % n], l1 = a.lambda[17 % n], l2 = a.lambda[18 % n];
    const double lf[3] = {l0, l1, l2};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) Lff[r * 3 + s] = a.dp->use_partial_update ? (lf[s] + lf[r] - lf[r] * lf[s]) : 1.0;
  }
  __syncthreads();

  int fixpar = 0;  // parity of the fix_depth mailbox being WRITTEN in the current phase

  // =====================================================================================
  // propagate (vi_ekf.cpp:262-318)
  // =====================================================================================
  if (do_prop) {
    double* Abb = smem + L.Abb;
    double* Gb = smem + L.Gb;
    double* Phibb = smem + L.Phibb;
    double* Mbb = smem + L.Mbb;
    double* Gdb = smem + L.Gdb;
    double* T16 = smem + L.T16;
    double* xdb = smem + L.xdb;
    BodyCtx* ctx = reinterpret_cast<BodyCtx*>(smem + L.ctx);
    double* Pbc = smem + L.Pbc;  // [nf][16]
    double* Pbr = smem + L.Pbr;  // [16][nf]  (becomes Ut)
    const double dt = dt_all[b];

#ifndef ABL_BODY
    if (tid == 0) res_body_phase(xs, u_all + (long)b * 6, a.dp, ctx, xdb, Abb, Gb);
#endif
    // strips -> LDS (inputs of U / Ut)
#pragma unroll
    for (int q = 0; q < SI; q++) {
      const int e = tc + TC * q, ia = e >> 4, k = e & 15, I = tr + TR * ia;
      if (own && e < RB * 16 && I < N)
#pragma unroll
        for (int r = 0; r < 3; r++) Pbc[(3 * I + r) * 16 + k] = sI[q][r];
    }
#pragma unroll
    for (int q = 0; q < SJ; q++) {
      const int e = tr + TR * q, ic = e >> 4, k = e & 15, J = tc + TC * ic;
      if (own && e < CB * 16 && J < N)
#pragma unroll
        for (int s = 0; s < 3; s++) Pbr[k * nf + 3 * J + s] = sJ[q][s];
    }
#pragma unroll
    for (int w = 0; w < BBE; w++) {
      const int e = tid + T * w;
      if (e < 256) Pbb[(e & 15) * 16 + (e >> 4)] = sbb[w];  // row-major copy
    }
    __syncthreads();

    // body transition blocks
    for (int e = tid; e < 256; e += T) {
      const int r = e >> 4, c = e & 15;
      double a2 = 0.0;
      for (int k = 0; k < 16; k++) a2 += Abb[r * 16 + k] * Abb[k * 16 + c];
      const double id = (r == c) ? 1.0 : 0.0, av = Abb[e];
      Mbb[e] = id + av * dt / 2.0 + a2 * dt * dt / 6.0;
      Phibb[e] = id + av * dt + a2 * dt * dt / 2.0;
    }
    // per feature: dynamics -> Phi_fb (into X[.,16..31] and Y[.,0..15]), Phi_ff, Gd (X[.,32..37]*Qu, Y[.,32..37])
#ifndef ABL_FEATP
    for (int f = tid; f < N; f += T) res_feature_phase(f, len, dt, xs, ctx, a.dp, Abb, Gb, X, Y, phiff);
#endif
    __syncthreads();
    if (tid == T - 1) {  // body state step (every feature thread has consumed the old body state)
      double* kt = Kt;   // borrow Kt as [16][2] scratch: dx = xdot*dt in slot 0, 0 in slot 1
      for (int i = 0; i < 16; i++) { kt[2 * i] = xdb[i] * dt; kt[2 * i + 1] = 0.0; }
#ifndef ABL_BUPD
      res_body_update(xs, kt, lam, false, 1.0, 0.0);
#endif
    }
    for (int e = tid; e < 96; e += T) {
      const int r = e / 6, k = e % 6;
      double s = 0.0;
      for (int c = 0; c < 16; c++) s += Mbb[r * 16 + c] * Gb[c * 6 + k];
      Gdb[e] = s * dt;
    }
    for (int e = tid; e < 256; e += T) {
      const int r = e >> 4, c = e & 15;
      double s = 0.0;
      for (int k = 0; k < 16; k++) s += Phibb[r * 16 + k] * Pbb[k * 16 + c];
      T16[e] = s;
    }
    // U (-> X[.,0..15]) and V (-> Y[.,16..31]), Ut (in place over Pbr); one (block, k) triple per item
    for (int e = tid; e < N * 16; e += T) {
      const int I = e >> 4, k = e & 15;
      double pc[3], pr[3];
      for (int m = 0; m < 3; m++) { pc[m] = Pbc[(3 * I + m) * 16 + k]; pr[m] = Pbr[k * nf + 3 * I + m]; }
      for (int r = 0; r < 3; r++) {
        double su = 0.0, st = 0.0;
        const double* phr = Y + (3 * I + r) * XK;  // Phi_fb[3I+r][0..15]
        for (int c = 0; c < 16; c++) {
          su += phr[c] * Pbb[c * 16 + k];
          st += Pbb[k * 16 + c] * phr[c];
        }
        double sv = 0.0;
        for (int m = 0; m < 3; m++) {
          const double pf = phiff[9 * I + r * 3 + m];
          su += pf * pc[m];
          sv += pr[m] * pf;
        }
        X[(3 * I + r) * XK + k] = su;        // U
        Y[(3 * I + r) * XK + 16 + k] = sv;   // V
        st += sv;
        Pbr[k * nf + 3 * I + r] = st;        // Ut, in place: pr[] of this (I,k) triple is already in registers
      }
    }
    __syncthreads();

    // ---- register-tiled contraction for the owned feature/feature blocks
#pragma unroll
    for (int ia = 0; ia < RB; ia++) {
      const int I = tr + TR * ia;
      const bool vi = own && I < N;
      double fi[9];
#pragma unroll
      for (int e = 0; e < 9; e++) fi[e] = vi ? phiff[9 * I + e] : 0.0;
#pragma unroll
      for (int ic = 0; ic < CB; ic++) {
        const int J = tc + TC * ic;
        const bool v = vi && J < N;
        double fj[9], t9[9], o9[9];
#pragma unroll
        for (int e = 0; e < 9; e++) fj[e] = v ? phiff[9 * J + e] : 0.0;
        mm<3, 3, 3>(fi, pb[ia][ic], t9);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int s = 0; s < 3; s++) o9[r * 3 + s] = t9[r * 3 + 0] * fj[s * 3 + 0] + t9[r * 3 + 1] * fj[s * 3 + 1] + t9[r * 3 + 2] * fj[s * 3 + 2];
#pragma unroll
        for (int e = 0; e < 9; e++) pb[ia][ic][e] = o9[e];
        if (v && I == J) {
          pb[ia][ic][0] += a.Qx[16 + 3 * I + 0];
          pb[ia][ic][4] += a.Qx[16 + 3 * I + 1];
          pb[ia][ic][8] += a.Qx[16 + 3 * I + 2];
        }
      }
    }
    for (int k = 0; k < XK; k += 2) {
      double2 yv[CB][3];
#pragma unroll
      for (int ic = 0; ic < CB; ic++) {
        const int J = min(tc + TC * ic, N - 1);
#pragma unroll
        for (int s = 0; s < 3; s++) yv[ic][s] = *reinterpret_cast<const double2*>(Y + (3 * J + s) * XK + k);
      }
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        const int I = min(tr + TR * ia, N - 1);
        double2 xv[3];
#pragma unroll
        for (int r = 0; r < 3; r++) xv[r] = *reinterpret_cast<const double2*>(X + (3 * I + r) * XK + k);
#pragma unroll
        for (int ic = 0; ic < CB; ic++)
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
              double acc = pb[ia][ic][r * 3 + s];
              acc = fma(xv[r].x, yv[ic][s].x, acc);
              acc = fma(xv[r].y, yv[ic][s].y, acc);
              pb[ia][ic][r * 3 + s] = acc;
            }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- strips and body block
#pragma unroll
    for (int q = 0; q < SI; q++) {   // P+[3I+r][k] = U[I][r,:] Phi_bb[k,:] + Gd_I Qu Gd_b[k]
      const int e = tc + TC * q, ia = e >> 4, k = e & 15, I = tr + TR * ia;
      if (own && e < RB * 16 && I < N) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const double* xr = X + (3 * I + r) * XK;
          double s = 0.0;
          for (int c = 0; c < 16; c++) s += xr[c] * Phibb[k * 16 + c];
          double g = 0.0;
          for (int c = 0; c < 6; c++) g += xr[32 + c] * Gdb[k * 6 + c];
          sI[q][r] = s + g;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < SJ; q++) {   // P+[k][3J+s] = Phi_bb[k,:] Ut[:,3J+s] + Gd_b[k] Qu Gd_J
      const int e = tr + TR * q, ic = e >> 4, k = e & 15, J = tc + TC * ic;
      if (own && e < CB * 16 && J < N) {
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) {
          double s = 0.0;
          for (int c = 0; c < 16; c++) s += Phibb[k * 16 + c] * Pbr[c * nf + 3 * J + s3];
          const double* xr = X + (3 * J + s3) * XK;
          double g = 0.0;
          for (int c = 0; c < 6; c++) g += xr[32 + c] * Gdb[k * 6 + c];
          sJ[q][s3] = s + g;
        }
      }
    }
#pragma unroll
    for (int w = 0; w < BBE; w++) {
      const int e = (tid + T * w) & 255, r = e & 15, c = e >> 4;
      double s = 0.0;
      for (int k = 0; k < 16; k++) s += T16[r * 16 + k] * Phibb[c * 16 + k];
      double g = 0.0;
      for (int k = 0; k < 6; k++) g += Gdb[r * 6 + k] * a.dp->Qu[k] * Gdb[c * 6 + k];
      s = s + g;
      if (r == c) s += a.Qx[r];
      sbb[w] = s;
    }
    // ---- fix_depth (vi_ekf.cpp:311): state here, covariance through the mailbox
    for (int f = tid; f < len; f += T)
#ifndef ABL_FUPD
      res_feature_update(xs + xZ + 5 * f, false, true, 0.0, 0.0, 0.0, a.dp, &fixadd[fixpar * N + f],
                         &fixset[fixpar * N + f], &flag, nullptr, nullptr);
#endif
    fixpar ^= 1;
    __syncthreads();
  }

  // applies the pending fix_depth covariance edits of mailbox `par` to the owned diagonal blocks
  auto apply_fixes = [&](int par) {
#pragma unroll
    for (int ia = 0; ia < RB; ia++)
#pragma unroll
      for (int ic = 0; ic < CB; ic++) {
        const int I = tr + TR * ia, J = tc + TC * ic;
        if (own && I == J && I < len) {
          const double ad = fixadd[par * N + I], st = fixset[par * N + I];
          if (ad != 0.0) { pb[ia][ic][8] += ad; fixadd[par * N + I] = 0.0; }
          if (st != 0.0) { pb[ia][ic][8] = a.dp->P0_feat[2]; fixset[par * N + I] = 0.0; }
        }
      }
  };

  // writes the two zeta columns of feature `slot` (raw P[:, j0], P[:, j0+1]) into Praw
  auto extract_cols = [&](int slot) {
    const int cc = slot / TC, ct = slot - cc * TC;
    if (own && tc == ct) {
#pragma unroll
      for (int ic = 0; ic < CB; ic++)
        if (ic == cc) {
#pragma unroll
          for (int ia = 0; ia < RB; ia++) {
            const int I = tr + TR * ia;
            if (I < N)
#pragma unroll
              for (int r = 0; r < 3; r++) {
                Praw[2 * (16 + 3 * I + r) + 0] = pb[ia][ic][r * 3 + 0];
                Praw[2 * (16 + 3 * I + r) + 1] = pb[ia][ic][r * 3 + 1];
              }
          }
        }
#pragma unroll
      for (int q = 0; q < SJ; q++) {
        const int e = tr + TR * q, ic = e >> 4, k = e & 15;
        if (e < CB * 16 && ic == cc) { Praw[2 * k + 0] = sJ[q][0]; Praw[2 * k + 1] = sJ[q][1]; }
      }
    }
  };

  // =====================================================================================
  // M sequential feature updates (vi_ekf_meas.cpp:196-278)
  // =====================================================================================
  const bool partial = a.dp->use_partial_update != 0;
  // prologue: first valid measurement's columns + prediction
  int m = 0;
  auto meas_valid = [&](int mm_, int& slot_out) -> int {   // 0 ok, else result code
    const int slot = slot_all[(long)b * M + mm_];
    slot_out = slot;
    if (slot < 0) return -1;
    if (slot >= len) return 3;
    const double* z = z_all + ((long)b * M + mm_) * 2;
    if (z[0] != z[0] || z[1] != z[1]) return 2;
    return 0;
  };
  auto next_valid = [&](int from) -> int {   // first m' >= from that will actually run an update; writes codes of skipped ones
    int mm_ = from;
    while (mm_ < M) {
      int slot;
      const int code = meas_valid(mm_, slot);
      if (code == 0) break;
      if (result_all && tid == 0) result_all[(long)b * M + mm_] = code;
      mm_++;
    }
    return mm_;
  };
  int smp = 0;   // which half of the prediction mailbox (Hb, residual) phase A reads
  m = next_valid(0);
  if (m < M) {
    const int slot = slot_all[(long)b * M + m];
    apply_fixes(fixpar ^ 1);
    extract_cols(slot);
#ifndef ABL_FUPD
    if (tid == 0)
      res_feature_update(xs + xZ + 5 * slot, false, false, 0.0, 0.0, 0.0, a.dp, nullptr, nullptr, &flag,
                         z_all + ((long)b * M + m) * 2, sm);
#endif
  }
  __syncthreads();

  while (m < M) {
    const int slot = slot_all[(long)b * M + m];
    const int j0 = 16 + 3 * slot;
    const double* R = R_all + (long)b * r_stride_b + (long)m * r_stride_m;
    const int mnext = next_valid(m + 1);
    const int slot_next = (mnext < M) ? slot_all[(long)b * M + mnext] : -1;
    // ---- phase A: S, gate, K
    const double* smr = sm + 8 * smp;
    double* smw = sm + 8 * (smp ^ 1);
    const double h00 = smr[2], h01 = smr[3], h10 = smr[4], h11 = smr[5];
    const double r0 = smr[6], r1 = smr[7];
    double S[4], Si[4];
    {
      const double a0 = Praw[2 * j0 + 0], a1 = Praw[2 * j0 + 1], b0 = Praw[2 * (j0 + 1) + 0], b1 = Praw[2 * (j0 + 1) + 1];
      // W rows j0, j0+1
      const double w00 = a0 * h00 + a1 * h01, w01 = a0 * h10 + a1 * h11;
      const double w10 = b0 * h00 + b1 * h01, w11 = b0 * h10 + b1 * h11;
      S[0] = h00 * w00 + h01 * w10 + R[0];
      S[1] = h00 * w01 + h01 * w11 + R[2];
      S[2] = h10 * w00 + h11 * w10 + R[1];
      S[3] = h10 * w01 + h11 * w11 + R[3];
    }
    inv2(S, Si);
    const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;
    const bool gated = mahal > 9.0;
    int bad = 0;
    if (!gated) {
      for (int i = tid; i < n; i += T) {
        const double p0 = Praw[2 * i], p1 = Praw[2 * i + 1];
        const double w0 = p0 * h00 + p1 * h01, w1 = p0 * h10 + p1 * h11;
        const double k0 = w0 * Si[0] + w1 * Si[2], k1 = w0 * Si[1] + w1 * Si[3];
        Wt[2 * i] = w0; Wt[2 * i + 1] = w1;
        Kt[2 * i] = k0; Kt[2 * i + 1] = k1;
        if (k0 != k0 || k1 != k1) bad = 1;
      }
      if (h00 != h00 || h01 != h01 || h10 != h10 || h11 != h11) bad = 1;
    }
    bad = __syncthreads_or(bad);
    // ---- phase B: sweep, state correction, next measurement's columns + prediction
    apply_fixes(fixpar ^ 1);   // edits posted by the previous phase B (or by propagate)
    if (!gated && !bad) {
      // feature/feature blocks
      double2 wJ[CB][3];
#pragma unroll
      for (int ic = 0; ic < CB; ic++) {
        const int J = min(tc + TC * ic, N - 1);
#pragma unroll
        for (int s = 0; s < 3; s++) wJ[ic][s] = *reinterpret_cast<const double2*>(Wt + 2 * (16 + 3 * J + s));
      }
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        const int I = min(tr + TR * ia, N - 1);
        double2 kI[3];
#pragma unroll
        for (int r = 0; r < 3; r++) kI[r] = *reinterpret_cast<const double2*>(Kt + 2 * (16 + 3 * I + r));
#pragma unroll
        for (int ic = 0; ic < CB; ic++)
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
              const double t = fma(kI[r].y, wJ[ic][s].y, kI[r].x * wJ[ic][s].x);
              pb[ia][ic][r * 3 + s] = fma(-Lff[r * 3 + s], t, pb[ia][ic][r * 3 + s]);
            }
        __builtin_amdgcn_sched_barrier(0);
      }
      // strips
#pragma unroll
      for (int q = 0; q < SI; q++) {
        const int e = tc + TC * q, ia = e >> 4, k = e & 15, I = tr + TR * ia;
        if (own && e < RB * 16 && I < N) {
          const double2 wk = *reinterpret_cast<const double2*>(Wt + 2 * k);
          const double lk = lam[k];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const double2 ki = *reinterpret_cast<const double2*>(Kt + 2 * (16 + 3 * I + r));
            const double li = lam[16 + 3 * I + r];
            const double Lm = partial ? (lk + li - li * lk) : 1.0;
            sI[q][r] = fma(-Lm, fma(ki.y, wk.y, ki.x * wk.x), sI[q][r]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < SJ; q++) {
        const int e = tr + TR * q, ic = e >> 4, k = e & 15, J = tc + TC * ic;
        if (own && e < CB * 16 && J < N) {
          const double2 kk = *reinterpret_cast<const double2*>(Kt + 2 * k);
          const double lk = lam[k];
#pragma unroll
          for (int s = 0; s < 3; s++) {
            const double2 wj = *reinterpret_cast<const double2*>(Wt + 2 * (16 + 3 * J + s));
            const double lj = lam[16 + 3 * J + s];
            const double Lm = partial ? (lj + lk - lk * lj) : 1.0;
            sJ[q][s] = fma(-Lm, fma(kk.y, wj.y, kk.x * wj.x), sJ[q][s]);
          }
        }
      }
#pragma unroll
      for (int w = 0; w < BBE; w++) {
        const int e = (tid + T * w) & 255, r = e & 15, c = e >> 4;
        const double2 kr = *reinterpret_cast<const double2*>(Kt + 2 * r);
        const double2 wc = *reinterpret_cast<const double2*>(Wt + 2 * c);
        const double lr = lam[r], lc = lam[c];
        const double Lm = partial ? (lc + lr - lr * lc) : 1.0;
        sbb[w] = fma(-Lm, fma(kr.y, wc.y, kr.x * wc.x), sbb[w]);
      }
      // state correction x <- x [+] (lambda o K r)  (vi_ekf_meas.cpp:254-255 / :262-263)
#ifndef ABL_BUPD
      if (tid == T - 1) res_body_update(xs, Kt, lam, partial, r0, r1);
#endif
    }
    for (int f = tid; f < len; f += T) {
      double* xf = xs + xZ + 5 * f;
      const int d = 16 + 3 * f;
      double dv[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double l = partial ? lam[d + q] : 1.0;
        dv[q] = (l * Kt[2 * (d + q)]) * r0 + (l * Kt[2 * (d + q) + 1]) * r1;
      }
#ifndef ABL_FUPD
      res_feature_update(xf, !gated && !bad, !gated, dv[0], dv[1], dv[2], a.dp, &fixadd[fixpar * N + f],
                         &fixset[fixpar * N + f], &flag,
                         (f == slot_next) ? (z_all + ((long)b * M + mnext) * 2) : nullptr, smw);
#endif
    }
    if (result_all && tid == 0) result_all[(long)b * M + m] = gated ? 1 : 0;
    fixpar ^= 1;
    smp ^= 1;
    if (slot_next >= 0) extract_cols(slot_next);   // reads the swept registers
    // NOTE: a fix_depth edit touches P(rho,rho) only, never the zeta columns just extracted
    __syncthreads();
    m = mnext;
  }
  apply_fixes(fixpar ^ 1);

  // ---------------- store ----------------
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
#pragma unroll
  for (int ia = 0; ia < RB; ia++)
#pragma unroll
    for (int ic = 0; ic < CB; ic++) {
      const int I = tr + TR * ia, J = tc + TC * ic;
      if (own && I < N && J < N)
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
          for (int r = 0; r < 3; r++) P[(16 + 3 * I + r) + (long)(16 + 3 * J + s) * ld] = pb[ia][ic][r * 3 + s];
    }
#pragma unroll
  for (int q = 0; q < SI; q++) {
    const int e = tc + TC * q, ia = e >> 4, k = e & 15, I = tr + TR * ia;
    if (own && e < RB * 16 && I < N)
#pragma unroll
      for (int r = 0; r < 3; r++) P[(16 + 3 * I + r) + (long)k * ld] = sI[q][r];
  }
#pragma unroll
  for (int q = 0; q < SJ; q++) {
    const int e = tr + TR * q, ic = e >> 4, k = e & 15, J = tc + TC * ic;
    if (own && e < CB * 16 && J < N)
#pragma unroll
      for (int s = 0; s < 3; s++) P[k + (long)(16 + 3 * J + s) * ld] = sJ[q][s];
  }
#pragma unroll
  for (int w = 0; w < BBE; w++) {
    const int e = tid + T * w;
    if (e < 256) P[(e & 15) + (long)(e >> 4) * ld] = sbb[w];
  }
  if (flag) atomicOr(&a.flags[b], flag);
}

}  // namespace viekf
