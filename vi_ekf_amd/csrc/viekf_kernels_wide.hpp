// viekf_kernels_wide.hpp -- the propagate of the wide-P path (P in HBM, 77 < N <= 160) in the fused kernel's K = 24 RECORD form.
//
// numeric core of VIEKF::propagate_state (src/vi_ekf/vi_ekf.cpp:262-318); dynamics src/vi_ekf/vi_ekf_dyn.cpp:14-135.
//
// With D = blockdiag(Phi_ff) the feature part is  P+_ff = D P_ff D^T + Ut Dc^T + Dc Ut^T + Gs Gs^T  (viekf_resident_prop.hpp:
// Phi_fb = Dc Psi with the 9 body directions Psi = [E_v; A_bb[vel rows]; E_g], Ut = Dc Pi / 2 + V Psi^T, V = Phi_ff P[feat, body],
// Gs = Gd_f sqrt(Qu)).  One record per feature row -- {Ut (9), Dc (9), Gs (6)} interleaved as in the fused kernel, 26 doubles --
// serves BOTH sides of the symmetric coupling, so the whole set (nf x 26 doubles: 94 KB at N = 150) stays in LDS for the sweep.
// r01/r02 (k_propagate_stream<512, true>) staged two K = 38 operand sets X, Y in global scratch: 0.28 GB written and ~1.6 GB
// re-read per step at B = 1024, N = 150, and 10 k-steps per 48 x 48 super-tile instead of 6.
//
// Phases of one workgroup (512 threads, one filter): state -> LDS; body dynamics (one lane); feature dynamics (a lane per
// feature) and P_bb; row expansion (a lane per feature row), Phi_bb, Psi P_bb; T16, Gs_b, Pi; Xi, body state step; per feature
// ROW: V, Ut -> record, P+[row, body] -> HBM; body block; then the super-tile sweep of k_propagate_stream (R = P D_J^T,
// O^T = R^T D_I^T with the accumulators as the A operand, + 6 k-steps of records from LDS), lower triangle only.
#pragma once
#include "viekf_resident_prop.hpp"

namespace viekf {

struct WideLds {   // offsets in doubles
  int xs, Abb, Gb, Phibb, PhibbT, Mbb, Gdb, Pbb, T16, xdb, ctx, AvG, PsiP, Pi, Xi, phiff, Z, sm, total;
  __host__ __device__ WideLds(int N, int nxs) {
    int o = 0;
    auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
    xs = take(nxs); Abb = take(256); Gb = take(96); Phibb = take(256); PhibbT = take(256); Mbb = take(256); Gdb = take(96);
    Pbb = take(256); T16 = take(256); xdb = take(16); ctx = take((int)((sizeof(BodyCtx) + 7) / 8)); AvG = take(18);
    PsiP = take(ZK * 16); Pi = take(ZK * ZK); Xi = take(ZK * 16); phiff = take(9 * N); Z = take(3 * N * ZS); sm = take(8);
    total = o;
  }
};

template <int T>
__global__ __launch_bounds__(T) void k_propagate_wide(StreamArgs a, const double* __restrict__ u_all, const double* __restrict__ dt_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (a.active && !a.active[b]) return;
  const int n = a.n, ld = a.ld, N = a.N;
  const WideLds L(N, a.nxs);
  double* xs = smem + L.xs;
  double* Abb = smem + L.Abb; double* Gb = smem + L.Gb; double* Phibb = smem + L.Phibb; double* PhibbT = smem + L.PhibbT;
  double* Mbb = smem + L.Mbb; double* Gdb = smem + L.Gdb; double* Pbb = smem + L.Pbb; double* T16 = smem + L.T16;
  double* xdb = smem + L.xdb;
  BodyCtx* ctx = reinterpret_cast<BodyCtx*>(smem + L.ctx);
  double* AvG = smem + L.AvG; double* PsiP = smem + L.PsiP; double* Pi = smem + L.Pi; double* Xi = smem + L.Xi;
  double* phiff = smem + L.phiff; double* Z = smem + L.Z;
  const DevParams& prm = *a.dp;

  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nf = 3 * len, nact = 16 + nf;
  const double dt = dt_all[b];

  // ---- state -> LDS, cleared body Jacobians, P_bb (both copies of a pair from the lower triangle)
  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  for (int i = tid; i < 256; i += T) { Abb[i] = 0.0; Pbb[i] = P[max(i >> 4, i & 15) + (long)min(i >> 4, i & 15) * ld]; }
  for (int i = tid; i < 96; i += T) Gb[i] = 0.0;
  if (tid < 16) xdb[tid] = 0.0;
  __syncthreads();
  if (tid == 0) res_body_phase(xs, u_all + (long)b * 6, a.dp, ctx, xdb, Abb, Gb);   // vi_ekf_dyn.cpp:42-80 (rotates u by q_b_u)
  __syncthreads();
  // ---- feature dynamics: raw Jacobian blocks -> the even record slots, Phi_ff, state step (one lane per feature);
  //      A_v G_b for the row expansion on lanes that carry no feature
  for (int f = tid; f < N; f += T) res_feature_phase(f, len, dt, xs, ctx, Z, phiff);
  if (tid >= T - 18) {
    const int e = tid - (T - 18), j = e / 6, k = e - 6 * j;
    double sv = 0.0;
#pragma unroll 4
    for (int c = 0; c < 16; c++) sv += Abb[(dxVEL + j) * 16 + c] * Gb[c * 6 + k];
    AvG[e] = sv;
  }
  __syncthreads();
  // ---- row expansion (raw blocks -> Dc in the odd slots, Gs in 18..23), body transition blocks (vi_ekf.cpp:302-303), Psi P_bb
  for (int e = tid; e < 3 * N; e += T) res_feature_expand_row(e / 3, e % 3, dt, Z, Gb, AvG, prm.sqrtQu);
  for (int e = T - 1 - tid; e < 256; e += T) {
    const int r = e >> 4, c = e & 15;
    double a2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; k++) a2 += Abb[r * 16 + k] * Abb[k * 16 + c];
    const double id = (r == c) ? 1.0 : 0.0, av = Abb[e];
    Mbb[e] = id + av * (0.5 * dt) + a2 * (dt * dt * (1.0 / 6.0));
    const double ph = id + av * dt + a2 * (0.5 * dt * dt);
    Phibb[e] = ph;
    PhibbT[c * 16 + r] = ph;
  }
  for (int e = T - 1 - tid; e < ZK * 16; e += T) {   // Psi P_bb: rows E_v, A_bb[vel rows], E_g of P_bb
    const int q = e >> 4, c = e & 15;
    double v;
    if (q < 3) v = Pbb[(dxVEL + q) * 16 + c];
    else if (q >= 6) v = Pbb[(dxB_G + q - 6) * 16 + c];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += Abb[(dxVEL + q - 3) * 16 + k] * Pbb[k * 16 + c];
    }
    PsiP[e] = v;
  }
  __syncthreads();
  // ---- T16 = Phi_bb P_bb, Gs_b = M_bb G_b dt sqrt(Qu), Pi = (Psi P_bb) Psi^T
  for (int e = tid; e < 256; e += T) {
    const int r = e >> 4, c = e & 15;
    double sv = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; k++) sv += Phibb[r * 16 + k] * Pbb[k * 16 + c];
    T16[e] = sv;
  }
  for (int e = T - 1 - tid; e < 96; e += T) {
    const int r = e / 6, k = e % 6;
    double sv = 0.0;
#pragma unroll 4
    for (int c = 0; c < 16; c++) sv += Mbb[r * 16 + c] * Gb[c * 6 + k];
    Gdb[e] = sv * dt * prm.sqrtQu[k];
  }
  for (int e = tid; e < ZK * ZK; e += T) {
    const int q = e / ZK, j = e - q * ZK;
    const double* pr = PsiP + q * 16;
    double v;
    if (j < 3) v = pr[dxVEL + j];
    else if (j >= 6) v = pr[dxB_G + j - 6];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += pr[k] * Abb[(dxVEL + j - 3) * 16 + k];
    }
    Pi[e] = v;
  }
  __syncthreads();
  // ---- Xi = Psi T16^T  (T16^T = P_bb Phi_bb^T); body state step (every feature lane has read the old body state through ctx)
  for (int e = T - 1 - tid; e < ZK * 16; e += T) {
    const int q = e >> 4, c = e & 15;
    double v;
    if (q < 3) v = T16[c * 16 + dxVEL + q];
    else if (q >= 6) v = T16[c * 16 + dxB_G + q - 6];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += Abb[(dxVEL + q - 3) * 16 + k] * T16[c * 16 + k];
    }
    Xi[e] = v;
  }
  if (tid == 0) {
    double dxb[16], xo[17];
#pragma unroll
    for (int i = 0; i < 16; i++) dxb[i] = xdb[i] * dt;
    body_boxplus_fast(xs, dxb, xo);
#pragma unroll
    for (int i = 0; i < 17; i++) xs[i] = xo[i];
  }
  __syncthreads();
  // ---- one lane per feature ROW:  V = Phi_ff[f][r] P[f rows, body]  (the body columns of P: lanes along the rows, coalesced),
  //      Ut = Dc Pi / 2 + V Psi^T -> the even record slots (the raw blocks there were consumed two barriers ago),
  //      P+[row, body] = V Phi_bb^T + Dc Xi + Gs Gs_b^T -> HBM (the lower triangle's body columns; nothing else reads them here)
  //      (nf <= T, checked on the host: one row per lane.  Every lane reads the three rows of its feature, so all the reads
  //       come before any store)
  {
    const int row = tid;
    double out[16];
    if (row < nf) {
      const int f = row / 3, r = row - 3 * f;
      const double f0 = phiff[9 * f + 3 * r], f1 = phiff[9 * f + 3 * r + 1], f2 = phiff[9 * f + 3 * r + 2];
      const double* pr = P + (16 + 3 * f);
      double V[16];
#pragma unroll
      for (int k = 0; k < 16; k++) V[k] = f0 * pr[(long)k * ld] + f1 * pr[(long)k * ld + 1] + f2 * pr[(long)k * ld + 2];
      double* zr = Z + row * ZS;
      double Dc[ZK], Gs[6];
#pragma unroll
      for (int q = 0; q < ZK; q++) Dc[q] = zr[2 * q + 1];
#pragma unroll
      for (int g = 0; g < 6; g++) Gs[g] = zr[18 + g];
#pragma unroll
      for (int j = 0; j < ZK; j++) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < ZK; q++) s += Dc[q] * Pi[q * ZK + j];
        s *= 0.5;
        if (j < 3) s += V[dxVEL + j];
        else if (j >= 6) s += V[dxB_G + j - 6];
        else {
#pragma unroll
          for (int c = 0; c < 16; c++) s += V[c] * Abb[(dxVEL + j - 3) * 16 + c];
        }
        zr[2 * j] = s;
      }
#pragma unroll
      for (int c = 0; c < 16; c++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) s += V[k] * Phibb[c * 16 + k];
#pragma unroll
        for (int q = 0; q < ZK; q++) s += Dc[q] * Xi[q * 16 + c];
#pragma unroll
        for (int g = 0; g < 6; g++) s += Gs[g] * Gdb[c * 6 + g];
        out[c] = s;
      }
    }
    __syncthreads();
    if (row < nf) {
#pragma unroll
      for (int c = 0; c < 16; c++) P[(16 + row) + (long)c * ld] = out[c];
    }
  }
  // ---- body block  P_bb+ = T16 Phi_bb^T + Gs_b Gs_b^T + Qx, one value per pair (the (lower, higher) expression for both copies)
  for (int e = tid; e < 256; e += T) {
    const int r = min(e >> 4, e & 15), c = max(e >> 4, e & 15);
    double s = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; k++) s += T16[r * 16 + k] * Phibb[c * 16 + k];
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) g += Gdb[r * 6 + k] * Gdb[c * 6 + k];
    s += g;
    if (r == c) s += a.Qx[r];
    P[(e >> 4) + (long)(e & 15) * ld] = s;
  }
  __syncthreads();   // the records are complete

  // ---- feature/feature part on the matrix cores, one wave per 48 x 48 super-tile on or below the diagonal (see
  //      k_propagate_stream for the operand chaining); the coupling is 6 k-steps of records from LDS
  {
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    constexpr int NWV = T / 64;
    const int nst = (nf + 47) / 48;
    auto dop = [&](int sup, int q, int s) -> double {   // D[16 q + lr][4 s + lk] of the block-diagonal 48 x 48 of super-tile `sup`
      const int jp = 16 * q + lr, j = 4 * s + lk;
      const int fp = jp / 3, fq = j / 3, F = 16 * sup + fp;
      return (fp == fq && F < len) ? phiff[9 * F + (jp - 3 * fp) * 3 + (j - 3 * fq)] : 0.0;
    };
    // record slot of contraction index k = 4 s + lk on the Y (rows of J) and the X (rows of I) side:
    //   k < 9: Ut_k . Dc_k      9 <= k < 18: Dc_{k-9} . Ut_{k-9}      k >= 18: Gs . Gs
    int yoff[6], xoff[6];
#pragma unroll
    for (int s = 0; s < 6; s++) {
      const int k = 4 * s + lk;
      yoff[s] = (k < ZK) ? 2 * k : ((k < 2 * ZK) ? 2 * (k - ZK) + 1 : k);
      xoff[s] = (k < ZK) ? 2 * k + 1 : ((k < 2 * ZK) ? 2 * (k - ZK) : k);
    }
    for (int st = wave; st < nst * nst; st += NWV) {
      const int I = st % nst, J = st / nst;
      if (I < J) continue;
      const int r0 = 16 + 48 * I, c0 = 16 + 48 * J;
      double pA[3][12];
#pragma unroll
      for (int aa = 0; aa < 3; aa++)
#pragma unroll
        for (int s = 0; s < 12; s++) {
          const int pi = min(r0 + 16 * aa + lr, nact - 1), pj = min(c0 + 4 * s + lk, nact - 1);
          pA[aa][s] = P[max(pi, pj) + (long)min(pi, pj) * ld];
        }
      v4f64 R[3][3], O[3][3];
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int aa = 0; aa < 3; aa++) { R[aa][q] = v4f64{0.0, 0.0, 0.0, 0.0}; O[q][aa] = v4f64{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int q = 0; q < 3; q++) {
        constexpr int S0[3] = {0, 3, 7}, S1[3] = {4, 8, 11};   // k-steps (4 columns each) that touch the features of tile q
#pragma unroll
        for (int s = 0; s < 12; s++) {
          if (s < S0[q] || s > S1[q]) continue;
          const double bd = dop(J, q, s);
#pragma unroll
          for (int aa = 0; aa < 3; aa++) R[aa][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(pA[aa][s], bd, R[aa][q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int aa = 0; aa < 3; aa++) {
        constexpr int S0[3] = {0, 3, 7}, S1[3] = {4, 8, 11};
#pragma unroll
        for (int s = 0; s < 12; s++) {
          if (s < S0[aa] || s > S1[aa]) continue;
          const double bi = dop(I, aa, s);
#pragma unroll
          for (int q = 0; q < 3; q++) O[q][aa] = __builtin_amdgcn_mfma_f64_16x16x4f64(R[s / 4][q][s % 4], bi, O[q][aa], 0, 0, 0);
        }
      }
#pragma unroll 2
      for (int s = 0; s < 6; s++) {
        double yv[3], xv[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          yv[q] = Z[min(48 * J + 16 * q + lr, nf - 1) * ZS + yoff[s]];
          xv[q] = Z[min(48 * I + 16 * q + lr, nf - 1) * ZS + xoff[s]];
        }
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
          for (int aa = 0; aa < 3; aa++) O[q][aa] = __builtin_amdgcn_mfma_f64_16x16x4f64(yv[q], xv[aa], O[q][aa], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int aa = 0; aa < 3; aa++) {
          const int i = r0 + 16 * aa + lr;
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            const int j = c0 + 16 * q + lk + 4 * rg;
            if (i < nact && j < nact && i >= j) {
              double v = O[q][aa][rg];
              if (i == j) v += a.Qx[i];
              P[i + (long)j * ld] = v;
            }
          }
        }
    }
  }
  // inactive slots: Phi = I and G = 0 there, so only Qx is added (vi_ekf.cpp:139-144,304)
  for (int d = nact + tid; d < n; d += T) P[d + (long)d * ld] += a.Qx[d];
  __syncthreads();

  // ---- fix_depth (vi_ekf.cpp:311) and write the state back
  unsigned flag = 0;
  for (int i = tid; i < len; i += T) fix_depth_one(xs, P, ld, i, prm, flag);
  __syncthreads();
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[b], flag);
}

}  // namespace viekf
