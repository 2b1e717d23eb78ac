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


// ------------------------------------------------------------------------------------------------
// k_update_feat_panelsvc (r04): the grouped feature update of the wide-P path (k_update_feat_blocked, viekf_kernels_stream.hpp:
// the same panel of zeta columns in LDS, the same one pass over P per group of 16 measurements) with the SEQUENTIAL part of a group
// reduced to what is sequential -- the way a blocked factorisation treats its diagonal block -- and run one group AHEAD, under the
// previous group's pass over P.
// VIEKF::update for active FEAT measurements, src/vi_ekf/vi_ekf_meas.cpp:196-278 (h_feat :354-367, gate :230-239, K :241, NaN
// guard :247, partial update :249-258, fix_depth :271), applied in the caller's order.
//
// k_update_feat_blocked walks a group measurement by measurement with the whole workgroup: every thread runs the prediction /
// S^-1 / gate chain, forms its W row, (barrier) applies the update to the later panel columns of its row, (barrier): 7,400 clk
// = 3.1 us per measurement at N = 150 (tools/stamps_wide.py), during which the CU moves no byte of P -- 150 x 3.1 us x 4 dispatch
// rounds = 1.9 of the 7.3 ms of a step.  But measurement g + 1 needs from update g only (i) the state of ITS feature and (ii) the
// 2 x 2 zeta-zeta block of ITS feature: the rows of the group's own features.  So, per group G:
//   S. ONE wave (the last) runs the chain of the whole group alone, without a barrier: lane l holds panel row 16 + 3 f_q + c of group
//      feature q = l / 3, component c = l % 3 -- all 32 columns of it, in registers -- and the state of feature q; per measurement:
//      prediction / S^-1 / gate from lane values (v_readlane), W and K of the 48 rows, the later columns of those rows, the state
//      correction and fix_depth of the 16 group features (one instruction stream for all of them).  It leaves {Hb, S^-1, r, gate}
//      per measurement and the W values of the later features' zeta rows (the coupling every other row needs) in LDS.
//      It does so WHILE THE OTHER SEVEN WAVES RUN THE PASS OF GROUP G - 1: its 48 x 32 entries of P are read before that pass starts
//      and brought up to date by the wave itself from the W rows of group G - 1 (the same rank-32 correction the pass applies), as
//      are the states of its 16 features.
//   A. (after the pass) the panel of group G is loaded;  C. every thread brings ITS panel row up to date on its own: the same
//      recurrence over g with the row in registers, the couplings read from LDS -- no barrier, no panel traffic -- and stores its W
//      row; the NaN guard (:247) is checked on every row's K here.
//   D. The state of the features outside groups G and G + 1 and of the body follows from the stored W rows, measurement by measurement,
//      on the lanes that own them (no barrier either), then the pass of group G (under which S of group G + 1 runs).
// S speculates that no update of the group is NaN-guarded (it cannot know: the verdict needs every row).  Nothing is committed before
// C has checked: if a K turned out NaN the group is run again as groups of ONE measurement, where the verdict is exact before anything
// is applied (the update is then skipped, fix_depth runs, :247,271).  Rows: one per thread (T >= n).  BG = 16: 48 rows fit the wave.
// The service wave's copy of its 48 x 32 entries equals what the pass writes up to rounding (it sums the same products in another
// order): the chain's S^-1 and couplings carry that rounding, nothing else does -- P itself is only ever written by the pass.
// ------------------------------------------------------------------------------------------------
struct PsvLds {
  int xs, lam, Wp, Si, mail, ctab, cbuf, cpre, flags, diag, gsl, win, total;   // offsets in doubles; mail .. win: [2] (group parity)
  int mail_sz, ctab_sz, cbuf_sz, gsl_sz, win_sz;
  __host__ __device__ PsvLds(int N, int n, int nxs, int BG) {
    const int nr = (n + 15) & ~15, BLD = 2 * BG + 2, BWIN = 2 * BG;
    int o = 0;
    auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
    xs = take(nxs); lam = take(n); Wp = take(nr * BLD); Si = take(4 * BG);
    mail_sz = 12 * BG; ctab_sz = 4 * BG * (BG - 1) / 2; cbuf_sz = 6 * BG; gsl_sz = BG; win_sz = 7 * BWIN;
    mail = take(2 * mail_sz);             // per measurement {Hb[4], S^-1[4], r[2], gate, -}: service wave -> everybody
    ctab = take(2 * ctab_sz);             // per pair g < g': W_g of the two zeta rows of feature g' (the coupling of update g into pair g')
    cbuf = take(2 * cbuf_sz);             // the group features' state {q_zeta[4], rho, P(rho, rho)} as the service wave leaves it:
                                          // committed only when the speculation held
    cpre = take(cbuf_sz);                 // the same BEFORE the group's own chain (through the previous group): what a failed
                                          // speculation falls back to -- nobody else applied the previous group to these features
    flags = take(2);                      // ints: [parity] a K came out NaN (speculation failed); [2] the pass's unit counter
    diag = take(N > 0 ? N : 1);
    gsl = take(2 * gsl_sz);               // ints: [parity]{slot[BG], measurement index[BG]}
    win = take(2 * win_sz);               // staged window of the measurement list: z [BWIN][2], R [BWIN][4], slot [BWIN] (ints)
    total = o;
  }
};

// W row = panel row x Hb^T and K row = W row x S^-1 with the multiply-adds written out: the service wave and the row owners
// round the same way
__device__ __forceinline__ void psv_wk(double px, double py, const double* hb, const double* si, double& w0, double& w1, double& k0, double& k1) {
  w0 = fma(px, hb[0], py * hb[1]);
  w1 = fma(px, hb[2], py * hb[3]);
  k0 = fma(w0, si[0], w1 * si[2]);
  k1 = fma(w0, si[1], w1 * si[3]);
}
__device__ __forceinline__ double psv_sub(double p, double nk0, double nk1, double wx, double wy) {   // p - L (K_i . W_c), nk = -L K_i
  return fma(nk0, wx, fma(nk1, wy, p));
}
__device__ __forceinline__ double psv_rl(double v, int src) {   // lane src's value for the whole wave (v_readlane, lands in SGPRs)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int psv_tri(int g, int gp, int BG) { return g * BG - g * (g + 1) / 2 + (gp - g - 1); }   // g < gp

// The pass of the look-ahead kernel:  P -= Lambda o (K W^T)  on the lower triangle, 16 x 16 tiles on the fp64 matrix cores, like
// blk_pass (viekf_kernels_stream.hpp) -- but r04 measured that pass to be LATENCY-bound, not HBM-bound (B = 64 filters, a quarter of the
// traffic and all of it cache-resident, take 1.66 ms per step against 1.76 ms at B = 256): a wave walks its tiles one by one, every
// k-step a chain of three LDS reads -> K -> one MFMA into the same accumulator.  Here a wave works on FOUR tiles of a unit at a time: the
// four accumulator chains are independent, and the W rows of the unit's column block and the S^-1 columns -- the same for every tile of
// the unit -- are read once per k-step instead of once per tile and k-step (6 LDS reads per 4 MFMAs instead of 12).  The diagonal tiles
// (both orientations, second accumulator) are units of their own and keep the one-tile form.  Units are drawn from an LDS counter.
template <int T, int BLD>
__device__ __forceinline__ void psv_pass(double* __restrict__ P, int ld, int nact, int Gn, const double* Wp, const double* SiL,
                                         const double* lam, bool partial, int lane, int* ticket,
                                         unsigned long long* st = nullptr) {
  const int nt = (nact + 15) >> 4;
  const int lr = lane & 15, lk = lane >> 4;
  const int ksteps = (2 * Gn + 3) >> 2;                   // (columns past 2 Gn are zero)
#ifdef VIEKF_STAMPS
  int sn = 0;                                             // diagnostic build: the first sixteen marks of this wave's pass
#define PASS_STAMP()                                                                      \
  do {                                                                                    \
    if (st && sn < 16) {                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      if (lane == 0) st[sn] = __builtin_amdgcn_s_memtime();                               \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      sn++;                                                                               \
    }                                                                                     \
  } while (0)
#else
#define PASS_STAMP() do {} while (0)
#endif
  constexpr int TPI = 4;                                  // tiles per unit: 4 vertically adjacent ones (512 B contiguous per column)
  int NU = 0;                                             // units strictly below the diagonal: tickets 0 .. NU - 1, then the nt diagonal tiles
  for (int tj = 0; tj < nt; tj++) NU += (nt - 1 - tj + TPI - 1) / TPI;
  auto draw = [&]() {
    int t = 0;
    if (lane == 0) t = atomicAdd(ticket, 1);
    return __builtin_amdgcn_readfirstlane(t);
  };
  // ---- a unit strictly below the diagonal: column block tj, tiles ti0 .. ti0 + 3
  auto decode = [&](int t, int& tj, int& ti0) {
    int rest = t;
    for (tj = 0; tj < nt; tj++) {
      const int nu = (nt - 1 - tj + TPI - 1) / TPI;       // units of this column block (rows tj + 1 .. nt - 1)
      if (rest < nu) break;
      rest -= nu;
    }
    ti0 = tj + 1 + TPI * rest;
  };
  // the matrix-core work of a unit and the Lambda scaling: pv -= Lambda o (K_I . W_J^T)
  auto compute = [&](double (&pv)[TPI][4], int tj, int ti0) {
    const int j0t = 16 * tj;
    v4f64 acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = v4f64{0.0, 0.0, 0.0, 0.0};
    int ib[4];
#pragma unroll
    for (int q = 0; q < 4; q++) ib[q] = min(16 * (ti0 + q), 16 * (nt - 1)) + lr;   // (a tile past the end redoes the last one)
    for (int sk = 0; sk < ksteps; sk++) {                  // (two waves of a SIMD in this loop at once are bound by the matrix core:
      const int c = 4 * sk + lk;                           //  reading the next k-step's operands ahead gained nothing, r04)
      const double wjc = Wp[(j0t + lr) * BLD + c];         // this lane's contraction index: column c of pair c >> 1
      const double2 sv = *reinterpret_cast<const double2*>(SiL + 4 * (c >> 1) + 2 * (c & 1));
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double2 wi = *reinterpret_cast<const double2*>(Wp + ib[q] * BLD + (c & ~1));
        const double kic = wi.x * sv.x + wi.y * sv.y;      // K[i][c]
        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(wjc, kic, acc[q], 0, 0, 0);   // K_i . W_j
      }
    }
    PASS_STAMP();
    double lj[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) lj[rg] = lam[min(j0t + lk + 4 * rg, nact - 1)];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double li = lam[min(ib[q], nact - 1)];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const double Lij = partial ? (lj[rg] + li - li * lj[rg]) : 1.0;
        pv[q][rg] -= Lij * acc[q][rg];
      }
    }
    PASS_STAMP();
  };
  PASS_STAMP();
  int t = draw();
  if (ld >= 16 * nt) {
    // Padded columns (the streaming family's layout, viekf_capi.hip: ld a multiple of 16): every row 16 ti + lr < 16 nt of a column
    // below n exists in memory, and the panel rows nact .. 16 nt - 1 are ZERO (written once per launch), so the ragged last tile row
    // needs neither clamps nor predicates -- the update of its rows past nact is  P -= Lambda o (0 . W^T): the value read is written
    // back.  A unit is then 16 loads and 16 stores at 32-bit offsets from the workgroup's base (the address arithmetic and the
    // predicates were a quarter of a unit's time), and the loop is software-pipelined over two register sets: the loads of the NEXT
    // unit are issued before the matrix-core work of the current one.  (A unit of fewer than four tiles redoes its last tile in the
    // spare slots: the same values stored twice.)  Nothing branches between a load and its use: a join makes the compiler's
    // wait-count bookkeeping wait for BOTH sets.
    auto offs = [&](int tj, int ti0, int q, int rg) {
      return (unsigned)(min(16 * (ti0 + q), 16 * (nt - 1)) + lr) + (unsigned)(16 * tj + lk + 4 * rg) * (unsigned)ld;
    };
    auto issue = [&](int tt, double (&pv)[TPI][4], int& tj, int& ti0) {
      decode(tt, tj, ti0);
#pragma unroll
      for (int q = 0; q < TPI; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) pv[q][rg] = P[offs(tj, ti0, q, rg)];
    };
    auto store = [&](const double (&pv)[TPI][4], int tj, int ti0) {
#pragma unroll
      for (int q = 0; q < TPI; q++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) P[offs(tj, ti0, q, rg)] = pv[q][rg];
      PASS_STAMP();
    };
    if (t < NU) {
      double pvA[TPI][4], pvB[TPI][4];
      int tjA, tiA, tjB, tiB;
      issue(t, pvA, tjA, tiA);
      PASS_STAMP();
      for (;;) {                                          // (the issue is unconditional -- past the last unit it reloads that one and
        t = draw();                                       //  drops it)
        PASS_STAMP();
        issue(min(t, NU - 1), pvB, tjB, tiB);
        PASS_STAMP();
        compute(pvA, tjA, tiA);
        store(pvA, tjA, tiA);
        if (t >= NU) break;
        t = draw();
        PASS_STAMP();
        issue(min(t, NU - 1), pvA, tjA, tiA);
        PASS_STAMP();
        compute(pvB, tjB, tiB);
        store(pvB, tjB, tiB);
        if (t >= NU) break;
      }
    }
  } else {
    // dense columns (the on-chip family's layout, when this kernel is forced at a small N): clamped loads, predicated stores
    for (; t < NU; t = draw()) {
      int tj, ti0;
      decode(t, tj, ti0);
      const int j0t = 16 * tj;
      double pv[TPI][4];
#pragma unroll
      for (int q = 0; q < TPI; q++) {
        const int i = min(16 * (ti0 + q) + lr, nact - 1);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) pv[q][rg] = P[i + (long)min(j0t + lk + 4 * rg, nact - 1) * ld];
      }
      compute(pv, tj, ti0);
#pragma unroll
      for (int q = 0; q < TPI; q++) {
        const int i = 16 * (ti0 + q) + lr;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int j = j0t + lk + 4 * rg;
          if (ti0 + q < nt && i < nact && j < nact) P[i + (long)j * ld] = pv[q][rg];
        }
      }
    }
  }
  // ---- the diagonal tiles: K_i . W_j for i >= j and the mirror expression K_j . W_i (same products, same order) for i < j
  for (; t < NU + nt; t = draw()) {
    const int j0t = 16 * (t - NU), i = j0t + lr;
    double pv[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) pv[rg] = P[min(i, nact - 1) + (long)min(j0t + lk + 4 * rg, nact - 1) * ld];
    v4f64 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    for (int sk = 0; sk < ksteps; sk++) {
      const int c = 4 * sk + lk;
      const double2 wi = *reinterpret_cast<const double2*>(Wp + i * BLD + (c & ~1));
      const double2 sv = *reinterpret_cast<const double2*>(SiL + 4 * (c >> 1) + 2 * (c & 1));
      const double wic = (c & 1) ? wi.y : wi.x;
      const double kic = wi.x * sv.x + wi.y * sv.y;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wic, kic, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(kic, wic, acc2, 0, 0, 0);
    }
    const double li = lam[min(i, nact - 1)];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int j = j0t + lk + 4 * rg;
      const double lj = lam[min(j, nact - 1)];
      const double Lij = partial ? (lj + li - li * lj) : 1.0;
      const double av = (i < j) ? acc2[rg] : acc[rg];
      if (i < nact && j < nact) P[i + (long)j * ld] = pv[rg] - Lij * av;
    }
  }
}

// Diagnostic build only (-DVIEKF_STAMPS, tools/stamps_wide.py): s_memtime stamps of filter 0, lane 0 of every wave, into the
// workspace: [16 wave + idx], in the SECOND trip of the group loop (the first overlapped one)
#ifdef VIEKF_STAMPS
#define PSV_STAMP(idx)                                                                                          \
  do {                                                                                                          \
    if (b == 0 && lane == 0 && stamp_iter == 1) {                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      reinterpret_cast<unsigned long long*>(a.ws)[16 * wave + (idx)] = __builtin_amdgcn_s_memtime();            \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
    }                                                                                                           \
  } while (0)
#else
#define PSV_STAMP(idx) do {} while (0)
#endif

template <int T, int BG>
__global__ __launch_bounds__(T) void k_update_feat_panelsvc(StreamArgs a, const double* __restrict__ z_all,
                                                            const int* __restrict__ slot_all, int M,
                                                            const double* __restrict__ R_all, long r_stride_b,
                                                            long r_stride_m, int* __restrict__ result_all) {
  static_assert(3 * BG <= 64, "the rows of a group's features have to fit one wave");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (a.active && !a.active[b]) return;
  const int n = a.n, ld = a.ld;
  constexpr int BLD = 2 * BG + 2, BWIN = 2 * BG, NC = 2 * BG;
  constexpr int NWV = T / 64;
  const PsvLds L(a.N, n, a.nxs, BG);
  double* xs = smem + L.xs;
  double* lam = smem + L.lam;
  double* Wp = smem + L.Wp;     // panel of raw columns; after phase C: the W rows
  double* SiL = smem + L.Si;    // per pair g: the two COLUMNS of S^-1 (zero if the update was skipped): operand of the pass
  int* flg = reinterpret_cast<int*>(smem + L.flags);
  double* diag = smem + L.diag; // running P(rho_f, rho_f), kept for the whole launch
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nact = 16 + 3 * len;
  const DevParams& prm = *a.dp;
  const bool partial = prm.use_partial_update != 0;
  unsigned flag = 0;
  const int lane = tid & 63, wave = tid >> 6;
  const bool svc = wave == NWV - 1;
  const double rho_reset = 1.0 / (2.0 * prm.min_depth), p0rr = prm.P0_feat[2];
#ifdef VIEKF_STAMPS
  int stamp_iter = -1;
#endif

  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  for (int i = tid; i < n; i += T) lam[i] = a.lambda[i];
  for (int f = tid; f < len; f += T) diag[f] = P[(16 + 3 * f + 2) + (long)(16 + 3 * f + 2) * ld];
  for (int i = nact * BLD + tid; i < ((n + 15) & ~15) * BLD; i += T) Wp[i] = 0.0;   // the panel rows past the active ones stay zero (psv_pass)
  __syncthreads();

  // fix_depth (vi_ekf_helper.cpp:128-156) of one feature: rho and its P(rho, rho), both in the caller's registers
  auto fix_depth_v = [&](double& rho, double& prr) {
    if (rho != rho) { rho = rho_reset; flag |= FLAG_NAN; }
    if (rho < 0.0) {
      const double err = rho_reset - rho;
      prr += err * err;
      rho = rho_reset;
      flag |= FLAG_NEGDEPTH;
    } else if (rho > 1e2) {
      prr = p0rr;
      rho = rho_reset;
    }
  };
  auto in_set = [](int f, unsigned long long s0, unsigned long long s1, unsigned long long s2) {
    return ((f < 64 ? s0 : (f < 128 ? s1 : s2)) >> (f & 63)) & 1ull;
  };
  auto krow = [&](int row, int g, const double* Si, double& q0, double& q1) {   // K row re-formed from the stored W row
    const double2 w = *reinterpret_cast<const double2*>(Wp + row * BLD + 2 * g);
    q0 = fma(w.x, Si[0], w.y * Si[2]); q1 = fma(w.x, Si[1], w.y * Si[3]);
  };

  // ---- group state: `cur` is the group being formed / processed, `prev` the one whose pass is pending
  int m = 0;                         // next entry of the measurement list
  int single_until = -1;             // below this index of the list the groups hold ONE measurement (a failed speculation is redone so)
  int pb = 0;                        // parity of cur's LDS buffers
  int Gn = 0, mbase = 0;             // cur: measurements, where its window starts in the list
  unsigned long long c0 = 0, c1 = 0, c2 = 0;        // cur's feature slots
  int Gp = 0;                        // prev: measurements (0: no pass pending)
  unsigned long long p0 = 0, p1 = 0, p2 = 0;        // prev's feature slots

  // forms the next group from a staged window of the list (as k_update_feat_blocked); all threads; two barriers inside
  auto form_group = [&]() {
    double* wz = smem + L.win + pb * L.win_sz;
    double* wR = wz + 2 * BWIN;
    int* wsl = reinterpret_cast<int*>(wR + 4 * BWIN);
    int* gsl = reinterpret_cast<int*>(smem + L.gsl + pb * L.gsl_sz);
    int* gml = gsl + BG;
    Gn = 0; c0 = 0; c1 = 0; c2 = 0;
    if (tid == 0) flg[2] = 0;                             // (the unit counter of the coming pass)
    while (Gn == 0 && m < M) {
      __syncthreads();
      if (tid < BWIN && m + tid < M) {
        const long mi = (long)b * M + m + tid;
        const int slot = slot_all[mi];
        const double z0 = z_all[2 * mi], z1 = z_all[2 * mi + 1];
        int code = 0;
        if (slot < 0) code = -1;
        else if (slot >= len) code = 3;                   // MEAS_INVALID
        else if (z0 != z0 || z1 != z1) code = 2;          // MEAS_NAN (vi_ekf_meas.cpp:136-137)
        wsl[tid] = code == 0 ? slot : -1;
        wz[2 * tid] = z0; wz[2 * tid + 1] = z1;
        const double* R = R_all + (long)b * r_stride_b + (long)(m + tid) * r_stride_m;   // column-major 2x2
        wR[4 * tid] = R[0]; wR[4 * tid + 1] = R[1]; wR[4 * tid + 2] = R[2]; wR[4 * tid + 3] = R[3];
        if (code != 0 && result_all) result_all[mi] = code;
      }
      if (tid == 0) flg[pb] = 0;
      __syncthreads();
      int mm = m;
      const int mend = min(M, m + BWIN);
      const int cap = (m < single_until) ? 1 : BG;
      while (mm < mend && Gn < cap) {
        const int slot = wsl[mm - m];
        if (slot >= 0) {
          unsigned long long& w = slot < 64 ? c0 : (slot < 128 ? c1 : c2);
          const unsigned long long bit = 1ull << (slot & 63);
          if (w & bit) break;                             // repeated slot: it opens the next group
          w |= bit;
          if (tid == 0) { gsl[Gn] = slot; gml[Gn] = mm - m; }   // (index into the window)
          Gn++;
        }
        mm++;
      }
      mbase = m;
      m = mm;
    }
    __syncthreads();                                      // gsl / gml visible
    // (every thread scanned the same LDS words: the results are wave-uniform, but the compiler cannot know -- say so, or every
    //  `g < Gn` below becomes an exec-mask branch and the unrolled rows go to scratch)
    Gn = __builtin_amdgcn_readfirstlane(Gn); m = __builtin_amdgcn_readfirstlane(m); mbase = __builtin_amdgcn_readfirstlane(mbase);
    auto uni64 = [](unsigned long long v) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
      return ((unsigned long long)hi << 32) | lo;
    };
    c0 = uni64(c0); c1 = uni64(c1); c2 = uni64(c2);
  };

  form_group();

  for (;;) {
#ifdef VIEKF_STAMPS
    stamp_iter++;
#endif
    PSV_STAMP(0);
    // the service wave's row of cur's 48 x 32 entries of P as they stand in memory NOW: read before the pending pass starts to
    // rewrite them.  (Assigned on every thread -- zeros on the other waves -- so that the registers are free again behind the barrier.)
    double sp[NC];
    {
      const int* gsl = reinterpret_cast<const int*>(smem + L.gsl + pb * L.gsl_sz);
      const bool mine = svc && lane < 3 * Gn;
      const int fq = gsl[min(lane / 3, max(Gn, 1) - 1)];
      const int row = 16 + 3 * fq + (lane - 3 * (lane / 3));
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int col = 16 + 3 * gsl[(c < 2 * Gn) ? (c >> 1) : 0] + (c & 1);
        sp[c] = (mine && c < 2 * Gn) ? P[max(row, col) + (long)min(row, col) * ld] : 0.0;
      }
    }
    __syncthreads();
    // ================= S: the service wave runs cur's chain  ||  the others: prev's state corrections (D) and pass
    if (svc) {
      if (Gn > 0) {
        // (the chain is ONE wave's serial work next to seven waves of pass: it gets the issue slots it asks for)
        __builtin_amdgcn_s_setprio(3);
        double* mail = smem + L.mail + pb * L.mail_sz;
        double* ctab = smem + L.ctab + pb * L.ctab_sz;
        double* cbuf = smem + L.cbuf + pb * L.cbuf_sz;
        const double* wz = smem + L.win + pb * L.win_sz;
        const double* wR = wz + 2 * BWIN;
        const int* gsl = reinterpret_cast<const int*>(smem + L.gsl + pb * L.gsl_sz);
        const int* gml = gsl + BG;
        const double* pmail = smem + L.mail + (pb ^ 1) * L.mail_sz;   // prev's
        const int sq_l = lane / 3, sc_l = lane - 3 * sq_l;
        const bool svalid = lane < 3 * Gn;
        const int fq = gsl[min(sq_l, Gn - 1)];
        const int row = 16 + 3 * fq + sc_l;
        const double li = lam[row];
        double La = 1.0, Lb = 1.0, l = 1.0, Lii = 1.0;
        if (partial) { const double la = lam[16], lb = lam[17]; La = la + li - li * la; Lb = lb + li - li * lb; l = li; Lii = li + li - li * li; }
        double sq[4], srho, sprr;                          // state and P(rho, rho) of this lane's feature (used on the lanes with c = 0)
        {
          const double* xf = xs + xZ + 5 * fq;
          sq[0] = xf[0]; sq[1] = xf[1]; sq[2] = xf[2]; sq[3] = xf[3]; srho = xf[4];
          sprr = diag[fq];
        }
        if (Gp > 0) {
          // prev's pass is pending: this lane's entries and its feature's state take prev's updates here (the same rank-2 terms the
          // pass applies / the corrections of D; a feature that was IN prev already has its state through prev)
          // (two loops -- the 48 x 32 entries, then the states -- so that the row, the operands in flight and the manifold step's
          //  temporaries are not all live at once)
#pragma unroll 1
          for (int g = 0; g < Gp; g++) {
            const double* ml = pmail + 12 * g;
            if (ml[10] != 0.0) continue;                   // gated: nothing was applied
            const double Si[4] = {ml[4], ml[5], ml[6], ml[7]};
            // (the W rows this correction couples with are the zeta rows of cur's features: this wave's own lanes hold them)
            const double2 w = *reinterpret_cast<const double2*>(Wp + row * BLD + 2 * g);
            const double k0 = fma(w.x, Si[0], w.y * Si[2]), k1 = fma(w.x, Si[1], w.y * Si[3]);
            const double a0 = -La * k0, a1 = -La * k1, b0 = -Lb * k0, b1 = -Lb * k1;
#pragma unroll
            for (int c = 0; c < NC; c += 2) {
              if (c < 2 * Gn) {                            // (uniform)
                const int Lc = 3 * (c >> 1);
                const double wa0 = psv_rl(w.x, Lc), wa1 = psv_rl(w.y, Lc), wb0 = psv_rl(w.x, Lc + 1), wb1 = psv_rl(w.y, Lc + 1);
                sp[c] = psv_sub(sp[c], a0, a1, wa0, wa1);
                sp[c + 1] = psv_sub(sp[c + 1], b0, b1, wb0, wb1);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          const bool fresh = !in_set(fq, p0, p1, p2);
#pragma unroll 1
          for (int g = 0; g < Gp; g++) {
            const double* ml = pmail + 12 * g;
            if (ml[10] != 0.0) continue;
            const double Si[4] = {ml[4], ml[5], ml[6], ml[7]};
            const double2 w = *reinterpret_cast<const double2*>(Wp + row * BLD + 2 * g);
            const double k0 = fma(w.x, Si[0], w.y * Si[2]), k1 = fma(w.x, Si[1], w.y * Si[3]);
            const double dv = (l * k0) * ml[8] + (l * k1) * ml[9];
            const double kw = Lii * (k0 * w.x + k1 * w.y);
            const double dv1 = __shfl_down(dv, 1), dv2 = __shfl_down(dv, 2), kw2 = __shfl_down(kw, 2);
            if (fresh) {
              double qn[4];
              q_feat_boxplus_fast(sq, dv, dv1, qn);
              sq[0] = qn[0]; sq[1] = qn[1]; sq[2] = qn[2]; sq[3] = qn[3];
              srho += dv2;
              sprr -= kw2;
              fix_depth_v(srho, sprr);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (svalid && sc_l == 0) {
          double* cp = smem + L.cpre + 6 * sq_l;
          cp[0] = sq[0]; cp[1] = sq[1]; cp[2] = sq[2]; cp[3] = sq[3]; cp[4] = srho; cp[5] = sprr;
        }
        PSV_STAMP(7);
        int sbad = 0;
#pragma unroll
        for (int g = 0; g < BG; g++) {                     // (unrolled: the column indices are compile-time, the row stays in registers)
          if (g < Gn) {
            const int wi = gml[g], L0 = 3 * g;
            // prediction, innovation, S^-1, gate of measurement g (:209-239): uniform values from the lanes of feature g
            double qg[4], zhat[2], hb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) qg[k] = psv_rl(sq[k], L0);
            {
              double t1[3], t2[3], zt[3];
              bearing_frame_fast(qg, t1, t2, zt);
              h_feat_frame(t1, t2, zt, prm, zhat, hb);
            }
            const double r0 = wz[2 * wi] - zhat[0], r1 = wz[2 * wi + 1] - zhat[1];
            const double* R = wR + 4 * wi;
            const double p00 = psv_rl(sp[2 * g], L0), p01 = psv_rl(sp[2 * g + 1], L0), p10 = psv_rl(sp[2 * g], L0 + 1), p11 = psv_rl(sp[2 * g + 1], L0 + 1);
            double S[4], Si[4];
            {
              const double w00 = p00 * hb[0] + p01 * hb[1], w01 = p00 * hb[2] + p01 * hb[3];
              const double w10 = p10 * hb[0] + p11 * hb[1], w11 = p10 * hb[2] + p11 * hb[3];
              S[0] = hb[0] * w00 + hb[1] * w10 + R[0];
              S[1] = hb[0] * w01 + hb[1] * w11 + R[2];
              S[2] = hb[2] * w00 + hb[3] * w10 + R[1];
              S[3] = hb[2] * w01 + hb[3] * w11 + R[3];
            }
            inv2_fast(S, Si);
            const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;   // :234
            const bool gate = mahal > 9.0;
            if (hb[0] != hb[0] || hb[1] != hb[1] || hb[2] != hb[2] || hb[3] != hb[3]) sbad = 1;
            if (lane == 0) {
              double* ml = mail + 12 * g;
              ml[0] = hb[0]; ml[1] = hb[1]; ml[2] = hb[2]; ml[3] = hb[3];
              ml[4] = Si[0]; ml[5] = Si[1]; ml[6] = Si[2]; ml[7] = Si[3];
              ml[8] = r0; ml[9] = r1; ml[10] = gate ? 1.0 : 0.0; ml[11] = 0.0;
            }
            if (!gate) {                                   // (a gated measurement changes nothing, not even through fix_depth: :235-239)
              double w0, w1, k0, k1;
              psv_wk(sp[2 * g], sp[2 * g + 1], hb, Si, w0, w1, k0, k1);
              if (svalid && (k0 != k0 || k1 != k1)) sbad = 1;
              const double a0 = -La * k0, a1 = -La * k1, b0 = -Lb * k0, b1 = -Lb * k1;
              // the later column pairs of this row, and the couplings for everybody else's rows
#pragma unroll
              for (int gp = g + 1; gp < BG; gp++) {
                if (gp < Gn) {                             // (uniform)
                  const int Lj = 3 * gp;
                  const double wa0 = psv_rl(w0, Lj), wa1 = psv_rl(w1, Lj), wb0 = psv_rl(w0, Lj + 1), wb1 = psv_rl(w1, Lj + 1);
                  sp[2 * gp] = psv_sub(sp[2 * gp], a0, a1, wa0, wa1);
                  sp[2 * gp + 1] = psv_sub(sp[2 * gp + 1], b0, b1, wb0, wb1);
                  if (lane == 0) {
                    double* ct = ctab + 4 * psv_tri(g, gp, BG);
                    *reinterpret_cast<double2*>(ct) = make_double2(wa0, wa1);
                    *reinterpret_cast<double2*>(ct + 2) = make_double2(wb0, wb1);
                  }
                }
              }
              // state correction of the group's features  x <- x [+] (lambda o K r)  (:254-255), rho-rho diagonal, fix_depth (:271)
              const double dv = (l * k0) * r0 + (l * k1) * r1;
              const double kw = Lii * (k0 * w0 + k1 * w1);
              const double dv1 = __shfl_down(dv, 1), dv2 = __shfl_down(dv, 2), kw2 = __shfl_down(kw, 2);
              double qn[4];
              q_feat_boxplus_fast(sq, dv, dv1, qn);
              sq[0] = qn[0]; sq[1] = qn[1]; sq[2] = qn[2]; sq[3] = qn[3];
              srho += dv2;
              sprr -= kw2;
              fix_depth_v(srho, sprr);
            }
          }
        }
        if (sbad) flg[pb] = 1;
        if (svalid && sc_l == 0) {
          double* cb = cbuf + 6 * sq_l;
          cb[0] = sq[0]; cb[1] = sq[1]; cb[2] = sq[2]; cb[3] = sq[3]; cb[4] = srho; cb[5] = sprr;
        }
        __builtin_amdgcn_s_setprio(0);
      }
    } else if (Gp > 0) {
      // ---- D of prev: the features outside prev and cur (prev's are committed, cur's ride with the service wave) and the body,
      //      measurement by measurement from the stored W rows, on the lanes that own them; then prev's pass on these seven waves
      const double* pmail = smem + L.mail + (pb ^ 1) * L.mail_sz;
      if (tid == T - 65) {                                 // (a worker thread without a feature of its own: T - 65 >= len)
#pragma unroll 1
        for (int g = 0; g < Gp; g++) {
          const double* ml = pmail + 12 * g;
          if (ml[10] != 0.0) continue;
          const double Si[4] = {ml[4], ml[5], ml[6], ml[7]}, r0 = ml[8], r1 = ml[9];
          double dxb[16], xo[17];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const double lq = partial ? lam[q] : 1.0;
            double q0, q1;
            krow(q, g, Si, q0, q1);
            dxb[q] = (lq * q0) * r0 + (lq * q1) * r1;
          }
          body_boxplus_fast(xs, dxb, xo);
#pragma unroll
          for (int q = 0; q < 17; q++) xs[q] = xo[q];
        }
      }
      for (int f = tid; f < len; f += T - 64) {
        if (in_set(f, p0, p1, p2) || in_set(f, c0, c1, c2)) continue;
        const int d = 16 + 3 * f;
        double* xf = xs + xZ + 5 * f;
        double qf[4] = {xf[0], xf[1], xf[2], xf[3]}, rho = xf[4], prr = diag[f];
        const double l0 = partial ? lam[d] : 1.0, l1 = partial ? lam[d + 1] : 1.0, l2 = partial ? lam[d + 2] : 1.0;
        const double Lii = partial ? (l2 + l2 - l2 * l2) : 1.0;
#pragma unroll 1
        for (int g = 0; g < Gp; g++) {
          const double* ml = pmail + 12 * g;
          if (ml[10] != 0.0) continue;
          const double Si[4] = {ml[4], ml[5], ml[6], ml[7]}, r0 = ml[8], r1 = ml[9];
          double a0, a1, b0, b1, e0, e1;
          krow(d, g, Si, a0, a1); krow(d + 1, g, Si, b0, b1); krow(d + 2, g, Si, e0, e1);
          const double dv0 = (l0 * a0) * r0 + (l0 * a1) * r1, dv1 = (l1 * b0) * r0 + (l1 * b1) * r1, dv2 = (l2 * e0) * r0 + (l2 * e1) * r1;
          double qn[4];
          q_feat_boxplus_fast(qf, dv0, dv1, qn);
          qf[0] = qn[0]; qf[1] = qn[1]; qf[2] = qn[2]; qf[3] = qn[3];
          rho += dv2;
          const double2 wr = *reinterpret_cast<const double2*>(Wp + (d + 2) * BLD + 2 * g);
          prr -= Lii * (e0 * wr.x + e1 * wr.y);
          fix_depth_v(rho, prr);
        }
        xf[0] = qf[0]; xf[1] = qf[1]; xf[2] = qf[2]; xf[3] = qf[3]; xf[4] = rho;
        diag[f] = prr;
      }
      PSV_STAMP(1);
    }
    PSV_STAMP(2);
    // prev's pass over P (psv_pass above); the units are drawn from a counter, so the waves that come late (the three with feature
    // lanes, the one with the body lane, the service wave) take what is left.  (The rho-rho diagonal is kept in LDS: written at the end.)
#ifdef VIEKF_STAMPS
    if (Gp > 0) psv_pass<T, BLD>(P, ld, nact, Gp, Wp, SiL, lam, partial, lane, flg + 2,
                                 (b == 0 && stamp_iter == 1) ? reinterpret_cast<unsigned long long*>(a.ws) + 128 + 16 * wave : nullptr);
#else
    if (Gp > 0) psv_pass<T, BLD>(P, ld, nact, Gp, Wp, SiL, lam, partial, lane, flg + 2);
#endif
    PSV_STAMP(6);
    __syncthreads();                                       // prev's pass is done; cur's mail / ctab / cbuf are ready
    PSV_STAMP(3);
    Gp = 0;
    if (Gn == 0) break;
    const double* mail = smem + L.mail + pb * L.mail_sz;
    const double* ctab = smem + L.ctab + pb * L.ctab_sz;
    const int* gsl = reinterpret_cast<const int*>(smem + L.gsl + pb * L.gsl_sz);
    const int* gml = gsl + BG;

    // ================= A: panel <- the zeta columns of cur's features from the lower triangle of P (one row per thread)
    for (int i = tid; i < nact; i += T) {
      double v[NC];                                        // (all 32 column loads of the row in flight: this phase is four memory
#pragma unroll                                             //  latencies long with eight at a time, and nothing else runs beside it)
      for (int c = 0; c < NC; c++) {
        const int col = 16 + 3 * gsl[(c < 2 * Gn) ? (c >> 1) : 0] + (c & 1);
        v[c] = (c < 2 * Gn) ? P[max(i, col) + (long)min(i, col) * ld] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < NC; c += 2) *reinterpret_cast<double2*>(Wp + i * BLD + c) = make_double2(v[c], v[c + 1]);
    }
    if (tid < 4 * BG) SiL[tid] = 0.0;
    PSV_STAMP(4);

    // ================= C: every row follows on its own (its raw values are its own: no barrier after A); W rows stored; NaN guard
    {
      const int i = tid;                                   // this thread's row of the panel (T >= n, checked on the host)
      if (i < nact) {
        double rw[NC];
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
          const double2 v = *reinterpret_cast<const double2*>(Wp + i * BLD + c);
          rw[c] = v.x; rw[c + 1] = v.y;
        }
        double La = 1.0, Lb = 1.0;
        if (partial) { const double li = lam[i], la = lam[16], lb = lam[17]; La = la + li - li * la; Lb = lb + li - li * lb; }
        int bad = 0;
#pragma unroll
        for (int g = 0; g < BG; g++) {                     // (unrolled: the row stays in registers under compile-time indices)
          if (g < Gn) {
            const double* ml = mail + 12 * g;
            if (ml[10] != 0.0) {                           // gated: a zero pair (nothing to apply in the pass)
              *reinterpret_cast<double2*>(Wp + i * BLD + 2 * g) = make_double2(0.0, 0.0);
            } else {
              double hb[4], Si[4], w0, w1, k0, k1;
#pragma unroll
              for (int k = 0; k < 4; k++) { hb[k] = ml[k]; Si[k] = ml[4 + k]; }
              psv_wk(rw[2 * g], rw[2 * g + 1], hb, Si, w0, w1, k0, k1);
              if (k0 != k0 || k1 != k1) bad = 1;
              *reinterpret_cast<double2*>(Wp + i * BLD + 2 * g) = make_double2(w0, w1);
              const double a0 = -La * k0, a1 = -La * k1, b0 = -Lb * k0, b1 = -Lb * k1;
#pragma unroll
              for (int gp = g + 1; gp < BG; gp++) {
                if (gp < Gn) {
                  const double* ct = ctab + 4 * psv_tri(g, gp, BG);
                  const double2 wa = *reinterpret_cast<const double2*>(ct), wb = *reinterpret_cast<const double2*>(ct + 2);
                  rw[2 * gp] = psv_sub(rw[2 * gp], a0, a1, wa.x, wa.y);
                  rw[2 * gp + 1] = psv_sub(rw[2 * gp + 1], b0, b1, wb.x, wb.y);
                }
              }
            }
          }
        }
        if (bad) flg[pb] = 1;
      }
    }
    __syncthreads();                                       // W rows complete; the verdict on the speculation
    PSV_STAMP(5);
    const bool failed = flg[pb] != 0;
    if (failed && tid < Gn) {                              // the group's features as they stood BEFORE its chain (the previous group's
      const double* cp = smem + L.cpre + 6 * tid;          // corrections reached them through the service wave only)
      const int fq = gsl[tid];
      double* xf = xs + xZ + 5 * fq;
      xf[0] = cp[0]; xf[1] = cp[1]; xf[2] = cp[2]; xf[3] = cp[3]; xf[4] = cp[4];
      diag[fq] = cp[5];
    }
    if (failed && Gn > 1) {                                // redo these measurements one per group: nothing else has been committed
      single_until = m;
      m = mbase;
      __syncthreads();                                     // (everybody has read the verdict before the flag is cleared)
      form_group();
      continue;
    }
    // ================= commit.  (failed with ONE measurement: the exact NaN-guard verdict -- update skipped, fix_depth runs, :247,271)
    if (tid < Gn) {
      const double* ml = mail + 12 * tid;
      const bool gate = ml[10] != 0.0;
      if (!gate && !failed) { SiL[4 * tid + 0] = ml[4]; SiL[4 * tid + 1] = ml[6]; SiL[4 * tid + 2] = ml[5]; SiL[4 * tid + 3] = ml[7]; }
      if (result_all) result_all[(long)b * M + mbase + gml[tid]] = gate ? 1 : 0;
    }
    if (failed) {
      __syncthreads();                                     // (the fall-back state above is in place)
      if (mail[10] == 0.0)
        for (int f = tid; f < len; f += T) { double rho = xs[xZ + 5 * f + 4], prr = diag[f]; fix_depth_v(rho, prr); xs[xZ + 5 * f + 4] = rho; diag[f] = prr; }
      Gp = 0; p0 = 0; p1 = 0; p2 = 0;                      // (no pass: nothing was applied)
    } else {
      if (tid < Gn) {                                      // cur's features: as the service wave left them
        const double* cb = smem + L.cbuf + pb * L.cbuf_sz + 6 * tid;
        const int fq = gsl[tid];
        double* xf = xs + xZ + 5 * fq;
        xf[0] = cb[0]; xf[1] = cb[1]; xf[2] = cb[2]; xf[3] = cb[3]; xf[4] = cb[4];
        diag[fq] = cb[5];
      }
      Gp = Gn; p0 = c0; p1 = c1; p2 = c2;
    }
    // ================= the next group; the service wave reads its entries of P before the pass of the group just committed starts
    pb ^= 1;
    form_group();                                          // (its barriers also publish the commit above)
  }
  // ---- the launch's last words: state, rho-rho diagonal, flags
  for (int f = tid; f < len; f += T) P[(16 + 3 * f + 2) + (long)(16 + 3 * f + 2) * ld] = diag[f];
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[b], flag);
}

}  // namespace viekf
