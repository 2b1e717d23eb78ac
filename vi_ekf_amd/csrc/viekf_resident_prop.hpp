// viekf_resident_prop.hpp -- resident family: the propagate (dynamics hand-over, low-rank coupling set-up, body strips).
#pragma once
#include "viekf_resident_common.hpp"

namespace viekf {


// Rarely-executed, register-hungry pieces are kept out of line so that they do not inflate the
// register allocation of the sweep loops that hold P.
__device__ RES_INLINE void res_body_phase(const double* xs, const double* u, const DevParams* p, BodyCtx* ctx,
                                            double* xdb, double* Abb, double* Gb) {
  double ub[6];
  q_rota(p->q_b_u, u, ub);
  q_rota(p->q_b_u, u + 3, ub + 3);
  // (the context is built in registers and handed to LDS once: worked on in place, every store to A / G -- which may alias
  //  it as far as the compiler knows -- forces the fields to be re-read from LDS, 10 k clk of single-lane latency)
  BodyCtx c;
  body_ctx(xs, ub, *p, c);
  body_dynamics<false>(c, *p, xdb, Abb, Gb);   // the service wave cleared xdb / Abb / Gb cooperatively
  *ctx = c;
}

// one feature's share of the propagate set-up on the SERVICE wave: dynamics, Phi_ff, state step.  The Jacobian blocks are
// handed to the worker waves RAW, in the EVEN slots of the feature's own three Z rows (row r, slot 2c: c = 0..2 Afv[r][c],
// 3..5 Afg[r][c-3], 6..8 Aff[r][c-6]); a worker thread per row expands them into D and Gs (res_feature_expand_row: odd slots
// and 18..23, so nothing it reads is overwritten), off this wave's serial path.  The even slots receive Ut afterwards.
__device__ RES_INLINE void res_feature_phase(int f, int len, double dt, double* xs, const BodyCtx* ctx, double* Z,
                                             double* phiff) {
  double* z0 = Z + (3 * f) * ZS;
  if (f < len) {
    double xd3[3], Afv[9], Afg[9], Aff[9];
    const double qz[4] = {xs[xZ + 5 * f], xs[xZ + 5 * f + 1], xs[xZ + 5 * f + 2], xs[xZ + 5 * f + 3]};
    const double rho = xs[xZ + 5 * f + 4];
    const BodyCtx c = *ctx;   // (a register copy: the stores below may alias the LDS one as far as the compiler knows)
    feature_dynamics(qz, rho, c, xd3, Afv, Afg, Aff);
    double Aff2[9];
    mm<3, 3, 3>(Aff, Aff, Aff2);
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
      phiff[9 * f + e] = id + Aff[e] * dt + Aff2[e] * (0.5 * dt * dt);
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c3 = 0; c3 < 3; c3++) {
        z0[r * ZS + 2 * c3] = Afv[r * 3 + c3]; z0[r * ZS + 2 * (3 + c3)] = Afg[r * 3 + c3]; z0[r * ZS + 2 * (6 + c3)] = Aff[r * 3 + c3];
      }
    double qn[4];
    q_feat_boxplus_fast(qz, xd3[0] * dt, xd3[1] * dt, qn);
    double* xf = xs + xZ + 5 * f;
    xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
    xf[4] = rho + xd3[2] * dt;
  } else {  // inactive slot: Phi = I, G = 0
    for (int r = 0; r < 3; r++)
      for (int e = 0; e < 9; e++) z0[r * ZS + 2 * e] = 0.0;
    for (int e = 0; e < 9; e++) phiff[9 * f + e] = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
  }
}

// Worker side of the hand-over, one thread per row r of feature f: the raw blocks (see above) -> row r of
//   D = [M1 | M3 | M2]  in the odd slots 2k+1,  Gs = Gd_f sqrt(Qu)  in slots 18..23, with
//   Gd_f = dt ((dt/2) (Afv + dt/3 Aff Afv) G_b[vel rows] + (dt^2/6) Afv (A_v G_b) + [0 | (I + Aff dt/2 + Aff^2 dt^2/6) Afg])
//   (vi_ekf.cpp:302 restricted to the feature rows; G_b has no bias rows, vi_ekf_dyn.cpp:74-79)
// (row-wise, operands re-read from LDS: the whole-feature form held ~100 doubles live next to the thread's blocks of P)
__device__ RES_INLINE void res_feature_expand_row(int f, int r, double dt, double* Z, const double* Gb, const double* AvG,
                                                  const double* sqrtQu) {
  const double* z0 = Z + (3 * f) * ZS;
  double* zr = Z + (3 * f + r) * ZS;
  auto raw = [&](int row, int c) { return z0[row * ZS + 2 * c]; };   // c: 0..2 Afv, 3..5 Afg, 6..8 Aff
  const double a0 = raw(r, 6), a1 = raw(r, 7), a2 = raw(r, 8);       // Aff[r][:]
  double AAv[3], AAg[3], A2r[3], afv[3], afg[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    afv[j] = raw(r, j); afg[j] = raw(r, 3 + j);
    AAv[j] = a0 * raw(0, j) + a1 * raw(1, j) + a2 * raw(2, j);
    AAg[j] = a0 * raw(0, 3 + j) + a1 * raw(1, 3 + j) + a2 * raw(2, 3 + j);
    A2r[j] = a0 * raw(0, 6 + j) + a1 * raw(1, 6 + j) + a2 * raw(2, 6 + j);
  }
  const double ar[3] = {a0, a1, a2};
  double mff[3];
#pragma unroll
  for (int j = 0; j < 3; j++) mff[j] = ((j == r) ? 1.0 : 0.0) + ar[j] * (0.5 * dt) + A2r[j] * (dt * dt * (1.0 / 6.0));
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      a += (afv[j] + (dt * (1.0 / 3.0)) * AAv[j]) * Gb[(dxVEL + j) * 6 + k];
      b += afv[j] * AvG[j * 6 + k];
    }
    double g = a * (0.5 * dt) + b * (dt * dt * (1.0 / 6.0));
    if (k >= 3) g += mff[0] * raw(0, k) + mff[1] * raw(1, k) + mff[2] * raw(2, k);   // (Mff Afg)[r][k-3]
    zr[18 + k] = g * dt * sqrtQu[k];
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    zr[2 * j + 1] = (afv[j] + (0.5 * dt) * AAv[j]) * dt;          // M1
    zr[2 * (3 + j) + 1] = afv[j] * (0.5 * dt * dt);               // M3
    zr[2 * (6 + j) + 1] = (afg[j] + (0.5 * dt) * AAg[j]) * dt;    // M2
  }
}

// fix_depth of one feature after the propagate (vi_ekf_helper.cpp:128-156, called at vi_ekf.cpp:311): the state here, the
// covariance edit through the fix mailbox (applied by the worker that owns the feature's diagonal block)
__device__ RES_INLINE void res_fix_depth(double* xf, const DevParams* p, double* fixadd_slot, double* fixset_slot,
                                         double* fixany, unsigned* flag) {
  double rho = xf[4];
  const double reset = 1.0 / (2.0 * p->min_depth);
  if (rho != rho) { rho = reset; *flag |= FLAG_NAN; }
  if (rho < 0.0) {
    const double err = reset - rho;
    *fixadd_slot = err * err;
    *fixany = 1.0;
    rho = reset;
    *flag |= FLAG_NEGDEPTH;
  } else if (rho > 1e2) {
    *fixset_slot = 1.0;
    *fixany = 1.0;
    rho = reset;
  }
  xf[4] = rho;
}


// Propagate set-up on the worker waves (LDS only; see the header of this file for the algebra).  Three intervals:
//   [B1p..B2p]  Phi_bb / M_bb, the per-feature expansion raw blocks -> D, Gs, V = Phi_ff P[feat, body] (in place), Psi P_bb
//   [B2p..B2q]  Gs_b = M_bb G_b dt sqrt(Qu), T16 = Phi_bb P_bb, Pi = Psi P_bb Psi^T
//   [B2q..B3p]  Ut = D Pi / 2 + V Psi^T  (the even Z slots), Xi = Psi T16^T
// Barriers B1p, B2p, B2q inside; the caller continues with B3p.
template <int TW>
__device__ __forceinline__ void res_prop_setup(const StreamArgs& a, const ResShared& S, int tid) {
  const int nf = S.nf, N = S.N;
  const DevParams& prm = *a.dp;
  double* Pbc = S.Pbc; double* Pbb = S.Pbb;
  double* Z = S.Z; double* phiff = S.phiff;
  double* Phibb = S.Phibb; double* Mbb = S.Mbb; double* Gdb = S.Gdb; double* T16 = S.T16;
  __syncthreads();  // B1p : body Jacobian, raw feature blocks and Phi_ff ready (service), and this propagate's dt
  const double dt = S.sm[42];

  // ---- [B1p..B2p]
  for (int e = tid; e < nf; e += TW) res_feature_expand_row(e / 3, e % 3, dt, Z, S.Gb, S.AvG, prm.sqrtQu);
  for (int e = TW - 1 - tid; e < 256; e += TW) {   // body transition blocks (vi_ekf.cpp:302-303), from the last threads
    const int r = e >> 4, c = e & 15;
    double a2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; k++) a2 += S.Abb[r * 16 + k] * S.Abb[k * 16 + c];
    const double id = (r == c) ? 1.0 : 0.0, av = S.Abb[e];
    Mbb[e] = id + av * (0.5 * dt) + a2 * (dt * dt * (1.0 / 6.0));
    const double ph = id + av * dt + a2 * (0.5 * dt * dt);
    Phibb[e] = ph;
    S.PhibbT[c * 16 + r] = ph;   // transposed copy: lanes that differ in the OUTPUT column read consecutive words
  }
  for (int e = TW - 1 - tid; e < ZK * 16; e += TW) {   // Psi P_bb: rows E_v, A_bb[vel rows], E_g of P_bb
    const int q = e >> 4, c = e & 15;
    double v;
    if (q < 3) v = Pbb[(dxVEL + q) * 16 + c];
    else if (q >= 6) v = Pbb[(dxB_G + q - 6) * 16 + c];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += S.Abb[(dxVEL + q - 3) * 16 + k] * Pbb[k * 16 + c];
    }
    S.PsiP[e] = v;
  }
  for (int e = tid; e < 16 * N; e += TW) {   // V = Phi_ff[f] P[f, body], in place: item = (feature f, body column k)
    const int f = e >> 4, k = e & 15;
    const double* ff = phiff + 9 * f;
    double* pc = Pbc + (3 * f) * 16 + k;
    const double p0 = pc[0], p1 = pc[16], p2 = pc[32];
    pc[0] = ff[0] * p0 + ff[1] * p1 + ff[2] * p2;
    pc[16] = ff[3] * p0 + ff[4] * p1 + ff[5] * p2;
    pc[32] = ff[6] * p0 + ff[7] * p1 + ff[8] * p2;
  }
  __syncthreads();  // B2p

  // ---- [B2p..B2q]
  for (int e = tid; e < 256; e += TW) {
    const int r = e >> 4, c = e & 15;
    double sv = 0.0;
#pragma unroll 4
    for (int k = 0; k < 16; k++) sv += Phibb[r * 16 + k] * Pbb[k * 16 + c];
    T16[e] = sv;
  }
  for (int e = TW - 1 - tid; e < 96; e += TW) {
    const int r = e / 6, k = e % 6;
    double sv = 0.0;
#pragma unroll 4
    for (int c = 0; c < 16; c++) sv += Mbb[r * 16 + c] * S.Gb[c * 6 + k];
    Gdb[e] = sv * dt * prm.sqrtQu[k];
  }
  for (int e = tid; e < ZK * ZK; e += TW) {   // Pi = (Psi P_bb) Psi^T
    const int q = e / ZK, j = e - q * ZK;
    const double* pr = S.PsiP + q * 16;
    double v;
    if (j < 3) v = pr[dxVEL + j];
    else if (j >= 6) v = pr[dxB_G + j - 6];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += pr[k] * S.Abb[(dxVEL + j - 3) * 16 + k];
    }
    S.Pi[e] = v;
  }
  __syncthreads();  // B2q

  // ---- [B2q..B3p]
  // Ut = [D | V] [Pi / 2 ; Psi^T]  (nf x 25)(25 x 9) on the matrix cores, one 16-row tile per wave and turn, 7 k-steps (as
  // 1350 dot products this interval was LDS-bound: 50 operand reads per output).  A result lane holds column lr = lane & 15
  // of rows lk + 4 r (lk = lane >> 4) of the tile; columns 9..15 of the right-hand side are zero.
  {
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    double bv[7];
    int aoff[7];     // where k = 4 sk + lk sits: D[k] in the Z row (odd slots), V[k - 9] in the Pbc row, k >= 25: nowhere
#pragma unroll
    for (int sk = 0; sk < 7; sk++) {
      const int k = 4 * sk + lk, c = k - ZK;
      double v = 0.0;
      if (lr < ZK) {
        if (k < ZK) v = 0.5 * S.Pi[k * ZK + lr];
        else if (c < 16) v = (lr < 3) ? ((c == dxVEL + lr) ? 1.0 : 0.0)
                           : ((lr >= 6) ? ((c == dxB_G + lr - 6) ? 1.0 : 0.0) : S.Abb[(dxVEL + lr - 3) * 16 + c]);
      }
      bv[sk] = v;
      aoff[sk] = (k < ZK) ? (2 * k + 1) : ((c < 16) ? (0x100 | c) : -1);
    }
    for (int t = wv; t * 16 < nf; t += TW / 64) {
      const int ar = min(16 * t + lr, nf - 1);
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sk = 0; sk < 7; sk++) {
        const int o = aoff[sk];
        const double av = (o < 0) ? 0.0 : ((o & 0x100) ? Pbc[ar * 16 + (o & 0xff)] : Z[ar * ZS + o]);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[sk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int row = 16 * t + lk + 4 * r4;
        if (row < nf && lr < ZK) Z[row * ZS + 2 * lr] = acc[r4];
      }
    }
  }
  for (int e = TW - 1 - tid; e < ZK * 16; e += TW) {   // Xi = Psi T16^T  (T16^T = P_bb Phi_bb^T)
    const int q = e >> 4, c = e & 15;
    double v;
    if (q < 3) v = T16[c * 16 + dxVEL + q];
    else if (q >= 6) v = T16[c * 16 + dxB_G + q - 6];
    else {
      v = 0.0;
#pragma unroll 4
      for (int k = 0; k < 16; k++) v += S.Abb[(dxVEL + q - 3) * 16 + k] * T16[c * 16 + k];
    }
    S.Xi[e] = v;
  }
}

// P+[feature rows, body columns] = V Phi_bb^T + D Xi + Gs Gs_b^T, in LDS and in place (a tile's rows are read before they are
// written, by the same wave), and the body block (-> Mbb).  (nf x 16)(16 x 16) + (nf x 16)(16 x 16) on the matrix cores, one
// 16-row tile per wave and turn, 4 + 4 k-steps: the second product's k runs over D[0..8], Gs[0..5] and one zero.
template <int TW>
__device__ __forceinline__ void res_prop_body(const StreamArgs& a, const ResShared& S, int tid) {
  const int nf = S.nf;
  double* Pbc = S.Pbc;
  const double* Z = S.Z; double* Gdb = S.Gdb; double* T16 = S.T16;
  {
    const int lane = tid & 63, wv = tid >> 6, lr = lane & 15, lk = lane >> 4;
    constexpr int step = TW / 64;
    // the right-hand operands do not depend on the tile: loaded once; two tiles per turn, so that one tile's dependent
    // MFMA chain runs in the shadow of the other's
    double bph[4], bdx[4];
    int zoff[4];   // slot of k = 4 sk + lk in a Z row: D[k] at 2k+1, Gs[k-9] at 18 + k - 9, k = 15: none
#pragma unroll
    for (int sk = 0; sk < 4; sk++) {
      const int k = 4 * sk + lk;
      bph[sk] = S.PhibbT[k * 16 + lr];
      bdx[sk] = (k < ZK) ? S.Xi[k * 16 + lr] : ((k < ZK + 6) ? Gdb[lr * 6 + (k - ZK)] : 0.0);
      zoff[sk] = (k < ZK) ? (2 * k + 1) : ((k < ZK + 6) ? (18 + k - ZK) : -1);
    }
    for (int t = wv; t * 16 < nf; t += 2 * step) {
      const int t1 = t + step;
      const int r0 = min(16 * t + lr, nf - 1), r1 = min(16 * t1 + lr, nf - 1);
      double a0[8], a1[8];
#pragma unroll
      for (int sk = 0; sk < 4; sk++) {
        a0[sk] = Pbc[r0 * 16 + 4 * sk + lk]; a1[sk] = Pbc[r1 * 16 + 4 * sk + lk];
        a0[4 + sk] = (zoff[sk] >= 0) ? Z[r0 * ZS + zoff[sk]] : 0.0;
        a1[4 + sk] = (zoff[sk] >= 0) ? Z[r1 * ZS + zoff[sk]] : 0.0;
      }
      v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sk = 0; sk < 8; sk++) {
        const double bv = sk < 4 ? bph[sk] : bdx[sk - 4];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[sk], bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[sk], bv, acc1, 0, 0, 0);
      }
      // (every lane of the wave has its operands before any lane stores: the stores depend on the MFMA results)
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int row0 = 16 * t + lk + 4 * r4, row1 = 16 * t1 + lk + 4 * r4;
        if (row0 < nf) Pbc[row0 * 16 + lr] = acc0[r4];
        if (row1 < nf) Pbc[row1 * 16 + lr] = acc1[r4];
      }
    }
  }
  // body block  P_bb+ = T16 Phi_bb^T + Gs_b Gs_b^T + Qx : one 16 x 16 tile, 4 + 2 k-steps, on the first wave
  if (tid < 64) {
    const int lr = tid & 15, lk = tid >> 4;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int sk = 0; sk < 4; sk++) {
      const int k = 4 * sk + lk;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T16[lr * 16 + k], S.PhibbT[k * 16 + lr], acc, 0, 0, 0);   // rows r = lr | cols c = lr
    }
#pragma unroll
    for (int sk = 0; sk < 2; sk++) {
      const int k = 4 * sk + lk;                                   // 0..7, the input-noise columns are k < 6
      const double gv = (k < 6) ? Gdb[lr * 6 + k] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gv, gv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      const int r = lk + 4 * r4, c = lr;                           // a result lane holds column lr of rows lk + 4 r4
      S.Mbb[r * 16 + c] = acc[r4] + ((r == c) ? a.Qx[r] : 0.0);    // P_bb+ staged in Mbb (T16 / Pbb are still being read)
    }
  }
}

}  // namespace viekf
