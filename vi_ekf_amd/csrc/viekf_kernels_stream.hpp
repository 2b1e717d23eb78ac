// viekf_kernels_stream.hpp -- "streaming" kernel family: one workgroup per filter, covariance stays
// in HBM/L2 (column-major, leading dimension ld) and is updated in place.  Works for any number of
// features; it is the path for wide P (N=150) and the correctness baseline for the resident family.
//
// Structure exploited (exact in real arithmetic):
//   A = [[A_bb, 0], [A_fb, blockdiag(A_ff)]]   (vi_ekf_dyn.cpp:55-71,121-128: body rows never depend on features)
//   => Phi = I + A dt + A^2 dt^2/2 has the same shape, so  P+ = Phi P Phi^T + Gd Qu Gd^T + Qx  (vi_ekf.cpp:302-304)
//      needs per 3x3 block only its own block plus the 16 body rows/columns.
//   H of a FEAT measurement has one 2x2 block (vi_ekf_meas.cpp:366) => W = P H^T is two columns of P and
//      (I-KH)P(I-KH)^T + KRK^T - P = -K W^T, so the partial update (vi_ekf_meas.cpp:254-257) is
//      P_ij -= Lambda_ij (K_i . W_j),  Lambda_ij = l_i + l_j - l_i l_j  (vi_ekf.cpp:83,146).
#pragma once
#include "viekf_device.hpp"
#include "viekf_fastmath.hpp"

namespace viekf {

struct StreamArgs {
  double* x;            // [B][nxs]
  double* P;            // [B][n][ld]
  int* len;             // [B]
  unsigned* flags;      // [B]
  const double* Qx;     // [n] diagonal
  const double* lambda; // [n]
  double* ws;           // [B][ws_stride]
  int B, N, nx, nxs, n, ld;
  long ws_stride;
  const DevParams* dp;  // device memory (uniform loads); NOT by value: indexing a by-value kernarg array spills it to scratch
  double* x_out;        // where the fused-step kernel stores the state / covariance: the same buffers (in place) or another
  double* P_out;        // slot of the history ring (viekf_batch_propagate_to: the propagate writes the NEXT slot, no copy)
  const unsigned char* active;   // [B] or NULL: filters with active[b] == 0 take no part in a propagate / feature-update launch
                                 // (viekf_batch_set_active: filters on different clocks share a batch)
  const int* resmap;    // fused-step kernel: ownership map of the 3x3 feature blocks, [RB][TW] entries I | J << 8 | owned << 16
  // Per-filter live ring slots (viekf_batch_select_filters: filters on clocks of their own share a batch, each with its own ring
  // position): x / P then point at the RING, [slot][B] entries, and filter b's state is entry smap[b] = slot_b * B + b of it.
  // smap_out: where the fused kernel stores filter b's state (viekf_batch_propagate_filters_to: slot i_b -> slot i_b + 1, no copy).
  int* smap;            // (device memory; the fused kernel moves a filter's entry to smap_out[b] when it stores it there)
  const int* smap_out;
  __device__ __forceinline__ long si(int b) const { return smap ? (long)smap[b] : (long)b; }
  __device__ __forceinline__ long so(int b) const { return smap_out ? (long)smap_out[b] : si(b); }
};

constexpr int WK = 40;   // contraction depth of the low-rank part (16 + 16 + 6, padded to whole MFMA k-steps of 4)
typedef double v4f64 __attribute__((ext_vector_type(4)));

// workspace carve-up (doubles) for one filter
struct WsLayout {
  long phi_fb, phi_ff, gd, U, pbr, pbc, Xw, Yw, total;
  __host__ __device__ WsLayout(int N, int n) {
    long o = 0;
    phi_fb = o; o += 3L * N * 16;   // [3N][16]
    phi_ff = o; o += 9L * N;        // [N][9]
    gd = o;     o += 6L * n;        // [n][6]
    U = o;      o += 3L * N * 16;   // [3N][16]   (Phi P)[feat rows, body cols]
    pbr = o;    o += 16L * n;       // [16][n]    copy of P[body rows, :]
    pbc = o;    o += 16L * n;       // [16][n]    copy of P[:, body cols], stored [k][i]
    const long nfp = ((3L * N + 47) / 48) * 48;
    Xw = o;     o += nfp * WK;      // [nfp][WK]  low-rank factors of the propagate's MFMA pass (k_propagate_stream<.., true>):
    Yw = o;     o += nfp * WK;      //            P+_ff = D P_ff D^T + Xw Yw^T, rows padded to whole 48-row super-tiles
    total = (o + 1) & ~1L;
  }
};

constexpr unsigned FLAG_NAN = 1u, FLAG_BLOWUP = 2u, FLAG_NEGDEPTH = 4u, FLAG_INTERNAL = 8u;

// fix_depth (vi_ekf_helper.cpp:128-156) for feature i of the block's filter; xs is the LDS copy of x
__device__ __forceinline__ void fix_depth_one(double* xs, double* P, int ld, int i, const DevParams& p, unsigned& flag) {
  const int xR = xZ + 5 * i + 4, dR = dxZ + 3 * i + 2;
  double rho = xs[xR];
  const double reset = 1.0 / (2.0 * p.min_depth);
  if (rho != rho) { rho = reset; flag |= FLAG_NAN; }
  if (rho < 0.0) {
    const double err = reset - rho;
    P[dR + (long)dR * ld] += err * err;
    rho = reset;
    flag |= FLAG_NEGDEPTH;
  } else if (rho > 1e2) {
    P[dR + (long)dR * ld] = p.P0_feat[2];
    rho = reset;
  }
  xs[xR] = rho;
}

// ------------------------------------------------------------------------------------------------
// propagate: numeric core of VIEKF::propagate_state (vi_ekf.cpp:262-318)
// ------------------------------------------------------------------------------------------------
// LDS written by some lanes of a wave and read by others of the SAME wave: order the accesses (the compiler treats lanes as
// independent threads and may otherwise move the loads above the stores)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// MF = true: the feature/feature part runs on the fp64 matrix cores (see the pass at the end); otherwise one 3x3 block per thread.
template <int T, bool MF>
__global__ __launch_bounds__(T) void k_propagate_stream(StreamArgs a, const double* __restrict__ u_all,
                                                        const double* __restrict__ dt_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (a.active && !a.active[b]) return;
  const int n = a.n, ld = a.ld;
  double* xs = smem;                    // [nxs]
  double* Abb = xs + a.nxs;             // 16x16 row-major
  double* Gb = Abb + 256;               // 16x6
  double* Phibb = Gb + 96;              // 16x16
  double* Mbb = Phibb + 256;            // 16x16
  double* Gdb = Mbb + 256;              // 16x6
  double* Pbb = Gdb + 96;               // 16x16 row-major copy of P_bb
  double* T16 = Pbb + 256;              // 16x16 scratch
  double* xdb = T16 + 256;              // 16
  BodyCtx* ctx = reinterpret_cast<BodyCtx*>(xdb + 16);
  double* Dl = xdb + 16 + (sizeof(BodyCtx) + 7) / 8;   // [N][9] Phi_ff blocks (MF only)

  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nf = 3 * len, nact = 16 + nf;
  const double dt = dt_all[b];
  const WsLayout L(a.N, n);
  double* ws = a.ws + (long)b * a.ws_stride;
  double* phi_fb = ws + L.phi_fb;
  double* phi_ff = ws + L.phi_ff;
  double* gd = ws + L.gd;
  double* U = ws + L.U;
  double* pbr = ws + L.pbr;
  double* pbc = ws + L.pbc;

  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  __syncthreads();

  if (tid == 0) {
    double ub[6];
    q_rota(a.dp->q_b_u, u_all + (long)b * 6, ub);          // vi_ekf.cpp:265-267
    q_rota(a.dp->q_b_u, u_all + (long)b * 6 + 3, ub + 3);
    body_ctx(xs, ub, (*a.dp), *ctx);
    body_dynamics(*ctx, (*a.dp), xdb, Abb, Gb);               // vi_ekf_dyn.cpp:42-80
  }
  __syncthreads();

  // ---- body transition blocks (vi_ekf.cpp:302-303 restricted to the 16x16 block)
  for (int e = tid; e < 256; e += T) {
    const int r = e >> 4, c = e & 15;
    double a2 = 0.0;
    for (int k = 0; k < 16; k++) a2 += Abb[r * 16 + k] * Abb[k * 16 + c];
    const double id = (r == c) ? 1.0 : 0.0, av = Abb[e];
    Mbb[e] = id + av * dt / 2.0 + a2 * dt * dt / 6.0;
    Phibb[e] = id + av * dt + a2 * dt * dt / 2.0;
  }
  __syncthreads();
  for (int e = tid; e < 96; e += T) {
    const int r = e / 6, k = e % 6;
    double s = 0.0;
    for (int c = 0; c < 16; c++) s += Mbb[r * 16 + c] * Gb[c * 6 + k];
    s *= dt;
    Gdb[e] = s;
    gd[r * 6 + k] = s;
  }

  // ---- per feature: dynamics, Phi_fb / Phi_ff / Gd_f, state step
  for (int i = tid; i < len; i += T) {
    double xd3[3], Afv[9], Afg[9], Aff[9];
    const double* qz = xs + xZ + 5 * i;
    const double rho = qz[4];
    feature_dynamics(qz, rho, *ctx, xd3, Afv, Afg, Aff);   // vi_ekf_dyn.cpp:96-134
    double Aff2[9];
    mm<3, 3, 3>(Aff, Aff, Aff2);
    double Mff[9];
    for (int e = 0; e < 9; e++) {
      const double id = (e == 0 || e == 4 || e == 8) ? 1.0 : 0.0;
      Mff[e] = id + Aff[e] * dt / 2.0 + Aff2[e] * dt * dt / 6.0;
      const double ph = id + Aff[e] * dt + Aff2[e] * dt * dt / 2.0;
      phi_ff[9 * i + e] = ph;
      if (MF) Dl[9 * i + e] = ph;
    }
    double gacc[18];
    for (int e = 0; e < 18; e++) gacc[e] = 0.0;
#pragma unroll
    for (int c = 0; c < 16; c++) {
#pragma unroll
      for (int r = 0; r < 3; r++) {
        // A_fb(r,c): non-zero only in the VEL and B_G columns
        double afb = 0.0;
        if (c >= dxVEL && c < dxVEL + 3) afb = Afv[r * 3 + (c - dxVEL)];
        else if (c >= dxB_G && c < dxB_G + 3) afb = Afg[r * 3 + (c - dxB_G)];
        // (A^2)_fb = A_fb A_bb + A_ff A_fb
        double a2 = 0.0;
        for (int k = 0; k < 3; k++) a2 += Afv[r * 3 + k] * Abb[(dxVEL + k) * 16 + c] + Afg[r * 3 + k] * Abb[(dxB_G + k) * 16 + c];
        if (c >= dxVEL && c < dxVEL + 3)
          for (int k = 0; k < 3; k++) a2 += Aff[r * 3 + k] * Afv[k * 3 + (c - dxVEL)];
        else if (c >= dxB_G && c < dxB_G + 3)
          for (int k = 0; k < 3; k++) a2 += Aff[r * 3 + k] * Afg[k * 3 + (c - dxB_G)];
        phi_fb[(3 * i + r) * 16 + c] = afb * dt + a2 * dt * dt / 2.0;
        const double mfb = afb * dt / 2.0 + a2 * dt * dt / 6.0;
        for (int k = 0; k < 6; k++) gacc[r * 6 + k] += mfb * Gb[c * 6 + k];
      }
    }
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) {   // G_f = [0 | Afg]  (vi_ekf_dyn.cpp:131-132)
        double s = 0.0;
        for (int m = 0; m < 3; m++) s += Mff[r * 3 + m] * Afg[m * 3 + k];
        gacc[r * 6 + 3 + k] += s;
      }
    for (int e = 0; e < 18; e++) gd[(16 + 3 * i) * 6 + e] = gacc[e] * dt;
    // state step for this feature (boxplus, vi_ekf_helper.cpp:93-97)
    double qn[4];
    q_feat_boxplus(qz, xd3[0] * dt, xd3[1] * dt, qn);
    xs[xZ + 5 * i + 0] = qn[0]; xs[xZ + 5 * i + 1] = qn[1]; xs[xZ + 5 * i + 2] = qn[2]; xs[xZ + 5 * i + 3] = qn[3];
    xs[xZ + 5 * i + 4] = rho + xd3[2] * dt;
  }
  // ---- save the body rows / columns of P before anything is overwritten
  // (the matrix-core variant reads nothing above the diagonal outside the diagonal 48 x 48 super-tiles: after a grouped update
  //  the rest of the upper triangle may be stale, see k_update_feat_blocked; P is symmetric, so the body rows are the body columns)
  for (int e = tid; e < 16 * nact; e += T) {
    const int k = e / nact, j = e % nact;
    const double pc = P[j + (long)k * ld];
    pbr[k * n + j] = MF ? pc : P[k + (long)j * ld];
    pbc[k * n + j] = pc;
  }
  for (int e = tid; e < 256; e += T) Pbb[e] = MF ? P[max(e >> 4, e & 15) + (long)min(e >> 4, e & 15) * ld] : P[(e >> 4) + (long)(e & 15) * ld];
  __syncthreads();
  if (tid == 0) {  // body state step (after every feature thread has read the old body state through ctx)
    double dxb[16], xo[17];
    for (int i = 0; i < 16; i++) dxb[i] = xdb[i] * dt;
    body_boxplus(xs, dxb, xo);
    for (int i = 0; i < 17; i++) xs[i] = xo[i];
  }

  // ---- U = (Phi P)[feat, body]  (its mirror (P Phi^T)[body, feat] is not formed: P+ is written symmetric),  T16 = Phi_bb P_bb
  for (int e = tid; e < nf * 16; e += T) {
    const int r = e >> 4, k = e & 15, I = r / 3, rr = r - 3 * I;
    double s = 0.0;
    for (int c = 0; c < 16; c++) s += phi_fb[r * 16 + c] * Pbb[c * 16 + k];
    double vt = 0.0;
    for (int m = 0; m < 3; m++) {
      const double pf = phi_ff[9 * I + rr * 3 + m];
      s += pf * pbc[k * n + 16 + 3 * I + m];
      vt += pbr[k * n + 16 + 3 * I + m] * pf;
    }
    U[r * 16 + k] = s;
    if (MF) {
      // rows of the low-rank factors:  P+_ff - D P_ff D^T = U Phi_fb^T + Phi_fb (P_bf D^T) + (Gd Qu) Gd^T = Xw Yw^T
      double* xr = ws + L.Xw + (long)r * WK;
      double* yr = ws + L.Yw + (long)r * WK;
      const double ph = phi_fb[r * 16 + k];
      xr[k] = s;       yr[k] = ph;
      xr[16 + k] = ph; yr[16 + k] = vt;
      if (k < 8) {
        const double g = (k < 6) ? gd[(16 + r) * 6 + k] : 0.0;
        xr[32 + k] = (k < 6) ? g * a.dp->Qu[k] : 0.0;
        yr[32 + k] = g;
      }
    }
  }
  for (int e = tid; e < 256; e += T) {
    const int r = e >> 4, c = e & 15;
    double s = 0.0;
    for (int k = 0; k < 16; k++) s += Phibb[r * 16 + k] * Pbb[k * 16 + c];
    T16[e] = s;
  }
  __syncthreads();

  // ---- write P+ : body block
  // P+ is written EXACTLY symmetric (every pair gets one value): the update kernels' rank-2 form equals the reference's
  // Joseph form only for symmetric P and amplifies an antisymmetric part instead of damping it (see the fused kernel's
  // sym_diag), so rounding-level asymmetry must not be allowed to build up over a flight.
  for (int e = tid; e < 256; e += T) {
    const int r = min(e >> 4, e & 15), c = max(e >> 4, e & 15);   // the (lower index, higher index) expression for both copies
    double s = 0.0;
    for (int k = 0; k < 16; k++) s += T16[r * 16 + k] * Phibb[c * 16 + k];
    double g = 0.0;
    for (int k = 0; k < 6; k++) g += Gdb[r * 6 + k] * a.dp->Qu[k] * Gdb[c * 6 + k];
    s = s + g;
    if (r == c) s += a.Qx[r];
    P[(e >> 4) + (long)(e & 15) * ld] = s;
  }
  // ---- body/feature cross blocks (the feature-row copy, mirrored)
  for (int e = tid; e < nf * 16; e += T) {
    const int r = e >> 4, k = e & 15;
    double s = 0.0;
    for (int c = 0; c < 16; c++) s += U[r * 16 + c] * Phibb[k * 16 + c];
    double g = 0.0;
    for (int q = 0; q < 6; q++) g += gd[(16 + r) * 6 + q] * a.dp->Qu[q] * Gdb[k * 6 + q];
    P[(16 + r) + (long)k * ld] = s + g;
    if (!MF) P[k + (long)(16 + r) * ld] = s + g;   // (the matrix-core variant keeps the lower triangle only, see below)
  }
  if constexpr (MF) {
    // ---- feature/feature part on the matrix cores.  With D = blockdiag(Phi_ff) the new block is
    //        P+_ff = D P_ff D^T + Xw Yw^T          (Xw, Yw: nf x 38, rows written above)
    //      One wave per 48 x 48 super-tile (16 features square, so D never straddles a tile), v_mfma_f64_16x16x4_f64:
    //        R   = P_IJ D_J^T              A = P (lane & 15 along the rows of P: coalesced loads), B = D_J^T built from LDS
    //        O^T = R^T D_I^T               A = R straight from the accumulators: register s of a 16x16 result IS the operand of
    //                                      k-step s (row (lane>>4) + 4 s), so the chain needs no data movement
    //        O^T += Yw_J Xw_I^T            10 k-steps
    //      O^T has lane & 15 along the rows of P again: coalesced stores.  D is block-diagonal, so only the k-steps that
    //      overlap a tile's features are issued (16 of 36 per product).  In place: a super-tile depends on itself only.
    __syncthreads();   // Xw / Yw rows and Dl are complete (this block's global writes are visible to it after the barrier)
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    constexpr int NWV = T / 64;
    const int nst = (nf + 47) / 48;
    const double* Xw = ws + L.Xw;
    const double* Yw = ws + L.Yw;
    // operand element D[16 q + lr][4 s + lk] of the 48 x 48 block-diagonal matrix of super-tile `sup`
    auto dop = [&](int sup, int q, int s) -> double {
      const int jp = 16 * q + lr, j = 4 * s + lk;
      const int fp = jp / 3, f = j / 3, F = 16 * sup + fp;
      return (fp == f && F < len) ? Dl[9 * F + (jp - 3 * fp) * 3 + (j - 3 * f)] : 0.0;
    };
    // Only the super-tiles on and below the diagonal are computed and only the elements i >= j stored: the hot kernels (this one,
    // the grouped update, the fused step) read the LOWER triangle of P only, so nothing above the diagonal is written here --
    // r02's transpose pass cost one more read and one more write of half the matrix per step.  The host marks the batch
    // (upper_stale) and mirrors the triangle up (k_mirror_upper) before anything reads P whole.
    for (int st = wave; st < nst * nst; st += NWV) {
      const int I = st % nst, J = st / nst;
      if (I < J) continue;
      const int r0 = 16 + 48 * I, c0 = 16 + 48 * J;
      double pA[3][12];
#pragma unroll
      for (int aa = 0; aa < 3; aa++)
#pragma unroll
        for (int s = 0; s < 12; s++)
        {   // (a diagonal super-tile: the elements above the diagonal through their mirrors -- only the lower triangle is valid)
          const int pi = min(r0 + 16 * aa + lr, nact - 1), pj = min(c0 + 4 * s + lk, nact - 1);
          pA[aa][s] = P[max(pi, pj) + (long)min(pi, pj) * ld];
        }
      v4f64 R[3][3], O[3][3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int aa = 0; aa < 3; aa++) { R[aa][q] = v4f64{0.0, 0.0, 0.0, 0.0}; O[q][aa] = v4f64{0.0, 0.0, 0.0, 0.0}; }
      }
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
      for (int s = 0; s < WK / 4; s++) {
        double yv[3], xv[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          yv[q] = Yw[(long)(48 * J + 16 * q + lr) * WK + 4 * s + lk];
          xv[q] = Xw[(long)(48 * I + 16 * q + lr) * WK + 4 * s + lk];
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
  } else {
    // ---- feature/feature 3x3 blocks, in place (each block needs only itself + saved body strips)
    for (int e = tid; e < len * len; e += T) {
      const int I = e % len, J = e / len;
      if (I < J) continue;                       // (the block below the diagonal is computed, its mirror written with it)
      const int r0 = 16 + 3 * I, c0 = 16 + 3 * J;
      double Pij[9], Mij[9], out[9];
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) Pij[r * 3 + c] = P[(r0 + r) + (long)(c0 + c) * ld];
      for (int r = 0; r < 3; r++)
        for (int m = 0; m < 3; m++) {
          double s = 0.0;
          for (int k = 0; k < 16; k++) s += phi_fb[(3 * I + r) * 16 + k] * pbr[k * n + c0 + m];
          for (int q = 0; q < 3; q++) s += phi_ff[9 * I + r * 3 + q] * Pij[q * 3 + m];
          Mij[r * 3 + m] = s;
        }
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          double s = 0.0;
          for (int k = 0; k < 16; k++) s += U[(3 * I + r) * 16 + k] * phi_fb[(3 * J + c) * 16 + k];
          for (int m = 0; m < 3; m++) s += Mij[r * 3 + m] * phi_ff[9 * J + c * 3 + m];
          double g = 0.0;
          for (int q = 0; q < 6; q++) g += gd[(r0 + r) * 6 + q] * a.dp->Qu[q] * gd[(c0 + c) * 6 + q];
          s = s + g;
          if (I == J && r == c) s += a.Qx[r0 + r];
          out[r * 3 + c] = s;
        }
      if (I == J) { out[1] = out[3]; out[2] = out[6]; out[5] = out[7]; }   // diagonal block: upper <- lower
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) {
          P[(r0 + r) + (long)(c0 + c) * ld] = out[r * 3 + c];
          if (I != J) P[(c0 + c) + (long)(r0 + r) * ld] = out[r * 3 + c];
        }
    }
  }
  // inactive slots: Phi = I and G = 0 there, so only Qx is added (vi_ekf.cpp:139-144,304)
  for (int d = nact + tid; d < n; d += T) P[d + (long)d * ld] += a.Qx[d];
  __syncthreads();

  // ---- fix_depth (vi_ekf.cpp:311) and write the state back
  unsigned flag = 0;
  for (int i = tid; i < len; i += T) fix_depth_one(xs, P, ld, i, (*a.dp), flag);
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
// M sequential active FEAT updates: VIEKF::update + h_feat (vi_ekf_meas.cpp:196-278, 354-367)
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_update_feat_stream(StreamArgs a, const double* __restrict__ z_all,
                                                          const int* __restrict__ slot_all, int M,
                                                          const double* __restrict__ R_all, long r_stride_b,
                                                          long r_stride_m, int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (a.active && !a.active[b]) return;
  const int n = a.n, ld = a.ld;
  double* xs = smem;           // [nxs]
  double* W = xs + a.nxs;      // [n][2]
  double* K = W + 2 * n;       // [n][2]
  double* lam = K + 2 * n;     // [n]
  double* sm = lam + n;        // small scratch: zhat(2) Hb(4) res(2) Sinv(4) = 12, dxb(16)
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nact = 16 + 3 * len;
  unsigned flag = 0;

  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  for (int i = tid; i < n; i += T) lam[i] = a.lambda[i];
  __syncthreads();

  for (int m = 0; m < M; m++) {
    const int slot = slot_all[(long)b * M + m];
    int* res = result_all ? &result_all[(long)b * M + m] : nullptr;
    if (slot < 0) { if (res && tid == 0) *res = -1; continue; }
    if (slot >= len) { if (res && tid == 0) *res = 3; continue; }  // MEAS_INVALID
    const double* z = z_all + ((long)b * M + m) * 2;
    const double* R = R_all + (long)b * r_stride_b + (long)m * r_stride_m;  // column-major 2x2
    const double z0 = z[0], z1 = z[1];
    if (z0 != z0 || z1 != z1) { if (res && tid == 0) *res = 2; continue; }  // MEAS_NAN (vi_ekf_meas.cpp:136-137)
    const int j0 = 16 + 3 * slot;
    if (tid == 0) {
      double zhat[2], Hb[4];
      h_feat(xs + xZ + 5 * slot, (*a.dp), zhat, Hb);
      sm[0] = zhat[0]; sm[1] = zhat[1];
      sm[2] = Hb[0]; sm[3] = Hb[1]; sm[4] = Hb[2]; sm[5] = Hb[3];
      sm[6] = z0 - zhat[0]; sm[7] = z1 - zhat[1];      // residual (vi_ekf_meas.cpp:220)
    }
    __syncthreads();
    const double h00 = sm[2], h01 = sm[3], h10 = sm[4], h11 = sm[5];
    // W = P H^T : two columns of P (coalesced along i)
    for (int i = tid; i < nact; i += T) {
      const double p0 = P[i + (long)j0 * ld], p1 = P[i + (long)(j0 + 1) * ld];
      W[2 * i + 0] = p0 * h00 + p1 * h01;
      W[2 * i + 1] = p0 * h10 + p1 * h11;
    }
    __syncthreads();
    // S = H W[j0:j0+2] + R ; every lane computes it (uniform)
    double S[4], Si[4];
    {
      const double w00 = W[2 * j0 + 0], w01 = W[2 * j0 + 1], w10 = W[2 * (j0 + 1) + 0], w11 = W[2 * (j0 + 1) + 1];
      S[0] = h00 * w00 + h01 * w10 + R[0];
      S[1] = h00 * w01 + h01 * w11 + R[2];
      S[2] = h10 * w00 + h11 * w10 + R[1];
      S[3] = h10 * w01 + h11 * w11 + R[3];
    }
    inv2(S, Si);
    const double r0 = sm[6], r1 = sm[7];
    const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;  // vi_ekf_meas.cpp:234
    if (mahal > 9.0) {                                   // gate (:235-239): returns before fix_depth
      if (res && tid == 0) *res = 1;
      __syncthreads();
      continue;
    }
    // K = W S^-1 (:241) and NaN guard (:247)
    int bad = 0;
    for (int i = tid; i < nact; i += T) {
      const double w0 = W[2 * i], w1 = W[2 * i + 1];
      const double k0 = w0 * Si[0] + w1 * Si[2], k1 = w0 * Si[1] + w1 * Si[3];
      K[2 * i] = k0; K[2 * i + 1] = k1;
      if (k0 != k0 || k1 != k1) bad = 1;
    }
    if (h00 != h00 || h01 != h01 || h10 != h10 || h11 != h11) bad = 1;
    bad = __syncthreads_or(bad);
    if (!bad) {
      const bool partial = a.dp->use_partial_update != 0;
      // state correction  x <- x [+] (lambda o K r)   (:254-255 / :262-263)
      if (tid == 0) {
        double dxb[16], xo[17];
        for (int i = 0; i < 16; i++) {
          const double l = partial ? lam[i] : 1.0;
          dxb[i] = (l * K[2 * i]) * r0 + (l * K[2 * i + 1]) * r1;
        }
        body_boxplus(xs, dxb, xo);
        for (int i = 0; i < 17; i++) xs[i] = xo[i];
      }
      for (int f = tid; f < len; f += T) {
        const int d = 16 + 3 * f;
        double dv[3];
        for (int q = 0; q < 3; q++) {
          const double l = partial ? lam[d + q] : 1.0;
          dv[q] = (l * K[2 * (d + q)]) * r0 + (l * K[2 * (d + q) + 1]) * r1;
        }
        double qn[4];
        q_feat_boxplus(xs + xZ + 5 * f, dv[0], dv[1], qn);
        double* xf = xs + xZ + 5 * f;
        xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
        xf[4] += dv[2];
      }
      // covariance sweep  P_ij -= Lambda_ij (K_i . W_j)   (:256-257, or :264-265 with Lambda = 1)
      const long tot = (long)nact * nact;
      for (long e = tid; e < tot; e += T) {
        const int i = (int)(e % nact), j = (int)(e / nact);
        // (i, j) and (j, i) both form K_hi . W_lo from their own, equal, copies: P stays EXACTLY symmetric (this rank-2 form
        //  equals the reference's Joseph form only for symmetric P, and it amplifies an antisymmetric part -- see the fused
        //  kernel's sym_diag)
        const int hi = max(i, j), lo = min(i, j);
        const double t = K[2 * hi] * W[2 * lo] + K[2 * hi + 1] * W[2 * lo + 1];
        const double li = lam[i], lj = lam[j];
        const double L = partial ? (lj + li - li * lj) : 1.0;
        P[i + (long)j * ld] -= L * t;
      }
    }
    __syncthreads();
    for (int f = tid; f < len; f += T) fix_depth_one(xs, P, ld, f, (*a.dp), flag);   // :271
    if (res && tid == 0) *res = 0;
    __syncthreads();
  }
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[b], flag);
}

// ------------------------------------------------------------------------------------------------
// M sequential active FEAT updates, BLOCKED for a covariance that does not fit on chip (wide P): the same arithmetic
// as k_update_feat_stream, but P crosses HBM once per GROUP of up to BG measurements instead of once per measurement.
//   1. panel:  the 2 zeta columns of every feature measured in the group, all rows, -> LDS  (n x 2 BG)
//   2. for each measurement of the group, in order: innovation, gate, W = panel_g Hb^T (in place), K = W S^-1 (in
//      registers; only S^-1 is kept), state correction, fix_depth; the rank-2 update is applied to the LATER panel columns
//      only (they are all that the following gains need) and to an exact running copy of the rho-rho diagonal (fix_depth
//      edits it, vi_ekf_helper.cpp:128-156, and a "set" does not commute with later updates).
//   3. one pass over P:  P -= Lambda o (K W^T)  with W = n x 2 BG from LDS and K = W S^-1 re-formed on the fly -- a dense
//      contraction, done per 16x16 tile by v_mfma_f64_16x16x4_f64 (BG/2 k-steps), Lambda applied to the accumulator; the
//      rho-rho diagonal takes the running copy.
// A slot that repeats inside a group starts a new group.  HBM traffic per step drops from (M+1) to (M/BG+1) passes over P;
// keeping W alone (K costs two multiply-adds per operand instead of 127 KB of LDS) is what lets BG = 16 fit at N = 160.
// LDS rows of W are 34 doubles apart: the MFMA operand reads (16 consecutive rows per k) then spread over the banks.
// ------------------------------------------------------------------------------------------------
// One pass over P:  P -= Lambda o (K W^T), 16 x 16 tiles on the fp64 matrix cores.
//      D[r][c] = sum_k W[j0+r][k] K[i0+c][k]:  D's column index (lane & 15) runs along the ROWS of P (contiguous in memory).
// (shared by the two grouped update kernels: k_update_feat_blocked below and k_update_feat_panelsvc in viekf_kernels_wide.hpp)
// ticket != NULL: the units are drawn from a counter in LDS instead of being dealt round-robin (waves that arrive late -- they had
// other work first -- then simply take fewer)
template <int T, int BLD>
__device__ __forceinline__ void blk_pass(double* __restrict__ P, int ld, int nact, int Gn, const double* Wp, const double* SiL,
                                         const double* lam, const double* diag, bool partial, int lane, int wave, int* ticket = nullptr) {
  constexpr int NWV = T / 64;
    const int nt = (nact + 15) >> 4;
    const int lr = lane & 15, lk = lane >> 4;
    const int ksteps = (2 * Gn + 3) >> 2;               // (columns past 2 Gn are zero)
    constexpr int TPI = 8;   // tiles per wave and unit: their 32 loads of P are in flight together (one tile at a time
                             // leaves 2 KB per wave on the wire and the pass latency-bound at a fraction of the HBM rate)
    // P stays EXACTLY symmetric and is read only once per pair: the tiles on and below the diagonal are processed; a diagonal
    // tile forms K_i . W_j for i >= j and the mirror expression K_j . W_i (same products, same order) for i < j.
    // Work unit of a wave: TPI vertically adjacent tiles of one column block (rows 16 ti0 .. +127): its loads cover 1024
    // contiguous bytes per column (HBM likes long runs: 128-byte runs scattered over the matrix reached 2.8 TB/s).  Units
    // are numbered column block by column block over the lower triangle and dealt to the waves round-robin.
    struct Unit { int tj, ti0; };
    auto next_unit = [&](Unit u, int steps) {            // advance `steps` units (past the end: tj == nt)
      for (int s2 = 0; s2 < steps && u.tj < nt; s2++) {
        u.ti0 += TPI;
        if (u.ti0 >= nt) { u.tj++; u.ti0 = u.tj; }
      }
      return u;
    };
    auto load_tiles = [&](Unit u, double (&pq)[TPI][4]) {
      const int tj = min(u.tj, nt - 1);                  // (clamped, unconditional loads: branch-free, so the compiler can
#pragma unroll                                             //  count them exactly instead of draining the queue at every use)
      for (int q = 0; q < TPI; q++) {
        const int i = min(16 * (u.ti0 + q) + lr, nact - 1);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int j = min(16 * tj + lk + 4 * rg, nact - 1);
          pq[q][rg] = P[i + (long)j * ld];
        }
      }
    };
    auto draw = [&]() {                                  // the unit of the next ticket (wave-uniform)
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1);
      return next_unit(Unit{0, 0}, __builtin_amdgcn_readfirstlane(t));
    };
    double pv[TPI][4];
    Unit u = ticket ? draw() : next_unit(Unit{0, 0}, wave);
    while (u.tj < nt) {
      load_tiles(u, pv);                // (the SIMD's other wave computes meanwhile; an explicit prefetch of the next unit
      const Unit un = ticket ? draw() : next_unit(u, NWV);   //  did not pay for its registers)
      const int tj = u.tj, j0t = 16 * tj;
#pragma unroll
      for (int q = 0; q < TPI; q++) {
        const int ti = u.ti0 + q;
        if (ti >= nt) break;
        const int i0 = 16 * ti;
        const bool diag_t = ti == tj;
        v4f64 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int sk = 0; sk < ksteps; sk++) {
          const int c = 4 * sk + lk;                       // this lane's contraction index: column c of pair c >> 1
          const double wjc = Wp[(j0t + lr) * BLD + c];
          const double2 wi = *reinterpret_cast<const double2*>(Wp + (i0 + lr) * BLD + (c & ~1));
          const double2 sv = *reinterpret_cast<const double2*>(SiL + 4 * (c >> 1) + 2 * (c & 1));
          const double kic = wi.x * sv.x + wi.y * sv.y;    // K[i][c], the same expression as in the panel phase
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wjc, kic, acc, 0, 0, 0);                          // K_i . W_j
          if (diag_t) {                                    // (wave-uniform)
            const double2 wj = *reinterpret_cast<const double2*>(Wp + (j0t + lr) * BLD + (c & ~1));
            const double wic = (c & 1) ? wi.y : wi.x;
            const double kjc = wj.x * sv.x + wj.y * sv.y;
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(kjc, wic, acc2, 0, 0, 0);                      // K_j . W_i
          }
        }
        if (diag_t) {
#pragma unroll
          for (int rg = 0; rg < 4; rg++)
            if (i0 + lr < j0t + lk + 4 * rg) acc[rg] = acc2[rg];   // upper triangle of the diagonal tile
        }
        const int i = i0 + lr;
        const double li = lam[min(i, nact - 1)];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {                   // (results stay in pv: every store is issued after the last
          const int j = j0t + lk + 4 * rg;                 //  wait on a load -- a store ahead of a load wait would be waited for too)
          const double lj = lam[min(j, nact - 1)];
          const double Lij = partial ? (lj + li - li * lj) : 1.0;
          double v = pv[q][rg] - Lij * acc[rg];
          if (i == j && i >= 16 && i < nact && (i - 16) % 3 == 2) v = diag[(i - 16) / 3];
          pv[q][rg] = v;
        }
      }
#pragma unroll
      for (int q = 0; q < TPI; q++) {
        const int ti = u.ti0 + q;
        const int i = 16 * ti + lr;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int j = 16 * tj + lk + 4 * rg;
          if (ti < nt && i < nact && j < nact) {
            P[i + (long)j * ld] = pv[q][rg];   // (no mirror store: nothing reads above the diagonal, see the panel load)
          }
        }
      }
      u = un;
    }
}

// ------------------------------------------------------------------------------------------------
// BG = measurements per group (template parameter: 16, 24 or 32 -- the largest whose panel fits the LDS, so that the narrower
// filters of this family cross HBM fewer times); BLD = 2 BG + 2 = LDS row stride (doubles) of the n x 2 BG panel; BWIN = 2 BG
// measurement-list entries staged per group

struct BlkLds {
  int xs, lam, Wp, Si, sm, diag, gsl, win, total;   // offsets in doubles
  __host__ __device__ BlkLds(int N, int n, int nxs, int BG) {
    const int nr = (n + 15) & ~15, BLD = 2 * BG + 2, BWIN = 2 * BG;
    int o = 0;
    auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
    xs = take(nxs); lam = take(n); Wp = take(nr * BLD); Si = take(4 * BG); sm = take(32);   // sm: pzz[2][4] (+ spare)
    diag = take(N > 0 ? N : 1);
    gsl = take(2 * BG);   // ints: slot[BG], measurement index[BG]
    win = take(7 * BWIN);   // staged window of the measurement list: z [BWIN][2], R [BWIN][4], slot [BWIN] (ints)
    total = o;
  }
};

template <int T, int BG>
__global__ __launch_bounds__(T) void k_update_feat_blocked(StreamArgs a, const double* __restrict__ z_all,
                                                           const int* __restrict__ slot_all, int M,
                                                           const double* __restrict__ R_all, long r_stride_b,
                                                           long r_stride_m, int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (a.active && !a.active[b]) return;
  const int n = a.n, ld = a.ld;
  constexpr int BLD = 2 * BG + 2, BWIN = 2 * BG;
  const BlkLds L(a.N, n, a.nxs, BG);
  double* xs = smem + L.xs;
  double* lam = smem + L.lam;
  double* Wp = smem + L.Wp;     // panel of raw columns, turned into W pair by pair
  double* SiL = smem + L.Si;    // per pair g: {Si00, Si10, Si01, Si11} = the two COLUMNS of S^-1 (zero if the update was skipped)
  double* pzz = smem + L.sm;    // [2][4] zeta-zeta block of the current / next measurement
  int* badf = reinterpret_cast<int*>(pzz + 8);   // [BG] NaN-guard verdict of each measurement of the group (set by any thread)
  double* diag = smem + L.diag; // running P(rho_f, rho_f)
  int* gsl = reinterpret_cast<int*>(smem + L.gsl);
  int* gml = gsl + BG;
  double* wz = smem + L.win;
  double* wR = wz + 2 * BWIN;
  int* wsl = reinterpret_cast<int*>(wR + 4 * BWIN);
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nact = 16 + 3 * len;
  const DevParams& prm = *a.dp;
  const bool partial = prm.use_partial_update != 0;
  unsigned flag = 0;
  constexpr int NWV = T / 64;
  const int lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  for (int i = tid; i < n; i += T) lam[i] = a.lambda[i];
  __syncthreads();

  int m = 0, mbase = 0;
  while (m < M) {
    // ---- stage a window of the measurement list in LDS (one parallel fetch instead of a chain of dependent global loads),
    //      then form the next group from it (every thread scans the same list: uniform control flow; tid 0 records it)
    __syncthreads();   // (the previous group is done with the window)
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
    __syncthreads();
    int Gn = 0;
    {
      int mm = m;
      const int mend = min(M, m + BWIN);
      unsigned long long seen0 = 0, seen1 = 0, seen2 = 0;   // slots already in the group (3 x 64 >= the 160-feature limit)
      while (mm < mend && Gn < BG) {
        const int slot = wsl[mm - m];
        if (slot >= 0) {
          unsigned long long& w = slot < 64 ? seen0 : (slot < 128 ? seen1 : seen2);
          const unsigned long long bit = 1ull << (slot & 63);
          if (w & bit) break;                             // repeated slot: it opens the next group
          w |= bit;
          if (tid == 0) { gsl[Gn] = slot; gml[Gn] = mm - m; }   // (index into the window)
          Gn++;
        }
        mm++;
      }
      const int m0 = m;
      m = mm;
      mbase = m0;
    }
    if (Gn == 0) continue;
    // ---- 1. panel <- the zeta columns of the group's features; unused columns <- 0.  Only the LOWER triangle of P is valid:
    //      element (i, c) of a column comes from the column where i >= c (coalesced along the rows) and from ROW c where i < c
    //      (one 8-byte read per thread, a column apart; the rows of a group's features are mostly neighbours -- measurements
    //      arrive in slot order -- so a thread's reads share 128-byte lines).  That costs about a quarter of a pass in fetched
    //      lines per group and saves the mirrored stores of every pass (r02: up to the whole upper triangle per pass, the
    //      kernel's largest single write stream) and the propagate's transpose pass.
    for (int i = tid; i < nact; i += T) {
#pragma unroll 1
      for (int c0 = 0; c0 < 2 * BG; c0 += 8) {           // eight independent column loads in flight per thread
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int c = c0 + k;
          const int col = 16 + 3 * gsl[(c < 2 * Gn) ? (c >> 1) : 0] + (c & 1);
          v[k] = (c < 2 * Gn) ? P[max(i, col) + (long)min(i, col) * ld] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; k += 2) *reinterpret_cast<double2*>(Wp + i * BLD + c0 + k) = make_double2(v[k], v[k + 1]);
      }
    }
    if (tid < 4 * BG) SiL[tid] = 0.0;
    if (tid < BG) badf[tid] = 0;
    for (int f = tid; f < len; f += T) diag[f] = P[(16 + 3 * f + 2) + (long)(16 + 3 * f + 2) * ld];
    // The zeta-zeta 2x2 of the measurement about to be processed is handed over in pzz[parity][4] by the two threads that own
    // its rows (they are the ones that keep those panel entries current), so nobody reads panel rows that are being rewritten.
    const int i = tid;                                      // this thread's row of the panel (T >= n, checked on the host)
    // Lambda of (this row, a bearing column):  lambda_i + lambda_c - lambda_i lambda_c  with the bearing components' lambda_c
    double Lza = 1.0, Lzb = 1.0;
    if (partial) { const double li = lam[min(i, n - 1)], la = lam[16], lb = lam[17]; Lza = la + li - li * la; Lzb = lb + li - li * lb; }
    auto publish_pzz = [&](int gn) {                        // for measurement gn of the group, from the thread's own row
      if (gn < Gn) {
        const int jn = 16 + 3 * gsl[gn];
        if (i == jn || i == jn + 1)
          *reinterpret_cast<double2*>(pzz + 4 * (gn & 1) + 2 * (i - jn)) = *reinterpret_cast<const double2*>(Wp + i * BLD + 2 * gn);
      }
    };
    auto fix_depth_own = [&]() {                            // fix_depth (:271) on the state and the running diagonal
      for (int f = tid; f < len; f += T) {
        const int xR = xZ + 5 * f + 4;
        double rho = xs[xR];
        const double reset = 1.0 / (2.0 * prm.min_depth);
        if (rho != rho) { rho = reset; flag |= FLAG_NAN; }
        if (rho < 0.0) {
          const double err = reset - rho;
          diag[f] += err * err;
          rho = reset;
          flag |= FLAG_NEGDEPTH;
        } else if (rho > 1e2) {
          diag[f] = prm.P0_feat[2];
          rho = reset;
        }
        xs[xR] = rho;
      }
    };
    publish_pzz(0);   // (own row: written by this thread in the load above)
    __syncthreads();
    // ---- 2. the measurements of the group, in order; two barriers each
    for (int g = 0; g < Gn; g++) {
      const int slot = gsl[g], wi = gml[g];
      int* res = result_all ? &result_all[(long)b * M + mbase + wi] : nullptr;
      const double* R = wR + 4 * wi;                                // column-major 2x2
      // prediction, innovation, S^-1 and the gate: every thread for itself (uniform data; cheaper than a broadcast + barrier)
      double zhat[2], Hb[4];
      {
        double t1[3], t2[3], zt[3];
        bearing_frame_fast(xs + xZ + 5 * slot, t1, t2, zt);
        h_feat_frame(t1, t2, zt, prm, zhat, Hb);
      }
      const double h00 = Hb[0], h01 = Hb[1], h10 = Hb[2], h11 = Hb[3];
      const double r0 = wz[2 * wi] - zhat[0], r1 = wz[2 * wi + 1] - zhat[1];   // residual (vi_ekf_meas.cpp:220)
      double S[4], Si[4];
      {   // S = Hb P_zz Hb^T + R
        const double* pz = pzz + 4 * (g & 1);
        const double p00 = pz[0], p01 = pz[1], p10 = pz[2], p11 = pz[3];
        const double w00 = p00 * h00 + p01 * h01, w01 = p00 * h10 + p01 * h11;      // W rows j0, j0+1
        const double w10 = p10 * h00 + p11 * h01, w11 = p10 * h10 + p11 * h11;
        S[0] = h00 * w00 + h01 * w10 + R[0];
        S[1] = h00 * w01 + h01 * w11 + R[2];
        S[2] = h10 * w00 + h11 * w10 + R[1];
        S[3] = h10 * w01 + h11 * w11 + R[3];
      }
      inv2_fast(S, Si);
      const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;   // vi_ekf_meas.cpp:234
      if (mahal > 9.0) {                                   // gate (:235-239): returns before fix_depth
        if (res && tid == 0) *res = 1;
        if (i < nact) *reinterpret_cast<double2*>(Wp + i * BLD + 2 * g) = make_double2(0.0, 0.0);
        publish_pzz(g + 1);
        __syncthreads();
        continue;
      }
      int bad = 0;
      double k0 = 0.0, k1 = 0.0;                           // this thread's row of K
      if (i < nact) {                                      // W = P H^T, K = W S^-1 (:241), NaN guard (:247)
        const double2 pr = *reinterpret_cast<const double2*>(Wp + i * BLD + 2 * g);
        const double w0 = pr.x * h00 + pr.y * h01, w1 = pr.x * h10 + pr.y * h11;
        k0 = w0 * Si[0] + w1 * Si[2]; k1 = w0 * Si[1] + w1 * Si[3];
        *reinterpret_cast<double2*>(Wp + i * BLD + 2 * g) = make_double2(w0, w1);
        if (k0 != k0 || k1 != k1) bad = 1;
      }
      if (h00 != h00 || h01 != h01 || h10 != h10 || h11 != h11) bad = 1;
      if (bad) badf[g] = 1;                                // (a flag word and ONE barrier: __syncthreads_or is a workgroup reduction,
      __syncthreads();                                     //  two barriers and an LDS round trip per measurement)
      bad = badf[g];                                       // barrier 1: the W pair is complete
      if (bad) {
        if (i < nact) *reinterpret_cast<double2*>(Wp + i * BLD + 2 * g) = make_double2(0.0, 0.0);
      } else {
        if (tid == 0) { SiL[4 * g + 0] = Si[0]; SiL[4 * g + 1] = Si[2]; SiL[4 * g + 2] = Si[1]; SiL[4 * g + 3] = Si[3]; }
        // state correction  x <- x [+] (lambda o K r)   (:254-255 / :262-263);  K rows re-formed from W where needed
        auto krow = [&](int row, double& q0, double& q1) {
          const double2 w = *reinterpret_cast<const double2*>(Wp + row * BLD + 2 * g);
          q0 = w.x * Si[0] + w.y * Si[2]; q1 = w.x * Si[1] + w.y * Si[3];
        };
        if (tid == T - 1) {                                // (a thread without a feature of its own: T - 1 >= n > len)
          double dxb[16], xo[17];
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const double l = partial ? lam[q] : 1.0;
            double q0, q1;
            krow(q, q0, q1);
            dxb[q] = (l * q0) * r0 + (l * q1) * r1;
          }
          body_boxplus_fast(xs, dxb, xo);
#pragma unroll
          for (int q = 0; q < 17; q++) xs[q] = xo[q];
        }
        for (int f = tid; f < len; f += T) {
          const int d = 16 + 3 * f;
          double dv[3], k2[2] = {0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const double l = partial ? lam[d + q] : 1.0;
            double q0, q1;
            krow(d + q, q0, q1);
            dv[q] = (l * q0) * r0 + (l * q1) * r1;
            k2[0] = q0; k2[1] = q1;
          }
          double qn[4];
          q_feat_boxplus_fast(xs + xZ + 5 * f, dv[0], dv[1], qn);
          double* xf = xs + xZ + 5 * f;
          xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
          xf[4] += dv[2];
          // running rho-rho diagonal:  P_ii -= Lambda_ii (K_i . W_i)
          const int ir = d + 2;
          const double li = lam[ir];
          const double Lii = partial ? (li + li - li * li) : 1.0;
          const double2 wr = *reinterpret_cast<const double2*>(Wp + ir * BLD + 2 * g);
          diag[f] -= Lii * (k2[0] * wr.x + k2[1] * wr.y);
        }
        // the later panel columns follow the update:  P_ic -= Lambda_ic (K_i . W_c), this thread's row
        // Four column pairs per trip, their operands loaded together: one pair at a time is a chain of LDS round trips (the
        // compiler keeps every load behind the previous pair's store), about a quarter of the measurement's latency at 16 per group.
        // (The mask of a (row, bearing column) pair is the same for every feature: lambda_feat is one triple for all slots.)
        if (i < nact) {
#pragma unroll 1
          for (int gc0 = g + 1; gc0 < Gn; gc0 += 4) {
            double2 wa[4], wb[4], pc[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int gc = min(gc0 + q, Gn - 1);
              const int cr = 16 + 3 * gsl[gc];
              wa[q] = *reinterpret_cast<const double2*>(Wp + cr * BLD + 2 * g);
              wb[q] = *reinterpret_cast<const double2*>(Wp + (cr + 1) * BLD + 2 * g);
              pc[q] = *reinterpret_cast<const double2*>(Wp + i * BLD + 2 * gc);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
              if (gc0 + q < Gn) {
                pc[q].x -= Lza * (k0 * wa[q].x + k1 * wa[q].y);
                pc[q].y -= Lzb * (k0 * wb[q].x + k1 * wb[q].y);
                *reinterpret_cast<double2*>(Wp + i * BLD + 2 * (gc0 + q)) = pc[q];
              }
            }
          }
        }
      }
      fix_depth_own();                                     // (not gated: runs after a NaN-guarded update too, :271)
      publish_pzz(g + 1);
      if (res && tid == 0) *res = 0;
      __syncthreads();                                     // barrier 2: state, panel and pzz are ready for the next measurement
    }
    // ---- 3. one pass over P:  P -= Lambda o (K W^T), 16 x 16 tiles on the fp64 matrix cores (blk_pass above)
    blk_pass<T, BLD>(P, ld, nact, Gn, Wp, SiL, lam, diag, partial, lane, wave);
    __syncthreads();
  }
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[b], flag);
}

// ------------------------------------------------------------------------------------------------
// generic measurement update: VIEKF::update for every measurement model of the reference's table
// (vi_ekf_meas.cpp:196-278 with h_acc/h_alt/h_att/h_pos/h_vel/h_qzeta/h_feat/h_depth/h_inv_depth, :281-386).
// Each H has at most 6 non-zero columns, so W = P H^T is a combination of <= 6 columns of P and the update is the same
// rank-r sweep  P_ij -= Lambda_ij (K_i . W_j)  as for FEAT.  One workgroup per filter, P in HBM/L2 (these models run at
// IMU / truth rate, one per propagate -- not the N-per-frame hot loop).
// ------------------------------------------------------------------------------------------------
constexpr int MT_ACC = 0, MT_ALT = 1, MT_ATT = 2, MT_POS = 3, MT_VEL = 4, MT_QZETA = 5, MT_FEAT = 6, MT_DEPTH = 8,
              MT_INV_DEPTH = 9;   // include/vi_ekf.h:113-124

__device__ __forceinline__ void q_log_dev(const double* q, double* o) {   // src/quat.cpp:82-98
  const double nv = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (nv < 1e-8) { o[0] = o[1] = o[2] = 0.0; }
  else { const double s = 2.0 * atan2(nv, q[0]) / nv; o[0] = s * q[1]; o[1] = s * q[2]; o[2] = s * q[3]; }
}
__device__ __forceinline__ void q_boxminus_dev(const double* q1, const double* q2, double* o) {   // src/quat.cpp:319-327
  const double inv[4] = {q2[0], -q2[1], -q2[2], -q2[3]};
  double dq[4];
  q_otimes(inv, q1, dq);
  if (dq[0] < 0.0) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; }
  q_log_dev(dq, o);
}
__device__ __forceinline__ void q_feat_boxminus_dev(const double* qj, const double* qi, double* o) {   // math_helper.h:25-43
  double t1[3], t2[3], zi[3], a1[3], a2[3], zj[3];
  bearing_frame(qi, t1, t2, zi);
  bearing_frame(qj, a1, a2, zj);
  const double d[3] = {zi[0] - zj[0], zi[1] - zj[1], zi[2] - zj[2]};
  if (sqrt(dot3(d, d)) > 1e-8) {
    double s[3];
    cross3(zi, zj, s);
    const double ns = sqrt(dot3(s, s));
    const double th = acos(dot3(zi, zj));
    s[0] = s[0] / ns * th; s[1] = s[1] / ns * th; s[2] = s[2] / ns * th;
    o[0] = dot3(t1, s); o[1] = dot3(t2, s);
  } else { o[0] = 0.0; o[1] = 0.0; }
}
// r x r inverse (r <= 3) by LU with partial pivoting, row-major (Eigen's dynamic-size inverse, vi_ekf_meas.cpp:232)
__device__ __forceinline__ void small_inverse_dev(int r, const double* S, double* Si) {
  double a[9], b[9];
  for (int i = 0; i < r; i++) for (int j = 0; j < r; j++) { a[i * 3 + j] = S[i * r + j]; b[i * 3 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < r; c++) {
    int piv = c; double best = fabs(a[c * 3 + c]);
    for (int i = c + 1; i < r; i++) if (fabs(a[i * 3 + c]) > best) { best = fabs(a[i * 3 + c]); piv = i; }
    if (piv != c) for (int j = 0; j < r; j++) {
      double t = a[c * 3 + j]; a[c * 3 + j] = a[piv * 3 + j]; a[piv * 3 + j] = t;
      t = b[c * 3 + j]; b[c * 3 + j] = b[piv * 3 + j]; b[piv * 3 + j] = t;
    }
    for (int i = c + 1; i < r; i++) {
      const double l = a[i * 3 + c] / a[c * 3 + c];
      for (int j = 0; j < r; j++) { a[i * 3 + j] -= l * a[c * 3 + j]; b[i * 3 + j] -= l * b[c * 3 + j]; }
    }
  }
  for (int j = 0; j < r; j++)
    for (int i = r - 1; i >= 0; i--) {
      double s = b[i * 3 + j];
      for (int k = i + 1; k < r; k++) s -= a[i * 3 + k] * Si[k * r + j];
      Si[i * r + j] = s / a[i * 3 + i];
    }
}

// ------------------------------------------------------------------------------------------------
// Read-only evaluations for the log writer (src/vi_ekf/vi_ekf_log.cpp): what the reference records next to a propagate
// (xdot = dx_ of VIEKF::dynamics, vi_ekf_dyn.cpp:6-134; the diagonal of P) and next to an update (zhat = h(x),
// vi_ekf_meas.cpp:281-386).  They do not touch the filter, so the hot kernels carry no logging arguments.
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_eval_xdot(StreamArgs a, const double* __restrict__ u_all, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  double* xs = smem;
  double* Abb = xs + a.nxs;
  double* Gb = Abb + 256;
  double* xdb = Gb + 96;
  BodyCtx* ctx = reinterpret_cast<BodyCtx*>(xdb + 16);
  const int len = a.len[b];
  const double* xg = a.x + a.si(b) * a.nxs;
  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  __syncthreads();
  if (tid == 0) {
    double ub[6];
    q_rota(a.dp->q_b_u, u_all + (long)b * 6, ub);          // vi_ekf.cpp:265-267
    q_rota(a.dp->q_b_u, u_all + (long)b * 6 + 3, ub + 3);
    body_ctx(xs, ub, (*a.dp), *ctx);
    body_dynamics(*ctx, (*a.dp), xdb, Abb, Gb);
  }
  __syncthreads();
  double* o = out + (long)b * a.n;
  for (int i = tid; i < a.n; i += T) {
    double v = 0.0;
    if (i < 16) v = xdb[i];
    else {
      const int f = (i - 16) / 3, q = (i - 16) - 3 * f;
      if (f < len) {
        double xd3[3], Afv[9], Afg[9], Aff[9];
        feature_dynamics(xs + xZ + 5 * f, xs[xZ + 5 * f + 4], *ctx, xd3, Afv, Afg, Aff);
        v = xd3[q];
      }
    }
    o[i] = v;
  }
}

// zhat [B][4] of measurement model `type` at the current state (one thread per filter); unused entries and filters whose
// slot is out of range get NaN
#ifndef VIEKF_INSTANCES_ONLY
__global__ void k_eval_h(StreamArgs a, int type, const int* __restrict__ slot_all, double* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const double* xs = a.x + a.si(b) * a.nxs;
  const DevParams& prm = *a.dp;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  double zhat[4] = {nan, nan, nan, nan};
  const bool needs_slot = type == 5 || type == 6 || type == 8 || type == 9;   // QZETA, FEAT, DEPTH, INV_DEPTH
  const int slot = (needs_slot && slot_all) ? slot_all[b] : 0;
  if (!needs_slot || (slot >= 0 && slot < a.len[b])) {
    if (type == 0) {                                          // ACC, vi_ekf_meas.cpp:281-306
      if (prm.use_drag_term) {
        const double mu = xs[xMU];
        zhat[0] = -mu * xs[xVEL] + xs[xB_A]; zhat[1] = -mu * xs[xVEL + 1] + xs[xB_A + 1];
      } else {
        const double g[3] = {0.0, 0.0, kGravity}; double gB[3];
        q_rotp(xs + xATT, g, gB);
        for (int i = 0; i < 3; i++) zhat[i] = xs[xB_A + i] - gB[i];
      }
    } else if (type == 1) zhat[0] = -xs[xPOS + 2];            // ALT
    else if (type == 2) { for (int i = 0; i < 4; i++) zhat[i] = xs[xATT + i]; }
    else if (type == 3) { for (int i = 0; i < 3; i++) zhat[i] = xs[xPOS + i]; }
    else if (type == 4) { for (int i = 0; i < 3; i++) zhat[i] = xs[xVEL + i]; }
    else if (type == 5) { for (int i = 0; i < 4; i++) zhat[i] = xs[xZ + 5 * slot + i]; }
    else if (type == 6) { double Hb[4], zh[2]; h_feat(xs + xZ + 5 * slot, prm, zh, Hb); zhat[0] = zh[0]; zhat[1] = zh[1]; }
    else if (type == 8) zhat[0] = 1.0 / xs[xZ + 5 * slot + 4];
    else if (type == 9) zhat[0] = xs[xZ + 5 * slot + 4];
  }
  for (int i = 0; i < 4; i++) out[(long)b * 4 + i] = zhat[i];
}
#endif

// P <- (P + P^T) / 2 of every filter (a covariance handed in through viekf_batch_set_state): the kernels keep P exactly
// symmetric from then on and rely on it
#ifndef VIEKF_INSTANCES_ONLY
__global__ void k_symmetrize(StreamArgs a) {
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B || e >= (long)a.n * a.n) return;
  const int i = (int)(e % a.n), j = (int)(e / a.n);
  if (i <= j) return;
  double* P = a.P + a.si(b) * a.n * a.ld;
  const double v = 0.5 * (P[i + (long)j * a.ld] + P[j + (long)i * a.ld]);
  P[i + (long)j * a.ld] = v;
  P[j + (long)i * a.ld] = v;
}
#endif

// upper triangle <- lower triangle (bit for bit), 32 x 32 tiles through LDS so that both sides are coalesced along the rows.
// Runs when a grouped update left the upper triangle stale (k_update_feat_blocked) and something is about to read all of P.
#ifndef VIEKF_INSTANCES_ONLY
__global__ __launch_bounds__(256) void k_mirror_upper(StreamArgs a) {
  __shared__ double t[32][33];
  const int b = blockIdx.y, n = a.n, nt = (n + 31) >> 5;
  // tile pairs (ti >= tj) numbered row by row: blockIdx.x = ti (ti + 1) / 2 + tj
  int ti = (int)((sqrtf(1.0f + 8.0f * (float)blockIdx.x) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) ti--;
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ti++;
  const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
  if (ti >= nt) return;
  double* P = a.P + a.si(b) * n * a.ld;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  for (int c = ly; c < 32; c += 8) {
    const int i = 32 * ti + lx, j = 32 * tj + c;
    t[c][lx] = (i < n && j < n) ? P[i + (long)j * a.ld] : 0.0;
  }
  __syncthreads();
  for (int c = ly; c < 32; c += 8) {
    const int i = 32 * tj + lx, j = 32 * ti + c;     // destination element (i, j) = source element (j, i)
    if (i < n && j < n && i < j) P[i + (long)j * a.ld] = t[lx][c];   // (a diagonal tile: its strictly upper part only)
  }
}
#endif

#ifndef VIEKF_INSTANCES_ONLY
__global__ void k_cov_diag(StreamArgs a, double* __restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < a.B && i < a.n) out[(long)b * a.n + i] = a.P[a.si(b) * a.n * a.ld + i + (long)i * a.ld];
}
#endif

// per-filter history: copies (x, P) of filter b between the batch's live buffers and ring slot slot[b] (< 0: filter skipped);
// to_ring != 0: live -> ring.  Filters on independent clocks advance and rewind their rings separately (viekf_seq, independent
// mode); the feature counts are not part of a slot, as in the reference's ring (include/vi_ekf.h:156-160).
#ifndef VIEKF_INSTANCES_ONLY
__global__ __launch_bounds__(256) void k_ring_copy(StreamArgs a, double* __restrict__ ring_x, double* __restrict__ ring_P,
                                                   const int* __restrict__ slot, int to_ring, int depth) {
  const int b = blockIdx.x;
  if (b >= a.B) return;
  const int sl = slot[b];
  if (sl < 0) return;
  if (sl >= depth) {   // (a device-resident slot list cannot be validated by the host: nothing is copied, the filter is flagged)
    if (threadIdx.x == 0) atomicOr(&a.flags[b], FLAG_INTERNAL);
    return;
  }
  const long nP = (long)a.n * a.ld;
  double* lx = a.x + a.si(b) * a.nxs;
  double* lP = a.P + a.si(b) * nP;
  double* rx = ring_x + ((long)sl * a.B + b) * a.nxs;
  double* rP = ring_P + ((long)sl * a.B + b) * nP;
  const double2* src = reinterpret_cast<const double2*>(to_ring ? lP : rP);   // (ld is even and the buffers 16-byte aligned)
  double2* dst = reinterpret_cast<double2*>(to_ring ? rP : lP);
  for (long e = threadIdx.x; e < nP / 2; e += 256) dst[e] = src[e];
  for (int e = threadIdx.x; e < a.nxs; e += 256) (to_ring ? rx : lx)[e] = (to_ring ? lx : rx)[e];
}
#endif

// per-filter live slots: smap[b] = slot[b] * B + b for the filters with slot[b] >= 0 (viekf_batch_select_filters: the rewind of
// filters on clocks of their own is this index, vi_ekf_meas.cpp:50-52 per filter)
#ifndef VIEKF_INSTANCES_ONLY
__global__ __launch_bounds__(256) void k_set_smap(int* __restrict__ smap, const int* __restrict__ slot, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B && slot[b] >= 0) smap[b] = slot[b] * B + b;
}
#endif

// a rectangular block P[r0 .. r0+nr, c0 .. c0+nc) of every filter -> out [B][nc][nr] (column-major per filter)
#ifndef VIEKF_INSTANCES_ONLY
__global__ void k_cov_block(StreamArgs a, int r0, int c0, int nr, int nc, double* __restrict__ out) {
  const int b = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B || e >= nr * nc) return;
  const int c = e / nr, r = e - c * nr;
  out[(long)b * nr * nc + e] = a.P[a.si(b) * a.n * a.ld + (r0 + r) + (long)(c0 + c) * a.ld];
}
#endif

// The reference's measurement models (src/vi_ekf/vi_ekf_meas.cpp:281-386): zhat = h(x) and the non-zero columns of H -- at most six
// (`cols`, their number `nc`), Hc [3][6] row-major = the entries of H in those columns.  One lane; xs = the filter's state.
__device__ __forceinline__ void meas_model(int type, const double* xs, int slot, const DevParams& prm, double* zhat, int* cols,
                                           double* Hc, int& nc) {
  nc = 0;
  for (int i = 0; i < 18; i++) Hc[i] = 0.0;
  auto col = [&](int c) { cols[nc] = c; return nc++; };
  if (type == MT_ACC) {                                       // vi_ekf_meas.cpp:281-306
    if (prm.use_drag_term) {
      const double mu = xs[xMU];
      zhat[0] = -mu * xs[xVEL] + xs[xB_A]; zhat[1] = -mu * xs[xVEL + 1] + xs[xB_A + 1];
      int c;
      c = col(dxVEL); Hc[0 * 6 + c] = -mu;  c = col(dxVEL + 1); Hc[1 * 6 + c] = -mu;
      c = col(dxB_A); Hc[0 * 6 + c] = 1.0;  c = col(dxB_A + 1); Hc[1 * 6 + c] = 1.0;
      c = col(dxMU); Hc[0 * 6 + c] = -xs[xVEL]; Hc[1 * 6 + c] = -xs[xVEL + 1];
    } else {
      const double g[3] = {0.0, 0.0, kGravity}; double gB[3], ng[3], Sk[9];
      q_rotp(xs + xATT, g, gB);
      for (int i = 0; i < 3; i++) { zhat[i] = xs[xB_A + i] - gB[i]; ng[i] = -1.0 * gB[i]; }
      skew3(ng, Sk);
      for (int j = 0; j < 3; j++) { const int c = col(dxATT + j); for (int i = 0; i < 3; i++) Hc[i * 6 + c] = Sk[i * 3 + j]; }
      for (int j = 0; j < 3; j++) { const int c = col(dxB_A + j); Hc[j * 6 + c] = 1.0; }
    }
  } else if (type == MT_ALT) { zhat[0] = -xs[xPOS + 2]; const int c = col(dxPOS + 2); Hc[c] = -1.0; }
  else if (type == MT_ATT) { for (int i = 0; i < 4; i++) zhat[i] = xs[xATT + i]; for (int j = 0; j < 3; j++) { const int c = col(dxATT + j); Hc[j * 6 + c] = 1.0; } }
  else if (type == MT_POS) { for (int j = 0; j < 3; j++) { zhat[j] = xs[xPOS + j]; const int c = col(dxPOS + j); Hc[j * 6 + c] = 1.0; } }
  else if (type == MT_VEL) { for (int j = 0; j < 3; j++) { zhat[j] = xs[xVEL + j]; const int c = col(dxVEL + j); Hc[j * 6 + c] = 1.0; } }
  else if (type == MT_QZETA) { for (int i = 0; i < 4; i++) zhat[i] = xs[xZ + 5 * slot + i]; for (int j = 0; j < 2; j++) { const int c = col(dxZ + 3 * slot + j); Hc[j * 6 + c] = 1.0; } }
  else if (type == MT_FEAT) {
    double Hb[4]; h_feat(xs + xZ + 5 * slot, prm, zhat, Hb);
    for (int j = 0; j < 2; j++) { const int c = col(dxZ + 3 * slot + j); Hc[0 * 6 + c] = Hb[0 * 2 + j]; Hc[1 * 6 + c] = Hb[1 * 2 + j]; }
  } else if (type == MT_DEPTH) { const double rho = xs[xZ + 5 * slot + 4]; zhat[0] = 1.0 / rho; const int c = col(dxZ + 3 * slot + 2); Hc[c] = -1.0 / (rho * rho); }
  else if (type == MT_INV_DEPTH) { zhat[0] = xs[xZ + 5 * slot + 4]; const int c = col(dxZ + 3 * slot + 2); Hc[c] = 1.0; }
}

template <int T>
__global__ __launch_bounds__(T) void k_update_generic(StreamArgs a, int type, int zdim, int rdim,
                                                      const double* __restrict__ z_all, const int* __restrict__ slot_all,
                                                      const double* __restrict__ R_all, long r_stride_b,
                                                      const unsigned char* __restrict__ active_all,
                                                      int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, ld = a.ld;
  const DevParams& prm = *a.dp;
  double* xs = smem;            // [nxs]
  double* W = xs + a.nxs;       // [n][3]
  double* K = W + 3 * n;        // [n][3]
  double* lam = K + 3 * n;      // [n]
  double* sm = lam + n;         // [64]: hcols (int as double) [0..5], Hc [6..23] (3 rows x 6 cols), res [24..26], ncol [27], Si [28..36], verdict [37]
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const int nact = 16 + 3 * len;
  const int slot = slot_all ? slot_all[b] : 0;
  const bool active = active_all ? active_all[b] != 0 : true;
  int* res_out = result_all ? &result_all[b] : nullptr;
  if (active_all && active_all[b] == 2) { if (res_out && tid == 0) *res_out = -1; return; }   // this filter skips the call
  unsigned flag = 0;
  const bool needs_slot = type == MT_QZETA || type == MT_FEAT || type == MT_DEPTH || type == MT_INV_DEPTH;
  if (needs_slot && (slot < 0 || slot >= len)) { if (res_out && tid == 0) *res_out = (slot < 0) ? -1 : 3; return; }
  const double* z = z_all + (long)b * zdim;
  {
    bool isnan_ = false;
    for (int i = 0; i < zdim; i++) isnan_ |= z[i] != z[i];
    if (isnan_) { if (res_out && tid == 0) *res_out = 2; return; }   // MEAS_NAN
  }
  for (int i = tid; i < xZ + 5 * len; i += T) xs[i] = xg[i];
  for (int i = tid; i < n; i += T) lam[i] = a.lambda[i];
  __syncthreads();

  if (tid == 0) {   // measurement model: zhat, the non-zero columns of H, residual
    int cols[6]; double Hc[18]; int nc = 0; double zhat[4] = {0, 0, 0, 0}; double r3[3] = {0, 0, 0};
    meas_model(type, xs, slot, prm, zhat, cols, Hc, nc);
    if (type == MT_QZETA) q_feat_boxminus_dev(z, zhat, r3);           // :210-213
    else if (type == MT_ATT) q_boxminus_dev(z, zhat, r3);             // :214-217
    else for (int i = 0; i < zdim && i < 3; i++) r3[i] = z[i] - zhat[i];
    for (int i = 0; i < 6; i++) sm[i] = (i < nc) ? (double)cols[i] : 0.0;
    for (int i = 0; i < 18; i++) sm[6 + i] = Hc[i];
    sm[24] = r3[0]; sm[25] = r3[1]; sm[26] = r3[2]; sm[27] = (double)nc;
  }
  __syncthreads();
  if (!active) {   // :230 -- an inactive measurement only runs fix_depth (and the logger)
    for (int f = tid; f < len; f += T) fix_depth_one(xs, P, ld, f, prm, flag);
    __syncthreads();
    for (int i = tid; i < xZ + 5 * len; i += T) xg[i] = xs[i];
    if (res_out && tid == 0) *res_out = 0;
    if (flag) atomicOr(&a.flags[b], flag);
    return;
  }
  const int nc = (int)sm[27];
  // W = P H^T  (n x r): combination of the nc non-zero columns
  for (int i = tid; i < nact; i += T) {
    double w[3] = {0.0, 0.0, 0.0};
    for (int c = 0; c < nc; c++) {
      const double pv = P[i + (long)((int)sm[c]) * ld];
      for (int q = 0; q < rdim; q++) w[q] += pv * sm[6 + q * 6 + c];
    }
    for (int q = 0; q < 3; q++) W[3 * i + q] = w[q];
  }
  __syncthreads();
  if (tid == 0) {   // S = H W[cols] + R, inverse, gate
    const double* R = R_all + (long)b * r_stride_b;   // column-major rdim x rdim
    double S[9], Si[9];
    for (int p = 0; p < rdim; p++)
      for (int q = 0; q < rdim; q++) {
        double s = 0.0;
        for (int c = 0; c < nc; c++) s += sm[6 + p * 6 + c] * W[3 * (int)sm[c] + q];
        S[p * rdim + q] = s + R[p + q * rdim];
      }
    small_inverse_dev(rdim, S, Si);
    double mahal = 0.0;
    for (int q = 0; q < rdim; q++) { double t = 0.0; for (int p = 0; p < rdim; p++) t += sm[24 + p] * Si[p * rdim + q]; mahal += t * sm[24 + q]; }
    for (int i = 0; i < 9; i++) sm[28 + i] = (i < rdim * rdim) ? Si[i] : 0.0;
    sm[37] = (mahal > 9.0) ? 1.0 : 0.0;   // :234-239
  }
  __syncthreads();
  if (sm[37] != 0.0) { if (res_out && tid == 0) *res_out = 1; return; }   // gated: returns before fix_depth
  int bad = 0;
  for (int i = tid; i < nact; i += T) {
    for (int q = 0; q < 3; q++) {
      double k = 0.0;
      if (q < rdim) for (int p = 0; p < rdim; p++) k += W[3 * i + p] * sm[28 + p * rdim + q];
      K[3 * i + q] = k;
      if (k != k) bad = 1;
    }
  }
  for (int i = 0; i < 18; i++) if (sm[6 + i] != sm[6 + i]) bad = 1;
  bad = __syncthreads_or(bad);
  if (!bad) {
    const bool partial = prm.use_partial_update != 0;
    if (tid == 0) {
      double dxb[16], xo[17];
      for (int i = 0; i < 16; i++) {
        const double l = partial ? lam[i] : 1.0;
        double s = 0.0;
        for (int q = 0; q < rdim; q++) s += (l * K[3 * i + q]) * sm[24 + q];
        dxb[i] = s;
      }
      body_boxplus(xs, dxb, xo);
      for (int i = 0; i < 17; i++) xs[i] = xo[i];
    }
    for (int f = tid; f < len; f += T) {
      const int d = 16 + 3 * f;
      double dv[3];
      for (int e = 0; e < 3; e++) {
        const double l = partial ? lam[d + e] : 1.0;
        double s = 0.0;
        for (int q = 0; q < rdim; q++) s += (l * K[3 * (d + e) + q]) * sm[24 + q];
        dv[e] = s;
      }
      double qn[4];
      q_feat_boxplus(xs + xZ + 5 * f, dv[0], dv[1], qn);
      double* xf = xs + xZ + 5 * f;
      xf[0] = qn[0]; xf[1] = qn[1]; xf[2] = qn[2]; xf[3] = qn[3];
      xf[4] += dv[2];
    }
    const long tot = (long)nact * nact;
    for (long e = tid; e < tot; e += T) {
      const int i = (int)(e % nact), j = (int)(e / nact);
      const int hi = max(i, j), lo = min(i, j);   // (exactly symmetric result, see k_update_feat_stream)
      double t = 0.0;
      for (int q = 0; q < rdim; q++) t += K[3 * hi + q] * W[3 * lo + q];
      const double li = lam[i], lj = lam[j];
      const double L = partial ? (lj + li - li * lj) : 1.0;
      P[i + (long)j * ld] -= L * t;
    }
  }
  __syncthreads();
  for (int f = tid; f < len; f += T) fix_depth_one(xs, P, ld, f, prm, flag);
  __syncthreads();
  for (int i = tid; i < xZ + 5 * len; i += T) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (res_out && tid == 0) *res_out = 0;
  if (flag) atomicOr(&a.flags[b], flag);
}

// ------------------------------------------------------------------------------------------------
// init_feature (vi_ekf_feat.cpp:6-47), one workgroup per filter
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_init_feature(StreamArgs a, const double* __restrict__ pix_all,
                                                    const double* __restrict__ depth_all,
                                                    const unsigned char* __restrict__ mask, int* __restrict__ ok) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (mask && !mask[b]) { if (ok && tid == 0) ok[b] = 0; return; }
  const int len = a.len[b];
  if (len >= a.N) { if (ok && tid == 0) ok[b] = 0; return; }   // :9-10
  const int n = a.n, ld = a.ld;
  double* P = a.P + a.si(b) * n * ld;
  const int d0 = 16 + 3 * len, dmax = d0 + 3;
  if (tid == 0) {
    double q[4], rho;
    init_feature_state(pix_all + 2L * b, depth_all ? depth_all[b] : NAN, (*a.dp), q, &rho);
    double* xf = a.x + a.si(b) * a.nxs + xZ + 5 * len;
    xf[0] = q[0]; xf[1] = q[1]; xf[2] = q[2]; xf[3] = q[3]; xf[4] = rho;
  }
  // zero the cross strips, set the 3x3 block to P0_feat (:39-42)
  for (int e = tid; e < 3 * d0; e += T) {
    const int j = e / 3, r = e % 3;
    P[(d0 + r) + (long)j * ld] = 0.0;
    P[j + (long)(d0 + r) * ld] = 0.0;
  }
  if (tid < 9) {
    const int r = tid % 3, c = tid / 3;
    P[(d0 + r) + (long)(d0 + c) * ld] = (r == c) ? a.dp->P0_feat[r] : 0.0;
  }
  (void)dmax;
  __syncthreads();
  if (tid == 0) {
    a.len[b] = len + 1;
    if (ok) ok[b] = 1;
  }
}

// ------------------------------------------------------------------------------------------------
// keep_only_features / clear_feature (vi_ekf_feat.cpp:50-73,81-117): drop the features whose keep flag is 0 and shift the
// survivors (state and covariance rows/columns) to the upper-left corner, zeroing what is left over.  One workgroup per
// filter; columns are moved through an LDS buffer in increasing order, so the in-place compaction never overwrites a
// column that is still needed.  (The keyframe-overlap bookkeeping of keep_only_features stays on the host.)
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_keep_features(StreamArgs a, const unsigned char* __restrict__ keep_all,
                                                     int* __restrict__ new_len) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  const int n = a.n, ld = a.ld, N = a.N;
  double* colbuf = smem;                                   // [n]
  int* srcrow = reinterpret_cast<int*>(smem + n);          // [n] old row index of new row i (or -1)
  int* cnt = srcrow + n;
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const int len = a.len[b];
  const unsigned char* keep = keep_all + (long)b * N;
  if (tid == 0) {
    for (int i = 0; i < 16; i++) srcrow[i] = i;
    int k = 0;
    for (int f = 0; f < len; f++)
      if (keep[f]) { for (int q = 0; q < 3; q++) srcrow[16 + 3 * k + q] = 16 + 3 * f + q; k++; }
    for (int i = 16 + 3 * k; i < n; i++) srcrow[i] = -1;
    *cnt = k;
  }
  __syncthreads();
  const int k = *cnt;
  if (k == len) { if (new_len && tid == 0) new_len[b] = len; return; }   // nothing to drop
  // state: features are 5 doubles each
  if (tid == 0) {
    int w = 0;
    for (int f = 0; f < len; f++)
      if (keep[f]) { if (w != f) for (int q = 0; q < 5; q++) xg[xZ + 5 * w + q] = xg[xZ + 5 * f + q]; w++; }
    for (int i = xZ + 5 * k; i < xZ + 5 * N; i++) xg[i] = 0.0;   // "clean up the rest" (vi_ekf_feat.cpp:66-69): EVERYTHING past the
  }                                                              // kept features, also what a replay from older history left there
  const int nk = 16 + 3 * k, nold = n;
  for (int jn = 0; jn < nold; jn++) {
    const int jo = (jn < nk) ? srcrow[jn] : -1;
    if (jo >= 0) for (int i = tid; i < nk; i += T) colbuf[i] = P[srcrow[i] + (long)jo * ld];
    __syncthreads();
    for (int i = tid; i < nold; i += T) P[i + (long)jn * ld] = (jo >= 0 && i < nk) ? colbuf[i] : 0.0;
    __syncthreads();
  }
  if (tid == 0) { a.len[b] = k; if (new_len) new_len[b] = k; }
}

// ------------------------------------------------------------------------------------------------
// keyframe reset ("Dan's way", vi_ekf_kfr.cpp:56-157): position <- 0, yaw <- 0, P <- N P N^T where N differs from I only
// in the position block (0) and the attitude block (RMEKF Eq. 81).  One workgroup per filter, in place: first the row
// operation N P (every thread owns columns), then the column operation (every thread owns rows).  edge [B][17] (optional):
// {t(3), q_yaw(4), cov_pos(9, column-major), cov_yaw} of the relative pose before the reset, for the caller's global-pose
// bookkeeping (:58-62,125-126,147-149 -- the Xformd algebra of the reference's `geometry` dependency stays with the caller).
// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void k_keyframe_reset(StreamArgs a, const unsigned char* __restrict__ mask,
                                                      double* __restrict__ edge_all) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= a.B) return;
  if (mask && !mask[b]) return;
  const int n = a.n, ld = a.ld;
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * n * ld;
  const double* q = xg + xATT;
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const double yaw = atan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz));     // src/quat.cpp:221-224
  const double roll = atan2(2.0 * (qw * qx + qy * qz), 1.0 - 2.0 * (qx * qx + qy * qy));    // :211-214
  const double pitch = asin(2.0 * (qw * qy - qz * qx));                                     // :216-219
  const double cp = cos(roll), sp = sin(roll), tt = tan(pitch);                             // vi_ekf_kfr.cpp:134-136
  const double Na[9] = {1.0, sp * tt, cp * tt, 0.0, cp * cp, -cp * sp, 0.0, -cp * sp, sp * sp};   // row-major (:139-142)
  if (edge_all && tid == 0) {
    double* e = edge_all + (long)b * 17;
    for (int i = 0; i < 3; i++) e[i] = xg[xPOS + i];
    e[3] = cos(yaw / 2.0); e[4] = 0.0; e[5] = 0.0; e[6] = sin(yaw / 2.0);                   // from_euler(0,0,yaw), :150-165
    for (int j = 0; j < 3; j++)
      for (int i = 0; i < 3; i++) e[7 + i + 3 * j] = P[(dxPOS + i) + (long)(dxPOS + j) * ld];
    e[16] = P[(dxATT + 2) + (long)(dxATT + 2) * ld];
  }
  __syncthreads();   // (the edge reads P and x before they change)
  // rows: T = N P
  for (int j = tid; j < n; j += T) {
    double* c = P + (long)j * ld;
    const double a0 = c[dxATT], a1 = c[dxATT + 1], a2 = c[dxATT + 2];
    c[dxATT] = Na[0] * a0 + Na[1] * a1 + Na[2] * a2;
    c[dxATT + 1] = Na[3] * a0 + Na[4] * a1 + Na[5] * a2;
    c[dxATT + 2] = Na[6] * a0 + Na[7] * a1 + Na[8] * a2;
    c[dxPOS] = 0.0; c[dxPOS + 1] = 0.0; c[dxPOS + 2] = 0.0;
  }
  __syncthreads();
  // columns: P = T N^T
  for (int i = tid; i < n; i += T) {
    const double a0 = P[i + (long)dxATT * ld], a1 = P[i + (long)(dxATT + 1) * ld], a2 = P[i + (long)(dxATT + 2) * ld];
    P[i + (long)dxATT * ld] = a0 * Na[0] + a1 * Na[1] + a2 * Na[2];
    P[i + (long)(dxATT + 1) * ld] = a0 * Na[3] + a1 * Na[4] + a2 * Na[5];
    P[i + (long)(dxATT + 2) * ld] = a0 * Na[6] + a1 * Na[7] + a2 * Na[8];
    P[i + (long)dxPOS * ld] = 0.0; P[i + (long)(dxPOS + 1) * ld] = 0.0; P[i + (long)(dxPOS + 2) * ld] = 0.0;
  }
  __syncthreads();
  // N P N^T of a symmetric P is symmetric; make it so bit for bit (the update kernels rely on it): the attitude rows take
  // the values of the attitude columns
  for (int i = tid; i < n; i += T)
    for (int c = 0; c < 3; c++)
      if (i < dxATT || i > dxATT + c) P[(dxATT + c) + (long)i * ld] = P[i + (long)(dxATT + c) * ld];
  if (tid == 0) {
    const double cr = cos(roll / 2.0), ct = cos(pitch / 2.0), sr = sin(roll / 2.0), st = sin(pitch / 2.0);
    xg[xPOS] = 0.0; xg[xPOS + 1] = 0.0; xg[xPOS + 2] = 0.0;                                  // :65
    xg[xATT] = cr * ct; xg[xATT + 1] = sr * ct; xg[xATT + 2] = cr * st; xg[xATT + 3] = -sr * st;   // from_euler(roll,pitch,0)
  }
}

// fill every filter with the initial state (vi_ekf.cpp:70-81 / :134-144)
#ifndef VIEKF_INSTANCES_ONLY
__global__ void k_reset(StreamArgs a, const double* __restrict__ x0 /*17*/, const double* __restrict__ Pdiag /*n*/) {
  const int b = blockIdx.x;
  if (b >= a.B) return;
  double* xg = a.x + a.si(b) * a.nxs;
  double* P = a.P + a.si(b) * a.n * a.ld;
  for (int i = threadIdx.x; i < a.nxs; i += blockDim.x) xg[i] = (i < 17) ? x0[i] : 0.0;
  const long tot = (long)a.n * a.ld;
  for (long e = threadIdx.x; e < tot; e += blockDim.x) {
    const int i = (int)(e % a.ld), j = (int)(e / a.ld);
    P[e] = (i == j) ? Pdiag[i] : 0.0;
  }
  if (threadIdx.x == 0) { a.len[b] = 0; a.flags[b] = 0; }
}
#endif

}  // namespace viekf
