// viekf_kernels_resident.hpp -- "resident" kernel family: one workgroup per filter, the whole
// covariance lives in VGPRs for the complete step (propagate + M sequential feature updates), so
// P crosses HBM once per step instead of (M+1) times.
//
// Ownership (DESIGN.md "resident layout"): the 3N x 3N feature part of P is a grid of 3x3 blocks (I,J).  P is
// symmetric, so of every unordered pair {I,J} only ONE block is kept, on wrapped diagonals J = (I + d) mod N,
// d = 0..N/2.  Worker threads form a TR x TD grid (tr = tid % TR, td = tid / TR); thread (tr,td) keeps the RB blocks
// I = tr + TR*a of diagonal d = td in registers.  The 16 body columns P[:,0:16] live in LDS for the whole step (the
// body rows are their mirror).  A rank-2 sweep needs K for the block's rows and W for its columns, and the propagation
// Phi P Phi^T + Gd Qu Gd^T becomes a register-tiled contraction
//     P+[I,J] = X_I Y_J^T + Phi_ff[I] P[I,J] Phi_ff[J]^T,   X_I = [U_I | Phi_fb[I] | Gd_I Qu],
//                                                          Y_J = [Phi_fb[J] | V_J | Gd_J]   (K = 38)
// with U = (Phi P)[feat, body], V_J = Phi_ff[J] P[J, body] staged in LDS.
// lambda_feat is the same for every feature slot (vi_ekf.cpp:139-144), so the partial-update
// mask Lambda (vi_ekf.cpp:83,146) is ONE 3x3 constant for every feature/feature block.
#pragma once
#include <type_traits>

#include "viekf_kernels_stream.hpp"

namespace viekf {

#ifndef RES_INLINE
#define RES_INLINE __forceinline__
#endif
// Propagate in low-rank coupling form.  A_fb[I] = Afv_I E_v + Afg_I E_g and the bias rows of A_bb are zero
// (vi_ekf_dyn.cpp:55-71,121-128), so  Phi_fb[I] = D_I Psi  with a per-feature 3x9  D_I = [M1 | M3 | M2],
//   M1 = (Afv + dt/2 Aff Afv) dt,  M3 = Afv dt^2/2,  M2 = (Afg + dt/2 Aff Afg) dt,   Psi = [E_v ; A_bb[vel rows] ; E_g]  (9 x 16),
// and with  Pi = Psi P_bb Psi^T,  V_I = Phi_ff[I] P[I, body],  Ut_I = D_I Pi / 2 + V_I Psi^T  (3x9):
//   P+[I,J] = Phi_ff[I] P[I,J] Phi_ff[J]^T + Ut_I D_J^T + D_I Ut_J^T + Gs_I Gs_J^T (+ Qx),   Gs = Gd sqrt(Qu)
//   P+[I,body] = V_I Phi_bb^T + D_I Xi + Gs_I Gs_b^T,   Xi = Psi (P_bb Phi_bb^T)
// -- a K = 24 contraction over ONE record per row (the symmetric form needs no separate X / Y operands):
//   Z[row] = { (Ut[k], D[k]) k = 0..8 interleaved | Gs[0..5] | pad }      ZS doubles per row
constexpr int ZK = 9;    // rank of the feature/body coupling
constexpr int ZS = 26;   // row stride of Z: 6 ZS = 28 (mod 64 dwords), consecutive features land on distinct 16-byte bank groups

struct ResLds {  // LDS carve-up in doubles, shared by host (size) and device (offsets)
  int xs, Kt, Wt, Praw, lam, sm, fixadd, fixset, Z, phiff, Abb, Gb, Phibb, Mbb, Gdb, Pbb, T16, xdb, ctx, Pbc, PhibbT, Pd, PsiP, Pi, Xi, AvG, Lbc, mslot, mseq, mz, mR, img_len, total;
  __host__ __device__ ResLds(int N, int n, int nxs) {
    const int nf = 3 * N;
    int o = 0;
    auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
    xs = take(nxs);
    lam = take(n);
    sm = take(64);   // [0..15],[16..31] two measurement mailboxes {Hb(4) res(2) Sinv(4) verdict}, [40..41] fix mailboxes
                     // non-empty, [42] dt, [44..46] NaN-guard words (phase mod 3), [49] count of worker waves that have
                     // published the next raw columns (int), [50..51] gate verdicts (phase parity)
    fixadd = take(2 * (N > 0 ? N : 1)); fixset = take(2 * (N > 0 ? N : 1));
    // Z, Phi_ff and the two-lives region are contiguous: at store time all of it is dead and holds the P image
    Z = take(nf * ZS > 4 * n ? nf * ZS : 4 * n);   // propagate: the records; updates: second gain-row buffer; store: P image
    phiff = take(9 * (N > 0 ? N : 1));
    // one region, two lives: the propagate's body-sized scratch | the update loop's gain rows, raw columns and zeta blocks
    const int u0 = o;
    Abb = take(256); Gb = take(96); Phibb = take(256); PhibbT = take(256); Gdb = take(96); T16 = take(256);
    PsiP = take(ZK * 16); Pi = take(ZK * ZK); Xi = take(ZK * 16); AvG = take(18);
    const int uprop = o;
    o = u0;
    Kt = take(2 * n); Wt = take(2 * n);
    Praw = take(4 * n > 256 ? 4 * n : 256);   // two buffers [n][2]: raw column pairs of the next two measurements
    Pd = take(4 * (N > 0 ? N : 1));           // zeta-zeta 2x2 diagonal blocks, handed from the workers to the service lanes
    if (uprop > o) o = uprop;
    img_len = o - Z;
    Mbb = take(256); Pbb = take(256);
    xdb = take(16);
    ctx = take((int)((sizeof(BodyCtx) + 7) / 8));
    Pbc = take(nf * 16);
    Lbc = take(48);   // Lambda of (feature row q, body column k): [3][16]
    mslot = take(32); mseq = take(64); mz = take(128); mR = take(256);   // MCAP = 64 measurements per launch
    total = o;
  }
};

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

// ------------------------------------------------------------------------------------------------
// fused step: [propagate] + M feature updates with P resident in registers, WARP-SPECIALISED:
//   worker waves (NW x 64 threads) own P and run only the lean contraction / sweep code;
//   one service wave runs the scalar-heavy math (dynamics, gain, manifold correction, h_feat).
// Both sides execute the same barrier sequence; their register footprints never mix, which is
// what keeps the sweep loops spill-free (a spill costs a ~1 us scratch round trip per use).
// ------------------------------------------------------------------------------------------------
// Returns v unchanged but opaque to the optimiser: values derived from it cannot be hoisted out of a loop and kept
// (or spilled) across iterations; recomputing a few integer ops per use is far cheaper than a scratch round trip.
// Orders LDS accesses of DIFFERENT lanes of one wave: a store under a lane predicate followed by loads on other lanes.  The
// hardware executes one wave's LDS instructions in order, but lanes are separate threads to the compiler, which otherwise
// hoists the other lanes' loads above the predicated store.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// Many blocks per thread: fences the operand loads of one group of blocks from the next (a compiler-level memory barrier plus
// a scheduling barrier) -- otherwise every block's (mutually independent) LDS reads are hoisted to the top and their results
// held live together, which does not fit the register file next to the blocks themselves.
// 16-byte LDS read as ONE vector load: through HIP's double2 struct the two halves are often split and re-paired as
// ds_read2_b64 (8 LDS cycles per wave instruction, 32-bank mapping) instead of ds_read_b128 (4 cycles, 64 banks)
typedef double v2f64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 lds_ld2(const double* p) {
  const v2f64 v = *reinterpret_cast<const v2f64*>(p);
  return make_double2(v.x, v.y);
}
template <bool ON>
__device__ __forceinline__ void group_fence() {
  if (ON) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ double uniform_f64(double v) {   // force a wave-uniform double into SGPRs
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// Diagnostic build only (-DVIEKF_STAMPS): s_memtime stamps of block 0 into the (otherwise unused) workspace.
#ifdef VIEKF_STAMPS
#define RES_STAMP(S_, who, idx)                                                                  \
  do {                                                                                           \
    if ((S_).b == 0 && (who)) {                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      reinterpret_cast<unsigned long long*>((S_).stamps)[(idx)] = __builtin_amdgcn_s_memtime();  \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  } while (0)
#else
#define RES_STAMP(S_, who, idx) do {} while (0)
#endif

constexpr int MCAP = 64;  // measurements per launch (the host chunks longer lists)
typedef __attribute__((address_space(3))) volatile int lds_vint_t;

struct ResShared {  // resolved LDS pointers + launch constants shared by both roles
  double *xs, *Kt, *Wt, *Praw, *lam, *sm, *fixadd, *fixset, *Z, *phiff, *Abb, *Gb, *Phibb, *Mbb, *Gdb, *Pbb, *T16,
      *xdb, *Pbc, *PhibbT, *Pd, *PsiP, *Pi, *Xi, *AvG, *Lbc, *mz, *mR;
  int* mslot;   // [MCAP] slot, or -1 for a measurement that is not run
  int2* mseq;   // [MCAP] {index of the next measurement that runs (or M), its slot (or -1)}: one LDS read per iteration
  BodyCtx* ctx;
  int N, n, nf, len, M, mstride, do_prop, b, dbg, kp, B, img_len;   // kp: propagates per launch (viekf_batch_step_n)
  double* stamps;
};

// first m' >= from whose update will actually run (mslot >= 0), else M
__device__ __forceinline__ int res_next_valid(const ResShared& S, int from) {
  int m = from;
  while (m < S.M && S.mslot[m] < 0) m++;
  return m;
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

// Store of P, cooperative part.  The workers scatter their 3x3 blocks (and the mirror images) into an LDS image of a chunk of
// feature columns -- the Z region, free after the propagate; [column][n rows] -- and the whole workgroup streams the chunk
// out with lanes along the rows: every wave instruction writes up to 512 contiguous bytes instead of 64 different cache
// lines (the direct 8-byte block stores were bound by the texture path's one line per clock: 25 k clk per step).
// Rows 0..15 of a feature column are the mirror of the LDS-resident body columns.
struct StoreChunks {
  int fc, nchunks;   // features per chunk, number of chunks
  __device__ StoreChunks(int N, int n, int img_len) {
    fc = max(1, min(N, img_len / (3 * n)));
    nchunks = (N + fc - 1) / fc;
  }
};
template <int T>
__device__ __forceinline__ void res_store_chunk(const StreamArgs& a, const ResShared& S, int f0, int f1, int tid) {
  const int n = S.n, ld = a.ld;
  double* P = a.P_out + (long)S.b * n * ld;
  const double* img = S.Z;
  const int ncol = 3 * (f1 - f0), lane = tid & 63, w = tid >> 6;
  constexpr int NWV = T / 64;
  if ((n & 1) == 0) {   // even n: row pairs are 16-byte aligned in the image, in Pbc and in P (ld is even)
#pragma unroll 4
    for (int c = w; c < ncol; c += NWV) {
      const int j = 16 + 3 * f0 + c;
      double2 v[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = 2 * (lane + 64 * u);
        const double* src = (i < 16) ? (S.Pbc + (j - 16) * 16 + i) : (img + c * n + min(i, n - 2));
        v[u] = lds_ld2(src);
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int i = 2 * (lane + 64 * u);
        if (i < n) *reinterpret_cast<double2*>(P + i + (long)j * ld) = v[u];
      }
    }
  } else {
#pragma unroll 2
    for (int c = w; c < ncol; c += NWV) {
      const int j = 16 + 3 * f0 + c;
      double v[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int i = lane + 64 * u;
        v[u] = (i < 16) ? S.Pbc[(j - 16) * 16 + i] : img[c * n + min(i, n - 1)];
      }
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int i = lane + 64 * u;
        if (i < n) P[i + (long)j * ld] = v[u];
      }
    }
  }
}

// Body columns of an update, in LDS: item = (feature g, k pair j) = the 3 rows of one feature x 2 body columns (6 elements),
// items [first, last) strided over `nthreads` callers; the mask Lambda of a (feature row, body column) pair comes from the
// table Lbc [3][16] (prologue).  The owner of the item of the feature measured two phases from now also adds its rows -- as
// they stand after this sweep -- to the raw column buffer `rawdst` (see res_worker).
__device__ __forceinline__ void res_body_items(const ResShared& S, const double* kP, bool run, int id, int nthreads,
                                               int first, int last, int slot2, double* rawdst) {
  const double* wP = kP + 2 * S.n;
  double* Pbc = S.Pbc;
#pragma unroll 1
  for (int item = first + id; item < last; item += nthreads) {
    const int g = item >> 3, j2 = (item & 7) * 2;
    double2 cpv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) cpv[q] = lds_ld2(Pbc + (3 * g + q) * 16 + j2);
    if (run) {
      const double2 cw0 = lds_ld2(wP + 2 * j2);
      const double2 cw1 = lds_ld2(wP + 2 * j2 + 2);
      double2 cki[3], cl[3];
#pragma unroll
      for (int q = 0; q < 3; q++) { cki[q] = lds_ld2(kP + 2 * (16 + 3 * g + q)); cl[q] = lds_ld2(S.Lbc + 16 * q + j2); }
#pragma unroll
      for (int q = 0; q < 3; q++) {
        cpv[q].x = fma(-cl[q].x, fma(cki[q].y, cw0.y, cki[q].x * cw0.x), cpv[q].x);
        cpv[q].y = fma(-cl[q].y, fma(cki[q].y, cw1.y, cki[q].x * cw1.x), cpv[q].y);
        *reinterpret_cast<double2*>(Pbc + (3 * g + q) * 16 + j2) = cpv[q];
      }
    }
    // rows j0, j0+1 of the feature measured two phases from now, as they stand after this phase's sweep
    if (g == slot2) {
      double* st = rawdst + 2 * j2;
      *reinterpret_cast<double2*>(st) = make_double2(cpv[0].x, cpv[1].x);
      *reinterpret_cast<double2*>(st + 2) = make_double2(cpv[0].y, cpv[1].y);
    }
  }
}
// how many of the 8 N body-column items of an update the service wave sweeps (after its chain), by worker-wave count
template <int NWV>
// (measured at N = 50, two workgroups per CU: 0 / 64 / 128 / 192 / 256 / 320 / 400 of the 400 items -> 0.456 / 0.428 / 0.431 /
//  0.424 / 0.415 / 0.431 / 0.457 ms per step)
__device__ __forceinline__ int res_service_items(int N) { return (NWV == 3) ? ((5 * N + 7) & ~7) : 0; }

template <int RB, int TW, bool MP, int T = TW + 64>
__device__ __forceinline__ void res_worker(const StreamArgs& a, const ResShared& S, int tid) {
  const int N = S.N, n = S.n, ld = a.ld, nf = S.nf, len = S.len;
  double* P = a.P + (long)S.b * n * ld;
  // SYMMETRIC ownership: of each unordered pair of feature blocks {I,J} only one is kept, on wrapped diagonals
  //   J = (I + d) mod N,  d = 0 .. N/2   (for even N the diagonal d = N/2 would hold every pair twice: only its rows
  //   I < N/2 are owned).  The N (N + 1) / 2 owned blocks are numbered  idx = d N + I  and dealt round-robin: thread t keeps
  //   idx = t + TW a, a < RB  -- any thread count, RB = ceil(N (N + 1) / 2 / TW) blocks per thread, and the lanes of a wave
  //   hold consecutive rows I of (mostly) one diagonal.  Slot a = 0 of the threads t < N is the diagonal d = 0.
  const int tid_ = tid;
  constexpr int NWV = TW / 64;
  const DevParams& prm = *a.dp;
  double* Pbc = S.Pbc;   // [nf][16]  P[16+row][k]: the body columns of P live in LDS for the whole step
  double* Pbb = S.Pbb;   // [16][16]  row-major P_bb
  // (P[body rows, feature cols] is NOT kept: P is symmetric up to rounding, the mirror is written at store time)
  const int nown = N * (N + 1) / 2;
  const float rcpN = 1.0f / (float)N;
  auto blk = [&](int t, int ia, int& I, int& J) -> bool {   // block ia of thread t; false = not owned
    int idx = t + TW * ia;
    const bool v = idx < nown;
    idx = min(idx, nown - 1);             // clamped: every LDS / global read stays in range, results never stored
    const int d = (int)(((float)idx + 0.5f) * rcpN);   // idx / N (exact: idx < 2^20, the margin 0.5 / N dwarfs the rounding)
    I = idx - d * N;
    J = I + d;
    if (J >= N) J -= N;
    return v;
  };
  const bool own_diag = tid_ < N;   // slot 0 of this thread is the diagonal block (I, I), I = tid
  static_assert(TW >= 64, "the diagonal d = 0 must sit in slot 0: TW >= N");

  double pb[RB][9];   // pb[a][r*3+s] = P[16+3I+r][16+3J+s]
  {
    const int tq = opaque(tid_);
#pragma unroll
    for (int ia = 0; ia < RB; ia++) {
      int I, J;
      blk(tq, ia, I, J);
      const double* pu = P + ((16 + 3 * I) + (long)(16 + 3 * J) * ld);
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int r = 0; r < 3; r++) pb[ia][r * 3 + s] = pu[r + (long)s * ld];
    }
    // body columns -> LDS (coalesced along rows)
    for (int e = tid; e < nf * 16; e += TW) {
      const int k = e / nf, row = e - k * nf;
      Pbc[row * 16 + k] = P[(16 + row) + (long)k * ld];
    }
    // (the body block is kept EXACTLY symmetric, like every other part of P here -- see sym_diag below: both copies of a pair
    //  are loaded from the upper triangle)
    for (int e = tid; e < 256; e += TW) {
      const int r = e & 15, c = e >> 4;
      Pbb[r * 16 + c] = P[min(r, c) + (long)max(r, c) * ld];
    }
  }

  // P is kept EXACTLY symmetric.  Off-diagonal feature blocks and the feature/body strips are symmetric by ownership (one
  // copy, mirrored at store time); the diagonal blocks (and the body block, in LDS) hold both triangles, and their lower one
  // is overwritten with the upper one after everything that changes them.  This is not cosmetic: the rank-2 form of the
  // update,  P -= Lambda o (K W^T)  with W from the COLUMNS of P, equals the reference's Joseph form (vi_ekf_meas.cpp:256-257)
  // only for symmetric P; on an antisymmetric part A it is  A_zz' = A_zz + K (Hb A_zz Hb^T) K^T  -- growth per update where
  // the Joseph form contracts -- and rounding-level asymmetry reaches 1e-7 within 3 s of flight (tests/test_sim_end_to_end.py).
  auto sym_diag = [&]() {
    if (own_diag) { pb[0][3] = pb[0][1]; pb[0][6] = pb[0][2]; pb[0][7] = pb[0][5]; }
  };
  sym_diag();
  // Lambda for feature/feature blocks: one 3x3 constant (lambda_feat identical for all slots), kept in SGPRs
  double Lff[9];
  {
    const double lf[3] = {a.lambda[16], a.lambda[17], a.lambda[18]};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) Lff[r * 3 + s] = uniform_f64(prm.use_partial_update ? (lf[s] + lf[r] - lf[r] * lf[s]) : 1.0);
  }
  const bool partial = prm.use_partial_update != 0;
  int par = 0;  // fix_depth mailbox parity (mirrors the service wave)
  RES_STAMP(S, tid == 0, 64);
  __syncthreads();  // B0
  RES_STAMP(S, tid == 0, 65);

  // K propagates per launch (viekf_batch_step_n: the IMU samples between two camera frames) keep P on chip in between:
  // bit for bit what K launches would give, without their HBM round trips
  // (MP = false -- one propagate, every launch but viekf_batch_step_n's -- is a separate instance: the loop costs the
  //  single-propagate kernel 2 % in registers kept alive across it)
  const int nkp = MP ? S.kp : 1;
  if (S.do_prop)
   for (int kp = 0; kp < nkp; kp++) {
    // (the thread index is laundered per propagate: otherwise everything derived from it is hoisted out of this loop and
    //  kept alive across it -- spills)
    const int tk = MP ? opaque(tid) : tid;
    const double* Z = S.Z; double* phiff = S.phiff;
    res_prop_setup<TW>(a, S, tk);
    RES_STAMP(S, tid == 0, 66);
    __syncthreads();  // B3p
    RES_STAMP(S, tid == 0, 67);

    // ---- local 3x3 transforms  Phi_ff[I] (P[I,J] Phi_ff[J]^T) (+ Qx on the diagonal), in place with 3 temporaries:
    //      first each row times Phi_ff[J]^T, then each column times Phi_ff[I]  (keeps the register peak low)
#pragma unroll
    for (int ia = 0; ia < RB; ia++) {
      const int tq = opaque(tid_);
      int I, J;
      const bool v = blk(tq, ia, I, J);
      const double* fj = phiff + 9 * J;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const double p0 = pb[ia][r * 3 + 0], p1 = pb[ia][r * 3 + 1], p2 = pb[ia][r * 3 + 2];
#pragma unroll
        for (int s = 0; s < 3; s++) pb[ia][r * 3 + s] = p0 * fj[s * 3 + 0] + p1 * fj[s * 3 + 1] + p2 * fj[s * 3 + 2];
      }
      const double* fi = phiff + 9 * I;
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const double p0 = pb[ia][0 * 3 + s], p1 = pb[ia][1 * 3 + s], p2 = pb[ia][2 * 3 + s];
#pragma unroll
        for (int r = 0; r < 3; r++) pb[ia][r * 3 + s] = fi[r * 3 + 0] * p0 + fi[r * 3 + 1] * p1 + fi[r * 3 + 2] * p2;
      }
      if (v && I == J) {
        pb[ia][0] += a.Qx[16 + 3 * I + 0];
        pb[ia][4] += a.Qx[16 + 3 * I + 1];
        pb[ia][8] += a.Qx[16 + 3 * I + 2];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (RB > 4) {   // pin the block's new values here: their arithmetic is otherwise sunk towards its first use, with the
                      // (twice as many) operands held live instead
#pragma unroll
        for (int e = 0; e < 9; e++) asm volatile("" : "+v"(pb[ia][e]));
      }
      group_fence<(RB > 4)>();
    }
    RES_STAMP(S, tid == 0, 68);
    // ---- register-tiled contraction  P[I,J] += Ut_I D_J^T + D_I Ut_J^T + Gs_I Gs_J^T  (K = 24): one 16-byte read per row
    //      and k gives the pair (Ut[k], D[k]) -- or two adjacent columns of Gs
    // (many blocks per thread: the Z-row offsets of a block's I and J are packed into one register per block ahead of the
    //  loop -- re-deriving them from the thread index cost as many instructions per k as the arithmetic)
    int zoff[RB > 4 ? RB : 1];
    if (RB > 4) {
      const int tq = opaque(tid_);
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        int I, J;
        blk(tq, ia, I, J);
        zoff[ia] = (3 * I * ZS) | ((3 * J * ZS) << 16);
      }
    }
    auto contract = [&](int k, auto crossed) {
      constexpr bool CROSS = decltype(crossed)::value;
      const int tq = opaque(tid_);
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        const double *zi, *zj;
        if (RB > 4) {
          zi = Z + (zoff[ia] & 0xffff) + 2 * k;
          zj = Z + (zoff[ia] >> 16) + 2 * k;
        } else {
          int I, J;
          blk(tq, ia, I, J);
          zi = Z + (3 * I) * ZS + 2 * k;
          zj = Z + (3 * J) * ZS + 2 * k;
        }
        double2 xv[3], yv[3];
#pragma unroll
        for (int r = 0; r < 3; r++) xv[r] = lds_ld2(zi + r * ZS);
#pragma unroll
        for (int s = 0; s < 3; s++) yv[s] = lds_ld2(zj + s * ZS);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int s = 0; s < 3; s++) {
            double acc = pb[ia][r * 3 + s];
            acc = fma(xv[r].x, CROSS ? yv[s].y : yv[s].x, acc);   // Ut_I . D_J + D_I . Ut_J   |   Gs_I . Gs_J
            acc = fma(xv[r].y, CROSS ? yv[s].x : yv[s].y, acc);
            pb[ia][r * 3 + s] = acc;
          }
        if (ia & 1) group_fence<(RB > 4)>();   // (many blocks per thread: the operands of two in flight)
      }
    };
#pragma unroll 1
    for (int k = 0; k < ZK; k++) contract(k, std::true_type{});
#pragma unroll 1
    for (int k = ZK; k < ZK + 3; k++) contract(k, std::false_type{});
    sym_diag();
    RES_STAMP(S, tid == 0, 69);
    if (MP) res_prop_body<TW>(a, S, tk);   // (single propagate: the service wave does this meanwhile, it would only wait)
    par ^= 1;   // the service wave posted propagate's fix_depth edits into mailbox 0
    RES_STAMP(S, tid == 0, 70);
    __syncthreads();  // B4p
    for (int e = tk; e < 256; e += TW) { const int r = e >> 4, c = e & 15; Pbb[e] = S.Mbb[min(r, c) * 16 + max(r, c)]; }
    if (MP && kp + 1 < nkp && own_diag && tid_ < len && S.sm[40 + (par ^ 1)] != 0.0) {   // this propagate's fix_depth edits of P(rho,rho), before the next
      const int mb = par ^ 1, I = tid_;
      const double ad = S.fixadd[mb * N + I], st = S.fixset[mb * N + I];
      if (ad != 0.0) { pb[0][8] += ad; S.fixadd[mb * N + I] = 0.0; }
      if (st != 0.0) { pb[0][8] = prm.P0_feat[2]; S.fixset[mb * N + I] = 0.0; }
    }
   }

  // block indices of this thread, computed once (symmetric ownership left enough registers to keep them)
  int Ib[RB], Jb[RB];
  bool vb[RB];
#pragma unroll
  for (int ia = 0; ia < RB; ia++) vb[ia] = blk(tid_, ia, Ib[ia], Jb[ia]);
  const double p0rr = uniform_f64(prm.P0_feat[2]);   // (read here: a global load inside the update loop would put vmcnt waits there)
  // applies the pending fix_depth covariance edits of mailbox `mb` to the owned diagonal blocks (diagonal d = 0)
  auto apply_fixes = [&](int mb, double pending) {
    if (pending == 0.0) return;   // nothing posted (the common case); the flag word was read ahead of the barrier
    if (own_diag && tid_ < len) {
      const int I = tid_;
      const double ad = S.fixadd[mb * N + I], st = S.fixset[mb * N + I];
      if (ad != 0.0) { pb[0][8] += ad; S.fixadd[mb * N + I] = 0.0; }
      if (st != 0.0) { pb[0][8] = p0rr; S.fixset[mb * N + I] = 0.0; }
    }
  };
  // Publishes the feature rows of the two zeta columns of feature `slot` (raw P[16.., j0], P[16.., j0+1]) into Praw for the
  // service wave, which turns them into the gain rows: the pair {I, slot} is held either as block (I, slot) (its columns
  // 0,1) or, transposed, as block (slot, J = I) (its rows 0,1).
  auto extract_cols = [&](int slot, double* Pw) {
#pragma unroll
    for (int ia = 0; ia < RB; ia++) {
      const int I = Ib[ia], J = Jb[ia];
      const bool asrow = J == slot;             // block (I, slot): its columns 0,1 are the wanted column pair
      const bool ascol = !asrow && I == slot;   // block (slot, J): its rows 0,1, transposed
      if (vb[ia] && (asrow || ascol)) {
        const int base = 16 + 3 * (asrow ? I : J);
        // plain selects on compile-time register indices (a data-dependent index would push the block to scratch; so did
        // select-free 8-byte stores of the two orientations, measured with 7 blocks per thread)
        const double a0 = pb[ia][0], a1 = asrow ? pb[ia][1] : pb[ia][3];
        const double b0 = asrow ? pb[ia][3] : pb[ia][1], b1 = pb[ia][4];
        const double c0 = asrow ? pb[ia][6] : pb[ia][2], c1 = asrow ? pb[ia][7] : pb[ia][5];
        *reinterpret_cast<double2*>(Pw + 2 * (base + 0)) = make_double2(a0, a1);
        *reinterpret_cast<double2*>(Pw + 2 * (base + 1)) = make_double2(b0, b1);
        *reinterpret_cast<double2*>(Pw + 2 * (base + 2)) = make_double2(c0, c1);
      }
    }
  };

  // ---------------- M sequential feature updates: covariance side ----------------
  int m = res_next_valid(S, 0);
  // hand the zeta-zeta 2x2 of every diagonal block to the service lanes (they keep it current from here on)
  if (own_diag) {
    *reinterpret_cast<double2*>(S.Pd + 4 * tid_) = make_double2(pb[0][0], pb[0][1]);
    *reinterpret_cast<double2*>(S.Pd + 4 * tid_ + 2) = make_double2(pb[0][3], pb[0][4]);
  }
  // Raw column pairs P[:, j0:j0+2] of a measured feature go to the service wave through two buffers [n][2] (Praw): the
  // columns of measurement m+2 are published in phase m, as they stand after the sweep of measurement m; the service wave
  // applies the one intervening update (m+1) to them itself when it forms the gain rows of m+2 -- so nothing it needs is
  // produced inside its own phase: no hand-shake, no polling, and the publishing sits off every critical path.
  int2 sq = S.mseq[min(m, MCAP - 1)];
  if (m < S.M) {
    apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
    const int s0 = S.mslot[m];
    extract_cols(s0, S.Praw);                                   // first measurement: buffer 0
    if (sq.y >= 0) extract_cols(sq.y, S.Praw + 2 * n);          // second one: buffer 1
    // body rows of those columns (P[k][j0+c] = P[j0+c][k]): 8 threads each, two body columns per thread
    const int e = opaque(tid);
    if (e < 16) {
      const int sf = (e < 8) ? s0 : sq.y, ijj = (e & 7) * 2;
      if (sf >= 0) {
        const double2 q0 = lds_ld2(Pbc + (3 * sf) * 16 + ijj);
        const double2 q1 = lds_ld2(Pbc + (3 * sf + 1) * 16 + ijj);
        double* d = S.Praw + ((e < 8) ? 0 : 2 * n) + 2 * ijj;
        *reinterpret_cast<double2*>(d) = make_double2(q0.x, q1.x);
        *reinterpret_cast<double2*>(d + 2) = make_double2(q0.y, q1.y);
      }
    }
  }
  RES_STAMP(S, tid == 0, 71);
  __syncthreads();  // Bp : Pd and the first measurement's raw columns are published
  __syncthreads();  // B1 : the service formed the first measurement's gain rows Kt / Wt, verdict and NaN word
  int it_ = 0;
  int cnt = 0;
  // ONE barrier per update, and no other hand-shake.  Inside a phase the worker waves (1) sweep their blocks with the gains
  // of measurement m, (2) publish the raw feature rows of measurement m+2's columns from the swept registers, (3) sweep the
  // LDS-resident body columns (the owner of the body-column item of that feature adds its body rows to the same buffer).
  // The service wave runs the state chain of measurement m meanwhile and forms the gain rows of measurement m+1 from the
  // columns published one phase earlier.
  while (m < S.M) {
    const int mnext = sq.x;
    // gain rows {K [n][2], W [n][2]} are double-buffered: the service wave forms those of measurement m+1 while step (3) of
    // this phase still reads those of measurement m.  The second buffer is the Z region (free outside the propagate).
    const double* kP = (cnt & 1) ? S.Z : S.Kt;
    const double* wP = kP + 2 * n;
    __builtin_amdgcn_s_setprio(1);
    const double fixpending = S.sm[40 + (par ^ 1)];   // posted before the barrier by the service wave
    const double gflag = S.sm[50 + (cnt & 1)];          // gate verdict of this measurement (service, previous phase)
    const double nanw = S.sm[44 + cnt % 3] + S.sm[52 + cnt % 3];   // (the second word: a second service wave's rows)
    sq = S.mseq[min(mnext, MCAP - 1)];                 // next iteration's table entry (static data)
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 1);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 0);
    const int it = tid;
    // ---- (1) feature/feature blocks (registers).  The operand rows of GB blocks are in flight together: all of them with
    //      few blocks per thread; two at a time with many, where holding every block's rows would not fit the register file
    constexpr int GB = (RB <= 4) ? RB : 2;
    const bool gated = gflag != 0.0;
    const bool run = !gated && nanw == 0.0 && !(S.dbg & 1);   // not gated, no NaN guard
    bool fixed = false;
#pragma unroll
    for (int g0 = 0; g0 < RB; g0 += GB) {
      double2 kI[GB][3], wJ[GB][3];
#pragma unroll
      for (int ig = 0; ig < GB; ig++) {
        const int ia = (g0 + ig < RB) ? g0 + ig : RB - 1;
#pragma unroll
        for (int r = 0; r < 3; r++) kI[ig][r] = lds_ld2(kP + 2 * (16 + 3 * Ib[ia] + r));
#pragma unroll
        for (int s = 0; s < 3; s++) wJ[ig][s] = lds_ld2(wP + 2 * (16 + 3 * Jb[ia] + s));
      }
      if (!fixed) { apply_fixes(par ^ 1, fixpending); fixed = true; }
      if (run) {
#pragma unroll
        for (int ig = 0; ig < GB; ig++) {
          if (g0 + ig >= RB) continue;
          const int ia = g0 + ig;
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
              const double t = fma(kI[ig][r].y, wJ[ig][s].y, kI[ig][r].x * wJ[ig][s].x);
              pb[ia][r * 3 + s] = fma(-Lff[r * 3 + s], t, pb[ia][r * 3 + s]);
            }
        }
      }
      group_fence<(RB > GB)>();
    }
    sym_diag();
    RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 0);
    // ---- (2) the raw feature rows of the measurement after next (a fix_depth edit touches P(rho,rho) only, never these
    //      columns), into the buffer the service wave is not reading in this phase
    double* rawdst = S.Praw + (cnt & 1) * 2 * n;
    if (sq.y >= 0 && !(S.dbg & 4)) extract_cols(sq.y, rawdst);
    __builtin_amdgcn_s_setprio(0);
    RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (3) body columns, in LDS (res_body_items); with few worker waves the tail of the items is the service wave's: it
    //      would only wait at the barrier, the workers are the longer side there
    res_body_items(S, kP, run, it, TW, 0, 8 * N - res_service_items<NWV>(N), sq.y, rawdst);
    if (run) {   // body block: 2 adjacent elements per thread, on the top 128 threads.  Element (r, c) and its mirror (c, r) are
                 // owned by different threads; both form  p - L (K_lo . W_hi), lo = min(r, c), hi = max(r, c)  from their own
                 // (equal) copies, so the block stays exactly symmetric without any exchange.
      const int ib = it - (TW - 128);
      if (ib >= 0) {
        const int br = ib >> 3, bc2 = (ib & 7) * 2;
        double2 bpv = *reinterpret_cast<double2*>(Pbb + br * 16 + bc2);
        const double blr = S.lam[br];
        const double2 blc = lds_ld2(S.lam + bc2);
        const double2 kr = lds_ld2(kP + 2 * br), wr = lds_ld2(wP + 2 * br);
        const double2 k0 = lds_ld2(kP + 2 * bc2), w0 = lds_ld2(wP + 2 * bc2);
        const double2 k1 = lds_ld2(kP + 2 * bc2 + 2), w1 = lds_ld2(wP + 2 * bc2 + 2);
        const double L0 = partial ? (blc.x + blr - blr * blc.x) : 1.0, L1 = partial ? (blc.y + blr - blr * blc.y) : 1.0;
        const bool up0 = br <= bc2, up1 = br <= bc2 + 1;
        const double2 ka = up0 ? kr : k0, wa = up0 ? w0 : wr;      // (K_lo, W_hi) of element (br, bc2)
        const double2 kb = up1 ? kr : k1, wb = up1 ? w1 : wr;      // ... of element (br, bc2 + 1)
        bpv.x = fma(-L0, fma(ka.y, wa.y, ka.x * wa.x), bpv.x);
        bpv.y = fma(-L1, fma(kb.y, wb.y, kb.x * wb.x), bpv.y);
        *reinterpret_cast<double2*>(Pbb + br * 16 + bc2) = bpv;
      }
    }
    par ^= 1;
    cnt++;
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 2);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 1);
    __syncthreads();  // B1 (the only barrier of an update): sweeps finished; next gain rows, verdict and NaN word complete
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 3);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 3);
    it_++;
    m = mnext;
  }
  apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
  RES_STAMP(S, tid == 0, 72);
  __syncthreads();  // B5 : every sweep of the LDS-resident body columns is finished

  // ---------------- store ----------------
  // (indices re-derived from opaque copies: otherwise the load addresses are kept alive -- spilled -- all kernel long)
  {
    P = a.P_out + (long)S.b * n * ld;   // in place, or the next slot of the history ring
    for (int e = opaque(tid); e < nf * 16; e += TW) {     // body columns, coalesced along rows
      const int k = e / nf, row = e - k * nf;
      P[(16 + row) + (long)k * ld] = Pbc[row * 16 + k];
    }
    for (int e = opaque(tid); e < 256; e += TW) P[(e >> 4) + (long)(e & 15) * ld] = Pbb[e];
    const StoreChunks sc(N, n, S.img_len);
    double* img = S.Z;
    const int gtid = threadIdx.x;
    RES_STAMP(S, tid == 0, 224);
    for (int ch = 0; ch < sc.nchunks; ch++) {
      const int f0 = ch * sc.fc, f1 = min(N, f0 + sc.fc);
      const int tq = opaque(tid_);
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        int I, J;
        if (blk(tq, ia, I, J)) {
          if (J >= f0 && J < f1) {                          // block (I,J): columns of feature J
            double* d = img + (3 * (J - f0)) * n + 16 + 3 * I;
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
              for (int r = 0; r < 3; r++) d[s * n + r] = pb[ia][r * 3 + s];
          }
          if (I != J && I >= f0 && I < f1) {                // its mirror (J,I): columns of feature I
            double* d = img + (3 * (I - f0)) * n + 16 + 3 * J;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int s = 0; s < 3; s++) d[r * n + s] = pb[ia][r * 3 + s];
          }
        }
      }
      RES_STAMP(S, tid == 0 && ch < 3, 225 + 4 * ch);
      __syncthreads();   // S1: the chunk image is complete
      RES_STAMP(S, tid == 0 && ch < 3, 226 + 4 * ch);
      res_store_chunk<T>(a, S, f0, f1, gtid);
      RES_STAMP(S, tid == 0 && ch < 3, 227 + 4 * ch);
      __syncthreads();   // S2: the image may be overwritten
      RES_STAMP(S, tid == 0 && ch < 3, 228 + 4 * ch);
    }
  }
  RES_STAMP(S, tid == 0, 73);
}

// ---- the service wave: everything that is not a sweep over P --------------------------------------
// ROLE 0: the one service wave of a workgroup (N + 14 <= 64 lanes: a lane per feature and 14 body lanes).  More features than
// that split the roles over TWO service waves: ROLE 1 = the feature lanes (and everything a single service wave does besides:
// dynamics, prediction, result codes, the state store), ROLE 2 = the 14 body lanes on a wave of their own.  The body wave
// receives each measurement's {Hb, residual, S^-1, gate} from the feature wave through an LDS mailbox (polled: the feature
// wave never waits for the body wave other than at the barriers); each wave writes the gain rows and the NaN-guard word of
// its own rows.
template <int T, bool MP, int ROLE = 0>
__device__ __forceinline__ void res_service(const StreamArgs& a, const ResShared& S, int lane, int nww,
                                            const double* __restrict__ u_all, const double* __restrict__ dt_all,
                                            int* __restrict__ result_all) {
  const int N = S.N, n = S.n, len = S.len, M = S.M;
  const DevParams& prm = *a.dp;
  double* xs = S.xs;
  double* sm = S.sm;
  unsigned flag = 0;
  const bool partial = prm.use_partial_update != 0;
  constexpr bool PRIMARY = ROLE != 2;
  int par = 0;
  // the per-update critical path runs on this wave: let it win the issue arbitration against its SIMD-mate worker wave
  __builtin_amdgcn_s_setprio(3);
  RES_STAMP(S, lane == 0, 0);
  // The dynamics of the propagate need only the state (in LDS since the prologue): they run BEFORE B0, while the worker
  // waves are still loading P from HBM, instead of holding every worker up afterwards.
  double dt = sm[42];
  // dynamics of propagate kp (of S.kp): body Jacobian on lane 0 (+ A_v G_b for the workers' expansion of the feature rows),
  // then one feature per lane
  auto dyn_body = [&](int kp) {
    for (int i = lane; i < 256; i += 64) S.Abb[i] = 0.0;
    for (int i = lane; i < 96; i += 64) S.Gb[i] = 0.0;
    if (lane < 16) S.xdb[lane] = 0.0;
    if (lane == 0) res_body_phase(xs, u_all + ((long)kp * S.B + S.b) * 6, a.dp, S.ctx, S.xdb, S.Abb, S.Gb);
    RES_STAMP(S, lane == 0 && kp == 0, 2);
    // (same wave: the LDS accesses of lane 0 above are complete before the other lanes read ctx / A_bb / G_b)
    wave_lds_sync();
    if (lane >= 64 - 18) {   // A_v G_b (3 x 6), one entry per lane, on lanes that carry no feature
      const int e = lane - (64 - 18), j = e / 6, k = e - 6 * j;
      double sv = 0.0;
#pragma unroll 4
      for (int c = 0; c < 16; c++) sv += S.Abb[(dxVEL + j) * 16 + c] * S.Gb[c * 6 + k];
      S.AvG[e] = sv;
    }
  };
  auto dyn_feat = [&](double dtk) {
    for (int f = lane; f < N; f += 64) res_feature_phase(f, len, dtk, xs, S.ctx, S.Z, S.phiff);
    RES_STAMP(S, lane == 0, 4);
  };
  if (S.do_prop && PRIMARY) { dyn_body(0); dyn_feat(dt); }
  __syncthreads();  // B0
  RES_STAMP(S, lane == 0, 1);

  const int nkp = MP ? S.kp : 1;
  if (S.do_prop)
   for (int kp = 0; kp < nkp; kp++) {
    __syncthreads();  // B1p
    RES_STAMP(S, lane == 0, 3);
    __syncthreads();  // B2p
    __syncthreads();  // B2q
    RES_STAMP(S, lane == 0, 5);
    if (PRIMARY && lane == 63) {   // body state step (every feature lane has consumed the old body state through ctx)
      double dxb[16], xo[17];
#pragma unroll
      for (int i = 0; i < 16; i++) dxb[i] = S.xdb[i] * dt;
      body_boxplus_fast(xs, dxb, xo);
#pragma unroll
      for (int i = 0; i < 17; i++) xs[i] = xo[i];
    }
    if (PRIMARY && lane == 0) sm[40 + par] = 0.0;
    for (int f = lane; PRIMARY && f < len; f += 64)   // fix_depth (vi_ekf.cpp:311): state here, covariance through the mailbox
      res_fix_depth(xs + xZ + 5 * f, a.dp, &S.fixadd[par * N + f], &S.fixset[par * N + f], &sm[40 + par], &flag);
    par ^= 1;
    RES_STAMP(S, lane == 0, 6);
    __syncthreads();  // B3p
    RES_STAMP(S, lane == 0, 7);
    // Several propagates per launch: the body part of the NEXT one's dynamics runs here, under the workers' contraction (this
    // wave would only wait for B4p).  The body state it needs is final (body step above) and nothing it writes (A_bb, G_b,
    // A_v G_b, xdot, ctx) is read again before the next B1p; the feature part writes Z rows and Phi_ff, which the workers
    // are still reading: it runs after B4p.
    double dt_next = 0.0;
    if (PRIMARY && MP && kp + 1 < nkp) {
      dt_next = dt_all[(long)(kp + 1) * S.B + S.b];
      dyn_body(kp + 1);
    }
    // One propagate per launch: this wave takes the body strips and the body block of P+ (res_prop_body: they need V, D, Xi,
    // ready since B3p, and write what no contraction reads) off the workers' path instead of waiting for them.
    if (PRIMARY && !MP) res_prop_body<64>(a, S, lane);
    __syncthreads();  // B4p (workers finish the contraction and publish the new body columns / block)
    RES_STAMP(S, lane == 0, 8);
    if (PRIMARY && MP && kp + 1 < nkp) {
      dt = dt_next;
      if (lane == 0) sm[42] = dt;
      dyn_feat(dt);
    }
   }

  // lane roles for the state correction (one instruction stream, no divergence):
  //   lane f < N           : feature f  -> rows 16+3f..+2 : bearing quaternion (2 rows) + inverse depth (1 row)
  //   lane N+j, j = 0..5   : body row j            (p, v)          linear state x[j]
  //   lane N+6             : body rows 6,7,8       (attitude)      quaternion x[6..9], right-multiplied
  //   lane N+j, j = 7..13  : body row j+2 = 9..15  (b_a, b_g, mu)  linear state x[j+3]
  const int jb = (ROLE == 2) ? lane : ((ROLE == 1) ? -1 : lane - N);
  const bool isfeat = PRIMARY && lane < N;
  const bool isatt = jb == 6;
  const bool hasq = (isfeat && lane < len) || isatt;
  const bool haslin = (isfeat && lane < len) || (jb >= 0 && jb < 14 && jb != 6);
  int rid0, rid1, rid2;
  if (isfeat) { rid0 = 16 + 3 * lane; rid1 = rid0 + 1; rid2 = rid0 + 2; }
  else if (isatt) { rid0 = 6; rid1 = 7; rid2 = 8; }
  else { const int r = (jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 2 : 0)); rid0 = rid1 = rid2 = r; }
  const bool rowlane = isfeat || (jb >= 0 && jb < 14);   // this lane owns rows of K / W (the others only tag along)
  double* qptr = isfeat ? (xs + xZ + 5 * lane) : (xs + xATT);
  double* linptr = isfeat ? (xs + xZ + 5 * lane + 4) : (xs + ((jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 3 : 0))));
  const double rho_reset = 1.0 / (2.0 * prm.min_depth);
  const double lam0 = partial ? S.lam[rid0] : 1.0, lam1 = partial ? S.lam[rid1] : 1.0, lam2 = partial ? S.lam[rid2] : 1.0;
  // Lambda of the zeta-zeta 2x2 block (lambda_feat[0], lambda_feat[1])
  const double lz0 = a.lambda[16], lz1 = a.lambda[17];
  const double L00 = partial ? (lz0 + lz0 - lz0 * lz0) : 1.0, L01 = partial ? (lz0 + lz1 - lz0 * lz1) : 1.0,
               L11 = partial ? (lz1 + lz1 - lz1 * lz1) : 1.0;

  // Each feature lane keeps its own P_zeta,zeta (2x2) current through the updates, so the lane of the NEXT measurement can
  // form  S = Hb P_zz Hb^T + R,  S^-1  and the gate verdict right after its prediction -- at the END of an iteration.
  // The next iteration then starts directly with the gain rows: no separate innovation phase, two barriers per update.
  int m = res_next_valid(S, 0);
  RES_STAMP(S, lane == 0, 9);
  __syncthreads();  // Bp : the workers published Pd (diagonal zeta blocks) and the first measurement's columns
  double pf00 = 0.0, pf01 = 0.0, pf10 = 0.0, pf11 = 0.0;
  if (isfeat) { const double* pd = S.Pd + 4 * lane; pf00 = pd[0]; pf01 = pd[1]; pf10 = pd[2]; pf11 = pd[3]; }
  // prediction + innovation of measurement mm (slot == this lane's feature) into mailbox half `hh`, from registers
  // Uniform per-measurement values {Hb, residual, S^-1, gate}: computed by the lane of the measured feature, handed to the
  // whole wave with v_readlane (they land in SGPRs; an LDS mailbox cost a store, a wave-level sync and a load on the
  // critical path of every update).
  struct Meas { double h0, h1, h2, h3, r0, r1, s0, s1, s2, s3, gate; };
  auto bcast = [&](double v, int src) -> double {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  };
  // prediction + innovation of measurement mm, whose feature is lane `src` (wave-uniform): EVERY lane runs the arithmetic
  // on its own registers (no divergence; the other lanes' results are discarded), lane src's values are broadcast
  auto predict = [&](const double* t1, const double* t2, const double* zt, int mm, int src, Meas& o) {
    double zhat[2], Hb[4], Sm[4], Si[4];
    h_feat_frame(t1, t2, zt, prm, zhat, Hb);
    const double2 zn = lds_ld2(S.mz + 2 * mm);
    const double* R = S.mR + 4 * mm;
    const double r0 = zn.x - zhat[0], r1 = zn.y - zhat[1];
    const double w00 = pf00 * Hb[0] + pf01 * Hb[1], w01 = pf00 * Hb[2] + pf01 * Hb[3];   // (P_zz Hb^T)
    const double w10 = pf10 * Hb[0] + pf11 * Hb[1], w11 = pf10 * Hb[2] + pf11 * Hb[3];
    Sm[0] = Hb[0] * w00 + Hb[1] * w10 + R[0];
    Sm[1] = Hb[0] * w01 + Hb[1] * w11 + R[2];
    Sm[2] = Hb[2] * w00 + Hb[3] * w10 + R[1];
    Sm[3] = Hb[2] * w01 + Hb[3] * w11 + R[3];
    inv2_fast(Sm, Si);
    const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;   // vi_ekf_meas.cpp:234
    o.h0 = bcast(Hb[0], src); o.h1 = bcast(Hb[1], src); o.h2 = bcast(Hb[2], src); o.h3 = bcast(Hb[3], src);
    o.r0 = bcast(r0, src); o.r1 = bcast(r1, src);
    o.s0 = bcast(Si[0], src); o.s1 = bcast(Si[1], src); o.s2 = bcast(Si[2], src); o.s3 = bcast(Si[3], src);
    o.gate = bcast((mahal > 9.0) ? 1.0 : 0.0, src);                                       // gate (:235-239)
  };
  // this lane's quaternion and linear state live in registers for the whole loop (written back once at the end)
  double qn[4] = {qptr[0], qptr[1], qptr[2], qptr[3]};
  double lin = *linptr;
  // the bearing frame of the CURRENT quaternion is kept alongside it: the prediction after a correction and the next
  // correction's T_zeta both use it, so it is computed once per update
  double f1[3], f2[3], fz[3];
  bearing_frame_fast(qn, f1, f2, fz);
  const double sgn = isatt ? -1.0 : 1.0;   // q (x) e instead of e (x) q flips the cross term only
  // Gain rows of a measurement for ALL n rows (three per lane) from its raw column pair pr (this lane's rows):
  //   W_i = P[i, j0:j0+2] Hb^T,  K_i = W_i S^-1   (vi_ekf_meas.cpp:241).
  // Also leaves the NaN guard (:247; a NaN in H makes every K row NaN, so testing K covers the H test) and the gate verdict
  // for the workers' next phase.
  const double lraw[3] = {S.lam[rid0], S.lam[rid1], S.lam[rid2]};
  // The rows are dealt by ROLE (a feature lane its three rows, the attitude lane rows 6..8, a linear body lane its one row),
  // so a lane's own rows of K and W -- all that its state correction needs in the next phase -- stay in registers.
  struct Rows { double2 kA, wA, kB, wB, kC; int bad; };
  const bool three = isfeat || isatt;   // lanes with three distinct rows (the others would write the same row three times)
  const int ridv[3] = {rid0, rid1, rid2};
  auto gain_rows = [&](const Meas& q, int nanword, int gateword, const double2 (&pr)[3], double* Kd, Rows& o) {
    double* Wd = Kd + 2 * n;                             // (Kd: destination buffer)
    int bad = 0;
    double2 wv[3], kv[3];
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const double w0 = pr[u].x * q.h0 + pr[u].y * q.h1, w1 = pr[u].x * q.h2 + pr[u].y * q.h3;
      const double k0 = w0 * q.s0 + w1 * q.s2, k1 = w0 * q.s1 + w1 * q.s3;
      wv[u] = make_double2(w0, w1); kv[u] = make_double2(k0, k1);
      if (rowlane && (u == 0 || three)) {
        *reinterpret_cast<double2*>(Wd + 2 * ridv[u]) = wv[u];
        *reinterpret_cast<double2*>(Kd + 2 * ridv[u]) = kv[u];
      }
      if (rowlane && (k0 != k0 || k1 != k1)) bad = 1;
    }
    bad = __any(bad);
    if (lane == 0) {   // (two service waves: each its own NaN word, 8 apart; the gate verdict is the feature wave's to publish)
      sm[nanword + (ROLE == 2 ? 8 : 0)] = bad ? 1.0 : 0.0;
      if (PRIMARY) sm[gateword] = q.gate;
    }
    o.kA = kv[0]; o.wA = wv[0]; o.kB = kv[1]; o.wB = wv[1]; o.kC = kv[2]; o.bad = bad;
  };
  // This lane's rows of the NEXT measurement's column pair: published by the worker waves one phase ago (buffer `rb`), as they
  // stood BEFORE the update being swept in this phase -- which is applied here (`swept`; gains Kc / Wc, this lane's own K rows
  // k3), with the workers' expression  p - Lambda (K . W):  feature rows i take K_i (own) and W of the column's feature,
  // body rows k take K of the column's feature and W_k, as the block sweep and the body-column sweep do.  Everything this
  // needs is complete at the top of a phase, so it runs there, off the critical path.
  const double lfz[3] = {a.lambda[16], a.lambda[17], a.lambda[18]};
  auto next_rows = [&](int rb, bool swept, int slot, const double* Kc, const double2 (&k3)[3], double2 (&o)[3]) {
    const double* raw = S.Praw + rb * 2 * n;
    const double* Wc = Kc + 2 * n;
    const double2 ka = lds_ld2(Kc + 2 * (16 + 3 * slot)), kb2 = lds_ld2(Kc + 2 * (16 + 3 * slot + 1));
    const double2 wa = lds_ld2(Wc + 2 * (16 + 3 * slot)), wb2 = lds_ld2(Wc + 2 * (16 + 3 * slot + 1));
    auto one = [&](int u) {
      const double2 st = lds_ld2(raw + 2 * ridv[u]);      // (P[i][j0], P[i][j0+1]) before the update
      double r0 = st.x, r1 = st.y;
      if (swept) {
        const double lamk = isfeat ? lfz[u] : lraw[u];
        const double La = partial ? (lamk + lz0 - lz0 * lamk) : 1.0, Lb = partial ? (lamk + lz1 - lz1 * lamk) : 1.0;
        const double2 wk = lds_ld2(Wc + 2 * ridv[u]);
        // (operands by role, no divergence: K_i . W_j0 | K_j0 . W_k)
        const double2 xa = isfeat ? k3[u] : ka, ya = isfeat ? wa : wk;
        const double2 xb = isfeat ? k3[u] : kb2, yb = isfeat ? wb2 : wk;
        r0 = fma(-La, fma(xa.y, ya.y, xa.x * ya.x), r0);
        r1 = fma(-Lb, fma(xb.y, yb.y, xb.x * yb.x), r1);
      }
      o[u] = make_double2(r0, r1);
    };
    one(0);
    if (three) { one(1); one(2); }
    else { o[1] = o[0]; o[2] = o[0]; }
    if (isfeat && lane == slot) o[1].x = o[0].y;   // the measured feature's own zeta block: lower = upper, as the workers keep it
  };
  // two service waves: the measurement's uniform values cross from the feature wave to the body wave through sm[16 mb ..],
  // published by a sequence number in sm[32 + mb] (an int; each mailbox sees increasing numbers)
  auto send = [&](const Meas& q, int mb, int seq) {
    if (lane == 0) {
      double* d = sm + 16 * mb;
      d[0] = q.h0; d[1] = q.h1; d[2] = q.h2; d[3] = q.h3; d[4] = q.r0; d[5] = q.r1;
      d[6] = q.s0; d[7] = q.s1; d[8] = q.s2; d[9] = q.s3; d[10] = q.gate;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) *(lds_vint_t*)(sm + 32 + mb) = seq;
  };
  auto recv = [&](Meas& q, int mb, int seq) {
    lds_vint_t* w = (lds_vint_t*)(sm + 32 + mb);
    int spins = 0;
    while (*w != seq && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(1); spins++; }
    if (spins >= (1 << 22)) flag |= FLAG_INTERNAL;   // (a bounded wait that gives up must say so)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const double* d = sm + 16 * mb;
    q.h0 = d[0]; q.h1 = d[1]; q.h2 = d[2]; q.h3 = d[3]; q.r0 = d[4]; q.r1 = d[5];
    q.s0 = d[6]; q.s1 = d[7]; q.s2 = d[8]; q.s3 = d[9]; q.gate = d[10];
  };
  Meas cur = {}, nxt = {};
  Rows crow = {}, nrow = {};
  if (m < M) {
    if (PRIMARY) predict(f1, f2, fz, m, __builtin_amdgcn_readfirstlane(S.mslot[m]), cur);
    if (ROLE == 1) send(cur, 0, 1);
    if (ROLE == 2) recv(cur, 0, 1);
    double2 pr0[3];
#pragma unroll
    for (int u = 0; u < 3; u++) pr0[u] = lds_ld2(S.Praw + 2 * ridv[u]);   // (the first raw columns: buffer 0, published before Bp)
    gain_rows(cur, 44, 50, pr0, S.Kt, crow);
  }
  int2 sq = S.mseq[min(m, MCAP - 1)];
  __syncthreads();  // B1
  RES_STAMP(S, lane == 0, 10);
  int it_ = 0, cnt = 0;

  while (m < M) {
    const int mnext = sq.x, slot_next = sq.y;
    // this lane's rows of the gain (formed by this wave at the end of the previous phase); rows 0,1 also feed its own P_zz
    const double* kP = (cnt & 1) ? S.Z : S.Kt;   // (double-buffered, see the worker side)
    const double2 kA = crow.kA, wA = crow.wA, kB = crow.kB, wB = crow.wB, kC = crow.kC;   // (own rows: from registers)
    sq = S.mseq[min(mnext, MCAP - 1)];   // next iteration's table entry (static data): its latency hides behind this update
    const bool gated = cur.gate != 0.0;
    // NaN guard (vi_ekf_meas.cpp:247), decided over every row of K in gain_rows (two service waves: the other one's rows too)
    const bool bad = crow.bad != 0 || (ROLE != 0 && sm[44 + cnt % 3 + (ROLE == 2 ? 0 : 8)] != 0.0);
    const double r0 = cur.r0, r1 = cur.r1;
    double2 prn[3] = {};
    if (slot_next >= 0) {
      const double2 k3[3] = {kA, kB, kC};
      next_rows((cnt + 1) & 1, !gated && !bad && !(S.dbg & 1), slot_next, kP, k3, prn);
    }
    RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 0);
    // correction lambda o (K r)   (vi_ekf_meas.cpp:249-255)
    const double dv0 = (lam0 * kA.x) * r0 + (lam0 * kA.y) * r1;
    const double dv1 = (lam1 * kB.x) * r0 + (lam1 * kB.y) * r1;
    const double dv2 = (lam2 * kC.x) * r0 + (lam2 * kC.y) * r1;
    const double kw[8] = {wA.x, wA.y, kA.x, kA.y, wB.x, wB.y, kB.x, kB.y};   // (w0,w1,k0,k1) of rows 0,1
    // rotation vector of the correction: bearing  T_zeta [d0 d1],  attitude  [d0 d1 d2]
    double v[3];
    v[0] = isatt ? dv0 : (f1[0] * dv0 + f2[0] * dv1);
    v[1] = isatt ? dv1 : (f1[1] * dv0 + f2[1] * dv1);
    v[2] = isatt ? dv2 : (f1[2] * dv0 + f2[2] * dv1);
    RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 1);
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 0);
    const bool corr = !gated && !bad && !(S.dbg & 2);
    // x <- x [+] dx  (vi_ekf_helper.cpp:88-98): bearing  exp(T_z d) (x) q ;  attitude  q (x) exp(d) ;  the rest adds.
    // The corrected quaternion / inverse depth stay in registers for fix_depth and the next prediction.
    if (corr) {
      double e[4];
      q_exp_fast(v, e);
      // e (x) q  and  q (x) e  share every term but the sign of the cross product (src/quat.cpp:304-312)
      const double ex = sgn * e[1], ey = sgn * e[2], ez = sgn * e[3];
      const double o0 = e[0] * qn[0] - e[1] * qn[1] - e[2] * qn[2] - e[3] * qn[3];
      const double o1 = e[0] * qn[1] + qn[0] * e[1] + (ey * qn[3] - ez * qn[2]);
      const double o2 = e[0] * qn[2] + qn[0] * e[2] + (ez * qn[1] - ex * qn[3]);
      const double o3 = e[0] * qn[3] + qn[0] * e[3] + (ex * qn[2] - ey * qn[1]);
      qn[0] = o0; qn[1] = o1; qn[2] = o2; qn[3] = o3;
      bearing_frame_fast(qn, f1, f2, fz);
      lin += isfeat ? dv2 : dv0;
      // this lane's copy of P_zz follows the sweep:  P_rs -= Lambda_rs (K_r . W_s)   (vi_ekf_meas.cpp:256-257)
      pf00 = fma(-L00, fma(kw[3], kw[1], kw[2] * kw[0]), pf00);
      pf01 = fma(-L01, fma(kw[3], kw[5], kw[2] * kw[4]), pf01);
      pf10 = pf01;   // (the workers keep the diagonal blocks exactly symmetric: lower = upper)
      pf11 = fma(-L11, fma(kw[7], kw[5], kw[6] * kw[4]), pf11);
    }
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 1);
    if (PRIMARY && lane == 0) sm[40 + par] = 0.0;
    // fix_depth (vi_ekf_meas.cpp:271; a gated update returns before it, :238): almost never fires -- one wave-wide test
    const bool odd_depth = !gated && isfeat && lane < len && !(lin >= 0.0 && lin <= 1e2);
    if (__any(odd_depth)) {
      if (odd_depth) {
        double rho = lin;
        if (rho != rho) { rho = rho_reset; flag |= FLAG_NAN; }
        if (rho < 0.0) {
          const double err = rho_reset - rho;
          S.fixadd[par * N + lane] = err * err;
          sm[40 + par] = 1.0;
          rho = rho_reset;
          flag |= FLAG_NEGDEPTH;
        } else if (rho > 1e2) {
          S.fixset[par * N + lane] = 1.0;
          sm[40 + par] = 1.0;
          rho = rho_reset;
        }
        lin = rho;
      }
    }
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 2);
    if (slot_next >= 0) {   // next measurement, from registers
      if (PRIMARY) predict(f1, f2, fz, mnext, __builtin_amdgcn_readfirstlane(slot_next), nxt);
      if (ROLE == 1) send(nxt, (cnt + 1) & 1, cnt + 2);
      if (ROLE == 2) recv(nxt, (cnt + 1) & 1, cnt + 2);
    }
    if (PRIMARY && result_all && lane == 0) result_all[(long)S.b * S.mstride + m] = gated ? 1 : 0;
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 3);
    if (slot_next >= 0) gain_rows(nxt, 44 + (cnt + 1) % 3, 50 + ((cnt + 1) & 1), prn, (cnt & 1) ? S.Kt : S.Z, nrow);
    {   // this wave's share of the body-column sweep of measurement m (few worker waves only), after its chain
      constexpr int NWV = T / 64 - 1;
      const int ns = res_service_items<NWV>(N);
      if (ns > 0) res_body_items(S, kP, !gated && !bad && !(S.dbg & 1), lane, 64, 8 * N - ns, 8 * N, sq.y, S.Praw + (cnt & 1) * 2 * n);
    }
    cur = nxt;
    crow = nrow;
    par ^= 1;
    cnt++;
    RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 2);
    __syncthreads();  // B1 (the only barrier of an update)
    RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 3);
    it_++;
    m = mnext;
  }

  if (hasq) { qptr[0] = qn[0]; qptr[1] = qn[1]; qptr[2] = qn[2]; qptr[3] = qn[3]; }
  if (haslin) *linptr = lin;
  RES_STAMP(S, lane == 0, 11);
  __syncthreads();  // B5
  RES_STAMP(S, lane == 0, 12);
  // ---------------- store x, status ----------------
  double* xg = a.x_out + (long)S.b * a.nxs;
  const int xend = (a.x_out != a.x) ? a.nxs : xZ + 5 * len;   // another ring slot gets the whole vector (zeros past the features)
  for (int i = lane; PRIMARY && i < xend; i += 64) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[S.b], flag);
  RES_STAMP(S, lane == 0, 13);
  {   // cooperative store of P (see res_store_chunk): this wave streams its share of every chunk
    const StoreChunks sc(N, n, S.img_len);
    for (int ch = 0; ch < sc.nchunks; ch++) {
      const int f0 = ch * sc.fc, f1 = min(N, f0 + sc.fc);
      __syncthreads();   // S1
      res_store_chunk<T>(a, S, f0, f1, threadIdx.x);
      __syncthreads();   // S2
    }
  }
}

// Common prologue of the fused-step kernels: LDS carve-up, state, lambdas, mailboxes and the measurement table (validity
// decided once, here).  T = workgroup size.  Ends with the table barriers.
template <int T>
__device__ __forceinline__ void res_prologue(const StreamArgs& a, ResShared& S, double* smem, int do_prop,
                                             const double* __restrict__ dt_all, const double* __restrict__ z_all,
                                             const int* __restrict__ slot_all, int M, int m_stride,
                                             const double* __restrict__ R_all, long r_stride_b, long r_stride_m,
                                             int* __restrict__ result_all) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const ResLds L(a.N, a.n, a.nxs);
  S.xs = smem + L.xs; S.Kt = smem + L.Kt; S.Wt = smem + L.Wt; S.Praw = smem + L.Praw; S.lam = smem + L.lam;
  S.sm = smem + L.sm; S.fixadd = smem + L.fixadd; S.fixset = smem + L.fixset; S.Z = smem + L.Z; S.img_len = L.img_len;
  S.phiff = smem + L.phiff; S.Abb = smem + L.Abb; S.Gb = smem + L.Gb; S.Phibb = smem + L.Phibb; S.Mbb = smem + L.Mbb;
  S.Gdb = smem + L.Gdb; S.Pbb = smem + L.Pbb; S.T16 = smem + L.T16; S.xdb = smem + L.xdb; S.Pbc = smem + L.Pbc; S.PhibbT = smem + L.PhibbT; S.Pd = smem + L.Pd; S.PsiP = smem + L.PsiP; S.Pi = smem + L.Pi; S.Xi = smem + L.Xi; S.AvG = smem + L.AvG; S.Lbc = smem + L.Lbc;
  S.mz = smem + L.mz; S.mR = smem + L.mR;
  S.mslot = reinterpret_cast<int*>(smem + L.mslot);
  S.mseq = reinterpret_cast<int2*>(smem + L.mseq);
  S.ctx = reinterpret_cast<BodyCtx*>(smem + L.ctx);
  S.N = a.N; S.n = a.n; S.nf = 3 * a.N; S.len = a.len[b]; S.M = M; S.mstride = m_stride; S.do_prop = do_prop & 1; S.dbg = (do_prop >> 8) & 0xff; S.kp = (do_prop >> 16) > 0 ? (do_prop >> 16) : 1; S.B = a.B; S.b = b; S.stamps = a.ws;
  {
    const double* xg = a.x + (long)b * a.nxs;
    for (int i = tid; i < a.nxs; i += T) S.xs[i] = (i < xZ + 5 * S.len) ? xg[i] : 0.0;
    for (int i = tid; i < a.n; i += T) S.lam[i] = a.lambda[i];
    for (int i = tid; i < 2 * a.N; i += T) { S.fixadd[i] = 0.0; S.fixset[i] = 0.0; }
    if (tid < 48) {
      const double lk = a.lambda[tid & 15], lq = a.lambda[16 + (tid >> 4)];
      S.Lbc[tid] = a.dp->use_partial_update ? (lk + lq - lq * lk) : 1.0;
    }
    if (tid == 0) { S.sm[42] = (do_prop & 1) ? dt_all[b] : 0.0; S.sm[40] = 0.0; S.sm[41] = 0.0; S.sm[44] = 0.0; S.sm[45] = 0.0; S.sm[46] = 0.0; S.sm[49] = 0.0; S.sm[50] = 0.0; S.sm[51] = 0.0; S.sm[32] = 0.0; S.sm[33] = 0.0; S.sm[52] = 0.0; S.sm[53] = 0.0; S.sm[54] = 0.0; }
    for (int mm_ = tid; mm_ < M; mm_ += T) {
      const int slot = slot_all[(long)b * m_stride + mm_];
      const double z0 = z_all[((long)b * m_stride + mm_) * 2], z1 = z_all[((long)b * m_stride + mm_) * 2 + 1];
      int code = 0;
      if (slot < 0) code = -1;                       // skipped
      else if (slot >= S.len) code = 3;              // MEAS_INVALID
      else if (z0 != z0 || z1 != z1) code = 2;       // MEAS_NAN (vi_ekf_meas.cpp:136-137)
      S.mslot[mm_] = (code == 0) ? slot : -1;
      S.mz[2 * mm_] = z0; S.mz[2 * mm_ + 1] = z1;
      const double* R = R_all + (long)b * r_stride_b + (long)mm_ * r_stride_m;
      S.mR[4 * mm_ + 0] = R[0]; S.mR[4 * mm_ + 1] = R[1]; S.mR[4 * mm_ + 2] = R[2]; S.mR[4 * mm_ + 3] = R[3];
      if (code != 0 && result_all) result_all[(long)b * m_stride + mm_] = code;
    }
  }
  RES_STAMP(S, tid == 0, 62);
  __syncthreads();
  for (int mm_ = tid; mm_ < M; mm_ += T) {   // successor table (each entry scans forward; M <= MCAP)
    int nx = mm_ + 1;
    while (nx < M && S.mslot[nx] < 0) nx++;
    S.mseq[mm_] = make_int2(nx, nx < M ? S.mslot[nx] : -1);
  }
  __syncthreads();
  RES_STAMP(S, tid == 0, 63);
}

template <int RB, int NW, bool MP = false, int NS = 1>
__global__ __launch_bounds__((NW + NS) * 64, (NW <= 3) ? 2 : 1) void k_step_resident(StreamArgs a, int TR, int TD, int do_prop,
                                                                const double* __restrict__ u_all,
                                                                const double* __restrict__ dt_all,
                                                                const double* __restrict__ z_all,
                                                                const int* __restrict__ slot_all, int M, int m_stride,
                                                                const double* __restrict__ R_all, long r_stride_b,
                                                                long r_stride_m, int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = (NW + NS) * 64, TW = NW * 64;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.B) return;
  if (a.active && !a.active[blockIdx.x]) return;   // (the whole workgroup: before any barrier)
  ResShared S;
  res_prologue<T>(a, S, smem, do_prop, dt_all, z_all, slot_all, M, m_stride, R_all, r_stride_b, r_stride_m, result_all);
  // The service wave's chain is the floor of an update, so it should not share its SIMD's issue slots with a worker wave.  A
  // workgroup's waves go to the four SIMDs round-robin: with 7 waves (NW = 6, one service wave) the 4th one is alone on its
  // SIMD -- that is the service wave.  Otherwise the last wave(s) serve.
  constexpr int SVC = (NW == 6 && NS == 1) ? 3 : NW;
  const int wave = tid >> 6;
  if (NS == 2) {
    if (wave == NW) res_service<T, MP, 1>(a, S, tid & 63, NW, u_all, dt_all, result_all);
    else if (wave == NW + 1) res_service<T, MP, 2>(a, S, tid & 63, NW, u_all, dt_all, result_all);
    else res_worker<RB, TW, MP, T>(a, S, tid);
  } else {
    if (wave == SVC) res_service<T, MP>(a, S, tid & 63, NW, u_all, dt_all, result_all);
    else res_worker<RB, TW, MP, T>(a, S, tid - (wave > SVC ? 64 : 0));
  }
}

}  // namespace viekf
