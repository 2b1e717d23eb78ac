// viekf_tiles_common.hpp -- "tile" kernel family (r03): the fused step with the WHOLE covariance held as 16 x 16 tiles in the
// accumulator layout of v_mfma_f64_16x16x4_f64, so that every sweep over P -- the rank-2 (with Lambda: rank-4) update of a
// measurement and the propagate's  Phi_ff P Phi_ff^T + low-rank coupling -- is issued as matrix-core instructions.
//
// Why: the resident family's update loop is bound by VALU ISSUE, not by the fp64 pipe or by HBM -- ~600 vector
// instructions per worker wave and update of which 168 are the fp64 floor, next to a ~970-instruction service chain on the same
// SIMD.  One v_mfma_f64_16x16x4_f64 applies  P_tile -= Lambda o (K W^T)  to 256 elements for ONE issue slot (the Lambda mask is
// rank 4:  Lambda o (K W^T) = K W^T - (mu o K)(mu o W)^T,  mu = 1 - lambda,  exactly the instruction's K = 4), the matrix pipe
// runs beside the vector pipe, and the service wave that shares the SIMD gets the issue slots back.
//
// Row space.  P's rows are re-indexed so that no feature straddles a tile:  tile 0 = the 16 body rows,  tile t >= 1 = the
// features 5 (t - 1) .. 5 (t - 1) + 4, three rows each, and one pad row (index 15: always zero).  q = 16 t + w  <->  P row
// w (t = 0)  or  16 + 15 (t - 1) + w (t >= 1, w < 15).  NT = 1 + ceil(N / 5) tiles per side, NQ = 16 NT rows.
// Ownership.  Only tiles (TI, TJ) with TI >= TJ are held (P is symmetric); tile -> (wave, slot) is a host-built table
// (build_tilemap, viekf_capi.hip), wave = (TI + TJ) mod NW so that the NT tiles that hold one feature's rows / columns spread
// evenly over the waves.  A diagonal tile holds both triangles, but only its LOWER one (P row >= P column) is ever read back
// (column extraction, store): P is exactly symmetric by construction, whatever the rounding of the two halves.
// Register content of a tile (TI, TJ), lane l, register r (the instruction's C/D layout: row = (l >> 4) + 4 r, col = l & 15):
//     X[r] = P[ prow(TI, l & 15) ][ prow(TJ, (l >> 4) + 4 r) ]
// i.e. the lane index runs along the ROWS of P (column-major in HBM: 128-byte runs per 16 lanes for loads and stores).
#pragma once
#include "viekf_resident_common.hpp"
#include "viekf_resident_prop.hpp"

namespace viekf {

__host__ __device__ inline int tile_nt(int N) { return 1 + (N + 4) / 5; }

struct TileLds {  // LDS carve-up in doubles, shared by host (size) and device (offsets)
  int xs, lam, sm, fixadd, fixset, Z, phiff, Abb, Gb, Phibb, PhibbT, Gdb, T16, PsiP, Pi, Xi, AvG, Cb, Eb, Pd, Mbb, Pbb, xdb, ctx, Pbc,
      mslot, mseq, mz, mR, total;
  __host__ __device__ TileLds(int N, int n, int nxs) {
    const int nf = 3 * N, NQ = 16 * tile_nt(N);
    int o = 0;
    auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
    xs = take(nxs);
    lam = take(n);
    sm = take(64);          // [0..15] two measurement mailboxes {g00, g01, g11, skip}; [40..41] fix mailboxes non-empty; [42] dt;
                            // [44..46] NaN-in-column words (phase mod 3)
    fixadd = take(2 * (N > 0 ? N : 1)); fixset = take(2 * (N > 0 ? N : 1));
    Z = take(nf * ZS);      // the propagate's records (viekf_resident_common.hpp)
    phiff = take(9 * (N > 0 ? N : 1));
    // one region, two lives: the propagate's body-sized scratch | the update loop's column buffers
    const int u0 = o;
    Abb = take(256); Gb = take(96); Phibb = take(256); PhibbT = take(256); Gdb = take(96); T16 = take(256);
    PsiP = take(ZK * 16); Pi = take(ZK * ZK); Xi = take(ZK * 16); AvG = take(18);
    const int uprop = o;
    o = u0;
    Cb = take(4 * NQ);      // two buffers [NQ][2]: the column pair of the CURRENT measurement's feature, every pending update applied
    Eb = take(4 * NQ);      // two buffers [NQ][2]: raw column pairs extracted from the tiles one update ahead
    Pd = take(4 * (N > 0 ? N : 1));   // zeta-zeta 2x2 diagonal blocks, handed from the workers to the service lanes
    if (uprop > o) o = uprop;
    Mbb = take(256); Pbb = take(256);
    xdb = take(16);
    ctx = take((int)((sizeof(BodyCtx) + 7) / 8));
    Pbc = take(nf * 16);    // the body columns P[16.., 0:16] during load and propagate (register tiles during the updates)
    const int mc = res_mcap(N);
    mslot = take(mc / 2); mseq = take(mc); mz = take(2 * mc); mR = take(4 * mc);
    total = o;
  }
};

struct TileShared : ResShared {   // (the propagate set-up routines of the resident family work on the ResShared part)
  double *Cb, *Eb;
  int NT, NQ;
};

// tile-space row -> row of P (or -1: pad row / past the last feature slot)
__device__ __forceinline__ int tile_prow(int T, int w, int nf) {
  if (T == 0) return w;
  const int r = 15 * (T - 1) + w;
  return (w < 15 && r < nf) ? 16 + r : -1;
}
// mu = 1 - lambda of a tile-space row (0 for a pad row and without the partial update):  Lambda_ij = 1 - mu_i mu_j
__device__ __forceinline__ double tile_mu(const double* lam, int q, int nf, bool partial) {
  const int pr = tile_prow(q >> 4, q & 15, nf);
  return (pr >= 0 && partial) ? 1.0 - lam[max(pr, 0)] : 0.0;
}
// tile-space row index of row r of feature f
__device__ __forceinline__ int tile_qrow(int f, int r) { return 16 * (1 + f / 5) + 3 * (f % 5) + r; }

// Common prologue: LDS carve-up, state, lambdas, mailboxes and the measurement table (validity decided once, here).
// T = number of threads that run it for this filter (tid = 0 .. T - 1), b = the filter.  present = false: a workgroup of the paired
// kernel whose second filter does not exist (odd batch) or takes no part (participation mask): only the barriers and the count.
template <int T>
__device__ __forceinline__ void tile_prologue(const StreamArgs& a, TileShared& S, double* smem, int b, int tid, bool present, int do_prop,
                                              const double* __restrict__ dt_all, const double* __restrict__ z_all,
                                              const int* __restrict__ slot_all, int M, int m_stride,
                                              const double* __restrict__ R_all, long r_stride_b, long r_stride_m,
                                              int* __restrict__ result_all) {
  const TileLds L(a.N, a.n, a.nxs);
  S.xs = smem + L.xs; S.lam = smem + L.lam; S.sm = smem + L.sm; S.fixadd = smem + L.fixadd; S.fixset = smem + L.fixset;
  S.Z = smem + L.Z; S.phiff = smem + L.phiff; S.Abb = smem + L.Abb; S.Gb = smem + L.Gb; S.Phibb = smem + L.Phibb; S.Mbb = smem + L.Mbb;
  S.Gdb = smem + L.Gdb; S.Pbb = smem + L.Pbb; S.T16 = smem + L.T16; S.xdb = smem + L.xdb; S.Pbc = smem + L.Pbc; S.PhibbT = smem + L.PhibbT;
  S.Pd = smem + L.Pd; S.PsiP = smem + L.PsiP; S.Pi = smem + L.Pi; S.Xi = smem + L.Xi; S.AvG = smem + L.AvG;
  S.Cb = smem + L.Cb; S.Eb = smem + L.Eb;
  S.Kt = nullptr; S.Wt = nullptr; S.Praw = nullptr; S.Lbc = nullptr; S.img_len = 0;
  S.mz = smem + L.mz; S.mR = smem + L.mR;
  S.mslot = reinterpret_cast<int*>(smem + L.mslot);
  S.mseq = reinterpret_cast<int2*>(smem + L.mseq);
  S.ctx = reinterpret_cast<BodyCtx*>(smem + L.ctx);
  S.N = a.N; S.mcap = res_mcap(a.N); S.n = a.n; S.nf = 3 * a.N; S.len = present ? a.len[b] : 0; S.M = present ? M : 0; S.mstride = m_stride; S.do_prop = do_prop & 1;
  S.dbg = (do_prop >> 8) & 0xff; S.kp = (do_prop >> 16) > 0 ? (do_prop >> 16) : 1; S.B = a.B; S.b = b; S.stamps = a.ws;
  S.NT = tile_nt(a.N); S.NQ = 16 * S.NT;
  S.si = present ? a.si(b) : 0; S.so = present ? a.so(b) : 0;
  if (present) {
    const double* xg = a.x + S.si * a.nxs;
    for (int i = tid; i < a.nxs; i += T) S.xs[i] = (i < xZ + 5 * S.len) ? xg[i] : 0.0;
    for (int i = tid; i < a.n; i += T) S.lam[i] = a.lambda[i];
    for (int i = tid; i < 2 * a.N; i += T) { S.fixadd[i] = 0.0; S.fixset[i] = 0.0; }
    for (int i = tid; i < 64; i += T) S.sm[i] = (i == 42 && (do_prop & 1)) ? dt_all[b] : 0.0;
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
  __syncthreads();
  if (present && a.smap_out && tid == 0) a.smap[b] = (int)S.so;   // (see res_prologue)
  for (int mm_ = tid; present && mm_ < M; mm_ += T) {   // successor table (each entry scans forward; M <= res_mcap(N))
    int nx = mm_ + 1;
    while (nx < M && S.mslot[nx] < 0) nx++;
    S.mseq[mm_] = make_int2(nx, nx < M ? S.mslot[nx] : -1);
  }
  if (tid == 0) {   // number of updates that will run (the paired kernel's two filters loop in step: the longer count rules)
    int c = 0;
    for (int i = 0; present && i < M; i++) c += S.mslot[i] >= 0;
    S.sm[60] = (double)c;
  }
  __syncthreads();
}

}  // namespace viekf
