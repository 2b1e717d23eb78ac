// Fused-step kernel, "tile" worker layout: few, fat worker threads.
//
// The wrapped-diagonal layout of viekf_kernels_resident.hpp keeps 3 scattered 3x3 blocks per thread: every block needs
// its own 3 K rows and 3 W rows from LDS (6 ds_read_b128 per 27 flops) and 7 worker waves repeat the per-wave overhead,
// so an update is bound by LDS reads and VALU issue of two waves per SIMD (~2,900 clk per update).  Here a worker
// thread owns a 2x3 TILE of blocks -- features {2A, 2A+1} x {3C, 3C+1, 3C+2} -- of the upper block triangle of P
// (every unordered feature pair appears in at least one tile; blocks with I > J inside a tile that straddles the diagonal
// are computed but never used), plus a 3x4 piece of the n x 16 body strip P[:, 0:16], all in registers:
//   per update 6 + 9 + 3 + 4 = 22 ds_read_b128 feed 6*27 + 12*3 = 198 fp64 FMAs, on ONE worker wave per SIMD.
// 233 tiles at N = 50 -> 4 worker waves + the service wave = 320 threads; up to 512 registers per lane are available.
// Everything else (service wave, mailboxes, barrier sequence, propagate set-up) is shared with the resident kernel.
#pragma once
#include "viekf_kernels_resident.hpp"

namespace viekf {

constexpr int TILE_NW = 4;   // worker waves

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// per-thread tile / strip-item coordinates
struct TileId {
  int A, C;        // block tile: features {2A, 2A+1} x {3C, 3C+1, 3C+2}
  int g, cg;       // strip item: rows 3g..3g+2 (of all n rows), columns 4cg..4cg+3 of P[:, 0:16]
  bool has_tile, has_item;
};

__device__ __forceinline__ TileId tile_id(const StreamArgs& a, int n, int tid) {
  TileId t;
  const int2 tc = a.tiles[min(tid, a.ntiles - 1)];
  t.A = tc.x; t.C = tc.y;
  t.has_tile = tid < a.ntiles;
  const int ng = (n + 2) / 3;
  t.has_item = tid < 4 * ng;
  t.g = min(tid >> 2, ng - 1); t.cg = tid & 3;
  return t;
}

// Tile <-> HBM.  P is column-major with an even leading dimension, and a tile's rows 16+6A .. 16+6A+5 are six consecutive,
// 16-byte aligned doubles of a column: a tile whose six blocks all lie strictly above the block diagonal moves as
// 16-byte pieces (27 loads; 27 + 30 stores with the mirror image, whose nine consecutive rows 16+9C .. start on an odd or
// even row with C).  Tiles that straddle the diagonal or the edge of P take the element-wise path.
template <int TW>
__device__ __forceinline__ void tile_load(const StreamArgs& a, const ResShared& S, int tid, const TileId& t,
                                          double (&pb)[2][3][9]) {
  const int N = S.N, n = S.n, ld = a.ld, nf = S.nf;
  const double* P = a.P + (long)S.b * n * ld;
  const int A = opaque(t.A), C = opaque(t.C);
  if (2 * A + 1 < N) {   // both feature rows exist: 16-byte pieces down each of the nine columns
#pragma unroll
    for (int ic = 0; ic < 3; ic++) {
      const int J = min(3 * C + ic, N - 1);
#pragma unroll
      for (int s = 0; s < 3; s++) {
        const double2* q = reinterpret_cast<const double2*>(P + (16 + 6 * A) + (long)(16 + 3 * J + s) * ld);
        const double2 v0 = q[0], v1 = q[1], v2 = q[2];
        pb[0][ic][0 * 3 + s] = v0.x; pb[0][ic][1 * 3 + s] = v0.y; pb[0][ic][2 * 3 + s] = v1.x;
        pb[1][ic][0 * 3 + s] = v1.y; pb[1][ic][1 * 3 + s] = v2.x; pb[1][ic][2 * 3 + s] = v2.y;
      }
    }
  } else {
#pragma unroll
    for (int ia = 0; ia < 2; ia++)
#pragma unroll
      for (int ic = 0; ic < 3; ic++) {
        const int I = min(2 * A + ia, N - 1), J = min(3 * C + ic, N - 1);
        const double* pu = P + ((16 + 3 * I) + (long)(16 + 3 * J) * ld);
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
          for (int r = 0; r < 3; r++) pb[ia][ic][r * 3 + s] = pu[r + (long)s * ld];
      }
  }
  // body strip -> LDS (coalesced along rows); it stays there for the whole step
  for (int e = tid; e < nf * 16; e += TW) {
    const int k = e / nf, row = e - k * nf;
    S.Pbc[row * 16 + k] = P[(16 + row) + (long)k * ld];
  }
  for (int e = tid; e < 256; e += TW) S.Pbb[(e & 15) * 16 + (e >> 4)] = P[(e & 15) + (long)(e >> 4) * ld];
}

template <int TW>
__device__ __forceinline__ void tile_store(const StreamArgs& a, const ResShared& S, int tid, const TileId& t,
                                           const double (&pb)[2][3][9]) {
  const int N = S.N, n = S.n, ld = a.ld, nf = S.nf;
  double* P = a.P + (long)S.b * n * ld;
  const int A = opaque(t.A), C = opaque(t.C);
  if (t.has_tile && 2 * A + 1 < 3 * C && 3 * C + 2 < N) {   // all six blocks strictly above the diagonal, inside P
#pragma unroll
    for (int ic = 0; ic < 3; ic++)
#pragma unroll
      for (int s = 0; s < 3; s++) {
        double2* q = reinterpret_cast<double2*>(P + (16 + 6 * A) + (long)(16 + 9 * C + 3 * ic + s) * ld);
        q[0] = make_double2(pb[0][ic][0 * 3 + s], pb[0][ic][1 * 3 + s]);
        q[1] = make_double2(pb[0][ic][2 * 3 + s], pb[1][ic][0 * 3 + s]);
        q[2] = make_double2(pb[1][ic][1 * 3 + s], pb[1][ic][2 * 3 + s]);
      }
    // mirror: column 16+6A+3ia+r of P, rows 16+9C .. 16+9C+8 = (ic, s) in order
#pragma unroll
    for (int ia = 0; ia < 2; ia++)
#pragma unroll
      for (int r = 0; r < 3; r++) {
        double* q = P + (16 + 9 * C) + (long)(16 + 6 * A + 3 * ia + r) * ld;
        double v[9];
#pragma unroll
        for (int ic = 0; ic < 3; ic++)
#pragma unroll
          for (int s = 0; s < 3; s++) v[3 * ic + s] = pb[ia][ic][r * 3 + s];
        if (C & 1) {   // 16 + 9C is odd: one element, then four aligned pairs
          q[0] = v[0];
#pragma unroll
          for (int u = 0; u < 4; u++) *reinterpret_cast<double2*>(q + 1 + 2 * u) = make_double2(v[1 + 2 * u], v[2 + 2 * u]);
        } else {
#pragma unroll
          for (int u = 0; u < 4; u++) *reinterpret_cast<double2*>(q + 2 * u) = make_double2(v[2 * u], v[2 * u + 1]);
          q[8] = v[8];
        }
      }
  } else {
#pragma unroll
    for (int ia = 0; ia < 2; ia++)
#pragma unroll
      for (int ic = 0; ic < 3; ic++) {
        const int Iu = 2 * A + ia, Ju = 3 * C + ic;
        if (t.has_tile && Iu < N && Ju < N && Iu <= Ju) {   // canonical blocks only; the mirror comes from the same registers
          double* pu = P + ((16 + 3 * Iu) + (long)(16 + 3 * Ju) * ld);
          double* pt = P + ((16 + 3 * Ju) + (long)(16 + 3 * Iu) * ld);
#pragma unroll
          for (int s = 0; s < 3; s++)
#pragma unroll
            for (int r = 0; r < 3; r++) pu[r + (long)s * ld] = pb[ia][ic][r * 3 + s];
          if (Iu != Ju) {
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
              for (int r = 0; r < 3; r++) pt[s + (long)r * ld] = pb[ia][ic][r * 3 + s];
          }
        }
      }
  }
  for (int e = opaque(tid); e < nf * 16; e += TW) {     // body strip columns, coalesced along rows
    const int k = e / nf, row = e - k * nf;
    P[(16 + row) + (long)k * ld] = S.Pbc[row * 16 + k];
  }
  for (int e = opaque(tid); e < nf * 16; e += TW) {     // mirrored body rows, coalesced along k
    const int row = e >> 4, k = e & 15;
    P[k + (long)(16 + row) * ld] = S.Pbc[e];
  }
  for (int e = opaque(tid); e < 256; e += TW) P[(e >> 4) + (long)(e & 15) * ld] = S.Pbb[e];
}

// rank-2 sweep of block column IC of the tile:  P_rs -= Lambda_rs (K_r . W_s)   (vi_ekf_meas.cpp:254-257)
// (a forceinline function with a compile-time column: a lambda taking the column as an argument is inlined only after the
//  first scalar-replacement pass, which leaves part of the tile in scratch memory)
template <int IC>
__device__ __forceinline__ void tile_sweep_col(double (&pb)[2][3][9], const double2 (&kI)[2][3], const double2 (&wv)[3],
                                               const double (&Lff)[9]) {
  // One worker wave per SIMD: nothing hides the latency of a dependent fp64 op, so the three dependent steps of an element
  // (mul, fma, fma) are issued block-wise -- 9 independent instructions between a value and its use.
#pragma unroll
  for (int ia = 0; ia < 2; ia++) {
    double t[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) t[r * 3 + s] = kI[ia][r].x * wv[s].x;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) t[r * 3 + s] = fma(kI[ia][r].y, wv[s].y, t[r * 3 + s]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) pb[ia][IC][r * 3 + s] = fma(-Lff[r * 3 + s], t[r * 3 + s], pb[ia][IC][r * 3 + s]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Publishing the raw columns of the next measurement from the tile registers.  The block column (row) that holds the
// feature is workgroup-uniform, so a uniform branch picks the register set.  Each variant ends in a distinct asm marker:
// without it the optimiser merges the variants' identical stores and selects the source by POINTER, which pins part of the
// tile in scratch memory.
template <int IC>
__device__ __forceinline__ void tile_pub_col(const double (&pb)[2][3][9], double* Pw, const bool (&ok)[2],
                                             const int (&Ib)[2]) {
#pragma unroll
  for (int ia = 0; ia < 2; ia++)
    if (ok[ia]) {
      double* d = Pw + 2 * (16 + 3 * Ib[ia]);
#pragma unroll
      for (int r = 0; r < 3; r++)
        *reinterpret_cast<double2*>(d + 2 * r) = make_double2(pb[ia][IC][r * 3 + 0], pb[ia][IC][r * 3 + 1]);
    }
  asm volatile("; tile column %0 published" ::"n"(IC));
}
template <int IA>
__device__ __forceinline__ void tile_pub_row(const double (&pb)[2][3][9], double* Pw, const bool (&ok)[3],
                                             const int (&Jb)[3]) {
#pragma unroll
  for (int ic = 0; ic < 3; ic++)
    if (ok[ic]) {
      double* d = Pw + 2 * (16 + 3 * Jb[ic]);
#pragma unroll
      for (int s = 0; s < 3; s++)
        *reinterpret_cast<double2*>(d + 2 * s) = make_double2(pb[IA][ic][0 * 3 + s], pb[IA][ic][1 * 3 + s]);
    }
  asm volatile("; tile row %0 published" ::"n"(IA));
}

template <int TW>
__device__ __forceinline__ void tile_worker(const StreamArgs& a, const ResShared& S, int tid, const TileId& tl,
                                            double (&pb)[2][3][9]) {
  const int N = S.N, n = S.n, len = S.len;
  const DevParams& prm = *a.dp;
  double* Pbc = S.Pbc;   // [nf][16]  staging of P[16+row][k] for the propagate; between the phases it lives in registers
  const bool has_tile = tl.has_tile, has_item = tl.has_item;
  const int A_ = tl.A, C_ = tl.C, g_ = tl.g, cg_ = tl.cg;
  // block (ia, ic) of the tile: features I = 2A+ia, J = 3C+ic (clamped for addressing; validity separately)
  auto fI = [&](int A, int ia) __attribute__((always_inline)) { return min(2 * A + ia, N - 1); };
  auto fJ = [&](int C, int ic) __attribute__((always_inline)) { return min(3 * C + ic, N - 1); };

  // Lambda for feature/feature blocks: one 3x3 constant (lambda_feat identical for all slots), kept in SGPRs
  const bool partial = prm.use_partial_update != 0;
  double Lff[9];
  {
    const double lf[3] = {a.lambda[16], a.lambda[17], a.lambda[18]};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) Lff[r * 3 + s] = uniform_f64(partial ? (lf[s] + lf[r] - lf[r] * lf[s]) : 1.0);
  }
  int par = 0;  // fix_depth mailbox parity (mirrors the service wave)
  RES_STAMP(S, tid == 0, 64);
  __syncthreads();  // B0
  RES_STAMP(S, tid == 0, 65);

  if (S.do_prop) {
    double* X = S.X; double* Y = S.Y; double* phiff = S.phiff;
    res_prop_setup<TW>(a, S, tid);
    RES_STAMP(S, tid == 0, 66);
    __syncthreads();  // B3p
    RES_STAMP(S, tid == 0, 67);
    // ---- local 3x3 transforms  Phi_ff[I] (P[I,J] Phi_ff[J]^T) (+ Qx on the diagonal), in place
#pragma unroll
    for (int ia = 0; ia < 2; ia++)
#pragma unroll
      for (int ic = 0; ic < 3; ic++) {
        const int A = opaque(A_), C = opaque(C_);
        const int I = fI(A, ia), J = fJ(C, ic);
        const double* fj = phiff + 9 * J;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const double p0 = pb[ia][ic][r * 3 + 0], p1 = pb[ia][ic][r * 3 + 1], p2 = pb[ia][ic][r * 3 + 2];
#pragma unroll
          for (int s = 0; s < 3; s++) pb[ia][ic][r * 3 + s] = p0 * fj[s * 3 + 0] + p1 * fj[s * 3 + 1] + p2 * fj[s * 3 + 2];
        }
        const double* fi = phiff + 9 * I;
#pragma unroll
        for (int s = 0; s < 3; s++) {
          const double p0 = pb[ia][ic][0 * 3 + s], p1 = pb[ia][ic][1 * 3 + s], p2 = pb[ia][ic][2 * 3 + s];
#pragma unroll
          for (int r = 0; r < 3; r++) pb[ia][ic][r * 3 + s] = fi[r * 3 + 0] * p0 + fi[r * 3 + 1] * p1 + fi[r * 3 + 2] * p2;
        }
        if (2 * A + ia == 3 * C + ic) {
          pb[ia][ic][0] += a.Qx[16 + 3 * I + 0];
          pb[ia][ic][4] += a.Qx[16 + 3 * I + 1];
          pb[ia][ic][8] += a.Qx[16 + 3 * I + 2];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    RES_STAMP(S, tid == 0, 68);
    // ---- register-tiled contraction  P[I,J] += X_I Y_J^T  (K = 38): 6 + 9 row reads feed 108 FMAs per k pair
#pragma unroll 1
    for (int k = 0; k < XK; k += 2) {
      const int A = opaque(A_), C = opaque(C_);
      double2 xv[2][3], yv[3][3];
#pragma unroll
      for (int ia = 0; ia < 2; ia++)
#pragma unroll
        for (int r = 0; r < 3; r++) xv[ia][r] = *reinterpret_cast<const double2*>(X + (3 * fI(A, ia) + r) * XK + k);
#pragma unroll
      for (int ic = 0; ic < 3; ic++)
#pragma unroll
        for (int s = 0; s < 3; s++) yv[ic][s] = *reinterpret_cast<const double2*>(Y + (3 * fJ(C, ic) + s) * XK + k);
#pragma unroll
      for (int ia = 0; ia < 2; ia++)
#pragma unroll
        for (int ic = 0; ic < 3; ic++)
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) {
              double acc = pb[ia][ic][r * 3 + s];
              acc = fma(xv[ia][r].x, yv[ic][s].x, acc);
              acc = fma(xv[ia][r].y, yv[ic][s].y, acc);
              pb[ia][ic][r * 3 + s] = acc;
            }
    }
    RES_STAMP(S, tid == 0, 69);
    res_prop_body<TW>(a, S, tid);
    par ^= 1;   // the service wave posted propagate's fix_depth edits into mailbox 0
    RES_STAMP(S, tid == 0, 70);
    __syncthreads();  // B4p
    for (int e = tid; e < 256; e += TW) S.Pbb[e] = S.Mbb[e];   // P_bb+ was staged in Mbb (T16 / Pbb were still being read)
  }

  // per-thread constants of the update loop
  bool vI[2], vJ[3];
  int Ib[2], Jb[3];
#pragma unroll
  for (int ia = 0; ia < 2; ia++) { vI[ia] = has_tile && (2 * A_ + ia < N); Ib[ia] = fI(A_, ia); }
#pragma unroll
  for (int ic = 0; ic < 3; ic++) { vJ[ic] = 3 * C_ + ic < N; Jb[ic] = fJ(C_, ic); }
  // body strip item: rows 3g..3g+2, columns 4cg..4cg+3 of P[:, 0:16], kept in the LDS arrays Pbb (rows < 16) / Pbc
  double* sp[3];
  bool sv[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const int ri = min(3 * g_ + q, n - 1);
    sp[q] = (ri < 16) ? (S.Pbb + ri * 16 + 4 * cg_) : (Pbc + (ri - 16) * 16 + 4 * cg_);
    sv[q] = has_item && (3 * g_ + q < n);
  }
  const double p0rr = uniform_f64(prm.P0_feat[2]);   // (read here: a global load inside the update loop would put vmcnt waits there)
  // applies the pending fix_depth covariance edits of mailbox `mb` to the diagonal blocks held by this tile
  auto apply_fixes = [&](int mb, double pending) __attribute__((always_inline)) {
    if (pending == 0.0) return;   // nothing posted (the common case)
#pragma unroll
    for (int ia = 0; ia < 2; ia++)
#pragma unroll
      for (int ic = 0; ic < 3; ic++) {
        const int I = Ib[ia];
        if (vI[ia] && 2 * A_ + ia == 3 * C_ + ic && I < len) {
          const double ad = S.fixadd[mb * N + I], st = S.fixset[mb * N + I];
          if (ad != 0.0) { pb[ia][ic][8] += ad; S.fixadd[mb * N + I] = 0.0; }
          if (st != 0.0) { pb[ia][ic][8] = p0rr; S.fixset[mb * N + I] = 0.0; }
        }
      }
  };
  // Publishes the two zeta columns of feature `slot` (raw P[:, j0], P[:, j0+1]; P is symmetric, so rows are used where the
  // column is not held) into the Praw buffer `buf`.  slot % 3 and slot % 2 are workgroup-uniform: uniform branches pick the
  // register set, no data-dependent register index.  Only CANONICAL blocks (I <= J) publish, so each row has one writer.
  auto extract = [&](int slot, int buf) __attribute__((always_inline)) {
    double* Pw = S.Praw + buf * 2 * n;
    const int cm = slot % 3, Cs = slot / 3, am = slot & 1, As = slot >> 1;
    // (1) tile column holds feature `slot`: block (I, slot), its columns 0,1 -> rows 16+3I+r
    if (has_tile && C_ == Cs) {
      const bool ok[2] = {vI[0] && 2 * A_ <= slot, vI[1] && 2 * A_ + 1 <= slot};
      if (cm == 0) tile_pub_col<0>(pb, Pw, ok, Ib);
      else if (cm == 1) tile_pub_col<1>(pb, Pw, ok, Ib);
      else tile_pub_col<2>(pb, Pw, ok, Ib);
    }
    // (2) tile row holds feature `slot`: block (slot, J), J > slot, its rows 0,1 transposed -> rows 16+3J+s
    if (has_tile && A_ == As) {
      const bool ok[3] = {vJ[0] && 3 * C_ > slot, vJ[1] && 3 * C_ + 1 > slot, vJ[2] && 3 * C_ + 2 > slot};
      if (am == 0) tile_pub_row<0>(pb, Pw, ok, Jb);
      else tile_pub_row<1>(pb, Pw, ok, Jb);
    }
    // (3) body rows k < 16: P[k][j0 + c] = P[j0 + c][k]; row j0 = 16 + 3 slot = 3 (5 + slot) + 1 sits at q = 1 of group 5 + slot
  };

  // ---------------- M sequential feature updates: covariance side ----------------
  int smp = 0, pp = 0;
  int m = res_next_valid(S, 0);
  // hand the zeta-zeta 2x2 of every diagonal block to the service lanes (they keep it current from here on)
#pragma unroll
  for (int ia = 0; ia < 2; ia++)
#pragma unroll
    for (int ic = 0; ic < 3; ic++)
      if (vI[ia] && 2 * A_ + ia == 3 * C_ + ic) {
        *reinterpret_cast<double2*>(S.Pd + 4 * Ib[ia]) = make_double2(pb[ia][ic][0], pb[ia][ic][1]);
        *reinterpret_cast<double2*>(S.Pd + 4 * Ib[ia] + 2) = make_double2(pb[ia][ic][3], pb[ia][ic][4]);
      }
  if (m < S.M) {
    apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
    extract(S.mslot[m], 0);
    if (has_item && g_ == 5 + S.mslot[m]) {   // body rows of the first measurement's columns, from the LDS strip
      const double2 a0 = *reinterpret_cast<const double2*>(sp[1]), a1 = *reinterpret_cast<const double2*>(sp[1] + 2);
      const double2 b0 = *reinterpret_cast<const double2*>(sp[2]), b1 = *reinterpret_cast<const double2*>(sp[2] + 2);
      double* d = S.Praw + 2 * (4 * cg_);
      *reinterpret_cast<double2*>(d + 0) = make_double2(a0.x, b0.x);
      *reinterpret_cast<double2*>(d + 2) = make_double2(a0.y, b0.y);
      *reinterpret_cast<double2*>(d + 4) = make_double2(a1.x, b1.x);
      *reinterpret_cast<double2*>(d + 6) = make_double2(a1.y, b1.y);
    }
  }
  RES_STAMP(S, tid == 0, 71);
  __syncthreads();  // Bp
  __syncthreads();  // B1 : the service published the first measurement's {Hb, res, S^-1, verdict}
  int it_ = 0;
  int2 sq = S.mseq[min(m, MCAP - 1)];
  const int irow = min(tid, n - 1);
  const double* kP = S.Kt;
  const double* wP = S.Wt;
  // LDS addresses of this thread's gain rows (loop constants)
  const double* kT = kP + 2 * (16 + 6 * min(A_, (N - 1) / 2));       // rows of features 2A, 2A+1 (6 rows)
  const double* wT = wP + 2 * (16 + 9 * min(C_, (N - 1) / 3));       // rows of features 3C..3C+2 (9 rows)
  while (m < S.M) {
    const int mnext = sq.x, slot_next = sq.y;
    const double* mbx = S.sm + 16 * smp;
    // every LDS read of the gain phase is issued up front and unconditionally (one latency, not a chain of dependent ones)
    const double2 hA = *reinterpret_cast<const double2*>(mbx + 0), hB = *reinterpret_cast<const double2*>(mbx + 2);
    const double2 sA = *reinterpret_cast<const double2*>(mbx + 6), sB = *reinterpret_cast<const double2*>(mbx + 8);
    const double2 pr = *reinterpret_cast<const double2*>(S.Praw + pp * 2 * n + 2 * irow);
    const double gflag = mbx[10];
    const double fixpending = S.sm[40 + (par ^ 1)];   // posted before B1 by the service wave: read it ahead of B2
    sq = S.mseq[min(mnext, MCAP - 1)];                 // next iteration's table entry (static data)
    // gain row i = tid:  W_i = P[i, j0:j0+2] Hb^T,  K_i = W_i S^-1   (vi_ekf_meas.cpp:241)
    const double w0 = pr.x * hA.x + pr.y * hA.y, w1 = pr.x * hB.x + pr.y * hB.y;
    const double k0 = w0 * sA.x + w1 * sB.x, k1 = w0 * sA.y + w1 * sB.y;
    const bool gated = gflag != 0.0;
    if (tid < n && !gated && !(S.dbg & 16)) {
      *reinterpret_cast<double2*>(S.Wt + 2 * irow) = make_double2(w0, w1);
      *reinterpret_cast<double2*>(S.Kt + 2 * irow) = make_double2(k0, k1);
      // a NaN in H makes every K row NaN (0 * NaN = NaN), so testing K covers the reference's H test (:247) as well
      if (k0 != k0 || k1 != k1) S.sm[44 + smp] = 1.0;
    }
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 0);
    __syncthreads();  // B2 : gain vectors Kt / Wt are in LDS
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 1);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 0);
    const bool nan = S.sm[44 + smp] != 0.0;
    apply_fixes(par ^ 1, fixpending);
    const bool run = !gated && !nan && !(S.dbg & 1);   // not gated, no NaN guard
    if (run) {
      // Operand reads are software-pipelined by hand, one block column ahead: all row reads in flight at once would need
      // ~90 registers on top of the 108 that hold the tile (the workgroup's 5 waves leave 256 per lane), and a spill to
      // scratch costs a memory round trip per use.
      double2 kI[2][3], w0[3], w1[3];
#pragma unroll
      for (int ia = 0; ia < 2; ia++)
#pragma unroll
        for (int r = 0; r < 3; r++) kI[ia][r] = *reinterpret_cast<const double2*>(kT + 2 * (3 * ia + r));
#pragma unroll
      for (int s = 0; s < 3; s++) w0[s] = *reinterpret_cast<const double2*>(wT + 2 * s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 3; s++) w1[s] = *reinterpret_cast<const double2*>(wT + 2 * (3 + s));
      tile_sweep_col<0>(pb, kI, w0, Lff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 3; s++) w0[s] = *reinterpret_cast<const double2*>(wT + 2 * (6 + s));
      tile_sweep_col<1>(pb, kI, w1, Lff);
      __builtin_amdgcn_sched_barrier(0);
      tile_sweep_col<2>(pb, kI, w0, Lff);
      RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- body strip item (LDS resident): its reads are issued here and land while the tile's columns are published
    double2 kS[3], wS[4], pv[3][2];
    double lr[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int ri = min(3 * g_ + q, n - 1);
      kS[q] = *reinterpret_cast<const double2*>(kP + 2 * ri);
      lr[q] = partial ? S.lam[ri] : 1.0;   // lambda = 1 everywhere gives Lambda = 1 + 1 - 1 = 1: no select per element
      pv[q][0] = *reinterpret_cast<const double2*>(sp[q]);
      pv[q][1] = *reinterpret_cast<const double2*>(sp[q] + 2);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) wS[j] = *reinterpret_cast<const double2*>(wP + 2 * (4 * cg_ + j));
    const double2 lcA = *reinterpret_cast<const double2*>(S.lam + 4 * cg_), lcB = *reinterpret_cast<const double2*>(S.lam + 4 * cg_ + 2);
    par ^= 1;
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 2);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 1);
    if (slot_next >= 0 && !(S.dbg & 8)) extract(slot_next, pp ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    {
      // rows and columns of the strip both carry their own lambda (vi_ekf.cpp:83,146)
      const double lc[4] = {partial ? lcA.x : 1.0, partial ? lcA.y : 1.0, partial ? lcB.x : 1.0, partial ? lcB.y : 1.0};
#pragma unroll
      for (int q = 0; q < 3; q++) {
        double e[4] = {pv[q][0].x, pv[q][0].y, pv[q][1].x, pv[q][1].y};
        if (run) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const double L = fma(-lr[q], lc[j], lr[q] + lc[j]);
            const double t = fma(kS[q].y, wS[j].y, kS[q].x * wS[j].x);
            e[j] = fma(-L, t, e[j]);
          }
          if (sv[q]) {
            *reinterpret_cast<double2*>(sp[q]) = make_double2(e[0], e[1]);
            *reinterpret_cast<double2*>(sp[q] + 2) = make_double2(e[2], e[3]);
          }
        }
        pv[q][0] = make_double2(e[0], e[1]); pv[q][1] = make_double2(e[2], e[3]);
      }
      // the next measurement's body rows P[k][j0 + c] = P[j0 + c][k]: row j0 = 16 + 3 slot sits at q = 1 of group 5 + slot
      if (has_item && slot_next >= 0 && g_ == 5 + slot_next) {
        double* d = S.Praw + (pp ^ 1) * 2 * n + 2 * (4 * cg_);
        *reinterpret_cast<double2*>(d + 0) = make_double2(pv[1][0].x, pv[2][0].x);
        *reinterpret_cast<double2*>(d + 2) = make_double2(pv[1][0].y, pv[2][0].y);
        *reinterpret_cast<double2*>(d + 4) = make_double2(pv[1][1].x, pv[2][1].x);
        *reinterpret_cast<double2*>(d + 6) = make_double2(pv[1][1].y, pv[2][1].y);
      }
      RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 1);
    }
    // NOTE: a fix_depth edit touches P(rho,rho) only, never the zeta columns just extracted
    pp ^= 1;
    smp ^= 1;
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 3);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 2);
    __syncthreads();  // B1
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 3);
    it_++;
    m = mnext;
  }
  apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
  RES_STAMP(S, tid == 0, 72);
  __syncthreads();  // B5

}

__global__ __launch_bounds__((TILE_NW + 1) * 64) void k_step_tile(StreamArgs a, int, int, int do_prop,
                                                                 const double* __restrict__ u_all,
                                                                 const double* __restrict__ dt_all,
                                                                 const double* __restrict__ z_all,
                                                                 const int* __restrict__ slot_all, int M, int m_stride,
                                                                 const double* __restrict__ R_all, long r_stride_b,
                                                                 long r_stride_m, int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = (TILE_NW + 1) * 64, TW = TILE_NW * 64;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.B) return;
  ResShared S;
  res_prologue<T>(a, S, smem, do_prop, dt_all, z_all, slot_all, M, m_stride, R_all, r_stride_b, r_stride_m, result_all);
  if (tid >= TW) {
    res_service(a, S, tid - TW, u_all, result_all);
  } else {
    const TileId tl = tile_id(a, a.n, tid);
    double pb[2][3][9];   // pb[ia][ic][r*3+s] = P[16+3I+r][16+3J+s]
    tile_load<TW>(a, S, tid, tl, pb);
    tile_worker<TW>(a, S, tid, tl, pb);
    const TileId ts = tile_id(a, a.n, opaque(tid));   // (re-derived: not kept live across the update loop)
    tile_store<TW>(a, S, tid, ts, pb);
    RES_STAMP(S, tid == 0, 73);
  }
}

}  // namespace viekf
