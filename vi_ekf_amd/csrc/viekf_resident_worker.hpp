// viekf_resident_worker.hpp -- resident family: the worker waves (blocks of P in registers: load, propagate contraction,
// rank-2 sweeps, column extraction, store).
#pragma once
#include "viekf_resident_prop.hpp"

namespace viekf {


// Store of P, cooperative part.  P is symmetric and only its LOWER triangle (and the diagonal 3x3 blocks) is stored: the
// part above the diagonal is left stale -- the host mirrors it up before anything that reads all of P (ensure_full_P; the
// fused kernel itself loads the lower triangle only), which halves the HBM bytes of the store phase.  The workers scatter
// their 3x3 blocks, each in its lower-triangle orientation, into an LDS image of a chunk of feature columns -- the Z region
// and its neighbours, free after the update loop; [column][rows rb .. n) -- and the whole workgroup streams the chunk out with
// lanes along the rows: every wave instruction writes up to 1 KB contiguous instead of 64 different cache lines (the direct
// 8-byte block stores were bound by the texture path's one line per clock: 25 k clk per step).  A chunk's image starts at
// its first column's diagonal, so later chunks hold more columns (N = 50: 13 + 17 + 20 features).
struct StoreChunk { int f0, f1, rb, h; };   // features [f0, f1), image rows rb .. rb + h - 1 (rb, h even)
__device__ __forceinline__ bool store_chunk_at(int f0, int N, int n, int img_len, StoreChunk& c) {
  if (f0 >= N) return false;
  c.f0 = f0;
  c.rb = (16 + 3 * f0) & ~1;
  c.h = ((n + 1) & ~1) - c.rb;
  c.f1 = min(N, f0 + max(1, img_len / (3 * c.h)));
  return true;
}
template <int T>
__device__ __forceinline__ void res_store_chunk(const StreamArgs& a, const ResShared& S, const StoreChunk& c, int tid) {
  const int n = S.n, ld = a.ld;
  double* P = a.P_out + S.so * n * ld;
  const double* img = S.Z;
  const int ncol = 3 * (c.f1 - c.f0), lane = tid & 63, w = tid >> 6;
  constexpr int NWV = T / 64;
  if ((n & 1) == 0) {   // even n: row pairs are 16-byte aligned in the image and in P (ld is even)
#pragma unroll 2
    for (int col = w; col < ncol; col += NWV) {
      const int j = 16 + 3 * c.f0 + col;
      const double* src = img + col * c.h - c.rb;
      for (int i0 = c.rb; i0 < n; i0 += 256) {
        double2 v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) v[u] = lds_ld2(src + min(i0 + 2 * (lane + 64 * u), n - 2));
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int i = i0 + 2 * (lane + 64 * u);
          if (i < n && i + 1 >= j) *reinterpret_cast<double2*>(P + i + (long)j * ld) = v[u];   // rows on and below the diagonal
        }
      }
    }
  } else {
#pragma unroll 2
    for (int col = w; col < ncol; col += NWV) {
      const int j = 16 + 3 * c.f0 + col;
      const double* src = img + col * c.h - c.rb;
      for (int i = c.rb + lane; i < n; i += 64)
        if (i >= j) P[i + (long)j * ld] = src[i];
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
  // (the caller count is a multiple of 8, so a caller's body-column pair j2 is the same for all its items: W of those two
  //  columns and their Lambda entries are read once, not per item)
  const int j2 = ((first + id) & 7) * 2;
  double2 cw0 = {}, cw1 = {}, cl[3] = {};
  if (run) {
    cw0 = lds_ld2(wP + 2 * j2);
    cw1 = lds_ld2(wP + 2 * j2 + 2);
#pragma unroll
    for (int q = 0; q < 3; q++) cl[q] = lds_ld2(S.Lbc + 16 * q + j2);
  }
#pragma unroll 1
  for (int item = first + id; item < last; item += nthreads) {
    const int g = item >> 3;
    double2 cpv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) cpv[q] = lds_ld2(Pbc + (3 * g + q) * 16 + j2);
    if (run) {
      double2 cki[3];
#pragma unroll
      for (int q = 0; q < 3; q++) cki[q] = lds_ld2(kP + 2 * (16 + 3 * g + q));
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
// (The service wave takes no share of the body-column items: with the tile map its chain is the longest wave of an update and
//  every share > 0 measured slower -- N = 50, B = 1024: 0 / N / 2 N items -> 0.348 / 0.347 / 0.353 ms per step.)
template <int RB, int TW, bool MP, int T = TW + 64, bool ZU = false>
__device__ __forceinline__ void res_worker(const StreamArgs& a, const ResShared& S, int tid) {
  const int N = S.N, n = S.n, ld = a.ld, nf = S.nf, len = S.len;
  double* P = a.P + S.si * n * ld;
  // SYMMETRIC ownership: of each unordered pair of feature blocks {I,J} only one is kept (I >= J); which thread keeps it in
  // which of its RB slots is a table built by the host (build_resmap, viekf_capi.hip: 8 x 8 tiles of blocks per (slot, wave)
  // group, so that the column pair of one feature is published from few groups).  Slot a = 0 of the threads t < N is the
  // diagonal block (t, t).
  const int tid_ = tid;
  constexpr int NWV = TW / 64;
  const DevParams& prm = *a.dp;
  double* Pbc = S.Pbc;   // [nf][16]  P[16+row][k]: the body columns of P live in LDS for the whole step
  double* Pbb = S.Pbb;   // [16][16]  row-major P_bb
  // (P[body rows, feature cols] is NOT kept: P is symmetric up to rounding, the mirror is written at store time)
  const int* __restrict__ resmap = a.resmap;
  auto blk = [&](int t, int ia, int& I, int& J) -> bool {   // block ia of thread t; false = not owned (then I = J = 0: every
    const int e = resmap[ia * TW + t];                       // LDS / global read stays in range, results are never stored)
    I = e & 0xff;
    J = (e >> 8) & 0xff;
    return (e >> 16) != 0;
  };
  const bool own_diag = tid_ < N;   // slot 0 of this thread is the diagonal block (I, I), I = tid

  double pb[RB][9];   // pb[a][r*3+s] = P[16+3I+r][16+3J+s]
  {
    const int tq = opaque(tid_);
#pragma unroll
    for (int ia = 0; ia < RB; ia++) {
      int I, J;
      blk(tq, ia, I, J);
      // only the lower triangle of P is valid in memory (see the store): a block above the diagonal is read as the transpose
      // of its mirror, a diagonal block takes its lower triangle for both
      const bool up = I < J;
      const int br = 16 + 3 * (up ? J : I), bc = 16 + 3 * (up ? I : J);
      const double* pu = P + (br + (long)bc * ld);
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const int rr = (I == J) ? max(r, s) : (up ? s : r), cc = (I == J) ? min(r, s) : (up ? r : s);
          pb[ia][r * 3 + s] = pu[rr + (long)cc * ld];
        }
    }
    // body columns -> LDS (coalesced along rows)
    for (int e = tid; e < nf * 16; e += TW) {
      const int k = e / nf, row = e - k * nf;
      Pbc[row * 16 + k] = P[(16 + row) + (long)k * ld];
    }
    // (the body block is kept EXACTLY symmetric, like every other part of P here -- see sym_diag below: both copies of a pair
    //  are loaded from the lower triangle)
    for (int e = tid; e < 256; e += TW) {
      const int r = e & 15, c = e >> 4;
      Pbb[r * 16 + c] = P[max(r, c) + (long)min(r, c) * ld];
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
  constexpr int NSV = (T - TW) / 64;   // service waves: with one, the second wave's flag words are not read at all
  auto fix_word = [&](int mb) -> double { return (NSV > 1) ? S.sm[40 + mb] + S.sm[36 + mb] : S.sm[40 + mb]; };
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
    int zoff[RB];
    {
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
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        const double* zi = Z + (zoff[ia] & 0xffff) + 2 * k;
        const double* zj = Z + (zoff[ia] >> 16) + 2 * k;
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
    if (MP && kp + 1 < nkp && own_diag && tid_ < len && fix_word(par ^ 1) != 0.0) {   // this propagate's fix_depth edits of P(rho,rho), before the next
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
  // Lambda of the body block's two elements of this thread's task (one task per thread where 128 threads share the 128 tasks):
  // constants of the launch, not re-formed from three LDS words in every update
  auto bb_mask = [&](int r, int c) -> double { const double lr_ = S.lam[r], lc_ = S.lam[c]; return partial ? (lc_ + lr_ - lr_ * lc_) : 1.0; };
  double bbL0 = 1.0, bbL1 = 1.0;
  {
    constexpr int BBT0 = (TW >= 128) ? 128 : TW;
    const int ib0 = tid - (TW - BBT0);
    if (ib0 >= 0 && ib0 < 128) { bbL0 = bb_mask(ib0 >> 3, (ib0 & 7) * 2); bbL1 = bb_mask(ib0 >> 3, (ib0 & 7) * 2 + 1); }
  }
  // applies the pending fix_depth covariance edits of mailbox `mb` to the owned diagonal blocks (diagonal d = 0)
  auto apply_fixes = [&](int mb, double pending) {
    if (pending == 0.0) return;   // nothing posted (the common case); the flag word was read ahead of the barrier
    if (own_diag && tid_ < len) {
      VIEKF_COLD_BEGIN();
      const int I = tid_;
      const double ad = S.fixadd[mb * N + I], st = S.fixset[mb * N + I];
      if (ad != 0.0) { pb[0][8] += ad; S.fixadd[mb * N + I] = 0.0; }
      if (st != 0.0) { pb[0][8] = p0rr; S.fixset[mb * N + I] = 0.0; }
      VIEKF_COLD_END();
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
  int2 sq = S.mseq[min(m, S.mcap - 1)];
  if (m < S.M) {
    apply_fixes(par ^ 1, fix_word(par ^ 1));
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
    RES_MARK("worker.phase_top");
    const double* kP = (cnt & 1) ? S.Z : S.Kt;
    const double* wP = kP + 2 * n;
    __builtin_amdgcn_s_setprio(1);
    const double fixpending = fix_word(par ^ 1);   // posted before the barrier by the service wave
    const double gflag = S.sm[50 + (cnt & 1)];          // gate verdict of this measurement (service, previous phase)
    const double nanw = (NSV > 1) ? S.sm[44 + cnt % 3] + S.sm[52 + cnt % 3] : S.sm[44 + cnt % 3];   // (the second word: a second service wave's rows)
    sq = S.mseq[min(mnext, S.mcap - 1)];                 // next iteration's table entry (static data)
    RES_STAMP(S, tid == 0 && it_ < 8, 80 + 4 * it_ + 1);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 0);
    const int it = tid;
    // ---- (1) feature/feature blocks (registers).  The operand rows of GB blocks are in flight together: all of them with
    //      few blocks per thread; two at a time with many, where holding every block's rows would not fit the register file
    constexpr int GB = (RB <= 4) ? RB : 2;
    RES_MARK("worker.block_sweep");
    const bool gated = gflag != 0.0;
    const bool run = !gated && nanw == 0.0 && !RES_ABLATE(S, 1);   // not gated, no NaN guard
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
              if (ZU && !(r == 2 && s == 2)) {   // Lambda = 1: two multiply-adds instead of three operations
                pb[ia][r * 3 + s] = fma(-kI[ig][r].y, wJ[ig][s].y, fma(-kI[ig][r].x, wJ[ig][s].x, pb[ia][r * 3 + s]));
              } else {
                const double t = fma(kI[ig][r].y, wJ[ig][s].y, kI[ig][r].x * wJ[ig][s].x);
                pb[ia][r * 3 + s] = fma(-Lff[r * 3 + s], t, pb[ia][r * 3 + s]);
              }
            }
        }
      }
      group_fence<(RB > GB)>();
    }
    sym_diag();
    RES_MARK("worker.column_extraction");
    RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 0);
    // ---- (2) the raw feature rows of the measurement after next (a fix_depth edit touches P(rho,rho) only, never these
    //      columns), into the buffer the service wave is not reading in this phase
    double* rawdst = S.Praw + (cnt & 1) * 2 * n;
    if (sq.y >= 0 && !RES_ABLATE(S, 4) && !(RES_ABLATE(S, 8) && (cnt & 1))) extract_cols(sq.y, rawdst);   // (bit 8: every other phase -- timing of a pairwise hand-over)
    __builtin_amdgcn_s_setprio(0);
    RES_STAMP(S, tid == 0 && it_ < 8, 160 + 4 * it_ + 1);
    RES_STAMP(S, (tid & 63) == 0 && it_ == 3, 192 + 4 * (tid >> 6) + 2);
    __builtin_amdgcn_sched_barrier(0);
    RES_MARK("worker.body_items");
    // ---- (3) body columns, in LDS (res_body_items).  Wave 0 takes the last places in the item order, so that the incomplete
    //      final round falls to the other waves: it is the longest of the workers whenever its SIMD-mate is the other workgroup's
    //      service wave.
    res_body_items(S, kP, run, (it >= 64) ? it - 64 : it + TW - 64, TW, 0, 8 * N, sq.y, rawdst);
    RES_MARK("worker.body_block");
    if (run) {   // body block: 2 adjacent elements per task, 128 tasks on the top 128 threads (a single worker wave: two tasks
                 // per thread).  Element (r, c) and its mirror (c, r) are owned by different tasks; both form
                 // p - L (K_lo . W_hi), lo = min(r, c), hi = max(r, c)  from their own (equal) copies, so the block stays exactly
                 // symmetric without any exchange.
      constexpr int BBT = (TW >= 128) ? 128 : TW;
#pragma unroll
      for (int ib = it - (TW - BBT); ib >= 0 && ib < 128; ib += BBT) {
        const int br = ib >> 3, bc2 = (ib & 7) * 2;
        double2 bpv = *reinterpret_cast<double2*>(Pbb + br * 16 + bc2);
        const double2 kr = lds_ld2(kP + 2 * br), wr = lds_ld2(wP + 2 * br);
        const double2 k0 = lds_ld2(kP + 2 * bc2), w0 = lds_ld2(wP + 2 * bc2);
        const double2 k1 = lds_ld2(kP + 2 * bc2 + 2), w1 = lds_ld2(wP + 2 * bc2 + 2);
        const double L0 = (BBT == 128) ? bbL0 : bb_mask(br, bc2), L1 = (BBT == 128) ? bbL1 : bb_mask(br, bc2 + 1);
        const bool up0 = br <= bc2, up1 = br <= bc2 + 1;
        const double2 ka = up0 ? kr : k0, wa = up0 ? w0 : wr;      // (K_lo, W_hi) of element (br, bc2)
        const double2 kb = up1 ? kr : k1, wb = up1 ? w1 : wr;      // ... of element (br, bc2 + 1)
        bpv.x = fma(-L0, fma(ka.y, wa.y, ka.x * wa.x), bpv.x);
        bpv.y = fma(-L1, fma(kb.y, wb.y, kb.x * wb.x), bpv.y);
        *reinterpret_cast<double2*>(Pbb + br * 16 + bc2) = bpv;
      }
    }
    RES_MARK("worker.phase_tail");
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
  RES_MARK("worker.loop_end");
  apply_fixes(par ^ 1, fix_word(par ^ 1));
  RES_STAMP(S, tid == 0, 72);
  __syncthreads();  // B5 : every sweep of the LDS-resident body columns is finished

  // ---------------- store ----------------
  // (indices re-derived from opaque copies: otherwise the load addresses are kept alive -- spilled -- all kernel long)
  {
    P = a.P_out + S.so * n * ld;   // in place, or the next slot of the history ring
    for (int e = opaque(tid); e < nf * 16; e += TW) {     // body columns, coalesced along rows
      const int k = e / nf, row = e - k * nf;
      P[(16 + row) + (long)k * ld] = Pbc[row * 16 + k];
    }
    for (int e = opaque(tid); e < 256; e += TW) P[(e >> 4) + (long)(e & 15) * ld] = Pbb[e];
    double* img = S.Z;
    const int gtid = threadIdx.x;
    RES_STAMP(S, tid == 0, 224);
    StoreChunk sc;
    int ch = 0;
    for (int f0 = 0; store_chunk_at(f0, N, n, S.img_len, sc); f0 = sc.f1, ch++) {
      const int f1 = sc.f1;
      const int tq = opaque(tid_);
#pragma unroll
      for (int ia = 0; ia < RB; ia++) {
        int I, J;
        if (blk(tq, ia, I, J)) {
          // the block in its lower-triangle orientation: (I, J) itself for I >= J, its mirror (J, I) otherwise
          if (I >= J && J >= f0 && J < f1) {                // rows of feature I in the columns of feature J
            double* d = img + (3 * (J - f0)) * sc.h + (16 + 3 * I - sc.rb);
#pragma unroll
            for (int s = 0; s < 3; s++)
#pragma unroll
              for (int r = 0; r < 3; r++) d[s * sc.h + r] = pb[ia][r * 3 + s];
          }
          if (I < J && I >= f0 && I < f1) {                 // rows of feature J in the columns of feature I
            double* d = img + (3 * (I - f0)) * sc.h + (16 + 3 * J - sc.rb);
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int s = 0; s < 3; s++) d[r * sc.h + s] = pb[ia][r * 3 + s];
          }
        }
      }
      RES_STAMP(S, tid == 0 && ch < 3, 225 + 4 * ch);
      __syncthreads();   // S1: the chunk image is complete
      RES_STAMP(S, tid == 0 && ch < 3, 226 + 4 * ch);
      res_store_chunk<T>(a, S, sc, gtid);
      RES_STAMP(S, tid == 0 && ch < 3, 227 + 4 * ch);
      __syncthreads();   // S2: the image may be overwritten
      RES_STAMP(S, tid == 0 && ch < 3, 228 + 4 * ch);
    }
  }
  RES_STAMP(S, tid == 0, 73);
}


}  // namespace viekf
