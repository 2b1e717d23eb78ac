// viekf_tiles_worker.hpp -- tile family: the worker waves (16 x 16 tiles of P in matrix-core accumulator layout: load, propagate,
// rank-4 MFMA sweeps, column extraction, store).  Overview: viekf_tiles_common.hpp.
#pragma once
#include <utility>

#include "viekf_tiles_common.hpp"

namespace viekf {

// Tile -> (wave, slot), fixed at COMPILE time: tile (TI, TJ), TI >= TJ, belongs to wave (TI + TJ) mod NW -- the NT tiles that
// hold one feature's rows and columns (tile row T and tile column T) then spread evenly over the waves, and every update
// extracts the next column pair from all of them; a wave's slots are ordered by (TI, TJ).  Each worker wave runs its own
// instantiation of the code below (tile_worker<.., W>), in which every tile index is a constant: addresses become immediates,
// the per-tile tests scalar compares against constants (with the map as run-time data -- the first version of this file -- the
// same loops compiled to 9,000 instructions per update and 1,500 spilled registers).
template <int NT, int NW>
struct TileMap {
  static constexpr int count(int w) {
    int c = 0;
    for (int TI = 0; TI < NT; TI++)
      for (int TJ = 0; TJ <= TI; TJ++)
        if ((TI + TJ) % NW == w) c++;
    return c;
  }
  static constexpr int max_count() {
    int m = 0;
    for (int w = 0; w < NW; w++) m = count(w) > m ? count(w) : m;
    return m;
  }
  static constexpr int find(int w, int s, bool want_i) {   // slot s of wave w -> TI or TJ (-1: the wave has fewer tiles)
    int c = 0;
    for (int TI = 0; TI < NT; TI++)
      for (int TJ = 0; TJ <= TI; TJ++)
        if ((TI + TJ) % NW == w) {
          if (c == s) return want_i ? TI : TJ;
          c++;
        }
    return -1;
  }
  static constexpr int ti(int w, int s) { return find(w, s, true); }
  static constexpr int tj(int w, int s) { return find(w, s, false); }
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int CNT, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, CNT>{}, f); }

// MFMA operand conventions (v_mfma_f64_16x16x4_f64; MI355X_MICROARCH.md "Matrix cores"):  D = A B + C  with, for lane l,
//   A: one double = A[i = l & 15][k = l >> 4]      B: one double = B[k = l >> 4][j = l & 15]
//   C / D: four doubles, register r = D[i = (l >> 4) + 4 r][j = l & 15]
// A tile register X[r] holds P[prow(TI, l & 15)][prow(TJ, (l >> 4) + 4 r)]: D's column index j runs along the ROWS of P inside
// tile row TI, D's row index i along the columns inside tile column TJ.  So the A operand carries the TJ-side factor and the
// B operand the TI-side factor, both indexed by l & 15, component k = l >> 4.
// PAIR: two filters share a 512-thread workgroup (k_step_tiles_pair) and run their update loops HALF A PHASE out of step: every
// update has two barriers -- [operands, MFMAs, next column pair] | [extraction] -- and filter 1 (half = 1) starts one barrier late,
// so one filter's matrix instructions run beside the other's extraction and scalar work on every SIMD they share.  (Two
// independent workgroups per CU fall into step instead: their MFMA bursts collide on the SIMD's one matrix pipe, then both
// extract while it idles -- measured 5,500 clocks per update pair against 2 x 1,408 of matrix work.)  trips = updates the longer
// of the two filters runs (both loop that often; a filter that is done idles through the barriers).
template <int NT, int NW, int W, bool MP, bool PAIR = false>
__device__ __forceinline__ void tile_worker(const StreamArgs& a, const TileShared& S, int tid, int half = 0, int trips = 0) {
  typedef TileMap<NT, NW> Map;
  constexpr int TW = NW * 64, TPW = Map::max_count(), CNT = Map::count(W);
  const int N = S.N, n = S.n, ld = a.ld, nf = S.nf, len = S.len, NQ = S.NQ;
  const int lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const DevParams& prm = *a.dp;
  double* P = a.P + S.si * n * ld;
  double* Pbc = S.Pbc;   // [nf][16]  P[16+row][k]: the body columns, in LDS during load and propagate
  double* Pbb = S.Pbb;   // [16][16]  row-major P_bb
  (void)TPW;

  // ---------------- load: feature/feature tiles from the lower triangle of P (a diagonal tile mirrors it), body columns -> LDS
  // Addressing: element (a = lg + 4 r, b = l15) of tile (TI, TJ) is P[rowbase(TI) + b + (rowbase(TJ) + a) ld]: the lane's part
  // l15 + lg ld is ONE 32-bit register for every load and store of the kernel, the tile's part and 4 r ld go into the scalar base
  // (four 64-bit scalar adds per tile instead of a 64-bit vector address per element: the first version of this file spent 76 k
  // clocks in its load phase and 62 k in its store, most of it address arithmetic and its spills).
  auto rowbase = [](int Tt) { return Tt == 0 ? 0 : 16 + 15 * (Tt - 1); };
  // (the lane parts are re-derived where they are used -- load and store, the two ends of the kernel -- from a laundered lane
  //  index: held live across the update loop they were spilled and re-loaded in every phase)
  struct LaneOff { unsigned vo, vom[4]; bool low[4]; };
  auto lane_off = [&]() {
    LaneOff o;
    const int ln = opaque(lane), b15 = ln & 15, g4 = ln >> 4;
    o.vo = (unsigned)b15 + (unsigned)g4 * (unsigned)ld;
    // a diagonal tile reads (and a store writes) only P's lower triangle: b >= a; the upper half comes from the mirrored element
#pragma unroll
    for (int r = 0; r < 4; r++) {
      o.low[r] = b15 >= g4 + 4 * r;
      o.vom[r] = o.low[r] ? (unsigned)b15 + (unsigned)(g4 + 4 * r) * (unsigned)ld : (unsigned)(g4 + 4 * r) + (unsigned)b15 * (unsigned)ld;
    }
    return o;
  };
  // validity of this lane's row b = l15 / column a = lg + 4 r inside a FULL feature tile (index 15 is the pad) and inside the LAST
  // one (NT - 1: features past N do not exist); body tile: everything valid
  auto row_ok = [&](int Tt) { return Tt == 0 ? true : (Tt == NT - 1 ? (l15 < 15 && 15 * (NT - 2) + l15 < nf) : l15 < 15); };
  auto col_ok = [&](int Tt, int r) {
    const int aa = lg + 4 * r;
    return Tt == 0 ? true : (Tt == NT - 1 ? (aa < 15 && 15 * (NT - 2) + aa < nf) : aa < 15);
  };
  v4f64 X[CNT];
  const LaneOff lo_ = lane_off();
  static_for<CNT>([&](auto sc) {
    constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
    v4f64 x = {0.0, 0.0, 0.0, 0.0};
    if (TJ >= 1) {
      const double* base = P + rowbase(TI) + (long)rowbase(TJ) * ld;   // (wave-uniform: scalar registers)
      if (row_ok(TI)) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (TI == TJ) { if (col_ok(TJ, r)) x[r] = base[lo_.vom[r]]; }
          else if (r < 3 && TJ != NT - 1) x[r] = (base + 4L * r * ld)[lo_.vo];   // (a < 15 for r < 3: no test)
          else if (col_ok(TJ, r)) x[r] = (base + 4L * r * ld)[lo_.vo];
        }
      }
    }
    X[s] = x;
    if ((s & 3) == 3) group_fence<true>();   // (four tiles' loads in flight; all 88 at once would hold 88 addresses live)
  });
  for (int e = tid; e < nf * 16; e += TW) {   // body columns -> LDS (coalesced along rows)
    const int k = e / nf, row = e - k * nf;
    Pbc[row * 16 + k] = P[(16 + row) + (long)k * ld];
  }
  for (int e = tid; e < 256; e += TW) {       // body block, both copies of a pair from the lower triangle
    const int r = e & 15, c = e >> 4;
    Pbb[r * 16 + c] = P[max(r, c) + (long)min(r, c) * ld];
  }
  int par = 0;  // fix_depth mailbox parity (mirrors the service wave)
  const bool st0 = (W == 0) && lane == 0;   // (-DVIEKF_STAMPS diagnostic build: who stamps)
  (void)st0;
  RES_STAMP(S, st0, 64);
  const double p0rr = uniform_f64(prm.P0_feat[2]);
  // applies the pending fix_depth covariance edits of mailbox `mb` (vi_ekf_helper.cpp:141-151: P(rho,rho) += err^2 / = P0) to the
  // diagonal tiles this wave holds: element a = b = 3 j + 2 of tile (T, T) is feature 5 (T - 1) + j
  auto apply_fixes = [&](int mb, double pending) {
    if (pending == 0.0) return;   // nothing posted (the common case)
    static_for<CNT>([&](auto sc) {
      constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
      if (TI != TJ || TI == 0) return;
      const int j = l15 / 3, f = 5 * (TI - 1) + j;
      const bool mine = (l15 == 3 * j + 2) && (lg == (l15 & 3)) && f < len;
      if (mine) {
        const double ad = S.fixadd[mb * N + f], st = S.fixset[mb * N + f];
#pragma unroll
        for (int r = 0; r < 4; r++)
          if ((l15 >> 2) == r) {
            if (ad != 0.0) X[s][r] += ad;
            if (st != 0.0) X[s][r] = p0rr;
          }
        if (ad != 0.0) S.fixadd[mb * N + f] = 0.0;
        if (st != 0.0) S.fixset[mb * N + f] = 0.0;
      }
    });
  };
  __syncthreads();  // B0
  RES_STAMP(S, st0, 65);

  // ---------------- propagate(s)
  const int nkp = MP ? S.kp : 1;
  if (S.do_prop) {
    // per-lane parts of the operand addresses:  Dblk_T[row = l15][col = lg + 4 s4]  (block diagonal of the five Phi_ff of tile T:
    // phiff + 45 (T - 1) + 9 (l15 / 3) + 3 (l15 % 3) + col % 3) and the slot of a Z record that k = 4 k6 + lg of the K = 24
    // coupling reads on either side (viekf_resident_common.hpp); the tile's part of an address is an immediate
    const int r3 = l15 / 3, rm = l15 - 3 * r3;
    for (int kp = 0; kp < nkp; kp++) {
      res_prop_setup<TW>(a, S, tid);   // (B1p, B2p, B2q inside)
      RES_STAMP(S, st0, 66);
      __syncthreads();  // B3p
      RES_STAMP(S, st0, 67);
      const double* Z = S.Z;
      const double* phiff = S.phiff;
      auto d_op = [&](auto Tc, int s4) -> double {   // Dblk_T[l15][lg + 4 s4]
        constexpr int Tt = decltype(Tc)::value;
        const int c = lg + 4 * s4;
        const bool ok = (l15 < 15) && (c < 15) && (c / 3 == r3) && (5 * (Tt - 1) + r3 < N);
        const double v = phiff[45 * (Tt - 1) + (ok ? 9 * r3 + 3 * rm + c % 3 : 0)];
        return ok ? v : 0.0;
      };
      auto z_op = [&](auto Tc, int k6, bool iside) -> double {
        constexpr int Tt = decltype(Tc)::value;
        const int kk = 4 * k6 + lg;
        int o = (kk < ZK) ? (2 * kk + 1) : ((kk < 2 * ZK) ? 2 * (kk - ZK) : kk);   // TJ side: D | Ut | Gs
        if (iside && kk < 2 * ZK) o ^= 1;                                           // TI side: Ut | D | Gs
        const bool ok = l15 < 15 && 15 * (Tt - 1) + l15 < nf;
        const double v = Z[(15 * (Tt - 1) + (ok ? l15 : 0)) * ZS + o];
        return ok ? v : 0.0;
      };
      static_for<CNT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
        if (TJ < 1) return;       // (body tiles: in LDS during the propagate, res_prop_body)
        std::integral_constant<int, TI> ci;
        std::integral_constant<int, TJ> cj;
        // X' = D_J X D_I^T + [D_J | Ut_J | Gs_J] [Ut_I | D_I | Gs_I]^T  (vi_ekf.cpp:304 on one tile):
        //   O1 = X^T D_J^T   (the accumulator registers of X are the A operand of k-step r: that reads X transposed)
        //   O2 = O1^T D_I^T = D_J X D_I^T, then the K = 24 coupling on the same accumulator
        v4f64 o1 = {0.0, 0.0, 0.0, 0.0}, o2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; r++) o1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X[s][r], d_op(cj, r), o1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) o2 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[r], d_op(ci, r), o2, 0, 0, 0);
#pragma unroll
        for (int k6 = 0; k6 < 6; k6++) o2 = __builtin_amdgcn_mfma_f64_16x16x4f64(z_op(cj, k6, false), z_op(ci, k6, true), o2, 0, 0, 0);
        if (TI == TJ) {   // + Qx on the diagonal (every slot, active or not: vi_ekf.cpp:139-144,304)
          const int pr = tile_prow(TI, l15, nf);
          const double qx = (pr >= 0) ? a.Qx[max(pr, 0)] : 0.0;
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (lg + 4 * r == l15) o2[r] += qx;
        }
        X[s] = o2;
        group_fence<true>();   // (one tile's operand loads are not hoisted over the previous tile's: they would all be held live)
      });
      RES_STAMP(S, st0, 68);
      if (MP) res_prop_body<TW>(a, S, tid);   // (single propagate: the service wave does this meanwhile)
      par ^= 1;   // the service wave posted this propagate's fix_depth edits into mailbox par ^ 1
      if (MP && kp + 1 < nkp) {
        // the next propagate reads whole diagonal tiles: upper triangle <- lower (what a store and a load in between would
        // leave: viekf_batch_step_n is bit for bit K propagates and a step), through lane shuffles
        static_for<CNT>([&](auto sc) {
          constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
          if (TI != TJ || TI == 0) return;
          v4f64 nx = X[s];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int aa = lg + 4 * r;                       // element (a = aa, b = l15): P[row b][col a]; upper triangle: b < a
            const int src = 16 * (l15 & 3) + aa;             // lane that holds (a = l15, b = aa)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
              const double v = __shfl(X[s][rr], src & 63, 64);
              if (l15 < aa && (l15 >> 2) == rr) nx[r] = v;
            }
          }
          X[s] = nx;
        });
      }
      __syncthreads();  // B4p
      for (int e = tid; e < 256; e += TW) { const int r = e >> 4, c = e & 15; Pbb[e] = S.Mbb[min(r, c) * 16 + max(r, c)]; }
      if (MP && kp + 1 < nkp) apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
    }
    RES_STAMP(S, st0, 70);
    __syncthreads();  // B4q : the body block copy above is complete
  }
  RES_STAMP(S, st0, 71);

  // ---------------- body tiles: LDS -> registers (they are swept on the matrix cores like every other tile)
  static_for<CNT>([&](auto sc) {
    constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
    if (TJ != 0) return;
    v4f64 x = {0.0, 0.0, 0.0, 0.0};
    if (TI == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) x[r] = Pbb[l15 * 16 + lg + 4 * r];
    } else {
      const int fr = 15 * (TI - 1) + l15;
      if (l15 < 15 && fr < nf) {
#pragma unroll
        for (int r = 0; r < 4; r++) x[r] = Pbc[fr * 16 + lg + 4 * r];
      }
    }
    X[s] = x;
  });

  // Publishes the column pair of feature g -- P[:, q*], P[:, q* + 1], q* = tile_qrow(g, 0) -- into dst [NQ][2], taken from the
  // lower triangle only (row i >= column j; the part above the diagonal through its mirror).  The NT tiles that hold it: tile
  // row T* (TI = T*: its P rows q*, q* + 1 are the wanted columns, mirrored) and tile column T* (TJ = T*).  One scalar jump on T*
  // reaches the code of exactly this wave's tiles of that cross.
  auto extract = [&](int g, double* dst) {
    const int Ts = 1 + g / 5, w0 = 3 * (g % 5);   // (wave-uniform)
    // (lane parts re-derived from a laundered lane index: everything computed from them here is otherwise hoisted out of the
    //  update loop, held across it -- and spilled)
    const int ln_ = opaque(lane), l15 = ln_ & 15, lg = ln_ >> 4;
    static_for<NT - 1>([&](auto tc) {
      constexpr int TS = decltype(tc)::value + 1;
      if (Ts != TS) return;
      static_for<CNT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
        if (TI == TS) {
          const int c = l15 - w0;
          if (c == 0 || c == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int aa = lg + 4 * r;
              if (TJ != TS || aa <= l15) dst[2 * (16 * TJ + aa) + c] = X[s][r];
            }
          }
        }
        if (TJ == TS) {
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const int aa = w0 + c, r = aa >> 2;
            const double v = (r == 0) ? X[s][0] : ((r == 1) ? X[s][1] : ((r == 2) ? X[s][2] : X[s][3]));
            if (lg == (aa & 3) && (TI != TS || l15 > aa)) dst[2 * (16 * TI + l15) + c] = v;
          }
        }
      });
    });
  };

  // ---------------- M sequential feature updates: covariance side ----------------
  // (measurement indices and slots are wave-uniform: kept in SGPRs, so that the tests on them are scalar branches)
  auto uni2 = [](int2 v) { return make_int2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y)); };
  int m = __builtin_amdgcn_readfirstlane(res_next_valid(S, 0));
  int2 sq = uni2(S.mseq[min(m, S.mcap - 1)]);     // {index of the measurement after m, its slot}
  if (m < S.M) {
    apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
    // hand the zeta-zeta 2x2 of every feature to the service lanes (they keep it current from here on): lower triangle, mirrored
    static_for<CNT>([&](auto sc) {
      constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
      if (TI != TJ || TI == 0) return;
      const int jb = l15 / 3, wb = l15 - 3 * jb, f = 5 * (TI - 1) + jb;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int aa = lg + 4 * r, ja = aa / 3, wa = aa - 3 * ja;
        if (ja == jb && l15 < 15 && wa < 2 && wb < 2 && wa <= wb && f < N) {   // P[row wb][col wa] of the feature's own block
          S.Pd[4 * f + 2 * wb + wa] = X[s][r];
          S.Pd[4 * f + 2 * wa + wb] = X[s][r];
        }
      }
    });
    extract(__builtin_amdgcn_readfirstlane(S.mslot[m]), S.Cb);   // first measurement: its columns ARE the current ones (nothing pending)
    if (sq.y >= 0) extract(sq.y, S.Eb + 2 * NQ);     // second one: raw, buffer 1
  }
  RES_STAMP(S, st0, 72);
  __syncthreads();  // Bp : Pd and the first columns are published
  if (m < S.M && tid < NQ) {   // NaN in the first column pair (the later ones are tested where they are formed)
    const double2 c0 = lds_ld2(S.Cb + 2 * tid);
    if (c0.x != c0.x || c0.y != c0.y) S.sm[44] = 1.0;
  }
  __syncthreads();  // B1 : the service published the first measurement's G and verdict
  RES_STAMP(S, st0, 73);
  int cnt = 0;
  // (mu = 1 - lambda of this lane's rows is read from the LDS copy of lambda inside every phase: held in registers across the
  //  loop, those six registers were what pushed an accumulator tile of the third worker wave into scratch)
  const bool partial = prm.use_partial_update != 0;
  const double mz0 = uniform_f64(tile_mu(S.lam, 16, nf, partial)), mz1 = uniform_f64(tile_mu(S.lam, 17, nf, partial));   // lambda_feat is the same for every slot
  // ONE barrier per update.  Inside a phase the worker waves (1) form their operands from the current column pair C_m and the
  // 2x2 G_m = Hb^T S^-1 Hb of the service wave and issue one MFMA per tile:  P -= Lambda o (C G C^T)  (= the reference's
  // (I-KH)P(I-KH)^T + KRK^T - P restricted by Lambda, vi_ekf_meas.cpp:254-257, for K = C Hb^T S^-1), (2) bring the NEXT
  // measurement's raw column pair -- extracted one phase ago, before this update -- up to date with this update (one row per
  // thread), (3) extract the raw column pair of the measurement after next from the swept tiles.
  if (PAIR && half) __syncthreads();   // (filter 1 runs half a phase behind)
  const int ntrip = PAIR ? trips : 0x7fffffff;
  for (int trip = 0; trip < ntrip && (PAIR || m < S.M); trip++) {
    if (PAIR && m >= S.M) { __syncthreads(); __syncthreads(); continue; }   // (done: keeps the other filter's barriers company)
    const int mnext = sq.x, snext = sq.y;
    const int2 sq2 = uni2(S.mseq[min(mnext, S.mcap - 1)]);
    const double* mb = S.sm + 8 * (cnt & 1);
    const double g00 = mb[0], g01 = mb[1], g11 = mb[2];
    const bool run = mb[3] == 0.0 && S.sm[44 + cnt % 3] == 0.0 && !RES_ABLATE(S, 1);   // not gated, no NaN guard
    const double fixpending = S.sm[40 + (par ^ 1)];
    const double* Cc = S.Cb + (cnt & 1) * 2 * NQ;
    double* Cn = S.Cb + ((cnt + 1) & 1) * 2 * NQ;
    if (tid == 0) S.sm[44 + (cnt + 2) % 3] = 0.0;   // (the word of the phase after next: nobody reads or sets it in this phase)
    apply_fixes(par ^ 1, fixpending);
    RES_STAMP(S, st0 && cnt < 8, 80 + 4 * cnt + 0);
    // (2) is spread over the MFMA sequence of (1): a wave issues in order, and a matrix instruction holds the next one back for
    // its 64 cycles -- whatever sits between two of them in program order is free.  Rows of tiles in turn (a wave's slots are
    // ordered by TI): the column pair's rows of tile row TR are read two rows ahead, the A-side operand of every tile index is
    // kept (a later row needs all TJ <= TI), the B-side one lives for its row only.
    const bool fix = snext >= 0 && tid < NQ;
    const int fq = min(tid, NQ - 1), qs = tile_qrow(max(snext, 0), 0);
    const double* En = S.Eb + ((cnt + 1) & 1) * 2 * NQ;
    if (run) {
      const int ln_ = opaque(lane), l15 = ln_ & 15, lg = ln_ >> 4;
      const bool odd = (lg & 1) != 0;
      const double muF = tile_mu(S.lam, 16 + l15, nf, partial), muB = tile_mu(S.lam, l15, nf, partial);
      const double fAF = (lg < 2) ? 1.0 : muF, fAB = (lg < 2) ? 1.0 : muB;        // TJ side:  C[k & 1] x {1, 1, mu, mu}
      const double fBF = (lg < 2) ? -1.0 : muF, fBB = (lg < 2) ? -1.0 : muB;      // TI side: Kg[k & 1] x {-1, -1, mu, mu}
      const double ga = odd ? g01 : g00, gb = odd ? g11 : g01;                  // column k & 1 of G
      const double gaF = ga * fBF, gbF = gb * fBF, gaB = ga * fBB, gbB = gb * fBB;
      const double* Cl = Cc + 2 * l15;
      const double* Ck = Cl + (lg & 1);
      // slot by slot (a wave's slots are ordered by TI): the A-side value of slot s + 2 and, at a row's first slot, the column-pair
      // rows of the row after next are read ahead -- a matrix instruction's operands are then in registers when its turn comes,
      // and only three of each are alive at a time (an array of all eleven A-side operands was spilled inside this loop)
      double ar[CNT + 2];
      double2 craw[NT + 2];
      double2 fe = {0.0, 0.0}, fc = fe;
      double bop = 0.0;
      constexpr int FC = CNT > 8 ? CNT / 2 : CNT - 1, FL = FC > 2 ? FC - 2 : 0;   // slots after which the next pair's inputs are read / it is formed
      static_for<2>([&](auto pc) {
        constexpr int s0 = decltype(pc)::value;
        if (s0 < CNT) ar[s0] = Ck[32 * Map::tj(W, s0 < CNT ? s0 : 0)];
      });
      // (rows this wave holds tiles of, in order: row r of them is Map::ti of the first slot of the r-th run)
      static_for<CNT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
        constexpr int prevTI = (s == 0) ? -1 : Map::ti(W, s == 0 ? 0 : s - 1);
        constexpr bool newrow = TI != prevTI;
        if (s + 2 < CNT) ar[s + 2 < CNT ? s + 2 : 0] = Ck[32 * Map::tj(W, s + 2 < CNT ? s + 2 : 0)];
        if (newrow) {
          // this row's pair was requested one row-start earlier (or here, for the wave's first row); request the next row's
          constexpr int nextTI = [] { for (int q = s + 1; q < CNT; q++) if (Map::ti(W, q) != TI) return Map::ti(W, q); return -1; }();
          if (s == 0) craw[TI] = lds_ld2(Cl + 32 * TI);
          if (nextTI >= 0) craw[nextTI >= 0 ? nextTI : 0] = lds_ld2(Cl + 32 * (nextTI >= 0 ? nextTI : 0));
          const double2 c = craw[TI];
          bop = (TI == 0) ? fma(c.y, gbB, c.x * gaB) : fma(c.y, gbF, c.x * gaF);
        }
        const double aop = ar[s] * ((TJ == 0) ? fAB : fAF);
        X[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, X[s], 0, 0, 0);
        // next measurement's column pair:  C_{m+1}[q] = E[q] - Lambda(q, zeta_c) (C G C^T)[q][zeta_c], one row per thread
        if (s == FL && snext >= 0) { fe = lds_ld2(En + 2 * fq); fc = lds_ld2(Cc + 2 * fq); }
        if (s == FC && snext >= 0) {
          const double2 fcs0 = lds_ld2(Cc + 2 * qs), fcs1 = lds_ld2(Cc + 2 * qs + 2), fes = lds_ld2(En + 2 * qs);
          const double kg0 = fma(fc.y, g01, fc.x * g00), kg1 = fma(fc.y, g11, fc.x * g01);
          const double mq = tile_mu(S.lam, opaque(fq), nf, partial);   // (read here: not held across the loop)
          fe.x = fma(-fma(-mq, mz0, 1.0), fma(kg1, fcs0.y, kg0 * fcs0.x), fe.x);
          fe.y = fma(-fma(-mq, mz1, 1.0), fma(kg1, fcs1.y, kg0 * fcs1.x), fe.y);
          // the feature's own 2x2 stays exactly symmetric: element (zeta1, zeta0) takes the value row zeta0 forms for (zeta0, zeta1)
          const double ks0 = fma(fcs0.y, g01, fcs0.x * g00), ks1 = fma(fcs0.y, g11, fcs0.x * g01);
          const double alt = fma(-fma(-mz0, mz1, 1.0), fma(ks1, fcs1.y, ks0 * fcs1.x), fes.y);
          if (fq == qs + 1) fe.x = alt;
          if (fix) {
            *reinterpret_cast<double2*>(Cn + 2 * fq) = fe;
            if (fe.x != fe.x || fe.y != fe.y) S.sm[44 + (cnt + 1) % 3] = 1.0;
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps this placement: the scheduler would otherwise cluster loads and arithmetic)
      });
    } else if (fix) {   // gated / NaN-guarded: nothing to apply, the raw pair is the current one
      const double2 e = lds_ld2(En + 2 * fq);
      *reinterpret_cast<double2*>(Cn + 2 * fq) = e;
      if (e.x != e.x || e.y != e.y) S.sm[44 + (cnt + 1) % 3] = 1.0;
    }
    RES_STAMP(S, st0 && cnt < 8, 80 + 4 * cnt + 1);
    RES_STAMP(S, st0 && cnt < 8, 80 + 4 * cnt + 2);
    if (PAIR) __syncthreads();   // mid-phase: from here the other filter of the pair has the matrix pipe
    // (3) raw column pair of the measurement after next, from the swept tiles
    if (sq2.y >= 0 && mnext < S.M && !RES_ABLATE(S, 4)) extract(sq2.y, S.Eb + (cnt & 1) * 2 * NQ);
    RES_STAMP(S, st0 && cnt < 8, 80 + 4 * cnt + 3);
    par ^= 1;
    cnt++;
    sq = sq2;
    __syncthreads();  // B1 (the only barrier of an update)
    RES_STAMP(S, st0 && cnt <= 8, 112 + cnt - 1);
    m = mnext;
  }
  if (PAIR && !half) __syncthreads();
  apply_fixes(par ^ 1, S.sm[40 + (par ^ 1)]);
  RES_STAMP(S, st0, 74);

  // ---------------- store: the lower triangle, straight from the tiles (lanes along the rows of P) ----------------
  {
    double* Po = a.P_out + S.so * n * ld;   // in place, or the next slot of the history ring
    const LaneOff so_ = lane_off();
    static_for<CNT>([&](auto sc) {
      constexpr int s = decltype(sc)::value, TI = Map::ti(W, s), TJ = Map::tj(W, s);
      double* base = Po + rowbase(TI) + (long)rowbase(TJ) * ld;
      if (row_ok(TI)) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (TI == TJ) { if (so_.low[r] && col_ok(TJ, r)) (base + 4L * r * ld)[so_.vo] = X[s][r]; }
          else if ((r < 3 && TJ != NT - 1) || TJ == 0) (base + 4L * r * ld)[so_.vo] = X[s][r];
          else if (col_ok(TJ, r)) (base + 4L * r * ld)[so_.vo] = X[s][r];
        }
      }
      if ((s & 3) == 3) group_fence<true>();
    });
  }
  RES_STAMP(S, st0, 75);
}

}  // namespace viekf
