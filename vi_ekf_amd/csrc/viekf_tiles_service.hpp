// viekf_tiles_service.hpp -- tile family: the service wave (dynamics, state correction, prediction, gate; the covariance side of
// an update reaches it as one column pair, never as gain rows).  Overview: viekf_tiles_common.hpp.
#pragma once
#include "viekf_tiles_common.hpp"

namespace viekf {

// One service wave (N + 14 <= 64 lanes: a lane per feature and 14 body lanes, as in the resident family).  Per update m it
//   reads its lanes' rows of the CURRENT column pair C_m (all pending updates applied by the worker threads),
//   corrects the state:  dx = lambda o (K r),  K r = C (Hb^T S^-1 r)  -- the 2-vector g_r is wave-uniform, so the gain rows are
//     never formed (vi_ekf_meas.cpp:241,249-255),
//   keeps each feature's zeta-zeta 2x2 of P current:  P_zz -= Lambda o (C_z G C_z^T),
//   runs fix_depth, predicts measurement m+1 on the lane of its feature and publishes its  G = Hb^T S^-1 Hb  and verdict.
template <int T, bool MP, bool PAIR = false>
__device__ __forceinline__ void tile_service(const StreamArgs& a, const TileShared& S, int lane, const double* __restrict__ u_all,
                                             const double* __restrict__ dt_all, int* __restrict__ result_all, int half = 0, int trips = 0) {
  const int N = S.N, len = S.len, M = S.M, NQ = S.NQ;
  const DevParams& prm = *a.dp;
  double* xs = S.xs;
  double* sm = S.sm;
  unsigned flag = 0;
  const bool partial = prm.use_partial_update != 0;
  int par = 0;
  __builtin_amdgcn_s_setprio(3);   // the per-update critical path runs on this wave
  double dt = sm[42];
  RES_STAMP(S, lane == 0, 0);
  // dynamics of propagate kp (of S.kp): body Jacobian on lane 0 (+ A_v G_b for the workers' expansion of the feature rows), then
  // one feature per lane -- before B0, while the worker waves are still loading P from HBM
  auto dyn_body = [&](int kp) {
    for (int i = lane; i < 256; i += 64) S.Abb[i] = 0.0;
    for (int i = lane; i < 96; i += 64) S.Gb[i] = 0.0;
    if (lane < 16) S.xdb[lane] = 0.0;
    if (lane == 0) res_body_phase(xs, u_all + ((long)kp * S.B + S.b) * 6, a.dp, S.ctx, S.xdb, S.Abb, S.Gb);
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
  };
  if (S.do_prop) { dyn_body(0); dyn_feat(dt); }
  RES_STAMP(S, lane == 0, 14);
  __syncthreads();  // B0
  RES_STAMP(S, lane == 0, 1);

  const int nkp = MP ? S.kp : 1;
  if (S.do_prop) {
    for (int kp = 0; kp < nkp; kp++) {
      __syncthreads();  // B1p
      __syncthreads();  // B2p
      __syncthreads();  // B2q
      if (lane == 63) {   // body state step (every feature lane has consumed the old body state through ctx)
        double dxb[16], xo[17];
#pragma unroll
        for (int i = 0; i < 16; i++) dxb[i] = S.xdb[i] * dt;
        body_boxplus_fast(xs, dxb, xo);
#pragma unroll
        for (int i = 0; i < 17; i++) xs[i] = xo[i];
      }
      if (lane == 0) sm[40 + par] = 0.0;
      for (int f = lane; f < len; f += 64)   // fix_depth (vi_ekf.cpp:311): state here, covariance through the mailbox
        res_fix_depth(xs + xZ + 5 * f, a.dp, &S.fixadd[par * N + f], &S.fixset[par * N + f], &sm[40 + par], &flag);
      par ^= 1;
      __syncthreads();  // B3p
      double dt_next = 0.0;
      if (MP && kp + 1 < nkp) {   // the body part of the NEXT propagate's dynamics, under the workers' tile products
        dt_next = dt_all[(long)(kp + 1) * S.B + S.b];
        dyn_body(kp + 1);
      }
      if (!MP) res_prop_body<64>(a, S, lane);   // body strips and body block of P+ off the workers' path
      __syncthreads();  // B4p
      if (MP && kp + 1 < nkp) {
        dt = dt_next;
        if (lane == 0) sm[42] = dt;
        dyn_feat(dt);
      }
    }
    __syncthreads();  // B4q
  }
  RES_STAMP(S, lane == 0, 2);

  // lane roles for the state correction (one instruction stream, no divergence):
  //   lane f < N           : feature f  -> rows 16+3f..+2 : bearing quaternion (2 rows) + inverse depth (1 row)
  //   lane N+j, j = 0..5   : body row j            (p, v)          linear state x[j]
  //   lane N+6             : body rows 6,7,8       (attitude)      quaternion x[6..9], right-multiplied
  //   lane N+j, j = 7..13  : body row j+2 = 9..15  (b_a, b_g, mu)  linear state x[j+3]
  const int fid = (lane < N) ? lane : -1;
  const int jb = lane - N;
  const bool isfeat = fid >= 0;
  const bool isatt = jb == 6;
  const bool hasq = (isfeat && fid < len) || isatt;
  const bool haslin = (isfeat && fid < len) || (jb >= 0 && jb < 14 && jb != 6);
  int q0, q1, q2;   // this lane's rows in tile space
  if (isfeat) { q0 = tile_qrow(fid, 0); q1 = q0 + 1; q2 = q0 + 2; }
  else if (isatt) { q0 = 6; q1 = 7; q2 = 8; }
  else { const int r = (jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 2 : 0)); q0 = q1 = q2 = r; }
  double* qptr = isfeat ? (xs + xZ + 5 * fid) : (xs + xATT);
  double* linptr = isfeat ? (xs + xZ + 5 * fid + 4) : (xs + ((jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 3 : 0))));
  const double rho_reset = uniform_f64(1.0 / (2.0 * prm.min_depth));
  const double lz0 = uniform_f64(a.lambda[16]), lz1 = uniform_f64(a.lambda[17]);
  const double L00 = uniform_f64(partial ? (lz0 + lz0 - lz0 * lz0) : 1.0), L01 = uniform_f64(partial ? (lz0 + lz1 - lz0 * lz1) : 1.0),
               L11 = uniform_f64(partial ? (lz1 + lz1 - lz1 * lz1) : 1.0);
  // lambda of this lane's rows (1 without the partial update); P row of a tile-space row q: q - (q >> 4) + 1 past the body tile
  auto prow_of = [](int q) { return q < 16 ? q : q - (q >> 4) + 1; };
  const double lam0 = partial ? S.lam[prow_of(q0)] : 1.0, lam1 = partial ? S.lam[prow_of(q1)] : 1.0, lam2 = partial ? S.lam[prow_of(q2)] : 1.0;

  int m = __builtin_amdgcn_readfirstlane(res_next_valid(S, 0));
  __syncthreads();  // Bp : the workers published Pd (zeta blocks) and the first measurement's columns
  RES_STAMP(S, lane == 0, 3);
  double pf00 = 0.0, pf01 = 0.0, pf11 = 0.0;
  if (isfeat) { const double* pd = S.Pd + 4 * fid; pf00 = pd[0]; pf01 = pd[1]; pf11 = pd[3]; }
  // Uniform per-measurement values, computed by the lane of the measured feature and handed to the whole wave with v_readlane
  struct Meas { double g00, g01, g11, gr0, gr1, skip; };
  auto bcast = [&](double v, int src) -> double {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
  };
  // prediction + innovation of measurement mm, whose feature is lane `src` (wave-uniform): EVERY lane runs the arithmetic on its
  // own registers (no divergence), lane src's values are broadcast.  h_feat vi_ekf_meas.cpp:354-367; S, gate :230-239;
  //   G = Hb^T S^-1 Hb  (K W^T = C G C^T),   g_r = Hb^T S^-1 r  (K r = C g_r);   skip = gated (1) or a NaN in G / g_r (2: :247)
  auto predict = [&](const double* t1, const double* t2, const double* zt, int mm, int src, Meas& o) {
    double zhat[2], Hb[4], Sm[4], Si[4];
    h_feat_frame(t1, t2, zt, prm, zhat, Hb);
    const double2 zn = lds_ld2(S.mz + 2 * mm);
    const double* R = S.mR + 4 * mm;
    const double r0 = zn.x - zhat[0], r1 = zn.y - zhat[1];
    const double w00 = pf00 * Hb[0] + pf01 * Hb[1], w01 = pf00 * Hb[2] + pf01 * Hb[3];   // (P_zz Hb^T)
    const double w10 = pf01 * Hb[0] + pf11 * Hb[1], w11 = pf01 * Hb[2] + pf11 * Hb[3];
    Sm[0] = Hb[0] * w00 + Hb[1] * w10 + R[0];
    Sm[1] = Hb[0] * w01 + Hb[1] * w11 + R[2];
    Sm[2] = Hb[2] * w00 + Hb[3] * w10 + R[1];
    Sm[3] = Hb[2] * w01 + Hb[3] * w11 + R[3];
    inv2_fast(Sm, Si);
    const double mahal = (r0 * Si[0] + r1 * Si[2]) * r0 + (r0 * Si[1] + r1 * Si[3]) * r1;   // vi_ekf_meas.cpp:234
    // M = S^-1 Hb (2x2), G = Hb^T M, g_r = Hb^T (S^-1 r)
    const double m00 = Si[0] * Hb[0] + Si[1] * Hb[2], m01 = Si[0] * Hb[1] + Si[1] * Hb[3];
    const double m10 = Si[2] * Hb[0] + Si[3] * Hb[2], m11 = Si[2] * Hb[1] + Si[3] * Hb[3];
    const double g00 = Hb[0] * m00 + Hb[2] * m10, g01 = Hb[0] * m01 + Hb[2] * m11, g11 = Hb[1] * m01 + Hb[3] * m11;
    const double s0 = Si[0] * r0 + Si[1] * r1, s1 = Si[2] * r0 + Si[3] * r1;
    const double gr0 = Hb[0] * s0 + Hb[2] * s1, gr1 = Hb[1] * s0 + Hb[3] * s1;
    const double chk = (g00 + g01) + (g11 + gr0) + gr1;
    const double skip = (chk != chk) ? 2.0 : ((mahal > 9.0) ? 1.0 : 0.0);                 // NaN guard first: a NaN never gates
    o.g00 = bcast(g00, src); o.g01 = bcast(g01, src); o.g11 = bcast(g11, src);
    o.gr0 = bcast(gr0, src); o.gr1 = bcast(gr1, src); o.skip = bcast(skip, src);
  };
  auto publish = [&](const Meas& q, int mbx) {
    if (lane == 0) {
      double* d = sm + 8 * mbx;
      d[0] = q.g00; d[1] = q.g01; d[2] = q.g11; d[3] = q.skip;
    }
  };
  // this lane's quaternion and linear state live in registers for the whole loop (written back once at the end)
  double qn[4] = {qptr[0], qptr[1], qptr[2], qptr[3]};
  double lin = *linptr;
  double f1[3], f2[3], fz[3];
  bearing_frame_fast(qn, f1, f2, fz);
  const double sgn = isatt ? -1.0 : 1.0;   // q (x) e instead of e (x) q flips the cross term only
  Meas cur = {}, nxt = {};
  if (m < M) {
    predict(f1, f2, fz, m, __builtin_amdgcn_readfirstlane(S.mslot[m]), cur);
    publish(cur, 0);
  }
  int2 sq = S.mseq[min(m, S.mcap - 1)];
  RES_STAMP(S, lane == 0, 4);
  __syncthreads();  // B1
  RES_STAMP(S, lane == 0, 5);
  int cnt = 0;

  if (PAIR && half) __syncthreads();   // (filter 1 of a pair runs half a phase behind: viekf_tiles_worker.hpp)
  const int ntrip = PAIR ? trips : 0x7fffffff;
  for (int trip = 0; trip < ntrip && (PAIR || m < M); trip++) {
    if (PAIR && m >= M) { __syncthreads(); __syncthreads(); continue; }
    const int mnext = __builtin_amdgcn_readfirstlane(sq.x), slot_next = __builtin_amdgcn_readfirstlane(sq.y);
    sq = S.mseq[min(mnext, S.mcap - 1)];
    const double* Cc = S.Cb + (cnt & 1) * 2 * NQ;
    const double2 c0 = lds_ld2(Cc + 2 * q0), c1 = lds_ld2(Cc + 2 * q1), c2 = lds_ld2(Cc + 2 * q2);
    RES_STAMP(S, lane == 0 && cnt < 8, 16 + 4 * cnt + 0);
    const bool gated = cur.skip == 1.0;
    // NaN guard (vi_ekf_meas.cpp:247: a NaN in K or H skips the update, not fix_depth): K = C Hb^T S^-1 has one iff the column
    // pair or the measurement's 2x2 factors have one -- the worker threads test every row of the pair where they form it
    const bool bad = cur.skip == 2.0 || sm[44 + cnt % 3] != 0.0;
    // correction lambda o (K r) = lambda o (C g_r)   (vi_ekf_meas.cpp:249-255)
    const double dv0 = lam0 * fma(c0.y, cur.gr1, c0.x * cur.gr0);
    const double dv1 = lam1 * fma(c1.y, cur.gr1, c1.x * cur.gr0);
    const double dv2 = lam2 * fma(c2.y, cur.gr1, c2.x * cur.gr0);
    // rotation vector of the correction: bearing  T_zeta [d0 d1],  attitude  [d0 d1 d2]
    double v[3];
    v[0] = isatt ? dv0 : (f1[0] * dv0 + f2[0] * dv1);
    v[1] = isatt ? dv1 : (f1[1] * dv0 + f2[1] * dv1);
    v[2] = isatt ? dv2 : (f1[2] * dv0 + f2[2] * dv1);
    const bool corr = !gated && !bad && !RES_ABLATE(S, 2);
    // x <- x [+] dx  (vi_ekf_helper.cpp:88-98): bearing  exp(T_z d) (x) q ;  attitude  q (x) exp(d) ;  the rest adds
    if (corr) {
      double e[4];
      q_exp_fast(v, e);
      const double ex = sgn * e[1], ey = sgn * e[2], ez = sgn * e[3];
      const double o0 = e[0] * qn[0] - e[1] * qn[1] - e[2] * qn[2] - e[3] * qn[3];
      const double o1 = e[0] * qn[1] + qn[0] * e[1] + (ey * qn[3] - ez * qn[2]);
      const double o2 = e[0] * qn[2] + qn[0] * e[2] + (ez * qn[1] - ex * qn[3]);
      const double o3 = e[0] * qn[3] + qn[0] * e[3] + (ex * qn[2] - ey * qn[1]);
      qn[0] = o0; qn[1] = o1; qn[2] = o2; qn[3] = o3;
      bearing_frame_fast(qn, f1, f2, fz);
      lin += isfeat ? dv2 : dv0;
      // this lane's copy of P_zz follows the sweep:  P_zz -= Lambda o (C_z G C_z^T)   (vi_ekf_meas.cpp:256-257)
      const double k00 = fma(c0.y, cur.g01, c0.x * cur.g00), k01 = fma(c0.y, cur.g11, c0.x * cur.g01);   // (C G) row zeta0
      const double k10 = fma(c1.y, cur.g01, c1.x * cur.g00), k11 = fma(c1.y, cur.g11, c1.x * cur.g01);   // row zeta1
      pf00 = fma(-L00, fma(k01, c0.y, k00 * c0.x), pf00);
      pf01 = fma(-L01, fma(k01, c1.y, k00 * c1.x), pf01);
      pf11 = fma(-L11, fma(k11, c1.y, k10 * c1.x), pf11);
    }
    RES_STAMP(S, lane == 0 && cnt < 8, 16 + 4 * cnt + 1);
    if (PAIR) __syncthreads();   // mid-phase barrier of the pair (no data of this wave's depends on it)
    if (lane == 0) sm[40 + par] = 0.0;
    // fix_depth (vi_ekf_meas.cpp:271; a gated update returns before it, :238): almost never fires -- one wave-wide test
    const bool odd_depth = !gated && isfeat && fid < len && !(lin >= 0.0 && lin <= 1e2);
    if (__any(odd_depth)) {
      if (odd_depth) {
        double rho = lin;
        if (rho != rho) { rho = rho_reset; flag |= FLAG_NAN; }
        if (rho < 0.0) {
          const double err = rho_reset - rho;
          S.fixadd[par * N + fid] = err * err;
          sm[40 + par] = 1.0;
          rho = rho_reset;
          flag |= FLAG_NEGDEPTH;
        } else if (rho > 1e2) {
          S.fixset[par * N + fid] = 1.0;
          sm[40 + par] = 1.0;
          rho = rho_reset;
        }
        lin = rho;
      }
    }
    if (slot_next >= 0) {   // next measurement, from registers
      predict(f1, f2, fz, mnext, slot_next, nxt);
      publish(nxt, (cnt + 1) & 1);
    }
    if (result_all && lane == 0) result_all[(long)S.b * S.mstride + m] = gated ? 1 : 0;
    RES_STAMP(S, lane == 0 && cnt < 8, 16 + 4 * cnt + 2);
    cur = nxt;
    par ^= 1;
    cnt++;
    __syncthreads();  // B1 (the only barrier of an update)
    RES_STAMP(S, lane == 0 && cnt <= 8, 16 + 4 * (cnt - 1) + 3);
    m = mnext;
  }
  if (PAIR && !half) __syncthreads();

  RES_STAMP(S, lane == 0, 12);
  if (hasq) { qptr[0] = qn[0]; qptr[1] = qn[1]; qptr[2] = qn[2]; qptr[3] = qn[3]; }
  if (haslin) *linptr = lin;
  wave_lds_sync();   // (the lanes below read what other lanes of this wave just wrote)
  // ---------------- store x, status ----------------
  double* xg = a.x_out + S.so * a.nxs;
  const int xend = (a.x_out != a.x || a.smap_out) ? a.nxs : xZ + 5 * len;   // another ring slot gets the whole vector (zeros past the features)
  for (int i = lane; i < xend; i += 64) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[S.b], flag);
  RES_STAMP(S, lane == 0, 13);
}

}  // namespace viekf
