// viekf_resident_service.hpp -- resident family: the service wave(s) (dynamics, state correction, prediction, gate, gain rows).
#pragma once
#include "viekf_resident_prop.hpp"

namespace viekf {

// ---- the service wave: everything that is not a sweep over P --------------------------------------
// ROLE 0: the one service wave of a workgroup (N + 14 <= 64 lanes: a lane per feature and 14 body lanes).  More features than
// that split the roles over TWO service waves: ROLE 1 = the lanes of features 0..63 (and everything a single service wave does
// besides: dynamics, result codes, the state store), ROLE 2 = the 14 body lanes on a wave of their own -- and, past 64 features
// (N <= 114 by lanes; the register file ends the family at N = 72), features 64.. on its lanes 14...  A measurement is
// predicted by the wave that holds its feature; the other one receives {Hb, residual, S^-1, gate} through an LDS mailbox
// (polled; bounded); each wave writes the gain rows, the NaN-guard word and the fix_depth mailbox flag of its own rows.
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
  // feature of this lane (or -1): ROLE 0 / 1: the lane number; ROLE 2: features 64.. on the lanes after the 14 body lanes
  const int fid = (ROLE == 2) ? ((lane >= 14 && 50 + lane < N) ? 50 + lane : -1) : ((lane < N) ? lane : -1);
  const int jb = (ROLE == 2) ? ((lane < 14) ? lane : -1) : ((ROLE == 1) ? -1 : lane - N);
  const bool isfeat = fid >= 0;
  // does this wave hold feature `slot` (it then predicts its measurement), and on which lane
  auto holds = [&](int slot) -> bool { return ROLE == 0 || ((ROLE == 1) == (slot < 64)); };
  auto lane_of = [&](int slot) -> int { return (ROLE == 2) ? slot - 50 : slot; };
  const bool isatt = jb == 6;
  const bool hasq = (isfeat && fid < len) || isatt;
  const bool haslin = (isfeat && fid < len) || (jb >= 0 && jb < 14 && jb != 6);
  int rid0, rid1, rid2;
  if (isfeat) { rid0 = 16 + 3 * fid; rid1 = rid0 + 1; rid2 = rid0 + 2; }
  else if (isatt) { rid0 = 6; rid1 = 7; rid2 = 8; }
  else { const int r = (jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 2 : 0)); rid0 = rid1 = rid2 = r; }
  const bool rowlane = isfeat || (jb >= 0 && jb < 14);   // this lane owns rows of K / W (the others only tag along)
  double* qptr = isfeat ? (xs + xZ + 5 * fid) : (xs + xATT);
  double* linptr = isfeat ? (xs + xZ + 5 * fid + 4) : (xs + ((jb < 0) ? 0 : ((jb < 6) ? jb : ((jb < 14) ? jb + 3 : 0))));
  // (wave-uniform constants of the update loop are forced into SGPRs: as VGPR pairs they were a fifth of the loop's live set)
  const double rho_reset = uniform_f64(1.0 / (2.0 * prm.min_depth));
  // Lambda of the zeta-zeta 2x2 block (lambda_feat[0], lambda_feat[1])
  const double lz0 = uniform_f64(a.lambda[16]), lz1 = uniform_f64(a.lambda[17]);
  const double L00 = uniform_f64(partial ? (lz0 + lz0 - lz0 * lz0) : 1.0), L01 = uniform_f64(partial ? (lz0 + lz1 - lz0 * lz1) : 1.0),
               L11 = uniform_f64(partial ? (lz1 + lz1 - lz1 * lz1) : 1.0);

  // Each feature lane keeps its own P_zeta,zeta (2x2) current through the updates, so the lane of the NEXT measurement can
  // form  S = Hb P_zz Hb^T + R,  S^-1  and the gate verdict right after its prediction -- at the END of an iteration.
  // The next iteration then starts directly with the gain rows: no separate innovation phase, two barriers per update.
  int m = res_next_valid(S, 0);
  RES_STAMP(S, lane == 0, 9);
  __syncthreads();  // Bp : the workers published Pd (diagonal zeta blocks) and the first measurement's columns
  double pf00 = 0.0, pf01 = 0.0, pf10 = 0.0, pf11 = 0.0;
  if (isfeat) { const double* pd = S.Pd + 4 * fid; pf00 = pd[0]; pf01 = pd[1]; pf10 = pd[2]; pf11 = pd[3]; }
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
  struct Rows { double2 kA, wA, kB, wB, kC, oA, oB, oC; int bad; };   // (oX: the row's operand of next_rows -- K for a feature row, W for a body row)
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
    o.oA = isfeat ? kv[0] : wv[0]; o.oB = isfeat ? kv[1] : wv[1]; o.oC = isfeat ? kv[2] : wv[2];
  };
  // (MERGE: next_rows and the state correction in one basic block -- see the update loop)
  constexpr bool MERGE = (T != 256);
  // This lane's rows of the NEXT measurement's column pair: published by the worker waves one phase ago (buffer `rb`), as they
  // stood BEFORE the update being swept in this phase -- which is applied here (`swept`; gains Kc / Wc, this lane's own K rows
  // k3), with the workers' expression  p - Lambda (K . W):  feature rows i take K_i (own) and W of the column's feature,
  // body rows k take K of the column's feature and W_k, as the block sweep and the body-column sweep do.  Everything this
  // needs is complete at the top of a phase, so it runs there, off the critical path.
  const double lfz[3] = {lz0, lz1, uniform_f64(a.lambda[18])};
  // The dot products are  K_i . W_j  for a feature row i (own K row, the column feature's W row -- uniform over the lanes) and
  // K_j . W_k  for a body row k (the column feature's K row -- uniform --, own W row): one form  own . uni  with the lane's own
  // operand kept from gain_rows (Rows::oX) and the uniform one read out of the registers of the column feature's lane (single
  // service wave; v_readlane, no LDS round trip on the wave that closes every update).  Two service waves: the body wave has no
  // feature lanes, it reads the column feature's K rows from the gain buffer.
  auto next_rows = [&](int rb, bool swept, int slot, const double* Kc, const Rows& cr, double2 (&o)[3]) {
    const double* raw = S.Praw + rb * 2 * n;
    double2 st[3];
    st[0] = lds_ld2(raw + 2 * ridv[0]);      // (P[i][j0], P[i][j0+1]) before the update
    if (MERGE || three) { st[1] = lds_ld2(raw + 2 * ridv[1]); st[2] = lds_ld2(raw + 2 * ridv[2]); }   // (MERGE: a lane with ONE row has it three times -- no branch)
    double2 ua, ub;                           // the uniform operands for columns j0, j0+1
    if (holds(slot)) {                        // (wave-uniform) the column feature's rows are in this wave's registers
      const int sl = lane_of(slot);
      const double2 ka = make_double2(bcast(cr.kA.x, sl), bcast(cr.kA.y, sl)), kb2 = make_double2(bcast(cr.kB.x, sl), bcast(cr.kB.y, sl));
      const double2 wa = make_double2(bcast(cr.wA.x, sl), bcast(cr.wA.y, sl)), wb2 = make_double2(bcast(cr.wB.x, sl), bcast(cr.wB.y, sl));
      ua = isfeat ? wa : ka; ub = isfeat ? wb2 : kb2;
    } else {                                  // the other service wave's: from the gain buffer it wrote before the barrier
      const double* Wc = Kc + 2 * n;
      const double2 ka = lds_ld2(Kc + 2 * (16 + 3 * slot)), kb2 = lds_ld2(Kc + 2 * (16 + 3 * slot + 1));
      const double2 wa = lds_ld2(Wc + 2 * (16 + 3 * slot)), wb2 = lds_ld2(Wc + 2 * (16 + 3 * slot + 1));
      ua = isfeat ? wa : ka; ub = isfeat ? wb2 : kb2;
    }
    const double2 own[3] = {cr.oA, cr.oB, cr.oC};
    auto one = [&](int u) {
      double r0 = st[u].x, r1 = st[u].y;
      if (swept) {
        const double lamk = isfeat ? lfz[u] : lraw[u];
        const double La = partial ? (lamk + lz0 - lz0 * lamk) : 1.0, Lb = partial ? (lamk + lz1 - lz1 * lamk) : 1.0;
        r0 = fma(-La, fma(own[u].y, ua.y, own[u].x * ua.x), r0);
        r1 = fma(-Lb, fma(own[u].y, ub.y, own[u].x * ub.x), r1);
      }
      o[u] = make_double2(r0, r1);
    };
    one(0);
    if (MERGE || three) { one(1); one(2); }
    else { o[1] = o[0]; o[2] = o[0]; }
    if (isfeat && fid == slot) o[1].x = o[0].y;   // the measured feature's own zeta block: lower = upper, as the workers keep it
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
#ifdef VIEKF_TEST_STALL   // test build only (tests/test_gpu_stall.py): filter 0 never publishes its third hand-over -- the body
    if (S.b == 0 && seq == 3) return;   // wave's bounded wait must give up, raise VIEKF_FLAG_INTERNAL and the launch must end
#endif
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
    {
      const int s0 = __builtin_amdgcn_readfirstlane(S.mslot[m]);
      if (holds(s0)) { predict(f1, f2, fz, m, lane_of(s0), cur); if (ROLE != 0) send(cur, 0, 1); }
      else recv(cur, 0, 1);
    }
    double2 pr0[3];
#pragma unroll
    for (int u = 0; u < 3; u++) pr0[u] = lds_ld2(S.Praw + 2 * ridv[u]);   // (the first raw columns: buffer 0, published before Bp)
    gain_rows(cur, 44, 50, pr0, S.Kt, crow);
  }
  int2 sq = S.mseq[min(m, S.mcap - 1)];
  __syncthreads();  // B1
  RES_STAMP(S, lane == 0, 10);
  int it_ = 0, cnt = 0;

  while (m < M) {
    RES_MARK("service.next_rows");
    const int mnext = sq.x, slot_next = sq.y;
    // this lane's rows of the gain (formed by this wave at the end of the previous phase); rows 0,1 also feed its own P_zz
    const double* kP = (cnt & 1) ? S.Z : S.Kt;   // (double-buffered, see the worker side)
    const double2 kA = crow.kA, wA = crow.wA, kB = crow.kB, wB = crow.wB, kC = crow.kC;   // (own rows: from registers)
    sq = S.mseq[min(mnext, S.mcap - 1)];   // next iteration's table entry (static data): its latency hides behind this update
    const bool gated = cur.gate != 0.0;
    // NaN guard (vi_ekf_meas.cpp:247), decided over every row of K in gain_rows (two service waves: the other one's rows too)
    const bool bad = crow.bad != 0 || (ROLE != 0 && sm[44 + cnt % 3 + (ROLE == 2 ? 0 : 8)] != 0.0);
    const double r0 = cur.r0, r1 = cur.r1;
    double2 prn[3] = {};
    if constexpr (MERGE) {
      // correction lambda o (K r)   (vi_ekf_meas.cpp:249-255)
      const double lam0 = partial ? lraw[0] : 1.0, lam1 = partial ? lraw[1] : 1.0, lam2 = partial ? lraw[2] : 1.0;
      const double dv0 = (lam0 * kA.x) * r0 + (lam0 * kA.y) * r1;
      const double dv1 = (lam1 * kB.x) * r0 + (lam1 * kB.y) * r1;
      const double dv2 = (lam2 * kC.x) * r0 + (lam2 * kC.y) * r1;
      const double kw[8] = {wA.x, wA.y, kA.x, kA.y, wB.x, wB.y, kB.x, kB.y};   // (w0,w1,k0,k1) of rows 0,1
      // rotation vector of the correction: bearing  T_zeta [d0 d1],  attitude  [d0 d1 d2]
      double v[3];
      v[0] = isatt ? dv0 : (f1[0] * dv0 + f2[0] * dv1);
      v[1] = isatt ? dv1 : (f1[1] * dv0 + f2[1] * dv1);
      v[2] = isatt ? dv2 : (f1[2] * dv0 + f2[2] * dv1);
      const bool corr = !gated && !bad && !RES_ABLATE(S, 2);
      // x <- x [+] dx  (vi_ekf_helper.cpp:88-98): bearing  exp(T_z d) (x) q ;  attitude  q (x) exp(d) ;  the rest adds.
      // The corrected quaternion / inverse depth stay in registers for fix_depth and the next prediction.
      auto correct = [&]() {
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
      };
      // The next measurement's rows and the state correction need the same inputs (this update's gain and residual) and nothing of
      // each other: in the common case -- an update that is applied, another measurement after it -- they sit in ONE basic block, so
      // that the scheduler fills the latencies of the one dependent chain with the other (a branch around either splits the block).
      // (Worth 1 - 2 % where this wave's chain sets the phase -- one workgroup per CU, the small instances; where the workers do --
      //  the three-worker-wave instances, two workgroups per CU: the headline -- the longer block costs 0.6 %: A/B on one box, r04.)
      if (slot_next >= 0 && corr && !RES_ABLATE(S, 1)) {
        next_rows((cnt + 1) & 1, true, __builtin_amdgcn_readfirstlane(slot_next), kP, crow, prn);
        correct();
        RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 0);
        RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 1);
        RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 0);
      } else {
        if (slot_next >= 0) next_rows((cnt + 1) & 1, !gated && !bad && !RES_ABLATE(S, 1), __builtin_amdgcn_readfirstlane(slot_next), kP, crow, prn);
        RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 0);
        RES_MARK("service.correction");
        RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 1);
        RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 0);
        if (corr) correct();
      }
    } else {   // (the three-worker-wave instances: the loop as it was -- their step time follows this code's layout to the per cent)
      if (slot_next >= 0) next_rows((cnt + 1) & 1, !gated && !bad && !RES_ABLATE(S, 1), __builtin_amdgcn_readfirstlane(slot_next), kP, crow, prn);
      RES_STAMP(S, lane == 0 && it_ < 8, 16 + 4 * it_ + 0);
      RES_MARK("service.correction");
      // correction lambda o (K r)   (vi_ekf_meas.cpp:249-255)
      const double lam0 = partial ? lraw[0] : 1.0, lam1 = partial ? lraw[1] : 1.0, lam2 = partial ? lraw[2] : 1.0;
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
      const bool corr = !gated && !bad && !RES_ABLATE(S, 2);
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
    }
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 1);
    RES_MARK("service.fix_depth");
    // (fix_depth mailbox flag: one word per service wave -- [40 + par] / [36 + par] -- each wave clears and sets its own)
    constexpr int FIXW = (ROLE == 2) ? 36 : 40;
    if (lane == 0) sm[FIXW + par] = 0.0;
    // fix_depth (vi_ekf_meas.cpp:271; a gated update returns before it, :238): almost never fires -- one wave-wide test
    const bool odd_depth = !gated && isfeat && fid < len && !(lin >= 0.0 && lin <= 1e2);
    if (__any(odd_depth)) {
      if (odd_depth) {
        VIEKF_COLD_BEGIN();
        double rho = lin;
        if (rho != rho) { rho = rho_reset; flag |= FLAG_NAN; }
        if (rho < 0.0) {
          const double err = rho_reset - rho;
          S.fixadd[par * N + fid] = err * err;
          sm[FIXW + par] = 1.0;
          rho = rho_reset;
          flag |= FLAG_NEGDEPTH;
        } else if (rho > 1e2) {
          S.fixset[par * N + fid] = 1.0;
          sm[FIXW + par] = 1.0;
          rho = rho_reset;
        }
        lin = rho;
        VIEKF_COLD_END();
      }
    }
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 2);
    RES_MARK("service.predict");
    if (slot_next >= 0) {   // next measurement, from registers, on the wave that holds its feature
      const int sn = __builtin_amdgcn_readfirstlane(slot_next);
      if (holds(sn)) { predict(f1, f2, fz, mnext, lane_of(sn), nxt); if (ROLE != 0) send(nxt, (cnt + 1) & 1, cnt + 2); }
      else recv(nxt, (cnt + 1) & 1, cnt + 2);
    }
    if (PRIMARY && result_all && lane == 0) result_all[(long)S.b * S.mstride + m] = gated ? 1 : 0;
    RES_STAMP(S, lane == 0 && it_ < 8, 128 + 4 * it_ + 3);
    RES_MARK("service.gain_rows");
    if (slot_next >= 0) gain_rows(nxt, 44 + (cnt + 1) % 3, 50 + ((cnt + 1) & 1), prn, (cnt & 1) ? S.Kt : S.Z, nrow);
    RES_MARK("service.phase_tail");
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
  RES_MARK("service.loop_end");

  if (hasq) { qptr[0] = qn[0]; qptr[1] = qn[1]; qptr[2] = qn[2]; qptr[3] = qn[3]; }
  if (haslin) *linptr = lin;
  RES_STAMP(S, lane == 0, 11);
  __syncthreads();  // B5
  RES_STAMP(S, lane == 0, 12);
  // ---------------- store x, status ----------------
  double* xg = a.x_out + S.so * a.nxs;
  const int xend = (a.x_out != a.x || a.smap_out) ? a.nxs : xZ + 5 * len;   // another ring slot gets the whole vector (zeros past the features)
  for (int i = lane; PRIMARY && i < xend; i += 64) {
    const double v = xs[i];
    if (v != v) flag |= FLAG_NAN;
    if (v > 1e6) flag |= FLAG_BLOWUP;
    xg[i] = v;
  }
  if (flag) atomicOr(&a.flags[S.b], flag);
  RES_STAMP(S, lane == 0, 13);
  {   // cooperative store of P (see res_store_chunk): this wave streams its share of every chunk
    StoreChunk sc;
    for (int f0 = 0; store_chunk_at(f0, N, n, S.img_len, sc); f0 = sc.f1) {
      __syncthreads();   // S1
      res_store_chunk<T>(a, S, sc, threadIdx.x);
      __syncthreads();   // S2
    }
  }
}

}  // namespace viekf
