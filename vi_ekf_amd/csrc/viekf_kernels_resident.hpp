// viekf_kernels_resident.hpp -- "resident" kernel family: one workgroup per filter, the whole
// covariance lives in VGPRs for the complete step (propagate + M sequential feature updates), so
// P crosses HBM once per step instead of (M+1) times.
//
// Files:  viekf_resident_common.hpp   LDS carve-up (ResLds), shared launch state (ResShared), small device helpers
//         viekf_resident_prop.hpp     the propagate: dynamics hand-over, low-rank coupling set-up (Z records), body strips
//         viekf_resident_worker.hpp   worker waves: blocks of P in registers -- load, contraction, rank-2 sweeps, extraction, store
//         viekf_resident_service.hpp  service wave(s): dynamics, state correction, prediction, gate, gain rows
//         this file                   prologue (state, measurement table) and the kernel
//
// Ownership: the 3N x 3N feature part of P is a grid of 3x3 blocks (I,J).  P is symmetric, so of every
// unordered pair {I,J} only ONE block is kept (I >= J); which worker thread keeps it in which of its RB register slots is a
// table built by the host (8 x 8 tiles of blocks per 64-lane group: build_resmap, viekf_capi.hip).  The 16 body columns
// P[:,0:16] live in LDS for the whole step (the body rows are their mirror).  A rank-2 sweep needs K for the block's rows and
// W for its columns; the propagation Phi P Phi^T + Gd Qu Gd^T is a register-tiled K = 24 contraction over one record per
// row (viekf_resident_common.hpp).  lambda_feat is the same for every feature slot (vi_ekf.cpp:139-144), so the
// partial-update mask Lambda (vi_ekf.cpp:83,146) is ONE 3x3 constant for every feature/feature block.
#pragma once
#include "viekf_resident_common.hpp"
#include "viekf_resident_prop.hpp"
#include "viekf_resident_worker.hpp"
#include "viekf_resident_service.hpp"

namespace viekf {


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
  S.N = a.N; S.mcap = res_mcap(a.N); S.n = a.n; S.nf = 3 * a.N; S.len = a.len[b]; S.M = M; S.mstride = m_stride; S.do_prop = do_prop & 1; S.dbg = (do_prop >> 8) & 0xff; S.kp = (do_prop >> 16) > 0 ? (do_prop >> 16) : 1; S.B = a.B; S.b = b; S.stamps = a.ws;
  S.si = a.si(b); S.so = a.so(b);
  {
    const double* xg = a.x + S.si * a.nxs;
    for (int i = tid; i < a.nxs; i += T) S.xs[i] = (i < xZ + 5 * S.len) ? xg[i] : 0.0;
    for (int i = tid; i < a.n; i += T) S.lam[i] = a.lambda[i];
    for (int i = tid; i < 2 * a.N; i += T) { S.fixadd[i] = 0.0; S.fixset[i] = 0.0; }
    if (tid < 48) {
      const double lk = a.lambda[tid & 15], lq = a.lambda[16 + (tid >> 4)];
      S.Lbc[tid] = a.dp->use_partial_update ? (lk + lq - lq * lk) : 1.0;
    }
    if (tid == 0) { S.sm[42] = (do_prop & 1) ? dt_all[b] : 0.0; S.sm[40] = 0.0; S.sm[41] = 0.0; S.sm[44] = 0.0; S.sm[45] = 0.0; S.sm[46] = 0.0; S.sm[49] = 0.0; S.sm[50] = 0.0; S.sm[51] = 0.0; S.sm[32] = 0.0; S.sm[33] = 0.0; S.sm[52] = 0.0; S.sm[53] = 0.0; S.sm[54] = 0.0; S.sm[36] = 0.0; S.sm[37] = 0.0; }
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
  // per-filter live slots: the slot this launch stores the filter into is its live slot from now on (every thread of the workgroup
  // has read the old entry above; later launches are ordered behind this one on the stream)
  if (a.smap_out && tid == 0) a.smap[b] = (int)S.so;
  for (int mm_ = tid; mm_ < M; mm_ += T) {   // successor table (each entry scans forward; M <= res_mcap(N))
    int nx = mm_ + 1;
    while (nx < M && S.mslot[nx] < 0) nx++;
    S.mseq[mm_] = make_int2(nx, nx < M ? S.mslot[nx] : -1);
  }
  __syncthreads();
  RES_STAMP(S, tid == 0, 63);
}

// ZU: lambda = 1 on the bearing components of every feature (the reference's parameter files: lambda_feat = [1, 1, x]; checked
// by the host), which makes Lambda = 1 for every element of a feature/feature block except (rho, rho).
template <int RB, int NW, bool MP = false, int NS = 1, bool ZU = false>
__global__ __launch_bounds__((NW + NS) * 64, (NW <= 3) ? 2 : 1) void k_step_resident(StreamArgs a, int do_prop,
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
#ifdef VIEKF_ABLATE
  {   // timing experiment (DESIGN.md 5.2, r04 xxxi): are the two workgroups of a CU better off OUT of step?  Half of the workgroups
      // start late by ((bits >> 4) & 7) x 8 us; bit 7 picks which half: 0 = the odd ones, 1 = every second block of 256
    const int bits = (do_prop >> 8) & 0xff, st = (bits >> 4) & 7;
    const bool late = (bits & 128) ? ((blockIdx.x >> 8) & 1) != 0 : (blockIdx.x & 1) != 0;
    if (st && late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)st * 19200ull) __builtin_amdgcn_s_sleep(64);
    }
  }
#endif
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
    else res_worker<RB, TW, MP, T, ZU>(a, S, tid);
  } else {
    if (wave == SVC) res_service<T, MP>(a, S, tid & 63, NW, u_all, dt_all, result_all);
    else res_worker<RB, TW, MP, T, ZU>(a, S, tid - (wave > SVC ? 64 : 0));
  }
}

}  // namespace viekf
