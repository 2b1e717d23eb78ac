// viekf_resident_common.hpp -- resident (fused-step) kernel family: LDS carve-up, shared launch state and the small device
// helpers both roles use.  Overview of the family: viekf_kernels_resident.hpp.
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

// measurements per launch (the host chunks longer lists: P then makes one more HBM round trip per chunk): a frame that measures
// every feature once fits one launch.  (64 up to 64 features: the headline instance is within 740 bytes of its 80 KB.)
__host__ __device__ inline int res_mcap(int N) { return N > 64 ? 80 : 64; }

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
    const int mc = res_mcap(N);   // measurements per launch
    mslot = take(mc / 2); mseq = take(mc); mz = take(2 * mc); mR = take(4 * mc);
    total = o;
  }
};
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

// Accounting build only (-DVIEKF_ISA_MARKS, tools/isa_regions.py): named marks in the ISA listing, fenced so that nothing is
// scheduled across them -- the per-region instruction counts of profiles/r03/isa_step_resident_7_3.json.
#ifdef VIEKF_ISA_MARKS
#define RES_MARK(name)                          \
  do {                                          \
    __builtin_amdgcn_sched_barrier(0);          \
    asm volatile("; @@MARK " name);             \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#else
#define RES_MARK(name) do {} while (0)
#endif

// Timing-only ablation bits (results become wrong) exist in a -DVIEKF_ABLATE diagnostic build only; the product build has none.
#ifdef VIEKF_ABLATE
#define RES_ABLATE(S_, bit) (((S_).dbg & (bit)) != 0)
#else
#define RES_ABLATE(S_, bit) false
#endif

typedef __attribute__((address_space(3))) volatile int lds_vint_t;

struct ResShared {  // resolved LDS pointers + launch constants shared by both roles
  double *xs, *Kt, *Wt, *Praw, *lam, *sm, *fixadd, *fixset, *Z, *phiff, *Abb, *Gb, *Phibb, *Mbb, *Gdb, *Pbb, *T16,
      *xdb, *Pbc, *PhibbT, *Pd, *PsiP, *Pi, *Xi, *AvG, *Lbc, *mz, *mR;
  int* mslot;   // [res_mcap(N)] slot, or -1 for a measurement that is not run
  int2* mseq;   // [res_mcap(N)] {index of the next measurement that runs (or M), its slot (or -1)}: one LDS read per iteration
  BodyCtx* ctx;
  int N, n, nf, len, M, mstride, do_prop, b, dbg, kp, B, img_len, mcap;   // kp: propagates per launch (viekf_batch_step_n)
  long si, so;   // the filter's entry of x / P it is loaded from and stored to (StreamArgs::si / so, read ONCE in the prologue)
  double* stamps;
};

// first m' >= from whose update will actually run (mslot >= 0), else M
__device__ __forceinline__ int res_next_valid(const ResShared& S, int from) {
  int m = from;
  while (m < S.M && S.mslot[m] < 0) m++;
  return m;
}

}  // namespace viekf
