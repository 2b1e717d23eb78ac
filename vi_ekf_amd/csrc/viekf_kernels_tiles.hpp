// viekf_kernels_tiles.hpp -- "tile" kernel family (r03): the fused step (propagate(s) + M sequential feature updates, P on chip for
// the whole step) with P held as 16 x 16 tiles in the fp64 matrix cores' accumulator layout.
//
// Files:  viekf_tiles_common.hpp   why, the row space, LDS carve-up (TileLds), prologue
//         viekf_tiles_worker.hpp   worker waves: load, tile propagate, rank-4 MFMA sweeps, column extraction, store
//         viekf_tiles_service.hpp  service wave: dynamics, state correction, prediction, gate
//         viekf_resident_prop.hpp  (shared with the resident family) the propagate's set-up and body strips
#pragma once
#include "viekf_tiles_common.hpp"
#include "viekf_tiles_worker.hpp"
#include "viekf_tiles_service.hpp"

namespace viekf {

// NT: tiles per side (the instance runs the feature counts with 1 + ceil(N / 5) == NT); NW worker waves + 1 service wave.
template <int NT, int NW, bool MP = false>
__global__ __launch_bounds__((NW + 1) * 64, (NW <= 3) ? 2 : 1) void k_step_tiles(StreamArgs a, int do_prop,
                                                                const double* __restrict__ u_all,
                                                                const double* __restrict__ dt_all,
                                                                const double* __restrict__ z_all,
                                                                const int* __restrict__ slot_all, int M, int m_stride,
                                                                const double* __restrict__ R_all, long r_stride_b,
                                                                long r_stride_m, int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int T = (NW + 1) * 64;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.B) return;
  if (a.active && !a.active[blockIdx.x]) return;   // (the whole workgroup: before any barrier)
  TileShared S;
  tile_prologue<T>(a, S, smem, (int)blockIdx.x, tid, true, do_prop, dt_all, z_all, slot_all, M, m_stride, R_all, r_stride_b, r_stride_m, result_all);
  static_assert(NW == 3, "one instantiation of the worker code per wave: extend the dispatch below");
  const int wave = tid >> 6;
  if (wave == NW) tile_service<T, MP>(a, S, tid & 63, u_all, dt_all, result_all);
  else if (wave == 0) tile_worker<NT, NW, 0, MP>(a, S, tid);
  else if (wave == 1) tile_worker<NT, NW, 1, MP>(a, S, tid);
  else tile_worker<NT, NW, 2, MP>(a, S, tid);
}

// Two filters per 512-thread workgroup, ONE workgroup per CU: filters 2 g and 2 g + 1 of the batch.  Waves 0..2 / 3: filter 0's
// workers / service; wave 4 / 5..7: filter 1's service / workers -- waves k and k + 4 share a SIMD, so each service wave sits
// beside ONE worker wave of the other filter (a matrix-bound wave that issues little), and the two filters' worker waves pair up on
// the remaining SIMDs, where the half-phase stagger of the update loop (viekf_tiles_worker.hpp) alternates their MFMA bursts.
// A filter that does not exist (odd batch) or is masked out idles through the barriers.
template <int MP>
__device__ __forceinline__ void tile_absent(const TileShared& S, int do_prop, int trips) {
  __syncthreads();  // B0
  if (do_prop & 1) {
    const int nkp = MP ? ((do_prop >> 16) > 0 ? (do_prop >> 16) : 1) : 1;
    for (int kp = 0; kp < nkp; kp++)
      for (int i = 0; i < 5; i++) __syncthreads();   // B1p B2p B2q B3p B4p
    __syncthreads();  // B4q
  }
  __syncthreads();  // Bp
  __syncthreads();  // B1
  for (int i = 0; i < 2 * trips + 1; i++) __syncthreads();
  (void)S;
}

template <int NT, bool MP = false>
__global__ __launch_bounds__(512, 1) void k_step_tiles_pair(StreamArgs a, int do_prop, const double* __restrict__ u_all,
                                                            const double* __restrict__ dt_all, const double* __restrict__ z_all,
                                                            const int* __restrict__ slot_all, int M, int m_stride,
                                                            const double* __restrict__ R_all, long r_stride_b, long r_stride_m,
                                                            int* __restrict__ result_all) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int NW = 3;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int half = wave >> 2;                              // which filter of the pair this wave serves
  const int b = 2 * (int)blockIdx.x + half;
  const int b0 = 2 * (int)blockIdx.x, b1 = b0 + 1;
  const bool on0 = b0 < a.B && (!a.active || a.active[b0]), on1 = b1 < a.B && (!a.active || a.active[b1]);
  if (!on0 && !on1) return;                                // (the whole workgroup: before any barrier)
  const bool present = half ? on1 : on0;
  const TileLds L(a.N, a.n, a.nxs);
  double* mine = smem + (size_t)half * L.total;
  TileShared S;
  const int tl = tid & 255;                                // thread index inside the filter's half
  tile_prologue<256>(a, S, mine, present ? b : 0, tl, present, do_prop, dt_all, z_all, slot_all, M, m_stride, R_all, r_stride_b, r_stride_m,
                     result_all);
  const int trips = max((int)smem[L.sm + 60], (int)smem[L.total + L.sm + 60]);
  if (!present) { tile_absent<MP>(S, do_prop, trips); return; }
  // half 0: waves 0, 1, 2 work, wave 3 serves;  half 1: wave 4 serves, waves 5, 6, 7 work
  const int wl = wave & 3;
  const bool service = half ? (wl == 0) : (wl == 3);
  const int w = half ? wl - 1 : wl;                        // worker index 0..2
  const int wt = 64 * w + (tid & 63);                      // worker thread index 0..191
  if (service) tile_service<256, MP, true>(a, S, tid & 63, u_all, dt_all, result_all, half, trips);
  else if (w == 0) tile_worker<NT, NW, 0, MP, true>(a, S, wt, half, trips);
  else if (w == 1) tile_worker<NT, NW, 1, MP, true>(a, S, wt, half, trips);
  else tile_worker<NT, NW, 2, MP, true>(a, S, wt, half, trips);
}

}  // namespace viekf
