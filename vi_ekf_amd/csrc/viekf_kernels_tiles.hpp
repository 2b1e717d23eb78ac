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
  tile_prologue<T>(a, S, smem, do_prop, dt_all, z_all, slot_all, M, m_stride, R_all, r_stride_b, r_stride_m, result_all);
  static_assert(NW == 3, "one instantiation of the worker code per wave: extend the dispatch below");
  const int wave = tid >> 6;
  if (wave == NW) tile_service<T, MP>(a, S, tid & 63, u_all, dt_all, result_all);
  else if (wave == 0) tile_worker<NT, NW, 0, MP>(a, S, tid);
  else if (wave == 1) tile_worker<NT, NW, 1, MP>(a, S, tid);
  else tile_worker<NT, NW, 2, MP>(a, S, tid);
}

}  // namespace viekf
