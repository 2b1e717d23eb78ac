// viekf_instances.hpp -- which instances of the fused-step kernels the library holds, and in which translation unit each is
// compiled (viekf_inst.hip, built once per group with -DVIEKF_INST_GROUP=g, in parallel).  viekf_capi.hip sees every instance
// as an `extern template`: it takes their addresses and launches them, the code lives in the group's object file.
#pragma once
#include "viekf_kernels_resident.hpp"
#include "viekf_kernels_tiles.hpp"

#define VIEKF_STEP_ARGS                                                                                                  \
  viekf::StreamArgs, int, const double*, const double*, const double*, const int*, int, int, const double*, long, long, int*

// resident family <RB, NW, NS> (four flavours each: several propagates per launch or one; unit-Lambda or general)
#define VIEKF_RES_LIST_0(X) X(2, 1, 1) X(2, 2, 1)
#define VIEKF_RES_LIST_1(X) X(3, 2, 1) X(4, 3, 1)
#define VIEKF_RES_LIST_2(X) X(5, 3, 1) X(6, 3, 1)
#define VIEKF_RES_LIST_3(X) X(7, 3, 1) X(1, 7, 1)
#define VIEKF_RES_LIST_4(X) X(2, 7, 1) X(3, 7, 1)
#define VIEKF_RES_LIST_5(X) X(5, 6, 2) X(6, 6, 2)
#define VIEKF_RES_LIST_6(X) X(7, 6, 2) X(8, 6, 2)
#define VIEKF_RES_LIST(X) \
  VIEKF_RES_LIST_0(X) VIEKF_RES_LIST_1(X) VIEKF_RES_LIST_2(X) VIEKF_RES_LIST_3(X) VIEKF_RES_LIST_4(X) VIEKF_RES_LIST_5(X) VIEKF_RES_LIST_6(X)
// tile family <NT, NW> (two flavours each)
#define VIEKF_TILE_LIST_7(X) X(11, 3)
#define VIEKF_TILE_LIST(X) VIEKF_TILE_LIST_7(X)
#define VIEKF_INST_GROUPS 8

#define VIEKF_RES_FLAVOURS(PFX, RB, NW, NS)                                                         \
  PFX template __global__ void viekf::k_step_resident<RB, NW, false, NS, false>(VIEKF_STEP_ARGS);  \
  PFX template __global__ void viekf::k_step_resident<RB, NW, false, NS, true>(VIEKF_STEP_ARGS);   \
  PFX template __global__ void viekf::k_step_resident<RB, NW, true, NS, false>(VIEKF_STEP_ARGS);   \
  PFX template __global__ void viekf::k_step_resident<RB, NW, true, NS, true>(VIEKF_STEP_ARGS);
#define VIEKF_TILE_FLAVOURS(PFX, NT, NW)                                              \
  PFX template __global__ void viekf::k_step_tiles<NT, NW, false>(VIEKF_STEP_ARGS);  \
  PFX template __global__ void viekf::k_step_tiles<NT, NW, true>(VIEKF_STEP_ARGS);   \
  PFX template __global__ void viekf::k_step_tiles_pair<NT, false>(VIEKF_STEP_ARGS); \
  PFX template __global__ void viekf::k_step_tiles_pair<NT, true>(VIEKF_STEP_ARGS);
