// viekf_capi.hip -- implementation of the C ABI declared in include/viekf.h (libviekf_hip.so).
// Host side of the MI355X-native batched VI-EKF core: owns the device buffers, stages
// host-pointer arguments, launches the gfx950 kernels.  No CPU fallback exists on purpose.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/viekf.h"
#include "viekf_host.hpp"
#include "viekf_instances.hpp"
#include "viekf_kernels_hooks.hpp"
#include "viekf_kernels_wide.hpp"

// (the fused-step kernels are compiled in viekf_inst.hip, one object file per group of instances)
#define RES_EXT(RB, NW, NS) VIEKF_RES_FLAVOURS(extern, RB, NW, NS)
#define TILE_EXT(NT, NW) VIEKF_TILE_FLAVOURS(extern, NT, NW)
VIEKF_RES_LIST(RES_EXT)
VIEKF_TILE_LIST(TILE_EXT)
#undef RES_EXT
#undef TILE_EXT

using namespace viekf;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return fail(VIEKF_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e_));            \
  } while (0)

constexpr int kThreads = 256;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to a device's copy of a kernel and is shared by every batch (and
// every host thread) that launches it: the per-device high-water marks below are raised under this lock, never lowered.
std::mutex g_attr_mutex;

}  // namespace

struct viekf_batch {
  int B = 0, N = 0, nx = 0, nxs = 0, n = 0, ld = 0, device = 0;
  viekf_params params;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double *d_x = nullptr, *d_P = nullptr, *d_Qx = nullptr, *d_lambda = nullptr, *d_ws = nullptr, *d_x0 = nullptr,
         *d_Pdiag = nullptr;
  int* d_len = nullptr;
  unsigned* d_flags = nullptr;
  long ws_stride = 0;
  char* d_stage = nullptr;
  size_t stage_bytes = 0, stage_used = 0;
  int family = 0;       // requested: 0 auto, 1 streaming, 2 resident
  int res_inst = -1;    // resident instance index (-1: N not covered by the resident family)
  bool res_zu = false;  // lambda = 1 on the bearing components (or no partial update): the fused kernel's ZU instances apply
  size_t res_lds = 0;
  DevParams dp;
  DevParams* d_dp = nullptr;
  // P is symmetric and the hot kernels keep only its LOWER triangle current; what is above the diagonal may be stale:
  //   0  all of P valid
  //   2  stale above the diagonal (left by the fused kernels, the matrix-core propagate and the grouped update: all of them read
  //      and write the lower triangle only)
  // ensure_full_P mirrors the lower triangle up before anything that reads all of P.
  int upper_stale = 0;
  int stale_ever = 0;         // the highest level any launch of this batch has left: a ring slot is taken to be that stale when
                              // it becomes (part of) the live state again -- slots carry no level of their own
  int hist_depth = 0;
  int live_slot = -1;        // >= 0: the live (x, P) ARE this slot of the history ring (d_x / d_P point into it)
  double *home_x = nullptr, *home_P = nullptr;   // the batch's own buffers (live state while live_slot < 0)
  double *h_x = nullptr, *h_P = nullptr;
  int* h_len = nullptr;
  unsigned char* d_active = nullptr;   // [B] participation mask of the next propagate / feature-update launches (NULL: all)
  bool active_on = false;
  int* d_resmap = nullptr;             // fused-step kernel: block ownership map [RB][TW] of the chosen instance (build_resmap)
  int* d_ringslot = nullptr;           // [B] staging of per-filter ring slots (viekf_batch_snapshot_filters / _restore_filters)
  // per-filter live ring slots (viekf_batch_select_filters): every filter's live (x, P) is a slot of the ring of its own; d_x / d_P
  // then point at the ring's base and the kernels address filter b through smap[b] = slot_b * B + b (StreamArgs::si).  The map is
  bool per_filter = false;
  std::vector<int32_t> live_slots;     // [B] host mirror
  int* d_smap = nullptr;               // [B] device: smap[b] = live_slots[b] * B + b, kept current in stream order by k_set_smap and by
                                       // the fused kernel itself when it stores a filter into another slot
  int* d_zero = nullptr;               // [B] zeros (gather / scatter between the ring and the batch's own buffers)
  // viekf_batch_set_tuning (tests / experiments; the defaults are what a caller gets)
  // async host inputs (viekf_batch_set_async): pinned staging ring the arguments are copied into at call time
  bool async_host = false;
  char* d_pin = nullptr;       // the device's address of h_pin
  char* h_pin = nullptr;
  size_t pin_bytes = 0, pin_used = 0;
  int tile_inst = -1;          // tile family (P as MFMA accumulator tiles): index into kTileInst, -1 = not used for this batch
  size_t tile_lds = 0;
  int tune_tiles = 0;          // the tile family is opt-in (measured slower than the resident family): 2 single, 3 pair
  int tune_res_inst = -1;      // >= 0: only this index of kResInst is tried
  int tune_unit_lambda = 1;    // 0: never the unit-Lambda instances
  int tune_block_group = 0;    // 16 / 24 / 32: group size of the grouped update where its panel fits
  int tune_stream_mfma = 1;    // 0: the streaming kernels without matrix-core passes
  int tune_panel_svc = 1;      // 0: the grouped update without the service wave (k_update_feat_blocked)
};

namespace {

StreamArgs make_args(const viekf_batch* b) {
  StreamArgs a;
  a.smap = b->per_filter ? b->d_smap : nullptr;
  a.smap_out = nullptr;
  a.x = b->d_x; a.P = b->d_P; a.len = b->d_len; a.flags = b->d_flags;
  a.Qx = b->d_Qx; a.lambda = b->d_lambda; a.ws = b->d_ws;
  a.B = b->B; a.N = b->N; a.nx = b->nx; a.nxs = b->nxs; a.n = b->n; a.ld = b->ld;
  a.ws_stride = b->ws_stride;
  a.dp = b->d_dp;
  a.x_out = b->d_x; a.P_out = b->d_P;
  a.active = b->active_on ? b->d_active : nullptr;
  a.resmap = b->d_resmap;
  return a;
}

constexpr size_t kZeroCopyBytes = 4u << 20;

// bump allocator over one device staging region (host-pointer calls only)
int stage_begin(viekf_batch* b, size_t need) {
  need += 4096;
  if (need > b->stage_bytes) {
    if (b->d_stage) {
      HIP_TRY(hipStreamSynchronize(b->stream));
      HIP_TRY(hipFree(b->d_stage));
      b->d_stage = nullptr;
      b->stage_bytes = 0;
    }
    HIP_TRY(hipMalloc(&b->d_stage, need));
    b->stage_bytes = need;
  }
  b->stage_used = 0;
  if (b->async_host) {
    // A pinned RING on the host side (the copies out of it run later, in stream order) and a ring on the device side too: the
    // previous call's kernel may still be reading its staged arguments, and although the next call's copy is ordered behind it
    // on the stream, a ring lets the copy engine run ahead.  Both wrap after a stream synchronise.
    const size_t ring = std::max<size_t>(64 * need, 8u << 20);   // (a wrap drains the stream: 64 calls of this size apart)
    if (ring > b->pin_bytes) {
      HIP_TRY(hipStreamSynchronize(b->stream));
      if (b->h_pin) HIP_TRY(hipHostFree(b->h_pin));
      b->h_pin = nullptr; b->pin_bytes = 0;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_pin), ring, hipHostMallocMapped));
      void* dp = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dp, b->h_pin, 0));
      b->d_pin = static_cast<char*>(dp);
      b->pin_bytes = ring; b->pin_used = 0;
    }
    if (b->pin_used + need > b->pin_bytes) {   // wrap: everything queued so far has to have left the ring
      HIP_TRY(hipStreamSynchronize(b->stream));
      b->pin_used = 0;
    }
  }
  return VIEKF_OK;
}

void* stage_take(viekf_batch* b, size_t bytes) {
  const size_t off = (b->stage_used + 255) & ~size_t(255);
  b->stage_used = off + bytes;
  return b->d_stage + off;
}

// returns a device pointer for an input array: the pointer itself (device) or a staged copy (host)
template <typename Tp>
int in_ptr(viekf_batch* b, const Tp* src, size_t count, viekf_mem where, const Tp** out) {
  if (!src) { *out = nullptr; return VIEKF_OK; }
  if (where == VIEKF_DEVICE) { *out = src; return VIEKF_OK; }
  Tp* d = static_cast<Tp*>(stage_take(b, count * sizeof(Tp)));
  const void* from = src;
  if (b->async_host) {   // the caller's array may change as soon as the call returns: its bytes go through pinned memory now
    const size_t bytes = count * sizeof(Tp), off = (b->pin_used + 255) & ~size_t(255);
    if (off + bytes > b->pin_bytes) return fail(VIEKF_ERR_INVALID, "async staging overflow");   // (stage_begin sized it)
    std::memcpy(b->h_pin + off, src, bytes);
    b->pin_used = off + bytes;
    from = b->h_pin + off;
    // The kernels read their arguments (an IMU sample and a dt per filter; a frame's pixels and slots: about 1 KB per filter)
    // straight out of the pinned ring: every workgroup fetches its own few hundred bytes across the host link in its prologue,
    // which costs the launch less than copy commands between the kernels cost the stream (each one a switch of engines).
    if (bytes <= kZeroCopyBytes) { *out = reinterpret_cast<const Tp*>(b->d_pin + off); return VIEKF_OK; }
  }
  HIP_TRY(hipMemcpyAsync(d, from, count * sizeof(Tp), hipMemcpyHostToDevice, b->stream));
  *out = d;
  return VIEKF_OK;
}

size_t stage_size(size_t bytes) { return bytes + 256; }

int check_batch(const viekf_batch* b) {
  if (!b) return fail(VIEKF_ERR_INVALID, "null batch handle");
  return VIEKF_OK;
}

size_t lds_propagate(const viekf_batch* b) {
  return sizeof(double) * (size_t)(b->nxs + 256 + 96 + 256 + 256 + 96 + 256 + 256 + 16) + sizeof(BodyCtx) + 16;
}
size_t lds_update(const viekf_batch* b) { return sizeof(double) * (size_t)(b->nxs + 5 * b->n + 32); }

// A grouped update (k_update_feat_blocked) keeps only the lower triangle of P current; the matrix-core propagate reads only
// that and rewrites all of P.  Everything else reads P whole: mirror the lower triangle up first.
int ensure_full_P(viekf_batch* b, int tolerate = 0) {
  if (b->upper_stale <= tolerate) return VIEKF_OK;
  StreamArgs a = make_args(b);
  const int nt = (b->n + 31) / 32;
  hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)(nt * (nt + 1) / 2), b->B), dim3(256), 0, b->stream, a);
  HIP_TRY(hipGetLastError());
  b->upper_stale = 0;
  return VIEKF_OK;
}

// (VIEKF_TUNE_STREAM_MFMA = 0 keeps the kernels without matrix-core passes: experiments)
bool stream_mfma_ok(const viekf_batch* b) { return b->tune_stream_mfma != 0; }

int launch_propagate(viekf_batch* b, const double* d_u, const double* d_dt) {
  if (int rc = ensure_full_P(b, stream_mfma_ok(b) ? 2 : 0)) return rc;
  StreamArgs a = make_args(b);
  if (stream_mfma_ok(b)) {   // feature/feature part on the fp64 matrix cores: reads and writes the lower triangle only
    const size_t wlds = sizeof(double) * (size_t)WideLds(b->N, b->nxs).total;
    if (b->tune_stream_mfma != 2 && 3 * b->N <= 512 && wlds <= 158 * 1024) {   // the K = 24 record form, records in LDS
      {
        std::lock_guard<std::mutex> lk(g_attr_mutex);
        static size_t have[64] = {};
        size_t& hw = have[b->device & 63];
        if (wlds > hw) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_propagate_wide<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
          hw = wlds;
        }
      }
      hipLaunchKernelGGL((k_propagate_wide<512>), dim3(b->B), dim3(512), wlds, b->stream, a, d_u, d_dt);
    } else {                                                                    // r02's K = 38 form, operands staged in global scratch
      hipLaunchKernelGGL((k_propagate_stream<512, true>), dim3(b->B), dim3(512), lds_propagate(b) + sizeof(double) * (9 * (size_t)b->N + 2),
                         b->stream, a, d_u, d_dt);
    }
    b->upper_stale = 2;
    b->stale_ever = 2;
  } else
    hipLaunchKernelGGL((k_propagate_stream<kThreads, false>), dim3(b->B), dim3(kThreads), lds_propagate(b), b->stream, a, d_u,
                       d_dt);
  HIP_TRY(hipGetLastError());
  return VIEKF_OK;
}

// Group size of the grouped update (k_update_feat_blocked): the largest of 32 / 24 / 16 whose panel fits the LDS (fewer passes
// over P for the narrower filters); 0 = the grouped kernel does not apply (then one pass per measurement).
// (measured: at N = 64 groups of 32 are SLOWER than 16 -- 1.96 vs 1.83 ms per step, the sequential panel phase grows with the
//  group -- while N = 100 gains 4 % from 24: the wider groups only where the passes dominate)
// the look-ahead form of the grouped update (k_update_feat_panelsvc): groups of 16 -- the 48 rows of a group's features fit the
// serving wave -- where its double-buffered LDS layout fits next to the panel (N <= 154 at groups of 16)
bool panel_svc(const viekf_batch* b) {
  if (b->tune_panel_svc == 0 || !(b->tune_block_group == 0 || b->tune_block_group == 16) || b->n > 512) return false;
  const PsvLds PL(b->N, b->n, b->nxs, 16);
  return sizeof(double) * (size_t)PL.total + 1024 <= 160 * 1024;
}

int blocked_group(const viekf_batch* b, size_t* lds_bytes) {
  if (!stream_mfma_ok(b) || b->n > 512) return 0;
  if (panel_svc(b)) {
    if (lds_bytes) *lds_bytes = sizeof(double) * (size_t)PsvLds(b->N, b->n, b->nxs, 16).total;
    return 16;
  }
  auto fits = [&](int cand, size_t* bytes) {
    const BlkLds BL(b->N, b->n, b->nxs, cand);
    *bytes = sizeof(double) * (size_t)BL.total;
    return *bytes + 1024 <= 160 * 1024;   // (+ the kernel's small static LDS)
  };
  size_t bytes = 0;
  if (b->tune_block_group && fits(b->tune_block_group, &bytes)) { if (lds_bytes) *lds_bytes = bytes; return b->tune_block_group; }
  for (int cand : {32, 24, 16}) {
    if (cand > 16 && b->n <= 256) continue;
    if (fits(cand, &bytes)) { if (lds_bytes) *lds_bytes = bytes; return cand; }
  }
  return 0;
}

int launch_update(viekf_batch* b, const double* d_z, const int* d_slot, int M, const double* d_R, int r_mode,
                  int* d_res) {
  StreamArgs a = make_args(b);
  long rsb = 0, rsm = 0;
  if (r_mode == 1) rsb = 4;
  else if (r_mode == 2) { rsb = 4L * M; rsm = 4; }
  // wide P, several measurements: the blocked kernel (one HBM pass over P per group of BG measurements, fp64 MFMA pass)
  size_t blds = 0;
  const int bg = blocked_group(b, &blds);
  if (M >= 1 && bg > 0) {   // (a single measurement too: a group of one, no mirror pass before it)
    typedef void (*blk_kernel_t)(StreamArgs, const double*, const int*, int, const double*, long, long, int*);
    const bool sv = panel_svc(b);
    const blk_kernel_t kern = sv ? k_update_feat_panelsvc<512, 16>
                                 : (bg == 32 ? k_update_feat_blocked<512, 32> : (bg == 24 ? k_update_feat_blocked<512, 24> : k_update_feat_blocked<512, 16>));
    {
      std::lock_guard<std::mutex> lk(g_attr_mutex);
      static size_t attr_bytes[64][6] = {};
      size_t& have = attr_bytes[b->device & 63][(bg == 32 ? 2 : (bg == 24 ? 1 : 0)) + (sv ? 3 : 0)];
      if (blds > have) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
        have = blds;
      }
    }
    hipLaunchKernelGGL(kern, dim3(b->B), dim3(512), blds, b->stream, a, d_z, d_slot, M, d_R, rsb, rsm, d_res);
    b->upper_stale = 2;   // (reads and writes the lower triangle only)
    b->stale_ever = 2;
  } else {
    if (int rc = ensure_full_P(b)) return rc;   // (the one-measurement kernel reads whole columns)
    hipLaunchKernelGGL(k_update_feat_stream<kThreads>, dim3(b->B), dim3(kThreads), lds_update(b), b->stream, a, d_z,
                       d_slot, M, d_R, rsb, rsm, d_res);
  }
  HIP_TRY(hipGetLastError());
  return VIEKF_OK;
}

// Resident instances <RB, NW>: NW worker waves (+1 service wave) per workgroup.  Symmetric ownership: the N (N + 1) / 2 owned
// 3x3 blocks are dealt to the NW * 64 worker threads by build_resmap below, at most RB per thread.
struct ResInst { int RB, NW, nmin, nmax, max_lds_kb, NS; };   // NS: service waves (2: the body lanes on a wave of their own)
const ResInst kResInst[] = {
    {2, 1, 1, 15, 40, 1},   // the reference's own sizes (NUM_FEATURES 12, params 20 -> here up to 15): ONE worker wave + the service
                     // wave, four 128-thread workgroups per CU (LDS <= 40 KB) -- a small filter's step is its update chain's latency,
                     // so the CU is filled with chains
    {2, 2, 1, 22, 80, 1},
    {3, 2, 1, 25, 80, 1},   // small filters: 192-thread workgroups, two per CU (LDS <= 80 KB, <= 256 VGPRs): one filter's update chain
                     // runs under the other's sweeps
    {4, 3, 26, 38, 80, 1},   // (a 7-slot instance sweeps 7 slots per update however few the map fills: N = 32 needs 3)
    {5, 3, 39, 43, 80, 1},
    {6, 3, 44, 47, 80, 1},
    {7, 3, 26, 50, 80, 1},   // two 256-thread workgroups per CU at the headline size (7 blocks per thread, LDS <= 80 KB)
    {1, 7, 1, 29, 160, 1},   // one workgroup per CU (small batches): again the smallest instance that holds the size
    {2, 7, 30, 41, 160, 1},
    {3, 7, 1, 50, 160, 1},
    {5, 6, 51, 57, 160, 2},   // more features than one service wave has lanes for (N + 14 > 64): two service waves
    {6, 6, 51, 67, 160, 2},   // ... and, past 64, features 64.. on the body wave's free lanes
    {7, 6, 65, 72, 160, 2},
    {8, 6, 73, 77, 160, 2},   // 8 blocks per thread: the register file's end (3 scratch operations per update in the worker loop)
    // (<4, 5> -- two 384-thread workgroups per CU, three waves per SIMD at <= 168 VGPRs -- measured 32 % slower: dropped)
    // (<4, 6> -- 4 blocks per thread on 6 worker waves, the service wave alone on its SIMD -- measured 4 % slower: dropped)
};

// Ownership map of the fused-step kernel: which 3x3 feature block P[16+3I.., 16+3J..] (I >= J: one of each symmetric pair)
// lives in slot `a` of worker thread t.  Entry [a][t] = I | J << 8 | owned << 16.
//  * slot 0 of the threads t < N holds the diagonal blocks (t, t) (the kernel's own_diag convention);
//  * every other (slot, wave) pair is a GROUP of 64 lanes.  The strictly lower blocks are cut into 8 x 8 tiles of features;
//    a full tile fills one group with lane = 8 i + j  <->  block (8 TI + i, 8 TJ + j).  Every update publishes the column
//    pair of ONE feature s from the registers that hold it: the blocks {., s} then sit in the few groups whose tile row or
//    tile column contains s -- at N = 50 on three worker waves 2.6 groups per wave on average (at most 4) instead of 6.7
//    (at most 7) with the blocks dealt round-robin along wrapped diagonals (r01/r02a), and each group costs its wave the
//    whole extraction body whether one lane matches or eight.  The LDS reads of a tile's operand rows (K rows by i, W rows by
//    j: 48 bytes apart) are conflict-free in every 16-lane service group of ds_read_b128.
//  * what is left (the triangles of the diagonal tiles, the ragged last tile row when N is not a multiple of 8) is packed
//    unit by unit into the remaining lanes, best fit first.
// Returns false when the blocks do not fit RB slots of TW threads.
bool build_resmap(int N, int RB, int NWV, std::vector<int>& map, int* used_slots = nullptr) {
  const int TW = 64 * NWV;
  map.assign((size_t)RB * TW, 0);
  if (N > TW || N > 255) return false;
  auto put = [&](int slot, int t, int I, int J) { map[(size_t)slot * TW + t] = I | (J << 8) | (1 << 16); };
  for (int t = 0; t < N; t++) put(0, t, t, t);
  struct Group { int slot, wave; std::vector<int> free_lanes; };
  std::vector<Group> groups;
  for (int s = 0; s < RB; s++)
    for (int w = 0; w < NWV; w++) {
      Group g{s, w, {}};
      for (int l = 0; l < 64; l++)
        if (!(s == 0 && 64 * w + l < N)) g.free_lanes.push_back(l);
      groups.push_back(g);
    }
  typedef std::vector<std::pair<int, int>> Unit;
  std::vector<Unit> ragged;
  const int kf = N / 8, r = N % 8;
  for (int TI = 0; TI < kf; TI++)
    for (int TJ = 0; TJ < TI; TJ++) {
      // a full group, preferably on wave (TI + TJ) mod NWV: the tiles of one tile row / column then spread over the waves
      const int pref = (TI + TJ) % NWV;
      int best = -1, bestkey = 1 << 30;
      for (int g = 0; g < (int)groups.size(); g++) {
        if (groups[g].free_lanes.size() != 64) continue;
        const int key = ((groups[g].wave - pref + NWV) % NWV) * 64 + groups[g].slot;
        if (key < bestkey) { bestkey = key; best = g; }
      }
      if (best < 0) {
        Unit u;
        for (int i = 0; i < 8; i++)
          for (int j = 0; j < 8; j++) u.push_back({8 * TI + i, 8 * TJ + j});
        ragged.push_back(u);
        continue;
      }
      for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) put(groups[best].slot, 64 * groups[best].wave + 8 * i + j, 8 * TI + i, 8 * TJ + j);
      groups[best].free_lanes.clear();
    }
  for (int TD = 0; TD < kf; TD++) {
    Unit u;
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < i; j++) u.push_back({8 * TD + i, 8 * TD + j});
    ragged.push_back(u);
  }
  if (r) {
    for (int TJ = 0; TJ < kf; TJ++) {
      Unit u;
      for (int i = 0; i < r; i++)
        for (int j = 0; j < 8; j++) u.push_back({8 * kf + i, 8 * TJ + j});
      ragged.push_back(u);
    }
    Unit u;
    for (int i = 0; i < r; i++)
      for (int j = 0; j < i; j++) u.push_back({8 * kf + i, 8 * kf + j});
    if (!u.empty()) ragged.push_back(u);
  }
  std::stable_sort(ragged.begin(), ragged.end(), [](const Unit& x, const Unit& y) { return x.size() > y.size(); });
  for (const Unit& u : ragged) {
    int best = -1;
    size_t bestslack = ~(size_t)0;
    for (int g = 0; g < (int)groups.size(); g++) {
      const size_t f = groups[g].free_lanes.size();
      if (f >= u.size() && f - u.size() < bestslack) { bestslack = f - u.size(); best = g; }
    }
    size_t k = 0;
    if (best >= 0) {
      Group& G = groups[best];
      for (; k < u.size(); k++) { put(G.slot, 64 * G.wave + G.free_lanes.front(), u[k].first, u[k].second); G.free_lanes.erase(G.free_lanes.begin()); }
      continue;
    }
    // no group takes the unit whole: split it over the emptiest ones
    while (k < u.size()) {
      int big = -1;
      for (int g = 0; g < (int)groups.size(); g++)
        if (!groups[g].free_lanes.empty() && (big < 0 || groups[g].free_lanes.size() > groups[big].free_lanes.size())) big = g;
      if (big < 0) return false;
      Group& G = groups[big];
      while (k < u.size() && !G.free_lanes.empty()) {
        put(G.slot, 64 * G.wave + G.free_lanes.front(), u[k].first, u[k].second);
        G.free_lanes.erase(G.free_lanes.begin());
        k++;
      }
    }
  }
  // Which WAVE a group sits on is still free (a wave sweeps all its slots alike): exchange whole groups between waves while
  // that lowers, in this order, the largest number of groups any wave has to publish from for one feature (the update's
  // waves meet at a barrier: the slowest one counts), the sum over the features of that maximum, and the sum of squares.
  // N = 50 on three waves: at most 4 -> 3 groups; N = 64 on six: 4 -> 2.
  if (N <= 128) {
    typedef unsigned __int128 fmask_t;
    const int ng = RB * NWV;
    std::vector<fmask_t> mask(ng, (fmask_t)0);   // features with a block in group (slot, wave) = index slot * NWV + wave
    auto remask = [&](int g) {
      fmask_t mk = 0;
      const int slot = g / NWV, wave = g % NWV;
      for (int l = 0; l < 64; l++) {
        const int e = map[(size_t)slot * TW + 64 * wave + l];
        if (e >> 16) mk |= ((fmask_t)1 << (e & 0xff)) | ((fmask_t)1 << ((e >> 8) & 0xff));
      }
      mask[g] = mk;
    };
    for (int g = 0; g < ng; g++) remask(g);
    struct Cost { long mx, summx, sq; bool operator<(const Cost& o) const { return mx != o.mx ? mx < o.mx : (summx != o.summx ? summx < o.summx : sq < o.sq); } };
    auto cost = [&]() {
      Cost c{0, 0, 0};
      for (int f = 0; f < N; f++) {
        long fm = 0;
        for (int w = 0; w < NWV; w++) {
          long cnt = 0;
          for (int sl = 0; sl < RB; sl++) cnt += (long)((mask[sl * NWV + w] >> f) & 1);
          fm = std::max(fm, cnt);
          c.sq += cnt * cnt;
        }
        c.mx = std::max(c.mx, fm);
        c.summx += fm;
      }
      return c;
    };
    auto whole = [&](int g) { return !(g / NWV == 0 && 64 * (g % NWV) < N); };   // (not sharing its lanes with the diagonal blocks)
    Cost best = cost();
    for (bool improved = true; improved;) {
      improved = false;
      for (int g1 = 0; g1 < ng; g1++)
        for (int g2 = g1 + 1; g2 < ng; g2++) {
          if (!whole(g1) || !whole(g2) || g1 % NWV == g2 % NWV) continue;
          std::swap(mask[g1], mask[g2]);
          const Cost c = cost();
          if (c < best) {
            best = c;
            improved = true;
            int* p1 = &map[(size_t)(g1 / NWV) * TW + 64 * (g1 % NWV)];
            int* p2 = &map[(size_t)(g2 / NWV) * TW + 64 * (g2 % NWV)];
            for (int l = 0; l < 64; l++) std::swap(p1[l], p2[l]);
          } else {
            std::swap(mask[g1], mask[g2]);
          }
        }
    }
  }
  // Which SLOT of its wave a group sits in is free as well: every wave's non-empty groups move to its lowest slots (slot 0
  // keeps the diagonal blocks), and the kernel's per-slot loops stop at the highest slot any wave uses -- a 7-slot instance
  // that holds N = 32 (3 slots' worth of blocks) then sweeps 3 slots per update, not 7.
  int used = 1;
  for (int w = 0; w < NWV; w++) {
    int dst = (64 * w < N) ? 1 : 0;   // (slot 0 of a wave that holds diagonal blocks stays where it is)
    for (int sl = dst; sl < RB; sl++) {
      bool any = false;
      for (int l = 0; l < 64 && !any; l++) any = (map[(size_t)sl * TW + 64 * w + l] >> 16) != 0;
      if (!any) continue;
      if (sl != dst)
        for (int l = 0; l < 64; l++) std::swap(map[(size_t)sl * TW + 64 * w + l], map[(size_t)dst * TW + 64 * w + l]);
      dst++;
    }
    used = std::max(used, dst);
  }
  if (used_slots) *used_slots = used;
  return true;
}

typedef void (*res_kernel_t)(StreamArgs, int, const double*, const double*, const double*, const int*, int, int,
                             const double*, long, long, int*);
// multi: several propagates per launch (viekf_batch_step_n); zu: the unit-Lambda instances (both flavours: step_n must stay
// bit for bit what K propagates and a step give)
template <int RB, int NW, int NS>
res_kernel_t res_pick(bool multi, bool zu) {
  if (multi) return zu ? k_step_resident<RB, NW, true, NS, true> : k_step_resident<RB, NW, true, NS, false>;
  return zu ? k_step_resident<RB, NW, false, NS, true> : k_step_resident<RB, NW, false, NS, false>;
}
res_kernel_t res_kernel(int inst, bool multi = false, bool zu = false) {
  switch (inst) {
    case 0: return res_pick<2, 1, 1>(multi, zu);
    case 1: return res_pick<2, 2, 1>(multi, zu);
    case 2: return res_pick<3, 2, 1>(multi, zu);
    case 3: return res_pick<4, 3, 1>(multi, zu);
    case 4: return res_pick<5, 3, 1>(multi, zu);
    case 5: return res_pick<6, 3, 1>(multi, zu);
    case 6: return res_pick<7, 3, 1>(multi, zu);
    case 7: return res_pick<1, 7, 1>(multi, zu);
    case 8: return res_pick<2, 7, 1>(multi, zu);
    case 9: return res_pick<3, 7, 1>(multi, zu);
    case 10: return res_pick<5, 6, 2>(multi, zu);
    case 11: return res_pick<6, 6, 2>(multi, zu);
    case 12: return res_pick<7, 6, 2>(multi, zu);
    case 13: return res_pick<8, 6, 2>(multi, zu);
  }
  return nullptr;
}

// Tile family instances <NT, NW>: NT tiles per side (an instance runs the feature counts with 1 + ceil(N / 5) == NT: its
// tile -> wave map is compile-time), NW worker waves + 1 service wave (N + 14 <= 64 lanes).  pair: TWO filters per 512-thread
// workgroup, one workgroup per CU, their update loops half a phase out of step (k_step_tiles_pair) -- the form for batches
// beyond one filter per CU; the single form is one filter per 256-thread workgroup.
struct TileInst { int NT, NW, max_lds_kb, pair; };
const TileInst kTileInst[] = {
    {11, 3, 160, 1},   // N = 46 .. 50, pairs: the headline instance
    {11, 3, 80, 0},    // N = 46 .. 50, one filter per workgroup
};

res_kernel_t tile_kernel(int inst, bool multi) {
  switch (inst) {
    case 0: return multi ? k_step_tiles_pair<11, true> : k_step_tiles_pair<11, false>;
    case 1: return multi ? k_step_tiles<11, 3, true> : k_step_tiles<11, 3, false>;
  }
  return nullptr;
}

int setup_tiles(viekf_batch* b) {
  b->tile_inst = -1;
  if (!b->tune_tiles || b->N + 14 > 64 || b->N < 1) return VIEKF_OK;
  const int NT = 1 + (b->N + 4) / 5;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess || cus <= 0) cus = 256;
  for (int i = 0; i < (int)(sizeof(kTileInst) / sizeof(kTileInst[0])); i++) {
    const TileInst& r = kTileInst[i];
    if (r.NT != NT || 16 * NT > 64 * r.NW) continue;   // (one worker thread per tile-space row brings the next column pair up to date)
    // tune_tiles: 0 / 1 the resident family (on the MI355X it is the faster one at every batch size measured, so "automatic"
    // never picks a tile instance), 2 the single form, 3 the paired form -- whatever the batch size
    if (b->tune_tiles == 1) continue;
    if (b->tune_tiles == 2 && r.pair) continue;
    if (b->tune_tiles == 3 && !r.pair) continue;
    const TileLds L(b->N, b->n, b->nxs);
    const size_t lds = sizeof(double) * (size_t)L.total * (r.pair ? 2 : 1);
    if (lds > (size_t)r.max_lds_kb * 1024) continue;
    {
      std::lock_guard<std::mutex> lk(g_attr_mutex);
      static size_t have[64][sizeof(kTileInst) / sizeof(kTileInst[0])] = {};
      size_t& hw = have[b->device & 63][i];
      if (lds > hw) {
        for (int fl = 0; fl < 2; fl++)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_kernel(i, fl != 0)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hw = lds;
      }
    }
    b->tile_inst = i; b->tile_lds = lds;
    break;
  }
  return VIEKF_OK;
}

int setup_resident(viekf_batch* b) {
  b->res_inst = -1;
  const bool force = b->tune_res_inst >= 0;   // (VIEKF_TUNE_RES_INSTANCE: pick an instance by index)
  for (int i = 0; i < (int)(sizeof(kResInst) / sizeof(kResInst[0])); i++) {
    const ResInst& r = kResInst[i];
    if (force && b->tune_res_inst != i) continue;
    if (b->N < r.nmin || b->N > r.nmax) continue;
    if (b->N * (b->N + 1) / 2 > r.RB * r.NW * 64 || b->N > r.NW * 64) continue;
    const ResLds L(b->N, b->n, b->nxs);
    const size_t lds = sizeof(double) * (size_t)L.total;
    if (lds > (size_t)r.max_lds_kb * 1024) continue;
    if (r.max_lds_kb <= 80 && !force) {   // two small workgroups per CU only pay when the batch fills the CUs more than once
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess || cus <= 0) cus = 256;
      if (b->B <= cus) continue;
      if (r.max_lds_kb <= 40 && b->B <= 2 * cus) continue;   // (four per CU: only when two per CU would leave filters waiting)
    }
    std::vector<int> map;
    if (!build_resmap(b->N, r.RB, r.NW, map)) continue;
    if (b->d_resmap) { HIP_TRY(hipFree(b->d_resmap)); b->d_resmap = nullptr; }
    HIP_TRY(hipMalloc(&b->d_resmap, sizeof(int) * map.size()));
    HIP_TRY(hipMemcpy(b->d_resmap, map.data(), sizeof(int) * map.size(), hipMemcpyHostToDevice));
    // (the attribute belongs to the DEVICE's copy of the kernel and is shared by every batch that runs this instance: it is a
    //  high-water mark, never lowered -- a second batch with fewer features must not take the first one's LDS away)
    {
      std::lock_guard<std::mutex> lk(g_attr_mutex);
      static size_t have[64][sizeof(kResInst) / sizeof(kResInst[0])] = {};
      size_t& hw = have[b->device & 63][i];
      if (lds > hw) {
        for (int fl = 0; fl < 4; fl++)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(res_kernel(i, (fl & 1) != 0, (fl & 2) != 0)),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hw = lds;
      }
    }
    b->res_inst = i; b->res_lds = lds;
    break;
  }
  return VIEKF_OK;
}

// Timing-only ablation (results become wrong) exists in a -DVIEKF_ABLATE diagnostic build only (tools/build_variant.sh): the
// bits 1 skip sweeps, 2 skip state correction, 4 skip the column extraction come from VIEKF_DEBUG_ABLATE there.  The product
// library has no such switch.
#ifdef VIEKF_ABLATE
int dbg_bits() {
  static const int v = []() { const char* e = getenv("VIEKF_DEBUG_ABLATE"); return e ? atoi(e) : 0; }();
  return v;
}
#else
constexpr int dbg_bits() { return 0; }
#endif

bool use_tiles(const viekf_batch* b) { return b->tile_inst >= 0 && b->family != 1; }
bool use_resident(const viekf_batch* b) { return (b->res_inst >= 0 || b->tile_inst >= 0) && b->family != 1; }

// one launch handles at most res_mcap(N) measurements; longer lists are chunked (P makes one extra HBM round trip per chunk)
int launch_resident(viekf_batch* b, bool do_prop, const double* d_u, const double* d_dt, const double* d_z,
                    const int* d_slot, int M, const double* d_R, int r_mode, int* d_res, double* x_out = nullptr,
                    double* P_out = nullptr, int KP = 1, const int* smap_out = nullptr) {
  // (the fused kernel loads the lower triangle only and stores the lower triangle only: no symmetrisation before or after)
  StreamArgs a = make_args(b);
  if (x_out) { a.x_out = x_out; a.P_out = P_out; }   // (only meaningful for a single-chunk launch)
  a.smap_out = smap_out;
  long rsb = 0, rsm = 0;
  if (r_mode == 1) rsb = 4;
  else if (r_mode == 2) { rsb = 4L * M; rsm = 4; }
  const bool tiles = use_tiles(b);
  const res_kernel_t kern = tiles ? tile_kernel(b->tile_inst, KP > 1) : res_kernel(b->res_inst, KP > 1, b->res_zu);
  const bool pair = tiles && kTileInst[b->tile_inst].pair;
  const int threads = tiles ? (pair ? 512 : (kTileInst[b->tile_inst].NW + 1) * 64) : (kResInst[b->res_inst].NW + kResInst[b->res_inst].NS) * 64;
  const unsigned grid = pair ? (unsigned)((b->B + 1) / 2) : (unsigned)b->B;
  const size_t lds = tiles ? b->tile_lds : b->res_lds;
  int m0 = 0;
  do {
    const int cap = res_mcap(b->N);
    const int mc = (M - m0 < cap) ? (M - m0) : cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, b->stream, a,
                       ((do_prop && m0 == 0) ? (1 | (KP << 16)) : 0) | ((dbg_bits() & 0xff) << 8), d_u, d_dt, d_z ? d_z + 2L * m0 : nullptr,
                       d_slot ? d_slot + m0 : nullptr, mc, M, d_R ? d_R + rsm * m0 : nullptr, rsb, rsm,
                       d_res ? d_res + m0 : nullptr);
    HIP_TRY(hipGetLastError());
    m0 += mc;
  } while (m0 < M);
  b->upper_stale = 2;
  b->stale_ever = 2;
  return VIEKF_OK;
}

// per-filter mode: (x, P) of every filter between its live ring slot and the batch's own buffers (to_home != 0: ring -> home)
int gather_scatter_home(viekf_batch* b, int to_home) {
  if (!b->d_zero) {
    HIP_TRY(hipMalloc(&b->d_zero, sizeof(int) * (size_t)b->B));
    HIP_TRY(hipMemsetAsync(b->d_zero, 0, sizeof(int) * (size_t)b->B, b->stream));
  }
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_ring_copy, dim3(b->B), dim3(256), 0, b->stream, a, b->home_x, b->home_P, b->d_zero, to_home, 1);
  HIP_TRY(hipGetLastError());
  return VIEKF_OK;
}

size_t r_count(const viekf_batch* b, int M, int r_mode) {
  return r_mode == 0 ? 4 : (r_mode == 1 ? 4 * (size_t)b->B : 4 * (size_t)b->B * M);
}

}  // namespace

extern "C" {

int viekf_abi_version(void) { return VIEKF_ABI_VERSION; }

const char* viekf_last_error(void) { return g_last_error.c_str(); }

int viekf_device_count(int32_t* count) {
  if (!count) return fail(VIEKF_ERR_INVALID, "count is null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
  *count = c;
  return VIEKF_OK;
}

int viekf_params_default(viekf_params* p) {
  if (!p) return fail(VIEKF_ERR_INVALID, "params is null");
  std::memset(p, 0, sizeof(*p));
  p->x0[6] = 1.0;
  p->q_b_c[0] = 1.0;
  p->q_b_u[0] = 1.0;
  p->focal_len[0] = p->focal_len[1] = 1.0;
  for (int i = 0; i < 16; i++) p->lambda[i] = 1.0;
  for (int i = 0; i < 3; i++) p->lambda_feat[i] = 1.0;
  p->min_depth = 1.5;
  p->keyframe_overlap_threshold = 0.8;
  p->use_drag_term = 1;
  p->use_partial_update = 1;
  p->use_keyframe_reset = 1;
  std::snprintf(p->name, sizeof p->name, "ekf");
  return VIEKF_OK;
}

// VIEKF::load, reference src/vi_ekf/vi_ekf.cpp:101-131 (same keys, same required lengths)
int viekf_params_load_yaml(const char* path, viekf_params* p) {
  if (!path || !p) return fail(VIEKF_ERR_INVALID, "null argument");
  viekf_params_default(p);
  YamlMap m;
  std::string err, name;
  if (!yaml_parse_file(path, m, err)) return fail(VIEKF_ERR_YAML, err);
  double v = 0.0;
#define GET(key, dst, cnt) \
  if (!yaml_get_doubles(m, key, dst, cnt, err)) return fail(VIEKF_ERR_YAML, std::string(path) + ": " + err)
  if (!yaml_get_string(m, "name", name, err)) return fail(VIEKF_ERR_YAML, std::string(path) + ": " + err);
  std::snprintf(p->name, sizeof p->name, "%s", name.c_str());
  GET("min_depth", &p->min_depth, 1);
  GET("keyframe_overlap_threshold", &p->keyframe_overlap_threshold, 1);
  GET("use_drag_term", &v, 1); p->use_drag_term = v != 0.0;
  GET("use_partial_update", &v, 1); p->use_partial_update = v != 0.0;
  GET("use_keyframe_reset", &v, 1); p->use_keyframe_reset = v != 0.0;
  GET("x0", p->x0, 17);
  GET("P0", p->P0, 16);
  GET("Qx", p->Qx, 16);
  GET("Qu", p->Qu, 6);
  GET("lambda", p->lambda, 16);
  GET("P0_feat", p->P0_feat, 3);
  GET("Qx_feat", p->Qx_feat, 3);
  GET("lambda_feat", p->lambda_feat, 3);
  GET("cam_center", p->cam_center, 2);
  GET("focal_len", p->focal_len, 2);
  GET("q_b_c", p->q_b_c, 4);
  GET("p_b_c", p->p_b_c, 3);
  GET("q_b_u", p->q_b_u, 4);
#undef GET
  return VIEKF_OK;
}

int viekf_batch_create(int32_t batch, int32_t num_features, const viekf_params* p, int32_t device,
                       viekf_batch** out) {
  if (!p || !out) return fail(VIEKF_ERR_INVALID, "null argument");
  *out = nullptr;
  if (batch <= 0 || num_features < 0 || num_features > 4096)
    return fail(VIEKF_ERR_INVALID, "batch must be > 0 and 0 <= num_features <= 4096");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    return fail(VIEKF_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
  }
  if (device < 0 || device >= ndev) return fail(VIEKF_ERR_NO_DEVICE, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  viekf_batch* b = new viekf_batch();
  b->B = batch; b->N = num_features; b->device = device;
  b->nx = 17 + 5 * num_features;
  b->n = 16 + 3 * num_features;
  b->nxs = (b->nx + 1) & ~1;
  // Column stride of P.  The streaming family (N > 77: P crosses HBM several times per step) gets columns that start on whole 128-B
  // lines, so that the 16 rows of a tile column are ONE line instead of parts of two: tools/micro/cu_stream_rate moves 6.4 instead
  // of 5.2 TB/s with the wide-P pass's access pattern, the N = 150 step went from 5.66 to 5.3 ms.  The on-chip family reads and
  // writes P once per launch and is 0.9 % SLOWER with padded columns at N = 50 (0.3495 against 0.3464 ms, three alternating runs
  // of each on one box: 239 instead of 226 MB of P next to the 256 MiB Infinity Cache): it keeps the dense stride.
  b->ld = num_features > 77 ? (b->n + 15) & ~15 : (b->n + 1) & ~1;
#ifdef VIEKF_LD_PAD_ALL               // (diagnostic build, tools/build_variant.sh: A/B of the padded stride on the on-chip family)
  b->ld = (b->n + 15) & ~15;
#endif
  b->params = *p;
  DevParams& d = b->dp;
  std::memcpy(d.Qu, p->Qu, sizeof d.Qu);
  std::memcpy(d.P0_feat, p->P0_feat, sizeof d.P0_feat);
  std::memcpy(d.cam_center, p->cam_center, sizeof d.cam_center);
  std::memcpy(d.focal, p->focal_len, sizeof d.focal);
  std::memcpy(d.q_b_c, p->q_b_c, sizeof d.q_b_c);
  std::memcpy(d.p_b_c, p->p_b_c, sizeof d.p_b_c);
  std::memcpy(d.q_b_u, p->q_b_u, sizeof d.q_b_u);
  d.min_depth = p->min_depth;
  d.use_drag_term = p->use_drag_term;
  d.use_partial_update = p->use_partial_update;
  for (int i = 0; i < 6; i++) d.sqrtQu[i] = std::sqrt(p->Qu[i] > 0.0 ? p->Qu[i] : 0.0);
  const WsLayout L(b->N, b->n);
  b->ws_stride = L.total;
#define ALLOC(ptr, bytes)                                                        \
  do {                                                                           \
    hipError_t e_ = hipMalloc(&(ptr), (bytes));                                  \
    if (e_ != hipSuccess) {                                                      \
      viekf_batch_destroy(b);                                                    \
      return fail(VIEKF_ERR_HIP, std::string("hipMalloc failed: ") + hipGetErrorString(e_)); \
    }                                                                            \
  } while (0)
  ALLOC(b->d_x, sizeof(double) * (size_t)batch * b->nxs);
  ALLOC(b->d_P, sizeof(double) * (size_t)batch * b->n * b->ld);
  ALLOC(b->d_Qx, sizeof(double) * (size_t)b->n);
  ALLOC(b->d_lambda, sizeof(double) * (size_t)b->n);
  ALLOC(b->d_Pdiag, sizeof(double) * (size_t)b->n);
  ALLOC(b->d_x0, sizeof(double) * 17);
  ALLOC(b->d_ws, sizeof(double) * (size_t)batch * b->ws_stride);
  ALLOC(b->d_len, sizeof(int) * (size_t)batch);
  ALLOC(b->d_flags, sizeof(unsigned) * (size_t)batch);
  ALLOC(b->d_dp, sizeof(DevParams));
#undef ALLOC
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    viekf_batch_destroy(b);
    return fail(VIEKF_ERR_HIP, std::string("hipStreamCreate failed: ") + hipGetErrorString(e));
  }
  b->own_stream = true;
  // shared per-batch vectors: Qx diag, lambda, initial P diagonal (vi_ekf.cpp:134-146)
  std::vector<double> Qx(b->n), lam(b->n), Pd(b->n);
  for (int i = 0; i < 16; i++) { Qx[i] = p->Qx[i]; lam[i] = p->lambda[i]; Pd[i] = p->P0[i]; }
  for (int i = 0; i < b->N; i++)
    for (int k = 0; k < 3; k++) {
      Qx[16 + 3 * i + k] = p->Qx_feat[k];
      lam[16 + 3 * i + k] = p->lambda_feat[k];
      Pd[16 + 3 * i + k] = p->P0_feat[k];
    }
  int rc = VIEKF_OK;
  auto up = [&](double* dptr, const double* h, size_t cnt) {
    if (hipMemcpy(dptr, h, cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = VIEKF_ERR_HIP;
  };
  up(b->d_Qx, Qx.data(), b->n);
  up(b->d_lambda, lam.data(), b->n);
  up(b->d_Pdiag, Pd.data(), b->n);
  up(b->d_x0, p->x0, 17);
  if (hipMemcpy(b->d_dp, &b->dp, sizeof(DevParams), hipMemcpyHostToDevice) != hipSuccess) rc = VIEKF_ERR_HIP;
  b->res_zu = !p->use_partial_update || (p->lambda_feat[0] == 1.0 && p->lambda_feat[1] == 1.0);
  if (rc == VIEKF_OK) rc = setup_resident(b);
  if (rc == VIEKF_OK) rc = setup_tiles(b);
  if (rc == VIEKF_OK) rc = viekf_batch_reset(b);
  if (rc != VIEKF_OK) {
    viekf_batch_destroy(b);
    return rc == VIEKF_ERR_HIP ? fail(rc, "parameter upload failed") : rc;
  }
  *out = b;
  return VIEKF_OK;
}

int viekf_batch_destroy(viekf_batch* b) {
  if (!b) return VIEKF_OK;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->d_smap) (void)hipFree(b->d_smap);
  if (b->d_zero) (void)hipFree(b->d_zero);
  void* ptrs[] = {b->home_x ? b->home_x : b->d_x, b->home_P ? b->home_P : b->d_P, b->d_Qx, b->d_lambda, b->d_Pdiag, b->d_x0, b->d_ws, b->d_len, b->d_flags, b->d_stage, b->d_dp, b->h_x, b->h_P, b->h_len, b->d_active, b->d_ringslot, b->d_resmap};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (b->h_pin) (void)hipHostFree(b->h_pin);
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  return VIEKF_OK;
}

int viekf_batch_reset(viekf_batch* b) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_reset, dim3(b->B), dim3(256), 0, b->stream, a, b->d_x0, b->d_Pdiag);
  HIP_TRY(hipGetLastError());
  b->upper_stale = 0;
  HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_dims(const viekf_batch* b, int32_t* batch, int32_t* num_features, int32_t* nx, int32_t* n) {
  if (int rc = check_batch(b)) return rc;
  if (batch) *batch = b->B;
  if (num_features) *num_features = b->N;
  if (nx) *nx = b->nx;
  if (n) *n = b->n;
  return VIEKF_OK;
}

int viekf_batch_describe(const viekf_batch* b, char* out, int32_t cap) {
  if (int rc = check_batch(b)) return rc;
  if (!out || cap < 1) return fail(VIEKF_ERR_INVALID, "describe: no buffer");
  char buf[256];
  if (use_tiles(b)) {
    const TileInst& r = kTileInst[b->tile_inst];
    snprintf(buf, sizeof buf, "%s<%d>: P as %d 16x16 fp64-MFMA accumulator tiles on %d worker waves + 1 service wave per filter, %s (LDS %zu KB)",
             r.pair ? "k_step_tiles_pair" : "k_step_tiles", r.NT, r.NT * (r.NT + 1) / 2, r.NW,
             r.pair ? "two filters per 512-thread workgroup half a phase out of step, one workgroup per CU" : "one filter per 256-thread workgroup, 2 workgroups per CU",
             b->tile_lds / 1024);
  } else if (use_resident(b)) {
    const ResInst& r = kResInst[b->res_inst];
    snprintf(buf, sizeof buf, "k_step_resident<%d,%d>%s: %d worker waves x %d blocks + %d service wave%s, %s per CU (LDS %zu KB)",
             r.RB, r.NW, b->res_zu ? " ZU" : "", r.NW, r.RB, r.NS, r.NS > 1 ? "s" : "",
             r.max_lds_kb <= 40 ? "4 workgroups" : (r.max_lds_kb <= 80 ? "2 workgroups" : "1 workgroup"), b->res_lds / 1024);
  } else {
    const int bg = blocked_group(b, nullptr);
    if (bg > 0)
      snprintf(buf, sizeof buf, "%s + %s<512,%d> (P in HBM/L2, one pass per group of %d measurements, fp64 MFMA "
               "passes, lower triangle only)", (b->tune_stream_mfma != 2 && 3 * b->N <= 512) ? "k_propagate_wide (K = 24 records in LDS)" : "k_propagate_stream",
               panel_svc(b) ? "k_update_feat_panelsvc" : "k_update_feat_blocked", bg, bg);
    else
      snprintf(buf, sizeof buf, "k_propagate_stream + k_update_feat_stream (P in HBM/L2, one pass per measurement)");
  }
  snprintf(out, (size_t)cap, "%s", buf);
  return VIEKF_OK;
}

int viekf_batch_get_params(const viekf_batch* b, viekf_params* out) {
  if (!b || !out) return fail(VIEKF_ERR_INVALID, "null argument");
  *out = b->params;
  return VIEKF_OK;
}

int viekf_batch_set_stream(viekf_batch* b, void* hip_stream) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (b->own_stream && b->stream) { HIP_TRY(hipStreamDestroy(b->stream)); b->stream = nullptr; b->own_stream = false; }
  // NULL is HIP's default (null) stream, exactly as a hipStream_t of 0 means everywhere else
  b->stream = static_cast<hipStream_t>(hip_stream);
  b->own_stream = false;
  return VIEKF_OK;
}

int viekf_batch_sync(viekf_batch* b) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_set_async(viekf_batch* b, int32_t async_host) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->async_host = async_host != 0;
  return VIEKF_OK;
}

int viekf_batch_set_kernel(viekf_batch* b, int32_t family) {
  if (int rc = check_batch(b)) return rc;
  if (family < 0 || family > 2) return fail(VIEKF_ERR_INVALID, "kernel family must be 0, 1 or 2");
  if (family == 2 && b->res_inst < 0 && b->tile_inst < 0)
    return fail(VIEKF_ERR_UNSUPPORTED, "resident kernel family does not cover this num_features (1..77)");
  b->family = family;
  return VIEKF_OK;
}

int viekf_batch_set_tuning(viekf_batch* b, int32_t key, int32_t value) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  switch (key) {
    case VIEKF_TUNE_RES_INSTANCE:
      if (value < -1 || value >= (int)(sizeof(kResInst) / sizeof(kResInst[0]))) return fail(VIEKF_ERR_INVALID, "no such resident instance");
      b->tune_res_inst = value;
      break;
    case VIEKF_TUNE_UNIT_LAMBDA:
      b->tune_unit_lambda = value != 0;
      b->res_zu = b->tune_unit_lambda && (!b->params.use_partial_update || (b->params.lambda_feat[0] == 1.0 && b->params.lambda_feat[1] == 1.0));
      return VIEKF_OK;
    case VIEKF_TUNE_BLOCK_GROUP:
      if (value != 0 && value != 16 && value != 24 && value != 32) return fail(VIEKF_ERR_INVALID, "group size must be 0 (auto), 16, 24 or 32");
      b->tune_block_group = value;
      return VIEKF_OK;
    case VIEKF_TUNE_PANEL_SERVICE:
      b->tune_panel_svc = value != 0;
      return VIEKF_OK;
    case VIEKF_TUNE_TILES:
      if (value < 0 || value > 3) return fail(VIEKF_ERR_INVALID, "tile family: 0 off, 1 automatic, 2 single form, 3 paired form");
      b->tune_tiles = value;
      HIP_TRY(hipStreamSynchronize(b->stream));
      return setup_tiles(b);
    case VIEKF_TUNE_STREAM_MFMA:
      if (value < 0 || value > 2) return fail(VIEKF_ERR_INVALID, "stream MFMA: 0 off, 1 on, 2 on with r02's scratch-staged propagate");
      b->tune_stream_mfma = value;
      if (!b->tune_stream_mfma) { if (int rc = ensure_full_P(b)) return rc; }   // (the plain kernels read all of P)
      return VIEKF_OK;
    default:
      return fail(VIEKF_ERR_INVALID, "unknown tuning key");
  }
  HIP_TRY(hipStreamSynchronize(b->stream));   // (the ownership map of the old instance may still be in use)
  if (int rc = setup_resident(b)) return rc;
  if (b->tune_res_inst >= 0 && b->res_inst < 0) {
    b->tune_res_inst = -1;
    if (int rc = setup_resident(b)) return rc;
    return fail(VIEKF_ERR_UNSUPPORTED, "this resident instance does not hold the batch's num_features");
  }
  return VIEKF_OK;
}

int viekf_batch_get_state(viekf_batch* b, double* x, double* P, int32_t* len_features, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  const hipMemcpyKind kind = where == VIEKF_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const double *sx = b->d_x, *sP = b->d_P;
  if (P) if (int rc = ensure_full_P(b)) return rc;
  if (b->per_filter && (x || P)) {   // every filter's live slot -> the batch's own buffers, then out as usual
    if (int rc = gather_scatter_home(b, 1)) return rc;
    sx = b->home_x; sP = b->home_P;
  }
  if (x)
    HIP_TRY(hipMemcpy2DAsync(x, sizeof(double) * b->nx, sx, sizeof(double) * b->nxs, sizeof(double) * b->nx, b->B,
                             kind, b->stream));
  if (P)
    HIP_TRY(hipMemcpy2DAsync(P, sizeof(double) * b->n, sP, sizeof(double) * b->ld, sizeof(double) * b->n,
                             (size_t)b->B * b->n, kind, b->stream));
  if (len_features) HIP_TRY(hipMemcpyAsync(len_features, b->d_len, sizeof(int32_t) * b->B, kind, b->stream));
  if (where == VIEKF_HOST) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_set_state(viekf_batch* b, const double* x, const double* P, const int32_t* len_features,
                          viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  const hipMemcpyKind kind = where == VIEKF_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (len_features && where == VIEKF_HOST)
    for (int i = 0; i < b->B; i++)
      if (len_features[i] < 0 || len_features[i] > b->N)
        return fail(VIEKF_ERR_INVALID, "len_features out of range");
  double *tx = b->d_x, *tP = b->d_P;
  if (b->per_filter && (x || P)) {   // through the batch's own buffers: what is not given keeps its value
    if (int rc = ensure_full_P(b)) return rc;
    if (int rc = gather_scatter_home(b, 1)) return rc;
    tx = b->home_x; tP = b->home_P;
  }
  if (x)
    HIP_TRY(hipMemcpy2DAsync(tx, sizeof(double) * b->nxs, x, sizeof(double) * b->nx, sizeof(double) * b->nx, b->B,
                             kind, b->stream));
  if (P)
    HIP_TRY(hipMemcpy2DAsync(tP, sizeof(double) * b->ld, P, sizeof(double) * b->n, sizeof(double) * b->n,
                             (size_t)b->B * b->n, kind, b->stream));
  if (b->per_filter && (x || P))
    if (int rc = gather_scatter_home(b, 0)) return rc;
  if (P) {
    // the kernels keep P exactly symmetric and rely on it (their rank-2 update equals the reference's Joseph form only then)
    StreamArgs a = make_args(b);
    const long tot = (long)b->n * b->n;
    hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)((tot + 255) / 256), b->B), dim3(256), 0, b->stream, a);
    HIP_TRY(hipGetLastError());
    b->upper_stale = 0;
  }
  if (len_features) HIP_TRY(hipMemcpyAsync(b->d_len, len_features, sizeof(int32_t) * b->B, kind, b->stream));
  if (where == VIEKF_HOST) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_get_status(viekf_batch* b, uint32_t* flags, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!flags) return fail(VIEKF_ERR_INVALID, "flags is null");
  HIP_TRY(hipSetDevice(b->device));
  const hipMemcpyKind kind = where == VIEKF_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  HIP_TRY(hipMemcpyAsync(flags, b->d_flags, sizeof(uint32_t) * b->B, kind, b->stream));
  if (where == VIEKF_HOST) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_propagate(viekf_batch* b, const double* u, const double* dt, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || !dt) return fail(VIEKF_ERR_INVALID, "u and dt must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const double *d_u = nullptr, *d_dt = nullptr;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(sizeof(double) * 6 * b->B) + stage_size(sizeof(double) * b->B))) return rc;
  if (int rc = in_ptr(b, u, (size_t)6 * b->B, where, &d_u)) return rc;
  if (int rc = in_ptr(b, dt, (size_t)b->B, where, &d_dt)) return rc;
  if (use_resident(b)) {
    if (int rc = launch_resident(b, true, d_u, d_dt, nullptr, nullptr, 0, nullptr, 0, nullptr)) return rc;
  } else {
    if (int rc = launch_propagate(b, d_u, d_dt)) return rc;
  }
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_init_feature(viekf_batch* b, const double* pix, const double* depth, const uint8_t* mask, int32_t* ok,
                             viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!pix) return fail(VIEKF_ERR_INVALID, "pix must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const double *d_pix = nullptr, *d_depth = nullptr;
  const uint8_t* d_mask = nullptr;
  int* d_ok = nullptr;
  if (where == VIEKF_HOST) {
    if (int rc = stage_begin(b, stage_size(sizeof(double) * 2 * b->B) + stage_size(sizeof(double) * b->B) +
                                    stage_size(b->B) + stage_size(sizeof(int) * b->B)))
      return rc;
  }
  if (int rc = in_ptr(b, pix, (size_t)2 * b->B, where, &d_pix)) return rc;
  if (int rc = in_ptr(b, depth, (size_t)b->B, where, &d_depth)) return rc;
  if (int rc = in_ptr(b, mask, (size_t)b->B, where, &d_mask)) return rc;
  if (ok) d_ok = where == VIEKF_DEVICE ? ok : static_cast<int*>(stage_take(b, sizeof(int) * b->B));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_init_feature<kThreads>, dim3(b->B), dim3(kThreads), 0, b->stream, a, d_pix, d_depth, d_mask,
                     d_ok);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (ok) HIP_TRY(hipMemcpyAsync(ok, d_ok, sizeof(int) * b->B, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

static int update_or_step(viekf_batch* b, const double* u, const double* dt, bool with_propagate, const double* z,
                          const int32_t* slot, int32_t M, const double* R, int32_t r_mode, int32_t* result,
                          viekf_mem where, int K = 1) {
  if (int rc = check_batch(b)) return rc;
  if (K < 1 || K > 64) return fail(VIEKF_ERR_INVALID, "1 <= K <= 64 propagates per call");
  if (M < 0) return fail(VIEKF_ERR_INVALID, "M must be >= 0");
  if (r_mode < 0 || r_mode > 2) return fail(VIEKF_ERR_INVALID, "r_mode must be 0, 1 or 2");
  if (M > 0 && (!z || !slot || !R)) return fail(VIEKF_ERR_INVALID, "z, slot and R must not be null when M > 0");
  if (with_propagate && (!u || !dt)) return fail(VIEKF_ERR_INVALID, "u and dt must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t BM = (size_t)b->B * (size_t)M;
  const double *d_u = nullptr, *d_dt = nullptr, *d_z = nullptr, *d_R = nullptr;
  const int32_t* d_slot = nullptr;
  int32_t* d_res = nullptr;
  if (where == VIEKF_HOST) {
    size_t need = stage_size(sizeof(double) * 6 * b->B * K) + stage_size(sizeof(double) * b->B * K) +
                  stage_size(sizeof(double) * 2 * BM) + stage_size(sizeof(int32_t) * BM) * 2 +
                  stage_size(sizeof(double) * r_count(b, M, r_mode));
    if (int rc = stage_begin(b, need)) return rc;
  }
  if (with_propagate) {
    if (int rc = in_ptr(b, u, (size_t)6 * b->B * K, where, &d_u)) return rc;
    if (int rc = in_ptr(b, dt, (size_t)b->B * K, where, &d_dt)) return rc;
  }
  if (M > 0) {
    if (int rc = in_ptr(b, z, 2 * BM, where, &d_z)) return rc;
    if (int rc = in_ptr(b, slot, BM, where, &d_slot)) return rc;
    if (int rc = in_ptr(b, R, r_count(b, M, r_mode), where, &d_R)) return rc;
    if (result) d_res = where == VIEKF_DEVICE ? result : static_cast<int32_t*>(stage_take(b, sizeof(int32_t) * BM));
  }
  if (use_resident(b)) {
    if (with_propagate || M > 0)
      if (int rc = launch_resident(b, with_propagate, d_u, d_dt, d_z, d_slot, M, d_R, r_mode, d_res, nullptr, nullptr, K)) return rc;
  } else {
    if (with_propagate)
      for (int k = 0; k < K; k++)
        if (int rc = launch_propagate(b, d_u + (size_t)6 * b->B * k, d_dt + (size_t)b->B * k)) return rc;
    if (M > 0)
      if (int rc = launch_update(b, d_z, d_slot, M, d_R, r_mode, d_res)) return rc;
  }
  if (where == VIEKF_HOST) {
    if (result && M > 0) HIP_TRY(hipMemcpyAsync(result, d_res, sizeof(int32_t) * BM, hipMemcpyDeviceToHost, b->stream));
    if (!b->async_host || (result && M > 0)) HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

// diagnostic hook (not part of include/viekf.h): first `count` 8-byte words of the device workspace
int viekf_debug_read_ws(viekf_batch* b, void* out, int count) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out, b->d_ws, (size_t)count * 8, hipMemcpyDeviceToHost));
  return VIEKF_OK;
}

// diagnostic hook (not part of include/viekf.h; host arithmetic only, no device needed): the fused-step kernel's block
// ownership map for n_feat features on `nw` worker waves with `rb` slots per thread -> out[rb * 64 * nw] entries
// I | J << 8 | owned << 16; -1 when the blocks do not fit.  tests/test_resmap_cpu.py checks its invariants.
int viekf_debug_build_resmap(int n_feat, int rb, int nw, int32_t* out) {   // returns the number of slots in use (> 0), -1 on failure
  std::vector<int> map;
  int used = 0;
  if (n_feat < 1 || rb < 1 || nw < 1 || !out) return -1;
  if (n_feat * (n_feat + 1) / 2 > rb * nw * 64 || !build_resmap(n_feat, rb, nw, map, &used)) return -1;
  for (size_t i = 0; i < map.size(); i++) out[i] = map[i];
  return used;
}

int viekf_batch_keep_features(viekf_batch* b, const uint8_t* keep, int32_t* new_len, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (b->upper_stale) { HIP_TRY(hipSetDevice(b->device)); if (int rc = ensure_full_P(b)) return rc; }
  if (!keep) return fail(VIEKF_ERR_INVALID, "keep must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t BN = (size_t)b->B * b->N;
  const uint8_t* d_keep = nullptr;
  int32_t* d_nl = nullptr;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(BN) + stage_size(sizeof(int32_t) * b->B))) return rc;
  if (int rc = in_ptr(b, keep, BN, where, &d_keep)) return rc;
  if (new_len) d_nl = where == VIEKF_DEVICE ? new_len : static_cast<int32_t*>(stage_take(b, sizeof(int32_t) * b->B));
  StreamArgs a = make_args(b);
  const size_t lds = sizeof(double) * (size_t)b->n + sizeof(int) * (size_t)(b->n + 4);
  hipLaunchKernelGGL(k_keep_features<kThreads>, dim3(b->B), dim3(kThreads), lds, b->stream, a, d_keep, d_nl);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (new_len) HIP_TRY(hipMemcpyAsync(new_len, d_nl, sizeof(int32_t) * b->B, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_keyframe_reset(viekf_batch* b, const uint8_t* mask, double* edge, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (b->upper_stale) { HIP_TRY(hipSetDevice(b->device)); if (int rc = ensure_full_P(b)) return rc; }
  HIP_TRY(hipSetDevice(b->device));
  const uint8_t* d_mask = nullptr;
  double* d_edge = nullptr;
  const size_t eb = sizeof(double) * 17 * (size_t)b->B;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size((size_t)b->B) + stage_size(eb))) return rc;
  if (mask)
    if (int rc = in_ptr(b, mask, (size_t)b->B, where, &d_mask)) return rc;
  if (edge) d_edge = where == VIEKF_DEVICE ? edge : static_cast<double*>(stage_take(b, eb));
  if (edge && where == VIEKF_HOST) HIP_TRY(hipMemsetAsync(d_edge, 0, eb, b->stream));   // filters outside the mask report zeros
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_keyframe_reset<kThreads>, dim3(b->B), dim3(kThreads), 0, b->stream, a, d_mask, d_edge);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (edge) HIP_TRY(hipMemcpyAsync(edge, d_edge, eb, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

// ---- read-only evaluations for the log writer (see the kernels) ----
int viekf_batch_eval_xdot(viekf_batch* b, const double* u, double* xdot, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || !xdot) return fail(VIEKF_ERR_INVALID, "u / xdot is null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t ub = sizeof(double) * 6 * (size_t)b->B, ob = sizeof(double) * (size_t)b->B * b->n;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(ub) + stage_size(ob))) return rc;
  const double* d_u = nullptr;
  if (int rc = in_ptr(b, u, 6 * (size_t)b->B, where, &d_u)) return rc;
  double* d_o = where == VIEKF_DEVICE ? xdot : static_cast<double*>(stage_take(b, ob));
  StreamArgs a = make_args(b);
  const size_t lds = sizeof(double) * (size_t)(b->nxs + 256 + 96 + 16) + sizeof(BodyCtx) + 16;
  hipLaunchKernelGGL(k_eval_xdot<kThreads>, dim3(b->B), dim3(kThreads), lds, b->stream, a, d_u, d_o);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(xdot, d_o, ob, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_eval_h(viekf_batch* b, int32_t type, const int32_t* slot, double* zhat, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!zhat) return fail(VIEKF_ERR_INVALID, "zhat is null");
  if (type < 0 || type > 9 || type == 7) return fail(VIEKF_ERR_INVALID, "no measurement model for this type");
  const bool needs_slot = type == 5 || type == 6 || type == 8 || type == 9;
  if (needs_slot && !slot) return fail(VIEKF_ERR_INVALID, "this measurement model needs a feature slot");
  HIP_TRY(hipSetDevice(b->device));
  const size_t sb = sizeof(int32_t) * (size_t)b->B, ob = sizeof(double) * 4 * (size_t)b->B;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(sb) + stage_size(ob))) return rc;
  const int32_t* d_slot = nullptr;
  if (needs_slot)
    if (int rc = in_ptr(b, slot, (size_t)b->B, where, &d_slot)) return rc;
  double* d_o = where == VIEKF_DEVICE ? zhat : static_cast<double*>(stage_take(b, ob));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_eval_h, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, a, type, d_slot, d_o);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(zhat, d_o, ob, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

// ---- run-time drag switch (VIEKF::set_drag_term / get_drag_term, include/vi_ekf.h:290-291) ----
int viekf_batch_set_drag_term(viekf_batch* b, int32_t use_drag_term) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  b->params.use_drag_term = use_drag_term != 0;
  b->dp.use_drag_term = use_drag_term != 0;
  // (every kernel reads the flag from the device-resident parameter block at launch: ordered on the batch's stream)
  HIP_TRY(hipMemcpyAsync(b->d_dp, &b->dp, sizeof(DevParams), hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));   // (b->dp is pageable host memory: the copy has read it when this returns)
  return VIEKF_OK;
}
int viekf_batch_get_drag_term(const viekf_batch* b, int32_t* use_drag_term) {
  if (!b || !use_drag_term) return fail(VIEKF_ERR_INVALID, "null argument");
  *use_drag_term = b->params.use_drag_term;
  return VIEKF_OK;
}

// ---- the reference's public test hooks, evaluated on the device (viekf_kernels_hooks.hpp) ----
int viekf_batch_eval_jacobians(viekf_batch* b, const double* x, const double* u, double* xdot, double* A, double* G, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || (!xdot && !A && !G)) return fail(VIEKF_ERR_INVALID, "u and at least one output must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B, n = (size_t)b->n;
  const size_t xb = sizeof(double) * B * b->nx, ub = sizeof(double) * 6 * B, db = sizeof(double) * B * n, Ab = sizeof(double) * B * n * n,
               Gb = sizeof(double) * B * n * 6;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(xb) + stage_size(ub) + stage_size(db) + stage_size(Ab) + stage_size(Gb))) return rc;
  const double *d_x = nullptr, *d_u = nullptr;
  if (int rc = in_ptr(b, x, B * b->nx, where, &d_x)) return rc;
  if (int rc = in_ptr(b, u, 6 * B, where, &d_u)) return rc;
  auto outp = [&](double* p, size_t bytes) { return !p ? (double*)nullptr : (where == VIEKF_DEVICE ? p : static_cast<double*>(stage_take(b, bytes))); };
  double *d_xd = outp(xdot, db), *d_A = outp(A, Ab), *d_G = outp(G, Gb);
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_eval_jacobians, dim3(b->B), dim3(256), 0, b->stream, a, d_x, d_u, d_xd, d_A, d_G);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (xdot) HIP_TRY(hipMemcpyAsync(xdot, d_xd, db, hipMemcpyDeviceToHost, b->stream));
    if (A) HIP_TRY(hipMemcpyAsync(A, d_A, Ab, hipMemcpyDeviceToHost, b->stream));
    if (G) HIP_TRY(hipMemcpyAsync(G, d_G, Gb, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_eval_h_jacobian(viekf_batch* b, const double* x, int32_t type, const int32_t* slot, double* zhat, double* H, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!zhat || !H) return fail(VIEKF_ERR_INVALID, "zhat / H is null");
  if (type < 0 || type > 9 || type == 7) return fail(VIEKF_ERR_INVALID, "no measurement model for this type");
  const bool needs_slot = type == 5 || type == 6 || type == 8 || type == 9;
  if (needs_slot && !slot) return fail(VIEKF_ERR_INVALID, "this measurement model needs a feature slot");
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B;
  const size_t xb = sizeof(double) * B * b->nx, sb = sizeof(int32_t) * B, zb = sizeof(double) * 4 * B, Hb = sizeof(double) * 3 * B * b->n;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(xb) + stage_size(sb) + stage_size(zb) + stage_size(Hb))) return rc;
  const double* d_x = nullptr;
  const int32_t* d_slot = nullptr;
  if (int rc = in_ptr(b, x, B * b->nx, where, &d_x)) return rc;
  if (needs_slot)
    if (int rc = in_ptr(b, slot, B, where, &d_slot)) return rc;
  double* d_z = where == VIEKF_DEVICE ? zhat : static_cast<double*>(stage_take(b, zb));
  double* d_H = where == VIEKF_DEVICE ? H : static_cast<double*>(stage_take(b, Hb));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_eval_H, dim3((b->B + 63) / 64), dim3(64), 0, b->stream, a, d_x, type, d_slot, d_z, d_H);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(zhat, d_z, zb, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(H, d_H, Hb, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

static int boxop(viekf_batch* b, int minus, const double* x1, const double* v, double* out, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!x1 || !v || !out) return fail(VIEKF_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B, xb = sizeof(double) * B * b->nx, db = sizeof(double) * B * b->n;
  const size_t vb = minus ? xb : db, ob = minus ? db : xb;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(xb) + stage_size(vb) + stage_size(ob))) return rc;
  const double *d_x1 = nullptr, *d_v = nullptr;
  if (int rc = in_ptr(b, x1, xb / sizeof(double), where, &d_x1)) return rc;
  if (int rc = in_ptr(b, v, vb / sizeof(double), where, &d_v)) return rc;
  double* d_o = where == VIEKF_DEVICE ? out : static_cast<double*>(stage_take(b, ob));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_boxops, dim3(b->B), dim3(64), 0, b->stream, a, minus, d_x1, d_v, d_o);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(out, d_o, ob, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}
int viekf_batch_boxplus(viekf_batch* b, const double* x, const double* dx, double* out, viekf_mem where) { return boxop(b, 0, x, dx, out, where); }
int viekf_batch_boxminus(viekf_batch* b, const double* x1, const double* x2, double* out, viekf_mem where) { return boxop(b, 1, x1, x2, out, where); }

int viekf_batch_eval_reset_jacobian(viekf_batch* b, const double* xm, double* xp, double* N, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!xm || (!xp && !N)) return fail(VIEKF_ERR_INVALID, "xm and at least one output must not be null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B, xb = sizeof(double) * B * b->nx, Nb = sizeof(double) * B * b->n * b->n;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, 2 * stage_size(xb) + stage_size(Nb))) return rc;
  const double* d_xm = nullptr;
  if (int rc = in_ptr(b, xm, xb / sizeof(double), where, &d_xm)) return rc;
  double* d_xp = !xp ? nullptr : (where == VIEKF_DEVICE ? xp : static_cast<double*>(stage_take(b, xb)));
  double* d_N = !N ? nullptr : (where == VIEKF_DEVICE ? N : static_cast<double*>(stage_take(b, Nb)));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_eval_reset, dim3(b->B), dim3(256), 0, b->stream, a, d_xm, d_xp, d_N);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (xp) HIP_TRY(hipMemcpyAsync(xp, d_xp, xb, hipMemcpyDeviceToHost, b->stream));
    if (N) HIP_TRY(hipMemcpyAsync(N, d_N, Nb, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_get_cov_diag(viekf_batch* b, double* diag, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!diag) return fail(VIEKF_ERR_INVALID, "diag is null");
  HIP_TRY(hipSetDevice(b->device));
  const size_t ob = sizeof(double) * (size_t)b->B * b->n;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(ob))) return rc;
  double* d_o = where == VIEKF_DEVICE ? diag : static_cast<double*>(stage_take(b, ob));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_cov_diag, dim3((b->n + 63) / 64, b->B), dim3(64), 0, b->stream, a, d_o);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(diag, d_o, ob, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_get_cov_block(viekf_batch* b, int32_t row0, int32_t col0, int32_t nrows, int32_t ncols, double* out,
                              viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!out) return fail(VIEKF_ERR_INVALID, "out is null");
  if (b->upper_stale) { HIP_TRY(hipSetDevice(b->device)); if (int rc = ensure_full_P(b)) return rc; }
  if (row0 < 0 || col0 < 0 || nrows < 1 || ncols < 1 || row0 + nrows > b->n || col0 + ncols > b->n)
    return fail(VIEKF_ERR_INVALID, "block outside the covariance");
  HIP_TRY(hipSetDevice(b->device));
  const size_t ob = sizeof(double) * (size_t)b->B * nrows * ncols;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(ob))) return rc;
  double* d_o = where == VIEKF_DEVICE ? out : static_cast<double*>(stage_take(b, ob));
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_cov_block, dim3((nrows * ncols + 63) / 64, b->B), dim3(64), 0, b->stream, a, row0, col0, nrows, ncols, d_o);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    HIP_TRY(hipMemcpyAsync(out, d_o, ob, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

static size_t hist_nx(const viekf_batch* b) { return sizeof(double) * (size_t)b->B * b->nxs; }
static size_t hist_nP(const viekf_batch* b) { return sizeof(double) * (size_t)b->B * b->n * b->ld; }
static double* slot_x(const viekf_batch* b, int slot) { return reinterpret_cast<double*>(reinterpret_cast<char*>(b->h_x) + hist_nx(b) * slot); }
static double* slot_P(const viekf_batch* b, int slot) { return reinterpret_cast<double*>(reinterpret_cast<char*>(b->h_P) + hist_nP(b) * slot); }

int viekf_batch_history_resize(viekf_batch* b, int32_t depth) {
  if (int rc = check_batch(b)) return rc;
  if (depth < 0 || depth > 4096) return fail(VIEKF_ERR_INVALID, "0 <= depth <= 4096");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (b->per_filter) {       // every filter's live slot goes home
    if (int rc = gather_scatter_home(b, 1)) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->d_x = b->home_x; b->d_P = b->home_P; b->per_filter = false;
  }
  if (b->live_slot >= 0) {   // the live state lives in the ring: bring it home first
    HIP_TRY(hipMemcpy(b->home_x, b->d_x, hist_nx(b), hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(b->home_P, b->d_P, hist_nP(b), hipMemcpyDeviceToDevice));
    b->d_x = b->home_x; b->d_P = b->home_P; b->live_slot = -1;
  }
  if (b->h_x) { HIP_TRY(hipFree(b->h_x)); b->h_x = nullptr; }
  if (b->h_P) { HIP_TRY(hipFree(b->h_P)); b->h_P = nullptr; }
  if (b->h_len) { HIP_TRY(hipFree(b->h_len)); b->h_len = nullptr; }
  b->hist_depth = 0;
  if (depth == 0) return VIEKF_OK;
  HIP_TRY(hipMalloc(&b->h_x, sizeof(double) * (size_t)depth * b->B * b->nxs));
  HIP_TRY(hipMalloc(&b->h_P, sizeof(double) * (size_t)depth * b->B * b->n * b->ld));
  HIP_TRY(hipMalloc(&b->h_len, sizeof(int) * (size_t)depth * b->B));
  b->hist_depth = depth;
  return VIEKF_OK;
}

static int history_copy(viekf_batch* b, int32_t slot, bool save) {
  if (int rc = check_batch(b)) return rc;
  // (a covariance is copied as it stands, stale upper triangle included; what comes back from the ring is taken to be as stale
  //  as anything this batch ever produced)
  if (slot < 0 || slot >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "snapshot slot out of range (viekf_batch_history_resize first)");
  if (b->per_filter) return fail(VIEKF_ERR_INVALID, "whole-batch ring copies under per-filter live slots (viekf_batch_select_filters)");
  if (!save) b->upper_stale = b->stale_ever > b->upper_stale ? b->stale_ever : b->upper_stale;
  HIP_TRY(hipSetDevice(b->device));
  const size_t nl = sizeof(int) * (size_t)b->B;
  char* hl = reinterpret_cast<char*>(b->h_len) + nl * slot;
  const bool same = slot == b->live_slot;   // the live state already IS this slot: only the feature counts move
  if (save) {
    if (!same) {
      HIP_TRY(hipMemcpyAsync(slot_x(b, slot), b->d_x, hist_nx(b), hipMemcpyDeviceToDevice, b->stream));
      HIP_TRY(hipMemcpyAsync(slot_P(b, slot), b->d_P, hist_nP(b), hipMemcpyDeviceToDevice, b->stream));
    }
    HIP_TRY(hipMemcpyAsync(hl, b->d_len, nl, hipMemcpyDeviceToDevice, b->stream));
  } else {
    if (!same) {
      HIP_TRY(hipMemcpyAsync(b->d_x, slot_x(b, slot), hist_nx(b), hipMemcpyDeviceToDevice, b->stream));
      HIP_TRY(hipMemcpyAsync(b->d_P, slot_P(b, slot), hist_nP(b), hipMemcpyDeviceToDevice, b->stream));
    }
    HIP_TRY(hipMemcpyAsync(b->d_len, hl, nl, hipMemcpyDeviceToDevice, b->stream));
  }
  return VIEKF_OK;
}
int viekf_batch_snapshot(viekf_batch* b, int32_t slot) { return history_copy(b, slot, true); }
int viekf_batch_restore(viekf_batch* b, int32_t slot) { return history_copy(b, slot, false); }

int viekf_batch_set_active(viekf_batch* b, const uint8_t* mask, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  HIP_TRY(hipSetDevice(b->device));
  if (!mask) { b->active_on = false; return VIEKF_OK; }
  if (!b->d_active) HIP_TRY(hipMalloc(&b->d_active, (size_t)b->B));
  const void* from = mask;
  if (where == VIEKF_HOST && b->async_host) {   // through the pinned ring: nothing to wait for (the mask itself stays a DEVICE copy:
    if (int rc = stage_begin(b, stage_size((size_t)b->B))) return rc;   // it outlives any number of launches)
    const size_t off = (b->pin_used + 255) & ~size_t(255);
    std::memcpy(b->h_pin + off, mask, (size_t)b->B);
    b->pin_used = off + (size_t)b->B;
    from = b->h_pin + off;
  }
  HIP_TRY(hipMemcpyAsync(b->d_active, from, (size_t)b->B, where == VIEKF_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                         b->stream));
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));   // (the caller's buffer may go away)
  b->active_on = true;
  return VIEKF_OK;
}

static int ring_filters(viekf_batch* b, const int32_t* slot, viekf_mem where, int to_ring) {
  if (int rc = check_batch(b)) return rc;
  if (!slot) return fail(VIEKF_ERR_INVALID, "slot is null");
  if (b->hist_depth <= 0) return fail(VIEKF_ERR_INVALID, "no history ring (viekf_batch_history_resize first)");
  if (b->live_slot >= 0) return fail(VIEKF_ERR_INVALID, "per-filter ring copies need the live state in the batch's own buffers (viekf_batch_select(-1))");
  HIP_TRY(hipSetDevice(b->device));
  // (host slots are validated here; device slots by the kernel, which skips an out-of-range one and raises VIEKF_FLAG_INTERNAL)
  if (where == VIEKF_HOST)
    for (int i = 0; i < b->B; i++)
      if (slot[i] >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range");
  if (!to_ring) b->upper_stale = b->stale_ever > b->upper_stale ? b->stale_ever : b->upper_stale;   // (see history_copy)
  const int* d_slot = slot;
  if (where == VIEKF_HOST && b->async_host) {   // (read by this one launch: straight from the pinned ring, nothing to wait for)
    if (int rc = stage_begin(b, stage_size(sizeof(int) * (size_t)b->B))) return rc;
    if (int rc = in_ptr(b, slot, (size_t)b->B, where, &d_slot)) return rc;
  } else if (where == VIEKF_HOST) {
    if (!b->d_ringslot) HIP_TRY(hipMalloc(&b->d_ringslot, sizeof(int) * (size_t)b->B));
    HIP_TRY(hipMemcpyAsync(b->d_ringslot, slot, sizeof(int) * (size_t)b->B, hipMemcpyHostToDevice, b->stream));
    d_slot = b->d_ringslot;
  }
  StreamArgs a = make_args(b);
  hipLaunchKernelGGL(k_ring_copy, dim3(b->B), dim3(256), 0, b->stream, a, b->h_x, b->h_P, d_slot, to_ring, b->hist_depth);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}
int viekf_batch_snapshot_filters(viekf_batch* b, const int32_t* slot, viekf_mem where) { return ring_filters(b, slot, where, 1); }
int viekf_batch_restore_filters(viekf_batch* b, const int32_t* slot, viekf_mem where) { return ring_filters(b, slot, where, 0); }

int viekf_batch_select(viekf_batch* b, int32_t slot) {
  if (int rc = check_batch(b)) return rc;
  if (slot < -1 || slot >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range (viekf_batch_history_resize first)");
  if (b->per_filter) return fail(VIEKF_ERR_INVALID, "viekf_batch_select under per-filter live slots (viekf_batch_select_filters)");
  b->upper_stale = b->stale_ever > b->upper_stale ? b->stale_ever : b->upper_stale;   // (see history_copy)
  if (!b->home_x) { b->home_x = b->d_x; b->home_P = b->d_P; }
  b->live_slot = slot;
  b->d_x = slot < 0 ? b->home_x : slot_x(b, slot);
  b->d_P = slot < 0 ? b->home_P : slot_P(b, slot);
  return VIEKF_OK;
}

int viekf_batch_propagate_to(viekf_batch* b, const double* u, const double* dt, int32_t dst_slot, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || !dt) return fail(VIEKF_ERR_INVALID, "u and dt must not be null");
  if (dst_slot < 0 || dst_slot >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range (viekf_batch_history_resize first)");
  if (b->per_filter) return fail(VIEKF_ERR_INVALID, "viekf_batch_propagate_to under per-filter live slots (viekf_batch_propagate_filters_to)");
  if (dst_slot == b->live_slot) return viekf_batch_propagate(b, u, dt, where);
  // (a filter outside a participation mask would have nothing written into the destination slot, which then becomes the live
  //  state: the zero-copy ring is for filters that advance together)
  if (b->active_on) return fail(VIEKF_ERR_INVALID, "viekf_batch_propagate_to under a participation mask (viekf_batch_set_active(NULL) first)");
  HIP_TRY(hipSetDevice(b->device));
  const double *d_u = nullptr, *d_dt = nullptr;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(sizeof(double) * 6 * b->B) + stage_size(sizeof(double) * b->B))) return rc;
  if (int rc = in_ptr(b, u, (size_t)6 * b->B, where, &d_u)) return rc;
  if (int rc = in_ptr(b, dt, (size_t)b->B, where, &d_dt)) return rc;
  if (use_resident(b)) {   // the fused kernel loads P from the live slot and stores it into the destination: no copy at all
    if (int rc = launch_resident(b, true, d_u, d_dt, nullptr, nullptr, 0, nullptr, 0, nullptr, slot_x(b, dst_slot), slot_P(b, dst_slot)))
      return rc;
    if (int rc = viekf_batch_select(b, dst_slot)) return rc;
  } else {                 // streaming family works in place: copy, then propagate the copy
    HIP_TRY(hipMemcpyAsync(slot_x(b, dst_slot), b->d_x, hist_nx(b), hipMemcpyDeviceToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(slot_P(b, dst_slot), b->d_P, hist_nP(b), hipMemcpyDeviceToDevice, b->stream));
    if (int rc = viekf_batch_select(b, dst_slot)) return rc;
    if (int rc = launch_propagate(b, d_u, d_dt)) return rc;
  }
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_propagate_n_to(viekf_batch* b, int32_t K, const double* u, const double* dt, const int32_t* dst_slots,
                               int32_t* intermediates_written, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || !dt || !dst_slots) return fail(VIEKF_ERR_INVALID, "u, dt and dst_slots must not be null");
  if (K < 1 || K > 64) return fail(VIEKF_ERR_INVALID, "1 <= K <= 64 propagates per call");
  if (b->per_filter) return fail(VIEKF_ERR_INVALID, "viekf_batch_propagate_n_to under per-filter live slots");
  for (int k = 0; k < K; k++) {
    if (dst_slots[k] < 0 || dst_slots[k] >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range (viekf_batch_history_resize first)");
    if (dst_slots[k] == b->live_slot) return fail(VIEKF_ERR_INVALID, "a destination slot is the live slot");
    for (int j = 0; j < k; j++)
      if (dst_slots[j] == dst_slots[k]) return fail(VIEKF_ERR_INVALID, "destination slots must differ");
  }
  if (intermediates_written) *intermediates_written = 1;
  if (K == 1 || !use_resident(b) || b->active_on) {   // one slot at a time, every slot written
    for (int k = 0; k < K; k++)
      if (int rc = viekf_batch_propagate_to(b, u + (size_t)6 * b->B * k, dt + (size_t)b->B * k, dst_slots[k], where)) return rc;
    return VIEKF_OK;
  }
  HIP_TRY(hipSetDevice(b->device));
  const double *d_u = nullptr, *d_dt = nullptr;
  if (where == VIEKF_HOST)
    if (int rc = stage_begin(b, stage_size(sizeof(double) * 6 * b->B * K) + stage_size(sizeof(double) * b->B * K))) return rc;
  if (int rc = in_ptr(b, u, (size_t)6 * b->B * K, where, &d_u)) return rc;
  if (int rc = in_ptr(b, dt, (size_t)b->B * K, where, &d_dt)) return rc;
  const int last = dst_slots[K - 1];
  // ONE launch of the fused kernel: P is loaded from the live slot, stays on chip through the K propagates and is stored into the
  // last slot only
  if (int rc = launch_resident(b, true, d_u, d_dt, nullptr, nullptr, 0, nullptr, 0, nullptr, slot_x(b, last), slot_P(b, last), K)) return rc;
  if (int rc = viekf_batch_select(b, last)) return rc;
  if (intermediates_written) *intermediates_written = 0;
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_select_filters(viekf_batch* b, const int32_t* slot) {
  if (int rc = check_batch(b)) return rc;
  if (!slot) return fail(VIEKF_ERR_INVALID, "slot is null");
  if (b->hist_depth <= 0) return fail(VIEKF_ERR_INVALID, "no history ring (viekf_batch_history_resize first)");
  if (b->live_slot >= 0) return fail(VIEKF_ERR_INVALID, "per-filter live slots need the live state in the batch's own buffers (viekf_batch_select(-1))");
  for (int i = 0; i < b->B; i++) {
    if (slot[i] >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range");
    if (slot[i] < 0 && !b->per_filter) return fail(VIEKF_ERR_INVALID, "the first call has to name a slot for every filter");
  }
  HIP_TRY(hipSetDevice(b->device));
  if (!b->per_filter) {
    if (!b->d_smap) HIP_TRY(hipMalloc(&b->d_smap, sizeof(int) * (size_t)b->B));
    if (!b->home_x) { b->home_x = b->d_x; b->home_P = b->d_P; }
    b->live_slots.assign((size_t)b->B, 0);
    b->per_filter = true;
    b->d_x = b->h_x; b->d_P = b->h_P;
  }
  for (int i = 0; i < b->B; i++)
    if (slot[i] >= 0) b->live_slots[(size_t)i] = slot[i];
  // the device's copy of the map follows in stream order (a launch of one small kernel, not a copy command between two kernels)
  if (int rc = stage_begin(b, stage_size(sizeof(int32_t) * (size_t)b->B))) return rc;
  const int32_t* d_slot = nullptr;
  if (int rc = in_ptr(b, slot, (size_t)b->B, VIEKF_HOST, &d_slot)) return rc;
  hipLaunchKernelGGL(k_set_smap, dim3((unsigned)((b->B + 255) / 256)), dim3(256), 0, b->stream, b->d_smap, d_slot, b->B);
  HIP_TRY(hipGetLastError());
  if (!b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  b->upper_stale = b->stale_ever > b->upper_stale ? b->stale_ever : b->upper_stale;   // (see history_copy)
  return VIEKF_OK;
}

int viekf_batch_propagate_filters_to(viekf_batch* b, const double* u, const double* dt, const int32_t* dst_slot, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (!u || !dt || !dst_slot) return fail(VIEKF_ERR_INVALID, "u, dt and dst_slot must not be null");
  if (!b->per_filter) return fail(VIEKF_ERR_INVALID, "viekf_batch_select_filters first");
  bool any = false;
  for (int i = 0; i < b->B; i++) {
    if (dst_slot[i] >= b->hist_depth) return fail(VIEKF_ERR_INVALID, "ring slot out of range");
    if (dst_slot[i] >= 0 && dst_slot[i] == b->live_slots[(size_t)i]) return fail(VIEKF_ERR_INVALID, "a destination slot is the filter's live slot");
    any |= dst_slot[i] >= 0;
  }
  if (!any) return VIEKF_OK;
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B;
  // the participation mask and the destination map of THIS launch, through the pinned staging like the other per-call arguments
  std::vector<unsigned char> act(B);
  std::vector<int32_t> omap(B);
  for (size_t i = 0; i < B; i++) {
    act[i] = dst_slot[i] >= 0 ? 1 : 0;
    omap[i] = (dst_slot[i] >= 0 ? dst_slot[i] : b->live_slots[i]) * b->B + (int32_t)i;
  }
  if (int rc = stage_begin(b, stage_size(sizeof(double) * 6 * B) + stage_size(sizeof(double) * B) + stage_size(B) + stage_size(sizeof(int32_t) * B) * 2))
    return rc;
  const double *d_u = nullptr, *d_dt = nullptr;
  const unsigned char* d_act = nullptr;
  const int32_t* d_omap = nullptr;
  if (int rc = in_ptr(b, u, 6 * B, where, &d_u)) return rc;
  if (int rc = in_ptr(b, dt, B, where, &d_dt)) return rc;
  if (int rc = in_ptr(b, act.data(), B, VIEKF_HOST, &d_act)) return rc;
  if (int rc = in_ptr(b, omap.data(), B, VIEKF_HOST, &d_omap)) return rc;
  const bool saved_on = b->active_on;
  unsigned char* saved_mask = b->d_active;
  b->active_on = true; b->d_active = const_cast<unsigned char*>(d_act);
  int rc = VIEKF_OK;
  if (use_resident(b)) {     // the fused kernel loads filter b from its live slot and stores it into dst_slot[b]: no copy at all
    rc = launch_resident(b, true, d_u, d_dt, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 1, d_omap);
    for (size_t i = 0; i < B && rc == VIEKF_OK; i++)     // (the kernel moves the device's map entries itself)
      if (dst_slot[i] >= 0) b->live_slots[i] = dst_slot[i];
  } else {                   // the HBM-path family works in place: copy slot -> slot, then propagate the copy
    const int32_t* d_dst = nullptr;
    rc = in_ptr(b, dst_slot, B, VIEKF_HOST, &d_dst);
    if (rc == VIEKF_OK) {
      StreamArgs a = make_args(b);
      hipLaunchKernelGGL(k_ring_copy, dim3(b->B), dim3(256), 0, b->stream, a, b->h_x, b->h_P, d_dst, 1, b->hist_depth);
      if (hipGetLastError() != hipSuccess) rc = fail(VIEKF_ERR_HIP, "k_ring_copy launch failed");
    }
    if (rc == VIEKF_OK) {
      for (size_t i = 0; i < B; i++)
        if (dst_slot[i] >= 0) b->live_slots[i] = dst_slot[i];
      hipLaunchKernelGGL(k_set_smap, dim3((unsigned)((b->B + 255) / 256)), dim3(256), 0, b->stream, b->d_smap, d_dst, b->B);
      if (hipGetLastError() != hipSuccess) rc = fail(VIEKF_ERR_HIP, "k_set_smap launch failed");
    }
    if (rc == VIEKF_OK) rc = launch_propagate(b, d_u, d_dt);
  }
  b->active_on = saved_on; b->d_active = saved_mask;
  if (rc) return rc;
  if (where == VIEKF_HOST && !b->async_host) HIP_TRY(hipStreamSynchronize(b->stream));
  return VIEKF_OK;
}

int viekf_batch_update(viekf_batch* b, int32_t type, const double* z, int32_t zdim, const double* R, int32_t rdim,
                       int32_t r_mode, const int32_t* slot, const uint8_t* active, int32_t* result, viekf_mem where) {
  if (int rc = check_batch(b)) return rc;
  if (b->upper_stale) { HIP_TRY(hipSetDevice(b->device)); if (int rc = ensure_full_P(b)) return rc; }
  if (!z || !R) return fail(VIEKF_ERR_INVALID, "z and R must not be null");
  if (type < 0 || type >= VIEKF_TOTAL_MEAS || type == VIEKF_PIXEL_VEL)
    return fail(VIEKF_ERR_UNSUPPORTED, "measurement type not supported (PIXEL_VEL is an empty TODO in the reference)");
  if (zdim < 1 || zdim > 4 || rdim < 1 || rdim > 3 || r_mode < 0 || r_mode > 1)
    return fail(VIEKF_ERR_INVALID, "need 1 <= zdim <= 4, 1 <= rdim <= 3, r_mode 0 or 1");
  const bool needs_slot = type == VIEKF_QZETA || type == VIEKF_FEAT || type == VIEKF_DEPTH || type == VIEKF_INV_DEPTH;
  if (needs_slot && !slot) return fail(VIEKF_ERR_INVALID, "slot must not be null for feature measurements");
  HIP_TRY(hipSetDevice(b->device));
  const size_t B = (size_t)b->B, rr = (size_t)rdim * rdim;
  const double *d_z = nullptr, *d_R = nullptr;
  const int32_t* d_slot = nullptr;
  const uint8_t* d_act = nullptr;
  int32_t* d_res = nullptr;
  if (where == VIEKF_HOST) {
    if (int rc = stage_begin(b, stage_size(sizeof(double) * B * zdim) + stage_size(sizeof(double) * B * rr) +
                                    stage_size(sizeof(int32_t) * B) * 2 + stage_size(B)))
      return rc;
  }
  if (int rc = in_ptr(b, z, B * zdim, where, &d_z)) return rc;
  if (int rc = in_ptr(b, R, r_mode ? B * rr : rr, where, &d_R)) return rc;
  if (int rc = in_ptr(b, slot, B, where, &d_slot)) return rc;
  if (int rc = in_ptr(b, active, B, where, &d_act)) return rc;
  if (result) d_res = where == VIEKF_DEVICE ? result : static_cast<int32_t*>(stage_take(b, sizeof(int32_t) * B));
  StreamArgs a = make_args(b);
  const size_t lds = sizeof(double) * (size_t)(b->nxs + 7 * b->n + 64);
  hipLaunchKernelGGL(k_update_generic<kThreads>, dim3(b->B), dim3(kThreads), lds, b->stream, a, type, zdim, rdim, d_z,
                     d_slot, d_R, r_mode ? (long)rr : 0L, d_act, d_res);
  HIP_TRY(hipGetLastError());
  if (where == VIEKF_HOST) {
    if (result) HIP_TRY(hipMemcpyAsync(result, d_res, sizeof(int32_t) * B, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return VIEKF_OK;
}

int viekf_batch_update_feat(viekf_batch* b, const double* z, const int32_t* slot, int32_t M, const double* R,
                            int32_t r_mode, int32_t* result, viekf_mem where) {
  return update_or_step(b, nullptr, nullptr, false, z, slot, M, R, r_mode, result, where);
}

int viekf_batch_step(viekf_batch* b, const double* u, const double* dt, const double* z, const int32_t* slot,
                     int32_t M, const double* R, int32_t r_mode, int32_t* result, viekf_mem where) {
  return update_or_step(b, u, dt, true, z, slot, M, R, r_mode, result, where);
}

int viekf_batch_step_n(viekf_batch* b, int32_t K, const double* u, const double* dt, const double* z, const int32_t* slot,
                       int32_t M, const double* R, int32_t r_mode, int32_t* result, viekf_mem where) {
  return update_or_step(b, u, dt, true, z, slot, M, R, r_mode, result, where, K);
}

}  // extern "C"
