// Microbenchmark: what ONE workgroup of 512 threads (one per CU: 120 KB of LDS claimed) moves per second when it reads, updates and
// writes back a private 1.7 MB matrix the way the wide-P pass does -- units of four 16 x 16 fp64 tiles, lane = (row, column group),
// 16 global_load_dwordx2 + 16 global_store_dwordx2 per unit and wave -- against the same bytes in other shapes (loads of the next unit
// issued before the current one is finished; fully contiguous dwordx2; contiguous dwordx4), with 64, 256 and 1024 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream_rate cu_stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#ifndef LDV
#define LDV 466
#endif
#ifndef TPV
#define TPV 4
#endif
constexpr int TP = TPV;                                   // tiles per unit (-DTPV=8: 1 KB contiguous per column)
constexpr int N = 466, LD = LDV, NT = (N + 15) / 16;      // the N_feat = 150 covariance (-DLDV=480: columns padded to whole 128-B lines)

// MODE 0: tile units, load all -> update -> store all.  MODE 1: the same, software-pipelined over two register sets.
// MODE 2: contiguous dwordx2 (a unit = 16 x 512 B).      MODE 3: contiguous dwordx4 (a unit = 8 x 1 KB).
template <int MODE>
__global__ __launch_bounds__(512) void k_rmw(double* base, int passes, int* tickets) {
  extern __shared__ double smem[];
  double* P = base + (long)blockIdx.x * N * LD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  int* ticket = reinterpret_cast<int*>(smem);
  if (threadIdx.x == 0) *ticket = 0;
  __syncthreads();
  constexpr int NU = (NT * (NT - 1) / 2 / 4 + NT / 4) * 4 / TP;   // units of TP tiles in the lower triangle (about)
  for (int pass = 0; pass < passes; pass++) {
    auto draw = [&]() {
      int t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1);
      return __builtin_amdgcn_readfirstlane(t) - pass * (NU + 8);
    };
    if (MODE == 0 || MODE == 1) {
      auto addr = [&](int u, int q, int rg) {
        // unit u -> column block tj, tile rows ti0 .. ti0 + 3 (disjoint units; the triangle's shape does not matter here)
        const int tj = u % NT, ti0 = TP * (u / NT);
        const int i = min(16 * (ti0 + q) + lr, N - 1), j = min(16 * tj + lk + 4 * rg, N - 1);
        return P + i + (long)j * LD;
      };
      if (MODE == 0) {
        for (int u = draw(); u < NU; u = draw()) {
          double pv[TP][4];
#pragma unroll
          for (int q = 0; q < TP; q++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) pv[q][rg] = *addr(u, q, rg);
#pragma unroll
          for (int q = 0; q < TP; q++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) *addr(u, q, rg) = pv[q][rg] * 1.0000001 + 1.0;
        }
      } else {
        int u = draw();
        if (u < NU) {
          double pa[TP][4], pb[TP][4];
          int ua = u, ub;
#pragma unroll
          for (int q = 0; q < TP; q++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) pa[q][rg] = *addr(ua, q, rg);
          for (;;) {
            u = draw(); ub = min(u, NU - 1);
#pragma unroll
            for (int q = 0; q < TP; q++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++) pb[q][rg] = *addr(ub, q, rg);
#pragma unroll
            for (int q = 0; q < TP; q++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++) *addr(ua, q, rg) = pa[q][rg] * 1.0000001 + 1.0;
            if (u >= NU) break;
            u = draw(); ua = min(u, NU - 1);
#pragma unroll
            for (int q = 0; q < TP; q++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++) pa[q][rg] = *addr(ua, q, rg);
#pragma unroll
            for (int q = 0; q < TP; q++)
#pragma unroll
              for (int rg = 0; rg < 4; rg++) *addr(ub, q, rg) = pb[q][rg] * 1.0000001 + 1.0;
            if (u >= NU) break;
          }
        }
      }
    } else if (MODE == 2) {
      for (int u = draw(); u < NU; u = draw()) {
        double* p = P + (long)u * 1024 + lane;   // (8 KB units)
        double pv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) pv[k] = p[64 * k];
#pragma unroll
        for (int k = 0; k < 16; k++) p[64 * k] = pv[k] * 1.0000001 + 1.0;
      }
    } else {
      for (int u = draw(); u < NU; u = draw()) {
        double2* p = reinterpret_cast<double2*>(P + (long)u * 1024) + lane;
        double2 pv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) pv[k] = p[64 * k];
#pragma unroll
        for (int k = 0; k < 8; k++) p[64 * k] = make_double2(pv[k].x * 1.0000001 + 1.0, pv[k].y * 1.0000001 + 1.0);
      }
    }
    __syncthreads();
  }
  (void)wave;
}

template <int MODE>
static void run(double* buf, int blocks, int passes, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  constexpr int NU = (NT * (NT - 1) / 2 / 4 + NT / 4) * 4 / TP;
  const size_t lds = 120 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_rmw<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_rmw<MODE><<<blocks, 512, lds>>>(buf, 2, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_rmw<MODE><<<blocks, 512, lds>>>(buf, passes, nullptr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 2.0 * 2048.0 * TP * NU * passes * blocks;   // read + write, 2 KB per tile
  const int rounds = (blocks + 255) / 256, cus = blocks < 256 ? blocks : 256;
  printf("%-44s %4d workgroups: %7.3f ms  %7.1f GB/s  = %5.1f GB/s per active CU (%d round(s))\n", what, blocks, ms, bytes / ms * 1e-6,
         bytes / ms * 1e-6 / cus, rounds);
}

int main() {
  double* buf;
  const size_t per = (size_t)N * LD * sizeof(double);
  hipMalloc(&buf, per * 1024 + (1 << 20));
  hipMemset(buf, 0, per * 1024 + (1 << 20));
  for (int blocks : {64, 256, 1024}) {
    run<0>(buf, blocks, 10, "tile units, load -> update -> store");
    run<1>(buf, blocks, 10, "tile units, next unit's loads in flight");
    run<2>(buf, blocks, 10, "contiguous dwordx2");
    run<3>(buf, blocks, 10, "contiguous dwordx4");
  }
  return 0;
}
