// Microbenchmark: cost of ds_write_b128 against the number of ACTIVE lanes (does a store with 8 of 64 lanes enabled still pay the
// whole instruction's VGPR -> LDS transfer?).   hipcc --offload-arch=gfx950 -O3 -o lds_store_mask lds_store_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_store(double* out, long long* clk, int iters, int active_mod) {
  __shared__ __attribute__((aligned(16))) double buf[64 * 2 * 16 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double2 v = make_double2(1.0 + lane, 2.0 + lane);
  double* base = buf + wave * (64 * 2 * 16) + lane * 2;
  const bool on = (lane % active_mod) == 0;     // active_mod 1: all 64 lanes, 8: every 8th (8 lanes)
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (on) {
#pragma unroll
      for (int k = 0; k < 16; k++) { *reinterpret_cast<double2*>(base + k * 128) = v; asm volatile("" ::: "memory"); }
    }
    v.x += 1.0;
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x] + v.x;
  if (lane == 0) clk[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

int main() {
  double* out; long long* clk;
  if (hipMalloc(&out, sizeof(double) * 512 * 256) != hipSuccess || hipMalloc(&clk, sizeof(long long) * 8 * 256) != hipSuccess) return 1;
  const int iters = 2000;
  for (int threads : {64, 256, 512}) {
    for (int mod : {1, 2, 8, 64}) {
      hipLaunchKernelGGL(k_store, dim3(256), dim3(threads), 0, 0, out, clk, iters, mod);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      std::vector<long long> h(256 * threads / 64);
      if (hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
      double av = 0; for (auto v : h) av += v; av /= h.size();
      printf("ds_write_b128  %3d threads per CU, %2d active lanes per wave: %6.2f clk per store instruction per wave, %6.2f per CU\n", threads, 64 / mod,
             av / (iters * 16.0), av / (iters * 16.0) / (threads / 64));
    }
  }
  return 0;
}
