// Microbenchmark: cycles per v_mfma_f64_16x16x4_f64 on one SIMD (independent accumulators, back-to-back issue), with one and two
// waves per SIMD, and per v_fma_f64 for comparison.   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double* out, long long* clk, int iters) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 2.0 - threadIdx.x * 1e-3;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  double s = 0.0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_fma(double* out, long long* clk, int iters) {
  double acc[16];
  for (int i = 0; i < 16; i++) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = fma(acc[i], a, b);
  }
  double s = 0.0;
  for (int i = 0; i < 16; i++) s += acc[i];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
  double* out; long long* clk;
  hipMalloc(&out, sizeof(double) * 1024 * 512);
  hipMalloc(&clk, sizeof(long long) * 1024 * 8);
  const int iters = 2000;
  for (int threads : {64, 256, 512}) {          // 1 wave (one SIMD), 4 waves (one per SIMD), 8 waves (two per SIMD)
    for (int blocks : {1, 256}) {
      hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
      hipDeviceSynchronize();
      std::vector<long long> h(blocks * threads / 64);
      hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
      long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("mfma_f64_16x16x4: %3d threads x %3d blocks: %.1f clk per MFMA per wave (8 independent accumulators)\n", threads, blocks,
             (double)mx / (iters * 8.0));
      hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
      mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("v_fma_f64        : %3d threads x %3d blocks: %.1f clk per FMA per wave (16 independent chains)\n", threads, blocks,
             (double)mx / (iters * 16.0));
    }
  }
  return 0;
}
