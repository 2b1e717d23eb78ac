// Microbenchmark: sustained v_fma_f64 issue rate of one CU against the number of resident waves, beside v_fma_f32 and v_pk_fma_f32
// (is the fp64 vector pipe shared between the SIMDs of a CU?).   hipcc --offload-arch=gfx950 -O3 -o valu_f64_rate valu_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename T>
__global__ void k_fma(T* out, long long* clk, int iters, T a, T b) {
  T acc[16];
  for (int i = 0; i < 16; i++) acc[i] = (T)(i + threadIdx.x);
  a += (T)threadIdx.x * (T)1e-9;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = acc[i] * a + b;
  }
  T s = acc[0];
  for (int i = 1; i < 16; i++) s += acc[i];
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename T>
void run(const char* name, T* out, long long* clk) {
  const int iters = 4000;
  for (int threads : {64, 128, 192, 256, 512, 1024}) {
    for (int blocks : {1, 256, 512, 1024}) {
      if ((long long)threads * blocks > 512 * 1024) continue;
      hipLaunchKernelGGL(k_fma<T>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, (T)1.0, (T)1e-9);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
      std::vector<long long> h(blocks * threads / 64);
      if (hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return;
      long long mx = 0; double av = 0; for (auto v : h) { mx = v > mx ? v : mx; av += v; }
      av /= h.size();
      printf("%-12s %4d threads x %4d blocks: %6.2f clk per instruction per wave (max %6.2f)\n", name, threads, blocks,
             av / (iters * 16.0), (double)mx / (iters * 16.0));
    }
  }
}

int main() {
  void* out; long long* clk;
  if (hipMalloc(&out, sizeof(double) * 1024 * 512) != hipSuccess || hipMalloc(&clk, sizeof(long long) * 1024 * 16) != hipSuccess) return 1;
  run<double>("v_fma_f64", (double*)out, clk);
  run<float>("v_fma_f32", (float*)out, clk);
  return 0;
}
