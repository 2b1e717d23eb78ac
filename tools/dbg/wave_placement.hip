// Diagnostic: where do the waves of two co-resident 256-thread workgroups land?  (SIMD id per wave index)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
  extern __shared__ double lds[];
  const int w = threadIdx.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double acc = threadIdx.x;
  for (int i = 0; i < spin; i++) acc = acc * 1.0000001 + 1e-9;
  lds[threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + w) * 4 + 0] = hw;
    out[(blockIdx.x * 4 + w) * 4 + 1] = xcc;
    out[(blockIdx.x * 4 + w) * 4 + 2] = (unsigned)(t0 >> 8);
    out[(blockIdx.x * 4 + w) * 4 + 3] = (unsigned)lds[threadIdx.x];
  }
}
int main() {
  const int B = 1024;
  unsigned* d;
  hipMalloc(&d, B * 16 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 79 * 1024);
  hipLaunchKernelGGL(k, dim3(B), dim3(256), 79 * 1024, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(B * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int same = 0, hist[4][4] = {};
  for (int b = 0; b < B; b++) {
    for (int w = 0; w < 4; w++) hist[w][(h[(b * 4 + w) * 4] >> 4) & 3]++;
  }
  for (int b = 0; b < 16; b++) {
    printf("wg %4d xcc %u:", b, h[b * 16 + 1] & 15);
    for (int w = 0; w < 4; w++) { unsigned hw = h[(b * 4 + w) * 4]; printf("  w%d simd %u cu %2u se %u slot %2u", w, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, hw & 15); }
    printf("\n");
  }
  printf("wave index -> SIMD histogram:\n");
  for (int w = 0; w < 4; w++) printf("  w%d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // co-resident pairs: same (xcc, se, cu); the first two workgroups that landed on each CU
  {
    int first[8 * 8 * 16];
    for (auto& f : first) f = -1;
    int pairs = 0, coll[4] = {};
    for (int b = 0; b < B; b++) {
      unsigned hw = h[(b * 4) * 4], xcc = h[b * 16 + 1] & 15;
      int key = (xcc * 8 + ((hw >> 13) & 7)) * 16 + ((hw >> 8) & 15);
      if (first[key] < 0) { first[key] = b; continue; }
      if (first[key] == -2) continue;
      int a = first[key];
      first[key] = -2;
      pairs++;
      for (int w = 0; w < 4; w++)
        if (((h[(a * 4 + w) * 4] >> 4) & 3) == ((h[(b * 4 + w) * 4] >> 4) & 3)) coll[w]++;
      if (pairs <= 8) printf("cu key %d: wg %d and wg %d: w3 on simd %u / %u\n", key, a, b, (h[(a * 4 + 3) * 4] >> 4) & 3, (h[(b * 4 + 3) * 4] >> 4) & 3);
    }
    printf("pairs %d; same SIMD for wave index w in both: %d %d %d %d\n", pairs, coll[0], coll[1], coll[2], coll[3]);
  }
  (void)same;
  return 0;
}
