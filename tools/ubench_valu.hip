// ubench_valu.hip -- instruction-rate microbenchmarks for the ops pair_tiles_kernel is made of
// (v_and_b32, v_bitop3_b32, v_bcnt_u32_b32) on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;

// MODE 0: 8 independent v_bcnt chains; 1: v_and; 2: v_bitop3 (xor-and); 3: the kernel's mix per pair-dword;
// 4: the missing-call (general) kernel's mix per pair-dword: 9 logic ops + 7 v_bcnt
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t seed) {
  uint32_t a[8], x = seed + threadIdx.x, y = seed * 3 + threadIdx.x, z = seed * 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = k + threadIdx.x;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) {
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
      } else if (MODE == 1) {
        asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[k]) : "v"(x));
      } else if (MODE == 2) {
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x48" : "+v"(a[k]) : "v"(x), "v"(y));
      } else if (MODE == 4) {
        uint32_t h, t, iN, iP, n2, n3, n4, n5, n6;
        const uint32_t iH = a[(k + 1) & 7], iR = a[(k + 2) & 7];
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(iH));
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x48" : "=v"(t) : "v"(y), "v"(h), "v"(iR));
        asm volatile("v_or_b32 %0, %1, %2" : "=v"(iN) : "v"(iH), "v"(iR));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(iP) : "v"(iH), "v"(iR));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(n2) : "v"(iN), "v"(z));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(n3) : "v"(iN), "v"(x));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(n4) : "v"(iN), "v"(y));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(n5) : "v"(z), "v"(iH));
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(n6) : "v"(z), "v"(iP));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(h));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 3) & 7]) : "v"(t));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 4) & 7]) : "v"(n2));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 5) & 7]) : "v"(n3));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 6) & 7]) : "v"(n4));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 7) & 7]) : "v"(n5));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(n6));
      } else {
        uint32_t h, t;
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(a[(k + 1) & 7]));
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x48" : "=v"(t) : "v"(y), "v"(h), "v"(z));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[k]) : "v"(h));
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[(k + 4) & 7]) : "v"(t));
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int blocks_per_cu, int ops_per_inner) {
  const int blocks = 256 * blocks_per_cu;
  uint32_t* d;
  CK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double lane_ops = (double)blocks * 256 * kIters * 8 * ops_per_inner;
  printf("%-28s blocks/CU %d  %.3f ms  %.3e lane-ops/s  = %.1f%% of 256CU*128lanes*2.4GHz\n", name, blocks_per_cu, ms, lane_ops / (ms * 1e-3),
         100.0 * lane_ops / (ms * 1e-3) / (256.0 * 128 * 2.4e9));
  CK(hipFree(d));
  return 0;
}

int main() {
  for (int bpc : {1, 2, 4, 8}) {
    run<0>("v_bcnt_u32_b32", bpc, 1);
    run<1>("v_and_b32", bpc, 1);
    run<2>("v_bitop3_b32", bpc, 1);
    run<3>("and+bitop3+2bcnt mix", bpc, 4);
    run<4>("general mix (9 logic+7 bcnt)", bpc, 16);
  }
  return 0;
}
