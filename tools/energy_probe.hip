// energy_probe.hip -- what a v_mfma_scale_f32_32x32x64_f8f6f4 (FP4) costs on a POWER-LIMITED MI355X, by operand data and by the
// VALU work issued beside it.  The pair kernels of this repo run at the socket's power cap (rocm-smi during a step: 1,330-1,380 W
// of 1,400 W, shader clock pulled from 2.4 to ~2.1 GHz; profiles/r04_power_during_step.txt), so their throughput is set by the
// ENERGY per useful instruction, not by issue slots or stalls.  This probe measures that price list:
//   * operand encodings of the same genotypes (HWE draws, MAF ~ U(0.01, 0.5), like the bench generator):
//       x   : +2 / 0 / -2 (E2M1 0100 / 0000 / 1100): what the kernels feed today (hom-REF = +2)
//       g   : 0 / 0.5 / 1 (0000 / 0001 / 0010): allele counts, hom-REF = 0 -> most products are 0 x 0
//       mix : MFMAs alternate between the two (odd samples x-coded, even samples g-coded)
//     beside all-zero and uniformly random nibbles;
//   * K extra v_bitop3_b32 per MFMA (the operand expansion), K = 0 .. 6.
// Every run: 2 waves per SIMD, 8 accumulators per wave, operands cycled through 4 + 4 register fragments so that consecutive
// MFMAs see different data; prints ms, PFLOP/s, shader clock (clock64 / wall_clock64) and the time per MFMA relative to row 1.
// Build: hipcc --offload-arch=gfx950 -O3 tools/energy_probe.hip -o tools/_bin/energy_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <random>
#include <string>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

// src: [frag 0..7][dword 0..3][lane 0..255]: fragments 0-3 are A operands, 4-7 B operands; with `alt` the MFMAs of odd rounds take
// fragments 2,3 / 6,7 and those of even rounds 0,1 / 4,5 (the "mix" layout: two encodings, never mixed inside one MFMA)
template <int K>
__global__ __launch_bounds__(256, 2) void probe_kernel(const uint32_t* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int l = threadIdx.x;
  v16f acc[8];
  for (int p = 0; p < 8; ++p) {
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }
  uint32_t f[8][4];
  for (int q = 0; q < 8; ++q) {
    for (int d = 0; d < 4; ++d) {
      f[q][d] = src[(q * 4 + d) * 256 + l];
    }
  }
  uint32_t dummy[6];
  for (int q = 0; q < 6; ++q) {
    dummy[q] = src[q * 256 + l] ^ (0x9e3779b9u * (q + 1));
  }
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int a = 2 * round + (p & 1), b = 4 + 2 * round + ((p >> 1) & 1);
        const v8i A = {(int)f[a][0], (int)f[a][1], (int)f[a][2], (int)f[a][3], 0, 0, 0, 0};
        const v8i B = {(int)f[b][0], (int)f[b][1], (int)f[b][2], (int)f[b][3], 0, 0, 0, 0};
        acc[p] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc[p], 4, 4, 0, 0x7e7e7e7e, 0, 0x7e7e7e7e);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          dummy[k] = __builtin_amdgcn_bitop3_b32(dummy[k], 0x44444444u + it, 0xccccccccu, 0x28);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (K) {
          __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
        }
      }
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int p = 0; p < 8; ++p) {
    for (int g = 0; g < 16; ++g) {
      sum += acc[p][g];
    }
  }
  uint32_t x = 0;
  for (int q = 0; q < 6; ++q) {
    x ^= dummy[q];
  }
  out[blockIdx.x * 256 + l] = sum + static_cast<float>(x & 1);
  if ((blockIdx.x == 0) && (l == 0)) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

// the same contraction as v_mfma_scale_f32_16x16x128_f8f6f4: half the multiply-adds per instruction (16 x 16 pairs x 128 samples), a
// quarter of the accumulator registers (4 per lane): 16 accumulators here, two instructions for the work of one 32x32x64
typedef float v4f __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(256, 2) void probe16_kernel(const uint32_t* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int l = threadIdx.x;
  v4f acc[16];
  for (int p = 0; p < 16; ++p) {
    for (int g = 0; g < 4; ++g) {
      acc[p][g] = 0.f;
    }
  }
  uint32_t f[8][4];
  for (int q = 0; q < 8; ++q) {
    for (int d = 0; d < 4; ++d) {
      f[q][d] = src[(q * 4 + d) * 256 + l];
    }
  }
  uint32_t dummy[6];
  for (int q = 0; q < 6; ++q) {
    dummy[q] = src[q * 256 + l] ^ (0x9e3779b9u * (q + 1));
  }
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int a = 2 * round + (p & 1), b = 4 + 2 * round + ((p >> 1) & 1);
        const v8i A = {(int)f[a][0], (int)f[a][1], (int)f[a][2], (int)f[a][3], 0, 0, 0, 0};
        const v8i B = {(int)f[b][0], (int)f[b][1], (int)f[b][2], (int)f[b][3], 0, 0, 0, 0};
        acc[p] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[p], 4, 4, 0, 0x7e7e7e7e, 0, 0x7e7e7e7e);
        if (p & 1) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            dummy[k] = __builtin_amdgcn_bitop3_b32(dummy[k], 0x44444444u + it, 0xccccccccu, 0x28);
          }
        }
      }
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int p = 0; p < 16; ++p) {
    for (int g = 0; g < 4; ++g) {
      sum += acc[p][g];
    }
  }
  uint32_t x = 0;
  for (int q = 0; q < 6; ++q) {
    x ^= dummy[q];
  }
  out[blockIdx.x * 256 + l] = sum + static_cast<float>(x & 1);
  if ((blockIdx.x == 0) && (l == 0)) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

enum { kZero = 0, kRandom, kX, kG, kMix, kGminor, kNData };
static const char* const kDataName[kNData] = {"all zero", "uniform random nibbles", "genotypes, x coding (+2 / 0 / -2)", "genotypes, g coding (0 / 0.5 / 1)",
                                               "genotypes, MFMAs alternate x / g coding", "genotypes, g coding of the MINOR allele, MAF ~ U(0.01, 0.2)"};

// one lane's share of a fragment: 32 samples of one variant row (lane & 31), four dwords of eight nibbles
static void fill(std::vector<uint32_t>& h, int data, std::mt19937& rng) {
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int q = 0; q < 8; ++q) {
    double maf[32];
    for (int r = 0; r < 32; ++r) {
      maf[r] = (data == kGminor) ? (0.01 + 0.19 * U(rng)) : (0.01 + 0.49 * U(rng));
    }
    int enc = data;
    if (data == kMix) {
      enc = ((q & 3) < 2) ? kX : kG;
    }
    if (data == kGminor) {
      enc = kG;
    }
    for (int l = 0; l < 256; ++l) {
      for (int d = 0; d < 4; ++d) {
        uint32_t w = 0;
        for (int nb = 0; nb < 8; ++nb) {
          uint32_t nib = 0;
          if (data == kRandom) {
            nib = rng() & 15u;
          } else if (data != kZero) {
            const double p = maf[l & 31];
            const double u = U(rng);
            const int g = (u < (1 - p) * (1 - p)) ? 0 : ((u < 1 - p * p) ? 1 : 2);
            nib = (enc == kX) ? ((g == 0) ? 0x4u : ((g == 1) ? 0x0u : 0xcu)) : static_cast<uint32_t>(g);
          }
          w |= nib << (4 * nb);
        }
        h[(q * 4 + d) * 256 + l] = w;
      }
    }
  }
}

template <int K>
static int run(const uint32_t* src, float* out, unsigned long long* clk, int blocks, int iters, double* ms_out, double* mhz_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {  // (the last of three: the clock has settled)
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe_kernel<K>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  unsigned long long hc[2];
  CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
  *ms_out = ms;
  *mhz_out = hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0;
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return 0;
}

// `energy_probe long <data 0..5> <K 0|3|6> <seconds>`: ONE configuration launched back to back for a few seconds, so that a power
// sampler beside it (tools/attribution.py --probe: rocm-smi at a few Hz) sees that configuration alone; prints the rate and the in-kernel clock
static int long_run(int data, int K, double seconds) {
  uint32_t* src;
  float* out;
  unsigned long long* clk;
  const int blocks = 256 * 2 * 8, iters = 1500;
  CHECK(hipMalloc(&src, 8 * 4 * 256 * 4));
  CHECK(hipMalloc(&out, blocks * 256 * 4));
  CHECK(hipMalloc(&clk, 16));
  std::mt19937 rng(12345);
  std::vector<uint32_t> h(8 * 4 * 256);
  fill(h, data, rng);
  CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  const double mfmas = (double)blocks * 4 * iters * 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double total_ms = 0.0, mhz_sum = 0.0;
  int launches = 0;
  while (total_ms < seconds * 1e3) {
    CHECK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < 20; ++rep) {
      if (K == 0) {
        hipLaunchKernelGGL((probe_kernel<0>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
      } else if (K == 3) {
        hipLaunchKernelGGL((probe_kernel<3>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
      } else {
        hipLaunchKernelGGL((probe_kernel<6>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
      }
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[2];
    CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    total_ms += ms;
    launches += 20;
    mhz_sum += hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0;
  }
  printf("{\"probe\": \"%s\", \"valu_per_mfma\": %d, \"seconds\": %.2f, \"mfma_instructions_per_s\": %.4e, \"pflops\": %.3f, \"in_kernel_mhz\": %.0f}\n", kDataName[data], K,
         total_ms / 1e3, mfmas * launches / (total_ms * 1e-3), 2 * mfmas * 65536.0 * launches / (total_ms * 1e-3) / 1e15, mhz_sum / (launches / 20));
  return 0;
}

int main(int argc, char** argv) {
  if ((argc == 5) && (std::string(argv[1]) == "long")) {
    return long_run(atoi(argv[2]), atoi(argv[3]), atof(argv[4]));
  }
  uint32_t* src;
  float* out;
  unsigned long long* clk;
  const int blocks = 256 * 2 * 8;  // 2 workgroups of 4 waves per CU at a time (2 waves per SIMD), 8 rounds
  const int iters = 1500;          // x 16 MFMAs per wave
  CHECK(hipMalloc(&src, 8 * 4 * 256 * 4));
  CHECK(hipMalloc(&out, blocks * 256 * 4));
  CHECK(hipMalloc(&clk, 16));
  std::mt19937 rng(12345);
  std::vector<uint32_t> h(8 * 4 * 256);
  const double mfmas = (double)blocks * 4 * iters * 16;
  double base = 0.0;
  printf("energy_probe: %d workgroups x 4 waves x %d MFMAs, 2 waves per SIMD, 8 accumulators\n", blocks, iters * 16);
  for (int data = 0; data < kNData; ++data) {
    fill(h, data, rng);
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double ms, mhz;
    if (run<0>(src, out, clk, blocks, iters, &ms, &mhz)) {
      return 1;
    }
    if (data == kX) {
      base = ms;
    }
    printf("A. MFMA alone, %-62s: %7.3f ms = %5.2f PFLOP/s at %4.0f MHz\n", kDataName[data], ms, 2 * mfmas * 65536.0 / (ms * 1e-3) / 1e15, mhz);
  }
  for (int data : {kX, kG, kMix}) {
    fill(h, data, rng);
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    double ms[7], mhz[7];
    if (run<0>(src, out, clk, blocks, iters, &ms[0], &mhz[0]) || run<1>(src, out, clk, blocks, iters, &ms[1], &mhz[1]) ||
        run<2>(src, out, clk, blocks, iters, &ms[2], &mhz[2]) || run<3>(src, out, clk, blocks, iters, &ms[3], &mhz[3]) ||
        run<4>(src, out, clk, blocks, iters, &ms[4], &mhz[4]) || run<5>(src, out, clk, blocks, iters, &ms[5], &mhz[5]) ||
        run<6>(src, out, clk, blocks, iters, &ms[6], &mhz[6])) {
      return 1;
    }
    for (int k = 0; k < 7; ++k) {
      printf("B. %-40s + %d v_bitop3 per MFMA: %7.3f ms = %5.2f PFLOP/s at %4.0f MHz, time per MFMA x %.3f (vs x coding alone)\n", kDataName[data], k, ms[k],
             2 * mfmas * 65536.0 / (ms[k] * 1e-3) / 1e15, mhz[k], base > 0 ? ms[k] / base : 0.0);
    }
  }
  // C. the 16x16x128 form: two instructions per 65,536 multiply-adds, 4 accumulator registers per lane instead of 16
  for (int data : {kX, kG}) {
    fill(h, data, rng);
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int k : {0, 3}) {
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        if (k == 0) {
          hipLaunchKernelGGL((probe16_kernel<0>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
        } else {
          hipLaunchKernelGGL((probe16_kernel<3>), dim3(blocks), dim3(256), 0, 0, src, out, iters, clk);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      unsigned long long hc[2];
      CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
      printf("C. 16x16x128, %-40s + %d v_bitop3 per 65,536 multiply-adds: %7.3f ms = %5.2f PFLOP/s at %4.0f MHz, time per 65,536 multiply-adds x %.3f (vs 32x32x64, x coding alone)\n",
             kDataName[data], k, ms, 2 * mfmas * 65536.0 / (ms * 1e-3) / 1e15, hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0, base > 0 ? ms / base : 0.0);
    }
  }
  CHECK(hipFree(src));
  CHECK(hipFree(out));
  CHECK(hipFree(clk));
  return 0;
}
