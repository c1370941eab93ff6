// tile_shape_probe.hip -- the STAGE LOOP of pair_mfma_wide_kernel (plink-ng_amd/csrc/ldp_pair_wide.hip) in other tile / wave shapes, measured
// instead of argued (the round-5 review's item: "four waves x 512 VGPRs on an 8 x 12 tile").  Same pipeline as the kernel -- 16-byte LDS-DMA
// (global_load_lds) of 512-sample stages into a two-stage ring with the XOR-swizzled row layout, s_waitcnt vmcnt(n) + one workgroup barrier per
// stage, ds_read_b128 of two pieces per lane, the allele-count FP4 expansion of ldp_mfma_device.h, v_mfma_scale_f32_32x32x64_f8f6f4 -- and the
// same launch geometry (eight XCD streams of neighbouring tiles), but NO checkpoints and NO per-pair epilogue: every tile is a full rectangle off
// the diagonal and runs the whole length of the rows; a wave's accumulators end in a checksum (the sum of G over every pair of the plan), which
// must be the same number for every shape and equal to the one a plain kernel computes from the rows.  So the probe answers one question: what
// do the matrix pipe, the LDS and the L2 -> LDS leg deliver per MFMA when a wave owns RJ x RV products of a TJ x TV tile -- at which power and clock.
//   shape   waves  products / wave   tile     staged row-blocks / products      accumulators
//   2x4       8       2 x 4          8 x 8        16 / 64                         128 VGPRs   (the kernel's shape, the stage form of rounds 3-5)
//   4x4       4       4 x 4          8 x 8        16 / 64                         256 AGPRs   (one wave per SIMD, 512 registers)
//   2x6       8       2 x 6          8 x 12       20 / 96                         192
//   4x6       4       4 x 6          8 x 12       20 / 96                         384         (the review's: one wave per SIMD; two passes of two k-steps)
//   2x4p2 / 4x4p2 / 2x6p2                                                                     the same in two passes of two k-steps (ds_read_b64)
//   2x4pf / 4x4pf     the second half-stage's J fragments made during the first (what pair_mfma_wide_kernel's complete-data instantiation does since round 6)
//   2x4pfh            ... and a stage's last eight MFMAs held back for the head of the next stage
//   2x4pf2            ... or the first half-stage's J fragments expanded k-step by k-step in front of V block 0's MFMAs
//   2x4r4             256-sample stages in a ring of four, every stage's J fragments made during the stage before
// -DPROBE_ENC=1 / 2: other operand codings of the same genotypes (powers of two; the +-2 coding), see pexp() below.
// Results: profiles/r06_experiments.md section 4b.  bench.py runs `tile_shape_probe 2x4pf 2` beside its other ceilings (roofline.stage_loop_probe_frac_of_peak).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plink-ng_amd/csrc tools/tile_shape_probe.hip -o tools/_bin/tile_shape_probe
// Run:   tile_shape_probe <shape> [seconds] [samples] [J tiles] [reach in row-blocks]        (one JSON line; tools/tile_shape_summary.py adds the power samples)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "ldp_mfma_device.h"
#include "ldp_pair_device.h"

using namespace ldp;

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

// Operand coding of the probe (compile time, -DPROBE_ENC=n): 0 = the kernels' allele-count coding (g / 2 = 0 / 0.5 / 1.0 as E2M1 0000 / 0001 / 0010, block scale 2);
// 1 = the same counts as powers of two (g = 0 / 1 / 2 as 0000 / 0010 / 0100: no mantissa bit anywhere, block scale 1; one more VALU per code dword);
// 2 = the +-2 coding of the SPARSE instantiation (x = 1 - g = +2 / 0 / -2, block scale 1 / 2; the checksum is then the sum of x.x and is compared
// with nothing).  What an MFMA costs at the power cap depends on its operands' VALUES (tools/energy_probe.hip); this prices them inside the real loop.
#ifndef PROBE_ENC
#define PROBE_ENC 0
#endif
__device__ __forceinline__ void pexp(uint32_t c0, uint32_t c1, Frag& f) {
#if PROBE_ENC == 1
  f.d[0] = (c0 << 1) & 0x66666666u;
  f.d[1] = (c0 >> 1) & 0x66666666u;
  f.d[2] = (c1 << 1) & 0x66666666u;
  f.d[3] = (c1 >> 1) & 0x66666666u;
#elif PROBE_ENC == 2
  fp4_of_codes(c0, c1, f);
#else
  fp4_g_of_codes(c0, c1, f);
#endif
}
__device__ __forceinline__ mf_v16f pmfma(const Frag& a, const Frag& b, mf_v16f c) {
#if PROBE_ENC == 1
  const mf_v8i A = {static_cast<int>(a.d[0]), static_cast<int>(a.d[1]), static_cast<int>(a.d[2]), static_cast<int>(a.d[3]), 0, 0, 0, 0};
  const mf_v8i B = {static_cast<int>(b.d[0]), static_cast<int>(b.d[1]), static_cast<int>(b.d[2]), static_cast<int>(b.d[3]), 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#elif PROBE_ENC == 2
  return mfma_fp4(a, b, c);
#else
  return mfma_fp4g(a, b, c);
#endif
}

constexpr uint32_t kStageSamples = 512, kRowStageBytes = 128, kPieces = 8, kBlockUnits = 32 * kPieces;  // as ldp_pair_wide.hip
constexpr uint32_t kTJ = 8;

// ---- the image: n_rows variants x n_samples (a multiple of 512) 2-bit allele counts of the minor allele, HWE draws, MAF ~ U(0.01, 0.5) ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__global__ void gen_kernel(uint32_t* codes, uint32_t row_dwords, uint32_t n_rows) {
  const uint32_t row = blockIdx.x;
  const uint32_t dw = blockIdx.y * blockDim.x + threadIdx.x;
  if ((row >= n_rows) || (dw >= row_dwords)) {
    return;
  }
  const double p = 0.01 + 0.49 * (static_cast<double>(mix32(row * 0x9e3779b9u + 12345u)) / 4294967296.0);
  const uint32_t t2 = static_cast<uint32_t>(p * p * 4294967296.0);                    // hom-minor
  const uint32_t t1 = t2 + static_cast<uint32_t>(2.0 * p * (1.0 - p) * 4294967296.0);  // het
  uint32_t w = 0;
  for (uint32_t s = 0; s < 16; ++s) {
    const uint32_t u = mix32((row * 0x85ebca6bu) ^ mix32(dw * 16u + s + 0x1234567u));
    const uint32_t g = (u < t2) ? 2u : ((u < t1) ? 1u : 0u);
    w |= g << (2 * s);
  }
  codes[static_cast<uint64_t>(row) * row_dwords + dw] = w;
}

// ---- the plain checksum: sum over J tiles, over the samples, of (sum of g over the tile's 256 J rows) x (sum of g over its V rows) ----
__global__ void ref_kernel(const uint32_t* __restrict__ codes, uint32_t row_dwords, uint32_t n_jt, uint32_t reach_rows, unsigned long long* out) {
  const uint32_t tj = blockIdx.y;
  const uint32_t dw = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long sum = 0;
  if ((tj < n_jt) && (dw < row_dwords)) {
    uint32_t a[16], b[16];
    for (int s = 0; s < 16; ++s) {
      a[s] = 0;
      b[s] = 0;
    }
    const uint32_t j0 = reach_rows + tj * 256, v0 = tj * 256;
    for (uint32_t r = 0; r < 256; ++r) {
      const uint32_t w = codes[static_cast<uint64_t>(j0 + r) * row_dwords + dw];
      for (int s = 0; s < 16; ++s) {
        a[s] += (w >> (2 * s)) & 3u;
      }
    }
    for (uint32_t r = 0; r < reach_rows; ++r) {
      const uint32_t w = codes[static_cast<uint64_t>(v0 + r) * row_dwords + dw];
      for (int s = 0; s < 16; ++s) {
        b[s] += (w >> (2 * s)) & 3u;
      }
    }
    for (int s = 0; s < 16; ++s) {
      sum += static_cast<unsigned long long>(a[s]) * b[s];
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, sum);
  }
}

// ---- one half-stage (256 samples) of a wave's RJ x RV rectangle: the J fragments of all four k-steps in registers, the V blocks streamed past ----
template <int RJ, int RV>
__device__ __forceinline__ void probe_stage(const mf_u4* __restrict__ st4, const uint32_t (&joff)[RJ], const uint32_t (&voff)[RV], uint32_t oH, uint32_t oR,
                                            mf_v16f (&acc)[RJ * RV]) {
  mf_u4 vH[2], vR[2];
  vH[0] = st4[voff[0] + oH];
  vR[0] = st4[voff[0] + oR];
  Frag fj[RJ][4];
#pragma unroll
  for (int q = 0; q < RJ; ++q) {
    mf_u4 H = st4[joff[q] + oH], R = st4[joff[q] + oR];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      pexp(H[ks], R[ks], fj[q][ks]);
    }
  }
#pragma unroll
  for (int b = 0; b < RV; ++b) {
    __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from hoisting every block's reads and expansions to the front: registers)
    if (b + 1 < RV) {
      vH[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oH];
      vR[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oR];
    }
    opaque(vH[b & 1], vR[b & 1]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fv;
      pexp(vH[b & 1][ks], vR[b & 1][ks], fv);
#pragma unroll
      for (int q = 0; q < RJ; ++q) {
        acc[q * RV + b] = pmfma(fv, fj[q][ks], acc[q * RV + b]);
      }
    }
  }
}

// The same half-stage in two passes of two k-steps (for rectangles whose accumulators leave no room for sixteen J fragments: 4 x 6 = 384 registers
// of accumulators): a pass holds the J fragments of its two k-steps (RJ x 8 registers) and reads the matching HALF of every piece (ds_read_b64: k-step
// ks is dword ks of a piece), so the bytes read from LDS and the expansions are those of the one-pass form, in twice the read instructions.
typedef uint32_t mf_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void opaque2(mf_u2& a, mf_u2& b) { asm("" : "+v"(a), "+v"(b)); }
template <int RJ, int RV>
__device__ __forceinline__ void probe_stage_2pass(const mf_u4* __restrict__ st4, const uint32_t (&joff)[RJ], const uint32_t (&voff)[RV], uint32_t oH, uint32_t oR,
                                                  mf_v16f (&acc)[RJ * RV]) {
  const mf_u2* __restrict__ st2 = reinterpret_cast<const mf_u2*>(st4);
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    mf_u2 vH[2], vR[2];
    vH[0] = st2[2 * (voff[0] + oH) + kh];
    vR[0] = st2[2 * (voff[0] + oR) + kh];
    Frag fj[RJ][2];
#pragma unroll
    for (int q = 0; q < RJ; ++q) {
      mf_u2 H = st2[2 * (joff[q] + oH) + kh], R = st2[2 * (joff[q] + oR) + kh];
      opaque2(H, R);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        pexp(H[ks], R[ks], fj[q][ks]);
      }
    }
#pragma unroll
    for (int b = 0; b < RV; ++b) {
      __builtin_amdgcn_sched_barrier(0);
      if (b + 1 < RV) {
        vH[(b & 1) ^ 1] = st2[2 * (voff[(b + 1 < RV) ? b + 1 : b] + oH) + kh];
        vR[(b & 1) ^ 1] = st2[2 * (voff[(b + 1 < RV) ? b + 1 : b] + oR) + kh];
      }
      opaque2(vH[b & 1], vR[b & 1]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag fv;
        pexp(vH[b & 1][ks], vR[b & 1][ks], fv);
#pragma unroll
        for (int q = 0; q < RJ; ++q) {
          acc[q * RV + b] = pmfma(fv, fj[q][ks], acc[q * RV + b]);
        }
      }
    }
  }
}

// The half-stage with the NEXT half-stage's J fragments made beside it (for one wave per SIMD: nobody else fills the matrix pipe while a wave reads and
// expands its J blocks at the head of a half-stage).  fj: this half-stage's J fragments, ready; PREF: during V block b < RJ the pieces of J block b at
// (oHn, oRn) are read and expanded into fjn.  RJ x 16 more registers.
template <int RJ, int RV, bool PREF, bool HOLD = false>
__device__ __forceinline__ void probe_stage_pref(const mf_u4* __restrict__ st4, const mf_u4* __restrict__ stn4, const uint32_t (&joff)[RJ], const uint32_t (&voff)[RV], uint32_t oH,
                                                 uint32_t oR, uint32_t oHn, uint32_t oRn, const Frag (&fj)[RJ][4], Frag (&fjn)[RJ][4], mf_v16f (&acc)[RJ * RV], Frag* held = nullptr) {
  static_assert(RJ <= RV, "a J block of the next half-stage per V block");
  mf_u4 vH[2], vR[2], jH[2], jR[2];
  vH[0] = st4[voff[0] + oH];
  vR[0] = st4[voff[0] + oR];
  if constexpr (PREF) {
    jH[0] = stn4[joff[0] + oHn];
    jR[0] = stn4[joff[0] + oRn];
  }
#pragma unroll
  for (int b = 0; b < RV; ++b) {
    __builtin_amdgcn_sched_barrier(0);
    if (b + 1 < RV) {
      vH[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oH];
      vR[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oR];
    }
    if constexpr (PREF) {
      if (b + 1 < RJ) {
        jH[(b & 1) ^ 1] = stn4[joff[(b + 1 < RJ) ? b + 1 : 0] + oHn];
        jR[(b & 1) ^ 1] = stn4[joff[(b + 1 < RJ) ? b + 1 : 0] + oRn];
      }
    }
    opaque(vH[b & 1], vR[b & 1]);
    if constexpr (PREF) {
      if (b < RJ) {
        opaque(jH[b & 1], jR[b & 1]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (HOLD && (b == RV - 1)) {
        pexp(vH[b & 1][ks], vR[b & 1][ks], held[ks]);  // (the last V block's MFMAs wait for the head of the next stage)
        continue;
      }
      Frag fv;
      pexp(vH[b & 1][ks], vR[b & 1][ks], fv);
      if constexpr (PREF) {
        if (b < RJ) {
          pexp(jH[b & 1][ks], jR[b & 1][ks], fjn[(b < RJ) ? b : 0][ks]);
        }
      }
#pragma unroll
      for (int q = 0; q < RJ; ++q) {
        acc[q * RV + b] = pmfma(fv, fj[q][ks], acc[q * RV + b]);
      }
    }
  }
}

// The first half-stage of a stage without its head (shape "2x4pf2"): the J fragments of k-step ks are expanded right in front of V block 0's MFMAs of that k-step
// (18 VALU in front of the first MFMA instead of 54), and the next half-stage's J fragments are made during the LAST two V blocks instead of the first two.
template <int RV>
__device__ __forceinline__ void probe_stage_first(const mf_u4* __restrict__ st4, const uint32_t (&joff)[2], const uint32_t (&voff)[RV], uint32_t oH, uint32_t oR, uint32_t oHn,
                                                  uint32_t oRn, Frag (&fj)[2][4], Frag (&fjn)[2][4], mf_v16f (&acc)[2 * RV]) {
  mf_u4 vH[2], vR[2], jH[2], jR[2];
  jH[0] = st4[joff[0] + oH];
  jR[0] = st4[joff[0] + oR];
  jH[1] = st4[joff[1] + oH];
  jR[1] = st4[joff[1] + oR];
  vH[0] = st4[voff[0] + oH];
  vR[0] = st4[voff[0] + oR];
#pragma unroll
  for (int b = 0; b < RV; ++b) {
    __builtin_amdgcn_sched_barrier(0);
    if (b + 1 < RV) {
      vH[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oH];
      vR[(b & 1) ^ 1] = st4[voff[(b + 1 < RV) ? b + 1 : b] + oR];
    }
    if (b == RV - 3) {  // the next half-stage's J pieces, in time for the last two V blocks
      jH[0] = st4[joff[0] + oHn];
      jR[0] = st4[joff[0] + oRn];
      jH[1] = st4[joff[1] + oHn];
      jR[1] = st4[joff[1] + oRn];
    }
    opaque(vH[b & 1], vR[b & 1]);
    if ((b == 0) || (b >= RV - 2)) {
      opaque(jH[0], jR[0]);
      opaque(jH[1], jR[1]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fv;
      pexp(vH[b & 1][ks], vR[b & 1][ks], fv);
      if (b == 0) {
        pexp(jH[0][ks], jR[0][ks], fj[0][ks]);
        pexp(jH[1][ks], jR[1][ks], fj[1][ks]);
      } else if (b == RV - 2) {
        pexp(jH[0][ks], jR[0][ks], fjn[0][ks]);
      } else if (b == RV - 1) {
        pexp(jH[1][ks], jR[1][ks], fjn[1][ks]);
      }
      acc[b] = pmfma(fv, fj[0][ks], acc[b]);
      acc[RV + b] = pmfma(fv, fj[1][ks], acc[RV + b]);
    }
  }
}

// tile idx of the launch: J tile idx / n_vt (J blocks from row reach_rows + 256 tj), V tile idx % n_vt (V blocks from row 256 tj + 32 TV k): the
// n_vt V tiles of a J tile cover the `reach_rows` rows in front of it, whatever TV is
template <int WAVES, int RJ, int RV, int TV, int PASSES>
__global__ __launch_bounds__(WAVES * 64, 1) void tile_probe_kernel(const uint8_t* __restrict__ codes, uint32_t row_bytes, uint32_t n_stages, uint32_t n_tiles, uint32_t n_vt,
                                                                   uint32_t reach_rows, unsigned long long* out) {
  constexpr uint32_t NS = kTJ + TV;                    // staged row-blocks: slots 0..7 the J blocks, 8.. the V blocks
  constexpr uint32_t kStageDwords = NS * kBlockUnits * 4;
  constexpr uint32_t kDma = (NS * 4) / WAVES;          // DMA wave-instructions per wave and stage (an instruction = a quarter of a row-block: 8 rows x 8 pieces)
  constexpr uint32_t WJ = kTJ / RJ;
  static_assert((NS * 4) % WAVES == 0 && WJ * (TV / RV) == WAVES && kTJ % RJ == 0 && TV % RV == 0, "shape");
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63, r = lane & 31, h = lane >> 5;
  const uint32_t per_xcd = n_tiles >> 3;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const uint32_t tj = idx / n_vt, kv = idx - tj * n_vt;
  const uint32_t jv0 = reach_rows + tj * 256, vv0 = tj * 256 + kv * TV * 32;
  const uint32_t a0 = RJ * (wave % WJ), b0 = RV * (wave / WJ);

  // DMA instruction T = wave + WAVES t of a stage: a quarter (8 rows x 8 pieces) of row-block slot T / 4.  WAVES is a multiple of 4, so the quarter --
  // and with it this lane's row and piece -- is the same for every t: one per-lane offset, and uniform row-block bases the scalar unit steps along
  static_assert(WAVES % 4 == 0, "a wave always fetches the same quarter of a row-block");
  const uint32_t rr = (wave & 3) * 8 + (lane >> 3);
  const uint32_t src_off = rr * row_bytes + ((lane & 7) ^ ((rr >> 1) & 7u)) * 16;
  const uint8_t* jbase = codes + static_cast<uint64_t>(jv0) * row_bytes;
  const uint8_t* vbase = codes + static_cast<uint64_t>(vv0) * row_bytes;
  uint32_t joff[RJ], voff[RV];
#pragma unroll
  for (int q = 0; q < RJ; ++q) {
    joff[q] = (a0 + q) * kBlockUnits;
  }
#pragma unroll
  for (int b = 0; b < RV; ++b) {
    voff[b] = (kTJ + b0 + b) * kBlockUnits;
  }
  const uint32_t sw = (r >> 1) & 7u;
  const uint32_t oH0 = r * kPieces + (h ^ sw), oR0 = r * kPieces + ((2 + h) ^ sw);
  const uint32_t oH1 = r * kPieces + ((4 + h) ^ sw), oR1 = r * kPieces + ((6 + h) ^ sw);

  mf_v16f acc[RJ * RV];
#pragma unroll
  for (int p = 0; p < RJ * RV; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }
  auto dma_stage = [&](uint32_t s, uint32_t buf) {
    const uint32_t kbyte = s * kRowStageBytes;
    uint32_t* dst = lds + buf * kStageDwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kDma); ++t) {
      const uint32_t T = wave + WAVES * t;
      const uint32_t slot = T >> 2;
      const uint8_t* base = ((slot < kTJ) ? (jbase + static_cast<uint64_t>(slot) * 32 * row_bytes) : (vbase + static_cast<uint64_t>(slot - kTJ) * 32 * row_bytes)) + kbyte;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + src_off), (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
    }
  };
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  unsigned long long sum = 0;
  // the f32 accumulators are integer-exact to 2^24 = 4 x 4,194,304 samples: one pass over the rows needs no flush below that
  uint32_t issued = 0, issue_buf = 0, read_buf = 0;
  dma_stage(0, 0);
  issued = 1;
  issue_buf = 1;
  [[maybe_unused]] Frag held[4], fja[RJ][4], fjb[RJ][4];
  if constexpr (PASSES == 5) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        held[ks].d[d] = 0;
#pragma unroll
        for (int q = 0; q < RJ; ++q) {
          fjb[q][ks].d[d] = 0;
        }
      }
    }
  }
  for (uint32_t kc = 0; kc < n_stages; ++kc) {
    wait_dma_then_barrier(kDma * (issued - kc - 1));
    if (issued < n_stages) {
      dma_stage(issued, issue_buf);  // (the buffer every wave finished reading before the barrier)
      ++issued;
      issue_buf ^= 1;
    }
    const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * kStageDwords);
    read_buf ^= 1;
    if constexpr (PASSES == 6) {
      static_assert(RJ == 2 && RV >= 3, "two J blocks");
      Frag fa[2][4], fb[2][4];
      probe_stage_first<RV>(st4, joff, voff, oH0, oR0, oH1, oR1, fa, fb, acc);
      probe_stage_pref<RJ, RV, false>(st4, st4, joff, voff, oH1, oR1, oH1, oR1, fb, fa, acc);
    } else if constexpr (PASSES == 5) {
      // as PASSES == 3, and the MFMAs of the stage's LAST V block (second half-stage) are held back: they are issued at the head of the NEXT stage, between
      // the expansions of its first J fragments, so that the matrix pipe has work while the new stage's first pieces come from LDS.  held / fjb start as
      // zeros (the first head adds nothing); the last stage's are flushed behind the loop.  16 more registers.
      static_assert(RJ == 2, "two J blocks");
      const mf_u4 H0 = st4[joff[0] + oH0], R0 = st4[joff[0] + oR0];
      const mf_u4 H1 = st4[joff[1] + oH0], R1 = st4[joff[1] + oR0];
      mf_u4 h0 = H0, r0 = R0, h1 = H1, r1 = R1;
      opaque(h0, r0);
      opaque(h1, r1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        acc[RV - 1] = pmfma(held[ks], fjb[0][ks], acc[RV - 1]);
        acc[2 * RV - 1] = pmfma(held[ks], fjb[1][ks], acc[2 * RV - 1]);
        pexp(h0[ks], r0[ks], fja[0][ks]);
        pexp(h1[ks], r1[ks], fja[1][ks]);
      }
      probe_stage_pref<RJ, RV, true>(st4, st4, joff, voff, oH0, oR0, oH1, oR1, fja, fjb, acc);
      probe_stage_pref<RJ, RV, false, true>(st4, st4, joff, voff, oH1, oR1, oH1, oR1, fjb, fja, acc, held);
    } else if constexpr (PASSES == 3) {
      // J fragments of the second half-stage made during the first; those of the first at the head of the stage (the stage has just landed)
      Frag fja[RJ][4], fjb[RJ][4];
#pragma unroll
      for (int q = 0; q < RJ; ++q) {
        mf_u4 H = st4[joff[q] + oH0], R = st4[joff[q] + oR0];
        opaque(H, R);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          pexp(H[ks], R[ks], fja[q][ks]);
        }
      }
      probe_stage_pref<RJ, RV, true>(st4, st4, joff, voff, oH0, oR0, oH1, oR1, fja, fjb, acc);
      probe_stage_pref<RJ, RV, false>(st4, st4, joff, voff, oH1, oR1, oH1, oR1, fjb, fja, acc);
    } else if constexpr (PASSES == 2) {
      probe_stage_2pass<RJ, RV>(st4, joff, voff, oH0, oR0, acc);
      probe_stage_2pass<RJ, RV>(st4, joff, voff, oH1, oR1, acc);
    } else {
      probe_stage<RJ, RV>(st4, joff, voff, oH0, oR0, acc);
      probe_stage<RJ, RV>(st4, joff, voff, oH1, oR1, acc);
    }
  }
  if constexpr (PASSES == 5) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[RV - 1] = pmfma(held[ks], fjb[0][ks], acc[RV - 1]);
      acc[2 * RV - 1] = pmfma(held[ks], fjb[1][ks], acc[2 * RV - 1]);
    }
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
#pragma unroll
  for (int p = 0; p < RJ * RV; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      sum += static_cast<unsigned long long>(acc[p][g]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
  }
  if (lane == 0) {
    atomicAdd(out, sum);
    if ((blockIdx.x == 0) && (wave == 0)) {
      out[1] = t1 - t0;
      out[2] = w1 - w0;
    }
  }
}

// ---- 256-sample stages in a ring of FOUR, every stage's J fragments made during the stage before (shape "2x4r4") ----
// The two-stage ring of 512-sample stages cannot make a stage's first J fragments early: the stage lands as the one before it ends.  Here the DMA runs
// two stages ahead (the same 64 KiB in flight), stage s + 1 has landed when stage s starts, and no half-stage begins with reads and expansions in
// front of an idle matrix pipe -- at twice the workgroup barriers.  Geometry of a 256-sample stage as pair_mfma_kernel's (ldp_mfma_device.h
// StageGeom<4>): four 16-byte pieces per row, piece c at unit (32 b + r) * 4 + (c ^ ((r >> 2) & 3)), lane half h reads pieces h and 2 + h.
template <int WAVES, int RJ, int RV, int TV, int PASSES>
__global__ __launch_bounds__(WAVES * 64, 1) void tile_probe_ring4_kernel(const uint8_t* __restrict__ codes, uint32_t row_bytes, uint32_t n_stages512, uint32_t n_tiles, uint32_t n_vt,
                                                                         uint32_t reach_rows, unsigned long long* out) {
  constexpr uint32_t NS = kTJ + TV;
  constexpr uint32_t kUnits = 32 * 4;                     // 16-byte units per row-block and stage
  constexpr uint32_t kStageDwords = NS * kUnits * 4;      // 8,192 dwords = 32 KiB at 16 row-blocks
  constexpr uint32_t kDma = (NS * 2) / WAVES;             // DMA wave-instructions per wave and stage (an instruction = half a row-block: 16 rows x 4 pieces)
  constexpr uint32_t WJ = kTJ / RJ;
  static_assert((NS * 2) % WAVES == 0 && WAVES % 2 == 0 && WJ * (TV / RV) == WAVES, "shape");
  extern __shared__ uint32_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63, r = lane & 31, h = lane >> 5;
  const uint32_t per_xcd = n_tiles >> 3;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const uint32_t tj = idx / n_vt, kv = idx - tj * n_vt;
  const uint32_t jv0 = reach_rows + tj * 256, vv0 = tj * 256 + kv * TV * 32;
  const uint32_t a0 = RJ * (wave % WJ), b0 = RV * (wave / WJ);
  const uint32_t n_stages = 2 * n_stages512;
  const uint32_t rr = (wave & 1) * 16 + (lane >> 2);
  const uint32_t src_off = rr * row_bytes + ((lane & 3) ^ ((rr >> 2) & 3u)) * 16;
  const uint8_t* jbase = codes + static_cast<uint64_t>(jv0) * row_bytes;
  const uint8_t* vbase = codes + static_cast<uint64_t>(vv0) * row_bytes;
  uint32_t joff[RJ], voff[RV];
#pragma unroll
  for (int q = 0; q < RJ; ++q) {
    joff[q] = (a0 + q) * kUnits;
  }
#pragma unroll
  for (int b = 0; b < RV; ++b) {
    voff[b] = (kTJ + b0 + b) * kUnits;
  }
  const uint32_t sw = (r >> 2) & 3u;
  const uint32_t oH = r * 4 + (h ^ sw), oR = r * 4 + ((2 + h) ^ sw);
  mf_v16f acc[RJ * RV];
#pragma unroll
  for (int p = 0; p < RJ * RV; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }
  auto dma_stage = [&](uint32_t s) {
    const uint32_t kbyte = s * 64u;
    uint32_t* dst = lds + (s & 3u) * kStageDwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kDma); ++t) {
      const uint32_t T = wave + WAVES * t;
      const uint32_t slot = T >> 1;
      const uint8_t* base = ((slot < kTJ) ? (jbase + static_cast<uint64_t>(slot) * 32 * row_bytes) : (vbase + static_cast<uint64_t>(slot - kTJ) * 32 * row_bytes)) + kbyte;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + src_off), (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
    }
  };
  auto st_of = [&](uint32_t s) { return reinterpret_cast<const mf_u4*>(lds + (s & 3u) * kStageDwords); };
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  uint32_t issued = 0;
  for (; (issued < 3) && (issued < n_stages); ++issued) {
    dma_stage(issued);
  }
  wait_dma_then_barrier(kDma * (issued - 1));  // stage 0 has landed
  Frag fa[RJ][4], fb[RJ][4];
  {
    const mf_u4* __restrict__ st4 = st_of(0);
#pragma unroll
    for (int q = 0; q < RJ; ++q) {
      mf_u4 H = st4[joff[q] + oH], R = st4[joff[q] + oR];
      opaque(H, R);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        pexp(H[ks], R[ks], fa[q][ks]);
      }
    }
  }
  // top of stage s: stage s + 1 has landed (its J pieces are read during s), stage s + 2 may be in flight, stage s + 3 goes into the buffer of s - 1
  auto top = [&](uint32_t s) {
    const uint32_t landed = s + 2;  // stages < landed must be in LDS
    wait_dma_then_barrier((issued > landed) ? kDma * (issued - landed) : 0u);
    if (issued < n_stages) {
      dma_stage(issued);
      ++issued;
    }
  };
  for (uint32_t s = 0; s < n_stages; s += 2) {
    top(s);
    probe_stage_pref<RJ, RV, true>(st_of(s), st_of(s + 1), joff, voff, oH, oR, oH, oR, fa, fb, acc);
    top(s + 1);
    probe_stage_pref<RJ, RV, true>(st_of(s + 1), st_of(s + 2), joff, voff, oH, oR, oH, oR, fb, fa, acc);
  }
  const unsigned long long t1 = clock64(), w1 = wall_clock64();
  unsigned long long sum = 0;
#pragma unroll
  for (int p = 0; p < RJ * RV; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      sum += static_cast<unsigned long long>(acc[p][g]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
  }
  if (lane == 0) {
    atomicAdd(out, sum);
    if ((blockIdx.x == 0) && (wave == 0)) {
      out[1] = t1 - t0;
      out[2] = w1 - w0;
    }
  }
}

struct Shape {
  const char* name;
  int waves, rj, rv, tv;
  const void* fn;
};

template <int WAVES, int RJ, int RV, int TV, int PASSES = 1>
static Shape shape_of(const char* name) {
  return Shape{name, WAVES, RJ, RV, TV, reinterpret_cast<const void*>(&tile_probe_kernel<WAVES, RJ, RV, TV, PASSES>)};
}

int main(int argc, char** argv) {
  const Shape shapes[] = {shape_of<8, 2, 4, 8>("2x4"), shape_of<4, 4, 4, 8>("4x4"), shape_of<8, 2, 6, 12>("2x6"), shape_of<4, 4, 6, 12, 2>("4x6"), shape_of<8, 2, 4, 8, 2>("2x4p2"), shape_of<4, 4, 4, 8, 2>("4x4p2"), shape_of<8, 2, 6, 12, 2>("2x6p2"), shape_of<4, 4, 4, 8, 3>("4x4pf"), shape_of<8, 2, 4, 8, 3>("2x4pf"),
                          shape_of<8, 2, 4, 8, 5>("2x4pfh"), shape_of<8, 2, 4, 8, 6>("2x4pf2"),
                          Shape{"2x4r4", 8, 2, 4, 8, reinterpret_cast<const void*>(&tile_probe_ring4_kernel<8, 2, 4, 8, 4>)}};
  const std::string want = (argc > 1) ? argv[1] : "2x4";
  const double seconds = (argc > 2) ? atof(argv[2]) : 3.0;
  const uint32_t n_samples = ((argc > 3) ? static_cast<uint32_t>(atol(argv[3])) : 500224u) / kStageSamples * kStageSamples;
  const uint32_t n_jt = (argc > 4) ? static_cast<uint32_t>(atol(argv[4])) : 256u;
  const uint32_t reach_blocks = (argc > 5) ? static_cast<uint32_t>(atol(argv[5])) : 48u;
  const Shape* S = nullptr;
  for (const Shape& s : shapes) {
    if (want == s.name) {
      S = &s;
    }
  }
  if (!S || (reach_blocks % S->tv) || !n_samples || (n_samples > 4000000u)) {
    printf("usage: tile_shape_probe 2x4|4x4|2x6|4x6|2x4p2|4x4p2|2x6p2|4x4pf|2x4pf|2x4pfh|2x4pf2|2x4r4 [seconds] [samples] [J tiles] [reach in row-blocks: a multiple of the tile's V blocks]\n");
    return 2;
  }
  const uint32_t reach_rows = reach_blocks * 32, n_rows = reach_rows + n_jt * 256;
  const uint32_t row_bytes = n_samples / 4, row_dwords = row_bytes / 4, n_stages = n_samples / kStageSamples;
  const uint32_t n_vt = reach_blocks / static_cast<uint32_t>(S->tv), n_tiles = n_jt * n_vt;
  if (n_tiles % 8) {
    printf("J tiles x V tiles must be a multiple of 8\n");
    return 2;
  }
  uint32_t* d_codes = nullptr;
  unsigned long long *d_out = nullptr, *d_ref = nullptr;
  CHECK(hipMalloc(&d_codes, static_cast<size_t>(n_rows) * row_bytes));
  CHECK(hipMalloc(&d_out, 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_ref, sizeof(unsigned long long)));
  CHECK(hipMemset(d_ref, 0, sizeof(unsigned long long)));
  hipLaunchKernelGGL(gen_kernel, dim3(n_rows, (row_dwords + 255) / 256), dim3(256), 0, 0, d_codes, row_dwords, n_rows);
  hipLaunchKernelGGL(ref_kernel, dim3((row_dwords + 255) / 256, n_jt), dim3(256), 0, 0, d_codes, row_dwords, n_jt, reach_rows, d_ref);
  CHECK(hipDeviceSynchronize());
  unsigned long long h_ref = 0;
  CHECK(hipMemcpy(&h_ref, d_ref, sizeof(h_ref), hipMemcpyDeviceToHost));

  const uint32_t lds_bytes = 2u * (kTJ + static_cast<uint32_t>(S->tv)) * kBlockUnits * 16u;
  CHECK(hipFuncSetAttribute(S->fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
  hipFuncAttributes fa;
  CHECK(hipFuncGetAttributes(&fa, S->fn));
  void* args[] = {&d_codes, const_cast<uint32_t*>(&row_bytes), const_cast<uint32_t*>(&n_stages), const_cast<uint32_t*>(&n_tiles), const_cast<uint32_t*>(&n_vt),
                  const_cast<uint32_t*>(&reach_rows), &d_out};
  auto launch = [&]() { return hipLaunchKernel(S->fn, dim3(n_tiles), dim3(static_cast<uint32_t>(S->waves) * 64), args, lds_bytes, 0); };
  // one checked launch
  CHECK(hipMemset(d_out, 0, 4 * sizeof(unsigned long long)));
  CHECK(launch());
  CHECK(hipDeviceSynchronize());
  unsigned long long h_out[4];
  CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
  const unsigned long long checked = h_out[0];
  const bool ok = (PROBE_ENC == 2) || (checked == h_ref);
  // timed launches for `seconds`
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  CHECK(launch());
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float one_ms = 0.f;
  CHECK(hipEventElapsedTime(&one_ms, e0, e1));
  int reps = static_cast<int>(seconds * 1000.0 / one_ms) + 1;
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) {
    CHECK(launch());
  }
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
  const double per = ms / reps;
  const double mfmas = static_cast<double>(n_jt) * 8.0 * reach_blocks * (n_samples / 64.0);
  const double pflops = mfmas * 131072.0 / (per * 1e-3) / 1e15;
  const double clock_mhz = h_out[2] ? (100.0 * static_cast<double>(h_out[1]) / static_cast<double>(h_out[2])) : 0.0;
  const double staged_gb = static_cast<double>(n_tiles) * (kTJ + S->tv) * 32.0 * row_bytes / 1e9;
  printf("{\"operand_coding\": %d, \"shape\": \"%s\", \"waves\": %d, \"products_per_wave\": \"%d x %d\", \"tile\": \"8 x %d\", \"vgprs\": %d, \"scratch_bytes\": %d, \"lds_bytes\": %u, "
         "\"samples\": %u, \"rows\": %u, \"tiles\": %u, \"block_products\": %.0f, \"launches\": %d, \"ms_per_launch\": %.3f, \"pflops\": %.4f, \"of_fp4_peak\": %.4f, "
         "\"ns_per_mfma_per_cu\": %.3f, \"shader_clock_mhz_wg0\": %.0f, \"staged_gb_per_launch\": %.2f, \"l2_to_lds_tb_s\": %.2f, \"image_gb\": %.2f, "
         "\"checksum\": %llu, \"checksum_plain_kernel\": %llu, \"checksum_ok\": %s}\n",
         PROBE_ENC, S->name, S->waves, S->rj, S->rv, S->tv, fa.numRegs, static_cast<int>(fa.localSizeBytes), lds_bytes, n_samples, n_rows, n_tiles, static_cast<double>(n_jt) * 8.0 * reach_blocks,
         reps, per, pflops, pflops / 10.0, per * 1e6 / (mfmas / 256.0), clock_mhz, staged_gb, staged_gb / per, static_cast<double>(n_rows) * row_bytes / 1e9, checked,
         h_ref, ok ? "true" : "false");
  return ok ? 0 : 1;
}
