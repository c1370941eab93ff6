// ldp_synth.hip -- deterministic synthetic genotype generator (benchmark / test support).
//
// Produces the workload SURVEY.md section 8(d) describes: biallelic autosomal variants, per-variant
// allele frequency ~ U(0.01, 0.5) (ALT is the major allele for half of them), Hardy-Weinberg draws,
// and LD planted the way the reference's --dummy does it (plink2_import.cc:16387-16432): with
// probability 1/2 a variant copies its predecessor and re-draws 5 % of the samples.  Every genotype is a
// pure function of (seed, global variant index, sample index), so any shard can be generated on any GPU
// (or on the host) without materialising the rest of the matrix.  Output: REF-based 2-bit codes
// (LDP_GENO_REF: 0 hom-REF, 1 het, 2 hom-ALT, 3 missing), 4 samples per byte, variant-major.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ldprune_hip_debug.h"

namespace {

__host__ __device__ inline uint64_t mix64(uint64_t x) {  // splitmix64 finalizer
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

__host__ __device__ inline uint32_t rnd32(uint64_t seed, uint64_t variant, uint64_t sample, uint32_t stream) {
  const uint64_t h = mix64(seed ^ mix64(variant * 0x100000001b3ull + stream) ^ (sample * 0xd6e8feb86659fd93ull));
  return static_cast<uint32_t>(h >> 32);
}

// per-variant draws use sample id ~0
__host__ __device__ inline uint32_t variant_rnd(uint64_t seed, uint64_t variant, uint32_t stream) {
  return rnd32(seed, variant, 0xffffffffffffull, stream);
}

__host__ __device__ inline bool is_copy(uint64_t seed, uint64_t variant) {
  // chains never cross a multiple of 8, which bounds the walk-back below
  return ((variant & 7) != 0) && (variant_rnd(seed, variant, 1) < 0x80000000u);
}

__host__ __device__ inline uint32_t alt_threshold(uint64_t seed, uint64_t variant) {
  // alt allele frequency in (0.01, 0.5), mirrored above 0.5 for half of the variants; as a 32-bit threshold
  // (seed bits 56-62, a measurement aid like bit 63: a floor on the minor-allele frequency in percent -- what a `--maf` filter in front of
  // the pruning step leaves; 0 = SURVEY 8(d)'s 0.01)
  const double u = variant_rnd(seed, variant, 2) * (1.0 / 4294967296.0);
  const uint32_t floor_pct = static_cast<uint32_t>((seed >> 56) & 0x7f);
  const double lo = (floor_pct > 1) ? ((floor_pct < 49) ? 0.01 * floor_pct : 0.49) : 0.01;
  double f = lo + (0.5 - lo) * u;
  if ((!(seed >> 63)) && (variant_rnd(seed, variant, 3) & 1)) {  // (seed bit 63, a measurement aid: ALT is the minor allele everywhere)
    f = 1.0 - f;
  }
  return static_cast<uint32_t>(f * 4294967296.0);
}

__host__ __device__ inline uint32_t synth_code(uint64_t seed, uint64_t variant, uint64_t sample, uint32_t missing_threshold) {
  if (missing_threshold && (rnd32(seed, variant, sample, 7) < missing_threshold)) {
    return 3;
  }
  uint64_t cur = variant;
  // copy from the predecessor unless this sample is one of the 5 % re-drawn
  while (is_copy(seed, cur) && !(rnd32(seed, cur, sample, 4) < 214748365u)) {
    --cur;
  }
  const uint32_t thr = alt_threshold(seed, cur);
  return (rnd32(seed, cur, sample, 5) < thr) + (rnd32(seed, cur, sample, 6) < thr);
}

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_variant, uint32_t n_variants, uint32_t founder_ct,
                                                    uint32_t missing_threshold, uint8_t* out, uint64_t stride_bytes) {
  const uint32_t bytes_per_row = (founder_ct + 3) / 4;
  const uint64_t total = static_cast<uint64_t>(n_variants) * bytes_per_row;
  for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += static_cast<uint64_t>(gridDim.x) * 256) {
    const uint64_t v = idx / bytes_per_row;
    const uint32_t byte = static_cast<uint32_t>(idx % bytes_per_row);
    uint32_t b = 0;
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t s = byte * 4 + k;
      if (s < founder_ct) {
        b |= synth_code(seed, first_variant + v, s, missing_threshold) << (2 * k);
      }
    }
    out[v * stride_bytes + byte] = static_cast<uint8_t>(b);
  }
}

}  // namespace

extern "C" int ldp_synth_genotypes(uint64_t seed, uint64_t first_variant, uint32_t n_variants, uint32_t founder_ct,
                                   double missing_rate, void* out, uint64_t stride_bytes, int location, void* stream) {
  if (!out || (stride_bytes < (static_cast<uint64_t>(founder_ct) + 3) / 4) || !(missing_rate >= 0.0) || !(missing_rate < 1.0)) {
    return LDP_ERR_INVALID;
  }
  const uint32_t missing_threshold = static_cast<uint32_t>(missing_rate * 4294967296.0);
  uint8_t* o = static_cast<uint8_t*>(out);
  if (location == LDP_MEM_HOST) {
    const uint32_t bytes_per_row = (founder_ct + 3) / 4;
    for (uint32_t v = 0; v < n_variants; ++v) {
      for (uint32_t byte = 0; byte < bytes_per_row; ++byte) {
        uint32_t b = 0;
        for (uint32_t k = 0; k < 4; ++k) {
          const uint32_t s = byte * 4 + k;
          if (s < founder_ct) {
            b |= synth_code(seed, first_variant + v, s, missing_threshold) << (2 * k);
          }
        }
        o[static_cast<uint64_t>(v) * stride_bytes + byte] = static_cast<uint8_t>(b);
      }
    }
    return LDP_OK;
  }
  if (location != LDP_MEM_DEVICE) {
    return LDP_ERR_INVALID;
  }
  if (!n_variants) {
    return LDP_OK;
  }
  hipLaunchKernelGGL(synth_kernel, dim3(256 * 32), dim3(256), 0, static_cast<hipStream_t>(stream), seed, first_variant, n_variants,
                     founder_ct, missing_threshold, o, stride_bytes);
  return (hipGetLastError() == hipSuccess) ? LDP_OK : LDP_ERR_GPU;
}
