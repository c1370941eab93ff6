// ldp_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the --indep-pairwise hot path.
//
//   prepare_kernel      2-bit genotype rows -> hom / ref2het bit-planes in HBM + per-variant aggregates
//                       (SplitHomRef2het pgenlib_misc.cc:1797-1885, FillVaggs plink2_ld.cc:725-738,
//                        allele counts / major allele plink2_data.cc:2421-2443, plink2_filter.cc:2113-2153,
//                        plink2_common.h:559-567, GenovecInvertUnsafe pgenlib_misc.cc:1090,
//                        PgrPlink1ToPlink2InplaceUnsafe pgenlib_read.cc:2157)
//   pair_tiles_kernel   banded all-pairs statistics + FP64 prune predicate
//                       (DotprodWords/SumSsqWords/SumSsqNmWords plink2_ld.cc:189-602,
//                        ComputeIndepPairwiseR2Components :699-723, decision :1085-1090)
//   pair_stats_ref      one wave per arbitrary pair, same integer statistics (parity/inspection)
//
// No MFMA: this is AND/XOR/popcount work.  v_bcnt_u32_b32 (popcount with accumulate) is the inner op;
// tiles are staged through LDS as 16-byte words and reused from registers 4 x NA times per thread.
#include "ldp_device.h"
#include "ldp_pair_device.h"

#include <algorithm>
#include <cstdlib>

namespace ldp {

// ================================================================================================
// prepare
// ================================================================================================
// Three of the four stages of the 32-bit perfect unshuffle (Hacker's Delight 7-2).  Afterwards each byte of x holds
// [even bits of the low half, odd bits of the low half, even bits of the high half, odd bits of the high half];
// the last stage (a byte swap) is folded into the v_perm_b32 that merges two input words into one plane dword.
__device__ __forceinline__ uint32_t unshuffle_to_bytes(uint32_t x) {
  uint32_t t;
  t = (x ^ (x >> 1)) & 0x22222222u;
  x = x ^ t ^ (t << 1);
  t = (x ^ (x >> 2)) & 0x0c0c0c0cu;
  x = x ^ t ^ (t << 2);
  t = (x ^ (x >> 4)) & 0x00f000f0u;
  x = x ^ t ^ (t << 4);
  return x;
}

// 32 samples (two input dwords, 2 bits per sample) -> one dword of each plane: ~30 VALU ops instead of the ~60 of
// four separate even-bit gathers (matters once the conversion shares the CUs with the pair kernel).
// .pgen codes: hom = !lo, ref2het = !hi.  .bed code b1b0 -> pgen lo = b0^b1, hi = !b1: hom = !(b0^b1), ref2het = b1.
__device__ __forceinline__ void planes_of_32(uint32_t w0, uint32_t w1, int encoding, uint32_t* hom, uint32_t* r2h) {
  uint32_t x0, x1;
  if (encoding == LDP_GENO_BED) {
    x0 = ((~(w0 ^ (w0 >> 1))) & 0x55555555u) | (w0 & 0xaaaaaaaau);
    x1 = ((~(w1 ^ (w1 >> 1))) & 0x55555555u) | (w1 & 0xaaaaaaaau);
  } else {
    x0 = ~w0;
    x1 = ~w1;
  }
  x0 = unshuffle_to_bytes(x0);
  x1 = unshuffle_to_bytes(x1);
  *hom = __builtin_amdgcn_perm(x1, x0, 0x06040200u);
  *r2h = __builtin_amdgcn_perm(x1, x0, 0x07050301u);
}

__device__ __forceinline__ uint32_t load_geno_dword(const uint8_t* row, uint32_t nbytes, uint32_t didx, bool aligned4) {
  const uint32_t off = didx * 4;
  if (aligned4 && (off + 4 <= nbytes)) {
    return *reinterpret_cast<const uint32_t*>(row + off);
  }
  uint32_t w = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (off + k < nbytes) {
      w |= static_cast<uint32_t>(row[off + k]) << (8 * k);
    }
  }
  return w;
}

// One plane dword (32 samples) from two input dwords (16 samples each).
__device__ __forceinline__ void convert_plane_dword(const uint8_t* row, uint32_t nbytes, bool aligned4, int encoding,
                                                    uint32_t founder_ct, uint32_t p, uint32_t* hom_out, uint32_t* r2h_out) {
  const uint32_t first_sample = p * 32;
  if (first_sample >= founder_ct) {
    *hom_out = 0;
    *r2h_out = 0;
    return;
  }
  const uint32_t w0 = load_geno_dword(row, nbytes, 2 * p, aligned4);
  const uint32_t w1 = load_geno_dword(row, nbytes, 2 * p + 1, aligned4);
  uint32_t hom, r2h;
  planes_of_32(w0, w1, encoding, &hom, &r2h);
  const uint32_t remaining = founder_ct - first_sample;
  if (remaining < 32) {
    const uint32_t mask = (1u << remaining) - 1;
    hom &= mask;
    r2h &= mask;
  }
  *hom_out = hom;
  *r2h_out = r2h;
}


// Two adjacent plane dwords (64 samples) from four input dwords; one 16-byte load when the row allows it.
__device__ __forceinline__ void planes_from_words(uint32_t w0, uint32_t w1, int encoding, uint32_t founder_ct, uint32_t p, uint32_t* hom_out, uint32_t* r2h_out) {
  uint32_t hom, r2h;
  planes_of_32(w0, w1, encoding, &hom, &r2h);
  const uint32_t first_sample = p * 32;
  if (first_sample >= founder_ct) {
    hom = 0;
    r2h = 0;
  } else if (founder_ct - first_sample < 32) {
    const uint32_t mask = (1u << (founder_ct - first_sample)) - 1;
    hom &= mask;
    r2h &= mask;
  }
  *hom_out = hom;
  *r2h_out = r2h;
}

// --indep-pairphase input (LDP_GENO_PHASED, include/ldprune_hip.h): 16 samples (one dword of 2-bit codes + 16 phaseinfo
// bits) -> 32 haplotypes = one dword of each plane.  Haplotype h (1 = carries the counted allele) is stored as the
// genotype code 2h: hom plane = non-missing, ref2het plane = non-missing and h == 0, x = 1 - 2h.  Then, against the
// reference's nm / sum / dot of plink2_ld.cc:1456-1481:  N*dot_x - S1*S2 = 4*(nm*dot - sum1*sum2),
// N*ssq - S^2 = 4*sum*(nm - sum), so cov12^2 > thr*var1*var2 compares 16x both sides of :1713 -- the same decision,
// power-of-two scalings being exact.  Which of a sample's two haplotypes gets the het's counted allele follows
// HapsplitMustPhased (pgenlib_misc.cc:1917: het + phaseinfo -> the second one).
__device__ __forceinline__ void hap_planes_of_16(uint32_t w, uint32_t phase16, uint32_t* hom, uint32_t* r2h) {
  const uint32_t lo = w & 0x55555555u;
  const uint32_t hi = (w >> 1) & 0x55555555u;
  const uint32_t miss = lo & hi;
  const uint32_t het = lo & ~hi;
  const uint32_t two = hi & ~lo;
  uint32_t ph = phase16 & 0xffffu;  // bit s -> bit 2s
  ph = (ph | (ph << 8)) & 0x00ff00ffu;
  ph = (ph | (ph << 4)) & 0x0f0f0f0fu;
  ph = (ph | (ph << 2)) & 0x33333333u;
  ph = (ph | (ph << 1)) & 0x55555555u;
  const uint32_t first = two | (het & ~ph);
  const uint32_t second = two | (het & ph);
  const uint32_t carries = first | (second << 1);
  const uint32_t nm = ~(miss | (miss << 1));
  *hom = nm;
  *r2h = nm & ~carries;
}

__device__ __forceinline__ void convert_plane_pair(const uint8_t* row, uint32_t nbytes, bool aligned4, bool aligned16, int encoding,
                                                   uint32_t founder_ct, uint32_t p, uint32_t (&hom)[2], uint32_t (&r2h)[2]) {
  if (encoding & LDP_GENO_PHASED) {
    // plane dwords p, p+1 = haplotypes [32p, 32p+64) = samples [16p, 16p+32): code bytes [4p, 4p+8), phase bytes [2p, 2p+4)
    const uint32_t samples = founder_ct >> 1;
    const uint32_t code_bytes = (samples + 3) >> 2;
    const uint32_t phase_off = (code_bytes + 3) & ~3u;
    const uint32_t phase_bytes = (samples + 7) >> 3;
    const uint32_t ph = load_geno_dword(row + phase_off, phase_bytes, p >> 1, aligned4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t w = load_geno_dword(row, code_bytes, p + k, aligned4);
      uint32_t h, r;
      hap_planes_of_16(w, ph >> (16 * k), &h, &r);
      const uint32_t first_hap = (p + k) * 32;
      if (first_hap >= founder_ct) {
        h = 0;
        r = 0;
      } else if (founder_ct - first_hap < 32) {
        const uint32_t mask = (1u << (founder_ct - first_hap)) - 1;
        h &= mask;
        r &= mask;
      }
      hom[k] = h;
      r2h[k] = r;
    }
    return;
  }
  // p is even; input bytes [8p, 8p+16)
  // global_load_dwordx4 only needs dword alignment on gfx950, which is all a packed row guarantees
  typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
  (void)aligned16;
  if (aligned4 && (8u * p + 16u <= nbytes)) {
    // streamed once: non-temporal, so the rows do not displace what the pair kernel keeps in L2
    const u32x4_a4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(row + 8u * p));
    planes_from_words(w.x, w.y, encoding, founder_ct, p, &hom[0], &r2h[0]);
    planes_from_words(w.z, w.w, encoding, founder_ct, p + 1, &hom[1], &r2h[1]);
  } else {
    convert_plane_dword(row, nbytes, aligned4, encoding, founder_ct, p, &hom[0], &r2h[0]);
    convert_plane_dword(row, nbytes, aligned4, encoding, founder_ct, p + 1, &hom[1], &r2h[1]);
  }
}

__device__ __forceinline__ void store_plane_pair(uint32_t* dst, uint2 v) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 t;
  t.x = v.x;
  t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<u32x2*>(dst));
}

// One block per variant.  Pass 1 converts the row to bit-planes held in registers (MAXIT pairs of plane dwords
// per thread, 16-byte loads) while counting; after the block-wide reduction decides the major allele, pass 2
// writes the planes (with ref2het ^= hom when ALT is major, 8-byte stores) without touching the input again.
// MAXIT == 0: rows too long for the register budget are simply converted twice (second read comes from L2).
template <int THREADS, int MAXIT>
__global__ __launch_bounds__(THREADS) void prepare_kernel(PrepareArgs A) {
  constexpr int kWaves = THREADS / 64;
  __shared__ uint32_t red[kWaves][3 + 2 * kCheckpoints];
  __shared__ uint32_t s_alt_major;
  __shared__ int32_t s_sum;
  const uint32_t v = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint8_t* row = A.geno + static_cast<uint64_t>(v) * A.stride_bytes;
  const uint32_t nbytes = (A.founder_ct + 3) / 4;
  const bool aligned4 = ((reinterpret_cast<uintptr_t>(row) & 3) == 0);
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  const uint32_t plane_dwords = A.chunks * kChunkDwords;  // even

  uint32_t keep_hom[MAXIT ? MAXIT : 1][2], keep_r2h[MAXIT ? MAXIT : 1][2];
  uint32_t hom_ct = 0, r2h_ct = 0, both_ct = 0;
  // hom calls / code-0 calls in k-chunks >= checkpoint k (suffix counts for early termination)
  uint32_t rest[kCheckpoints], rest_both[kCheckpoints];
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    rest[k] = 0;
    rest_both[k] = 0;
  }
  if constexpr (MAXIT > 0) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const uint32_t p = 2 * (tid + it * THREADS);
      keep_hom[it][0] = keep_hom[it][1] = 0;
      keep_r2h[it][0] = keep_r2h[it][1] = 0;
      if (p < plane_dwords) {
        convert_plane_pair(row, nbytes, aligned4, aligned16, A.encoding, A.founder_ct, p, keep_hom[it], keep_r2h[it]);
      }
      const uint32_t hc = __popc(keep_hom[it][0]) + __popc(keep_hom[it][1]);
      hom_ct += hc;
      r2h_ct += __popc(keep_r2h[it][0]) + __popc(keep_r2h[it][1]);
      const uint32_t bc = __popc(keep_hom[it][0] & keep_r2h[it][0]) + __popc(keep_hom[it][1] & keep_r2h[it][1]);
      both_ct += bc;
      const uint32_t chunk = p / kChunkDwords;
      const uint32_t packed = hc | (bc << 16);  // per-thread totals stay below 2^16 here (<= 16 x 64 samples)
#pragma unroll
      for (int k = 0; k < kCheckpoints; ++k) {
        rest[k] += (chunk >= A.checkpoint_chunk[k]) ? packed : 0;
      }
    }
  } else {
    for (uint32_t p = 2 * tid; p < plane_dwords; p += 2 * THREADS) {
      uint32_t hom[2], r2h[2];
      convert_plane_pair(row, nbytes, aligned4, aligned16, A.encoding, A.founder_ct, p, hom, r2h);
      const uint32_t hc = __popc(hom[0]) + __popc(hom[1]);
      hom_ct += hc;
      r2h_ct += __popc(r2h[0]) + __popc(r2h[1]);
      const uint32_t bc = __popc(hom[0] & r2h[0]) + __popc(hom[1] & r2h[1]);
      both_ct += bc;
      const uint32_t chunk = p / kChunkDwords;
#pragma unroll
      for (int k = 0; k < kCheckpoints; ++k) {
        rest[k] += (chunk >= A.checkpoint_chunk[k]) ? hc : 0;
        rest_both[k] += (chunk >= A.checkpoint_chunk[k]) ? bc : 0;
      }
    }
  }
  hom_ct = wave_reduce_add(hom_ct);
  r2h_ct = wave_reduce_add(r2h_ct);
  both_ct = wave_reduce_add(both_ct);
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    if constexpr (MAXIT > 0) {
      rest_both[k] = rest[k] >> 16;  // (accumulated packed in the register-resident pass)
      rest[k] &= 0xffff;
    }
    rest[k] = wave_reduce_add(rest[k]);
    rest_both[k] = wave_reduce_add(rest_both[k]);
  }
  if ((tid & 63) == 0) {
    red[tid >> 6][0] = hom_ct;
    red[tid >> 6][1] = r2h_ct;
    red[tid >> 6][2] = both_ct;
#pragma unroll
    for (int k = 0; k < kCheckpoints; ++k) {
      red[tid >> 6][3 + k] = rest[k];
      red[tid >> 6][3 + kCheckpoints + k] = rest_both[k];
    }
  }
  __syncthreads();
  if (tid == 0) {
    hom_ct = 0;
    r2h_ct = 0;
    both_ct = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      hom_ct += red[w][0];
      r2h_ct += red[w][1];
      both_ct += red[w][2];
    }
    // raw genotype counts: code 0 = hom&r2h, code 1 = r2h only, code 2 = hom only
    const uint32_t n0 = both_ct;
    const uint32_t n1 = r2h_ct - both_ct;
    const uint32_t n2 = hom_ct - both_ct;
    // (sample-mapped rows: the het calls that were made missing still count as one allele each)
    const uint32_t n1_alleles = n1 + (A.extra_het ? A.extra_het[v] : 0u);
    uint32_t alt_major = 0;
    ldp_variant_rec rec;
    rec.n_homref = 0;
    rec.n_het = 0;
    rec.n_homalt = 0;
    rec.reserved = 0;
    if (((A.encoding & 3) != LDP_GENO_INVERSE) && !(A.row_inverse && A.row_inverse[v])) {
      // plink2_filter.cc:2137-2147: freq = ref * (1 / tot), 1/2 when nothing is observed;
      // major = REF iff freq >= 0.5 (plink2_common.h:559-567)
      const uint64_t ref_ct = 2ull * n0 + n1_alleles;
      const uint64_t alt_ct = 2ull * n2 + n1_alleles;
      const uint64_t tot = ref_ct + alt_ct;
      double ref_freq = 0.5;
      if (tot) {
        const double tot_recip = __ddiv_rn(1.0, static_cast<double>(tot));
        ref_freq = __dmul_rn(static_cast<double>(ref_ct), tot_recip);
      }
      alt_major = !(ref_freq >= 0.5);
      rec.n_homref = n0;
      rec.n_het = n1_alleles;
      rec.n_homalt = n2;
    }
    // after the (optional) 0<->2 inversion
    const uint32_t plus_ct = alt_major ? n2 : n0;
    const uint32_t minus_ct = alt_major ? n0 : n2;
    const uint32_t nm_ct = plus_ct + minus_ct + n1;
    rec.nm_ct = nm_ct;
    rec.sum = static_cast<int32_t>(plus_ct - minus_ct);
    rec.ssq = hom_ct;
    const uint32_t mono = ((!plus_ct) && (!minus_ct)) || (plus_ct == nm_ct) || (minus_ct == nm_ct);  // plink2_ld.cc:902
    rec.flags = alt_major | (mono << 1) | ((nm_ct != A.founder_ct) ? 4u : 0u);
    A.recs[v] = rec;
    s_alt_major = alt_major;
    s_sum = rec.sum;
  }
  __syncthreads();
  const uint32_t alt_major = s_alt_major;
  if (A.cp_stats && (tid < kCpSlots)) {
    // early-termination statistics (layout: ldp_device.h); only ever read for complete-data rows
    const double N = static_cast<double>(A.founder_ct);
    cp_slot slot;
    if (tid < kCheckpoints) {
      uint32_t hom_r = 0, both_r = 0;
      for (int w = 0; w < kWaves; ++w) {
        hom_r += red[w][3 + tid];
        both_r += red[w][3 + kCheckpoints + tid];
      }
      const uint32_t plus_r = alt_major ? (hom_r - both_r) : both_r;
      {
        // the interval bound's numbers (ldp_device.h): calls, sum z, sum z^2 over the remainder, z = 1 - x.
        // Het calls = remainder size - hom calls: right for a complete row; cp_gen_fix_kernel redoes the rows
        // with missing calls (keeping their het suffix counts here costs registers, hence occupancy, for every row).
        const uint64_t seen_k = static_cast<uint64_t>(A.checkpoint_chunk[tid]) * (kChunkDwords * 32);
        const uint32_t rs_k = (seen_k < A.founder_ct) ? static_cast<uint32_t>(A.founder_ct - seen_k) : 0;
        const uint32_t het_r = (rs_k >= hom_r) ? (rs_k - hom_r) : 0;
        const uint32_t minus_r = hom_r - plus_r;
        cp_gen_slot gs;
        gs.nm_r = hom_r + het_r;
        gs.zs_r = het_r + 2 * minus_r;
        gs.zq_r = het_r + 4 * minus_r;
        gs.pad = 0;
        A.cp_gen[static_cast<uint64_t>(v) * kCheckpoints + tid] = gs;
      }
      const double s_r = static_cast<double>(static_cast<int32_t>(2 * plus_r - hom_r));
      const uint64_t seen = static_cast<uint64_t>(A.checkpoint_chunk[tid]) * (kChunkDwords * 32);
      const double n_r = (seen < A.founder_ct) ? static_cast<double>(A.founder_ct - seen) : 1.0;
      const double v_r = fmax(static_cast<double>(hom_r) - s_r * s_r / n_r, 0.0);
      slot.a = s_r * sqrt(N / n_r);
      slot.b = sqrt(N * v_r);
    } else {
      uint32_t hom_all = 0;
      for (int w = 0; w < kWaves; ++w) {
        hom_all += red[w][0];
      }
      const double S = static_cast<double>(s_sum);
      slot.a = S;
      slot.b = sqrt(fmax(N * static_cast<double>(hom_all) - S * S, 0.0)) * A.cp_tv_scale;
    }
    A.cp_stats[static_cast<uint64_t>(v) * kCpStride + tid] = slot;
  }
  uint32_t* out_row = A.planes + static_cast<uint64_t>(v) * A.row_dwords;
  if constexpr (MAXIT > 0) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const uint32_t p = 2 * (tid + it * THREADS);
      if (p < plane_dwords) {
        const uint32_t off = (p / kChunkDwords) * kRowChunkDwords + (p % kChunkDwords);
        const uint2 hv = make_uint2(keep_hom[it][0], keep_hom[it][1]);
        // 0 <-> 2 flips ref2het on homozygous calls
        const uint2 rv = alt_major ? make_uint2(keep_r2h[it][0] ^ hv.x, keep_r2h[it][1] ^ hv.y) : make_uint2(keep_r2h[it][0], keep_r2h[it][1]);
        store_plane_pair(out_row + off, hv);
        store_plane_pair(out_row + off + kChunkDwords, rv);
      }
    }
  } else {
    for (uint32_t p = 2 * tid; p < plane_dwords; p += 2 * THREADS) {
      uint32_t hom[2], r2h[2];
      convert_plane_pair(row, nbytes, aligned4, aligned16, A.encoding, A.founder_ct, p, hom, r2h);
      if (alt_major) {
        r2h[0] ^= hom[0];
        r2h[1] ^= hom[1];
      }
      const uint32_t off = (p / kChunkDwords) * kRowChunkDwords + (p % kChunkDwords);
      store_plane_pair(out_row + off, make_uint2(hom[0], hom[1]));
      store_plane_pair(out_row + off + kChunkDwords, make_uint2(r2h[0], r2h[1]));
    }
  }
}

// One wave per variant with missing calls: het calls of the remainders from the written planes (het = ref2het and
// not hom, which the 0 <-> 2 inversion leaves alone), then the row's cp_gen slots again.  Complete rows leave at once.
__global__ __launch_bounds__(256) void cp_gen_fix_kernel(PrepareArgs A) {
  const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (v >= A.n_variants) {
    return;
  }
  const ldp_variant_rec rec = A.recs[v];
  if (!(rec.flags & 4u)) {
    return;
  }
  const uint32_t* row = A.planes + static_cast<uint64_t>(v) * A.row_dwords;
  uint32_t het[kCheckpoints], hom[kCheckpoints], minus[kCheckpoints];
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    het[k] = 0;
    hom[k] = 0;
    minus[k] = 0;
  }
  // four k-chunks per step: lane -> (chunk step + lane / 16, plane dword lane % 16)
  for (uint32_t c0 = 0; c0 < A.chunks; c0 += 64 / kChunkDwords) {
    const uint32_t chunk = c0 + lane / kChunkDwords;
    uint32_t h = 0, r = 0;
    if (chunk < A.chunks) {
      h = row[static_cast<uint64_t>(chunk) * kRowChunkDwords + (lane % kChunkDwords)];
      r = row[static_cast<uint64_t>(chunk) * kRowChunkDwords + kChunkDwords + (lane % kChunkDwords)];
    }
    const uint32_t hc = __popc(h), tc = __popc(r & ~h), mc = __popc(h & ~r);  // hom, het, homozygous minor (x = -1)
#pragma unroll
    for (int k = 0; k < kCheckpoints; ++k) {
      const bool in = (chunk >= A.checkpoint_chunk[k]);
      hom[k] += in ? hc : 0;
      het[k] += in ? tc : 0;
      minus[k] += in ? mc : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    hom[k] = wave_reduce_add(hom[k]);
    het[k] = wave_reduce_add(het[k]);
    minus[k] = wave_reduce_add(minus[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < kCheckpoints; ++k) {
      if (static_cast<uint32_t>(k) < A.n_checkpoints) {
        cp_gen_slot gs;
        gs.nm_r = hom[k] + het[k];
        gs.zs_r = het[k] + 2 * minus[k];
        gs.zq_r = het[k] + 4 * minus[k];
        gs.pad = 0;
        A.cp_gen[static_cast<uint64_t>(v) * kCheckpoints + k] = gs;
      }
    }
  }
}

// Sample-mapped rows (ldp_set_sample_map): one block per variant gathers the engine's columns from the file's row.
// The input row (<= a few hundred KB) is read through L2 by every thread of its block; the map is shared by all blocks.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ in, uint64_t in_stride, int in_is_bed, const uint32_t* __restrict__ map,
                                                          uint32_t out_ct, uint8_t* __restrict__ out, uint64_t out_stride, uint32_t* __restrict__ extra_het) {
  __shared__ uint32_t red[4];
  const uint32_t v = blockIdx.x;
  const uint8_t* row = in + static_cast<uint64_t>(v) * in_stride;
  uint32_t* orow = reinterpret_cast<uint32_t*>(out + static_cast<uint64_t>(v) * out_stride);
  const uint32_t out_dwords = static_cast<uint32_t>(out_stride / 4);
  uint32_t made_missing = 0;
  for (uint32_t d = threadIdx.x; d < out_dwords; d += 256) {
    uint32_t w = 0;
    // sixteen columns that are neighbours in the file too (a founder subset with few gaps, a block of one sex): one
    // unaligned 32-bit window of the input row instead of sixteen byte loads
    if (d * 16 + 15 < out_ct) {
      const uint32_t m0 = map[d * 16], m15 = map[d * 16 + 15];
      if (((m15 & 0x7fffffffu) - (m0 & 0x7fffffffu) == 15) && ((m0 >> 31) == (m15 >> 31))) {
        bool run = true;
#pragma unroll
        for (uint32_t k = 1; k < 15; ++k) {
          run = run && (map[d * 16 + k] == m0 + k);
        }
        if (run) {
          const uint32_t src = m0 & 0x7fffffffu;
          const uint8_t* at = row + (src >> 2);
          // (the row is read bytewise: no alignment or over-read assumptions about the caller's stride)
          uint64_t bits = 0;
#pragma unroll
          for (uint32_t b = 0; b < 5; ++b) {
            bits |= static_cast<uint64_t>(at[(b < 4 || (src & 3)) ? b : 0]) << (8 * b);
          }
          w = static_cast<uint32_t>(bits >> (2 * (src & 3)));
          if (in_is_bed) {
            // .bed -> .pgen codes, sixteen at a time: 00 -> 10, 01 -> 11, 10 -> 01, 11 -> 00
            const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
            w = ((hi ^ 0x55555555u) << 1) | (lo ^ hi);
          }
          if (m0 >> 31) {
            const uint32_t hets = (w & 0x55555555u) & ~((w >> 1) & 0x55555555u);
            w |= hets * 3u;
            made_missing += __popc(hets);
          }
          orow[d] = w;
          continue;
        }
      }
    }
#pragma unroll 4
    for (uint32_t k = 0; k < 16; ++k) {
      const uint32_t f = d * 16 + k;
      if (f < out_ct) {
        const uint32_t m = map[f];
        const uint32_t src = m & 0x7fffffffu;
        uint32_t c = (static_cast<uint32_t>(row[src >> 2]) >> (2 * (src & 3))) & 3u;
        if (in_is_bed) {
          c = (0x1eu >> (2 * c)) & 3u;  // .bed 0,1,2,3 -> .pgen 2,3,1,0 (pgenlib_read.cc:2157)
        }
        if ((m >> 31) && (c == 1)) {
          c = 3;
          ++made_missing;
        }
        w |= c << (2 * k);
      }
    }
    orow[d] = w;
  }
  made_missing = wave_reduce_add(made_missing);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = made_missing;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    extra_het[v] = red[0] + red[1] + red[2] + red[3];
  }
}

hipError_t launch_gather_rows(const uint8_t* in, uint64_t in_stride, uint32_t n_variants, int in_is_bed, const uint32_t* map, uint32_t out_ct, uint8_t* out,
                              uint64_t out_stride, uint32_t* extra_het, hipStream_t stream) {
  if (!n_variants) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_variants), dim3(256), 0, stream, in, in_stride, in_is_bed, map, out_ct, out, out_stride, extra_het);
  return hipGetLastError();
}

// What the rows of one conversion launch add to MissStats (ldp_device.h).  A kernel of its own behind prepare_kernel: one
// atomic per row from inside it costs 16 ns each -- they serialise at the memory side -- and quadrupled the conversion.
__global__ __launch_bounds__(256) void miss_stats_kernel(const ldp_variant_rec* __restrict__ recs, uint32_t n, uint32_t founder_ct, uint32_t miss_high,
                                                         MissStats* __restrict__ stats) {
  unsigned long long total = 0;
  uint32_t high = 0, mx = 0;
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
    const uint32_t miss = founder_ct - recs[v].nm_ct;
    total += miss;
    high += (miss > miss_high) ? 1u : 0u;
    mx = max(mx, miss);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    total += __shfl_down(total, off, 64);
    high += __shfl_down(high, off, 64);
    mx = max(mx, static_cast<uint32_t>(__shfl_down(mx, off, 64)));
  }
  __shared__ unsigned long long s_total[4];
  __shared__ uint32_t s_high[4], s_max[4];
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_total[wave] = total;
    s_high[wave] = high;
    s_max[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    total = s_total[0] + s_total[1] + s_total[2] + s_total[3];
    high = s_high[0] + s_high[1] + s_high[2] + s_high[3];
    mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    if (mx) {
      atomicMax(&stats->max_missing, mx);
      atomicAdd(&stats->total[blockIdx.x % kMissStripes], total);
      if (high) {
        atomicAdd(&stats->high_rows[blockIdx.x % kMissStripes], static_cast<unsigned long long>(high));
      }
    }
  }
}

// One wave: which kernel takes the launches queued behind this point (ldp_device.h: kRoute*).
__global__ __launch_bounds__(64) void route_kernel(const MissStats* __restrict__ stats, unsigned long long total_limit, unsigned long long high_limit,
                                                   int allow_sparse, uint32_t* __restrict__ route_out) {
  const uint32_t lane = threadIdx.x;
  unsigned long long total = stats->total[lane], high = stats->high_rows[lane];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    total += __shfl_down(total, off, 64);
    high += __shfl_down(high, off, 64);
  }
  if (lane == 0) {
    uint32_t route = kRouteComplete;
    if (stats->max_missing) {
      route = (allow_sparse && (total <= total_limit) && (high <= high_limit)) ? kRouteSparse : kRouteGeneral;
    }
    *route_out = route;
  }
}

hipError_t launch_miss_stats(const ldp_variant_rec* recs, uint32_t n, uint32_t founder_ct, uint32_t miss_high, MissStats* stats, hipStream_t stream) {
  if (!n) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(miss_stats_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 256)), dim3(256), 0, stream, recs, n, founder_ct, miss_high, stats);
  return hipGetLastError();
}

hipError_t launch_route(const MissStats* stats, unsigned long long total_limit, unsigned long long high_limit, int allow_sparse, uint32_t* route_out,
                        hipStream_t stream) {
  static_assert(kMissStripes == 64, "one lane per stripe");
  hipLaunchKernelGGL(route_kernel, dim3(1), dim3(64), 0, stream, stats, total_limit, high_limit, allow_sparse, route_out);
  return hipGetLastError();
}

hipError_t launch_prepare(const PrepareArgs& a, hipStream_t stream) {
  if (!a.n_variants) {
    return hipSuccess;
  }
  const uint32_t pairs = (a.chunks * kChunkDwords + 1) / 2;  // pairs of plane dwords per row
  // Few threads with many 16-byte loads in flight each beat many threads with few (config 2: 128 x 7 runs at
  // 4.75 ms, 256 x 4 at 5.0, 512 x 2 at 7.1, one wave x 13 at 5.6): up to 8 register-resident iterations per thread.
#define LDP_PREP(T, I) hipLaunchKernelGGL((prepare_kernel<T, I>), dim3(a.n_variants), dim3(T), 0, stream, a)
  if (pairs <= 128 * 2) {
    LDP_PREP(128, 2);
  } else if (pairs <= 128 * 4) {
    LDP_PREP(128, 4);
  } else if (pairs <= 128 * 6) {
    LDP_PREP(128, 6);
  } else if (pairs <= 128 * 8) {
    LDP_PREP(128, 8);
  } else if (pairs <= 128 * 16) {  // longer rows: 16 iterations (N = 500,000: 512 x 16 runs at 5.75 ms, 1024 x 8 at 6.6)
    LDP_PREP(128, 16);
  } else if (pairs <= 256 * 16) {
    LDP_PREP(256, 16);
  } else if (pairs <= 512 * 16) {
    LDP_PREP(512, 16);
  } else if (pairs <= 1024 * 16) {
    LDP_PREP(1024, 16);
  } else {
    LDP_PREP(1024, 0);
  }
#undef LDP_PREP
  if (a.miss_stats) {
    hipLaunchKernelGGL(miss_stats_kernel, dim3(std::min<uint32_t>((a.n_variants + 255) / 256, 256)), dim3(256), 0, stream, a.recs, a.n_variants, a.founder_ct,
                       a.miss_high, a.miss_stats);
  }
  if (a.cp_stats && a.n_checkpoints && a.fix_cp_gen) {
    hipLaunchKernelGGL(cp_gen_fix_kernel, dim3((a.n_variants + 3) / 4), dim3(256), 0, stream, a);
  }
  return hipGetLastError();
}

// ================================================================================================
// pair tiles
// ================================================================================================
constexpr int kLdsRowSlots = kLdsRowDwords / 4;  // 9 16-byte slots per LDS row (8 data + 1 pad)

__device__ __forceinline__ uint32_t popc4(const uint4& v) { return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
__device__ __forceinline__ uint4 and4(const uint4& a, const uint4& b) { return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w); }
__device__ __forceinline__ uint4 or4(const uint4& a, const uint4& b) { return make_uint4(a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w); }
__device__ __forceinline__ uint4 xor4(const uint4& a, const uint4& b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }

// acc += popcount(x) as ONE v_bcnt_u32_b32 (popcount with accumulate).  Written as inline asm because the
// optimizer otherwise reassociates chains of ctpop+add into ctpop(.,0) trees plus v_add3.
__device__ __forceinline__ void bcnt_acc(uint32_t& acc, uint32_t x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}
__device__ __forceinline__ void bcnt_acc4(uint32_t& acc, const uint4& v) {
  bcnt_acc(acc, v.x);
  bcnt_acc(acc, v.y);
  bcnt_acc(acc, v.z);
  bcnt_acc(acc, v.w);
}

__device__ __forceinline__ uint4 lds_row_slot(const uint4* __restrict__ l4, int row, int slot) { return l4[row * kLdsRowSlots + slot]; }

// ---- the inner loop ---------------------------------------------------------------------------------------
// Column layout: wave w of a block owns second-variant group b = w (j = j0 + tx + 8w) of ALL the block's distance
// units a (d = d0 + ty + 8a), one pair per lane and unit, NA <= kMaxUnitsPerBlock units.  Whatever the unit count
// the four waves carry the same load, a wave's accumulators are few (2 or 7 per unit), and early termination only
// ever shortens the unit list from the far end.  Per 16-byte k-group a lane reads its J row once and streams the NA
// I rows (row of unit a = irow - 8a), the next one in flight while the current one is consumed.
//
// Complete data (2 counts): dot = popcnt(hom1&hom2) - 2*popcnt(hom1&hom2&(r2h1^r2h2))          (plink2_ld.cc:244-250)
// Missing calls (7 counts, plink2_ld.cc:244-250, :329-334, :590-601), nm = hom | r2h:
//   [0] popcnt(hom_i & hom_j)              [1] popcnt(hom_i & hom_j & (r2h_i ^ r2h_j))
//   [2] popcnt(nm_i & nm_j)                [3] popcnt(nm_i & hom_j)   [4] popcnt(nm_i & hom_j & r2h_j)
//   [5] popcnt(nm_j & hom_i)               [6] popcnt(nm_j & hom_i & r2h_i)
template <bool GENERAL>
struct Counts {
  static constexpr int n = GENERAL ? 7 : 2;
};

template <bool GENERAL, int NA>
__device__ __forceinline__ void tile_chunk(const uint4* __restrict__ l4, int jrow, int irow, uint32_t (&acc)[kMaxUnitsPerBlock][Counts<GENERAL>::n]) {
#pragma unroll 1
  for (int g = 0; g < kChunkDwords / 4; ++g) {
    const uint4 jH = lds_row_slot(l4, jrow, g);
    const uint4 jR = lds_row_slot(l4, jrow, (kChunkDwords / 4) + g);
    uint4 iH = lds_row_slot(l4, irow, g);
    uint4 iR = lds_row_slot(l4, irow, (kChunkDwords / 4) + g);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      uint4 nH = iH, nR = iR;
      if (a + 1 < NA) {
        nH = lds_row_slot(l4, irow - 8 * (a + 1), g);
        nR = lds_row_slot(l4, irow - 8 * (a + 1), (kChunkDwords / 4) + g);
      }
      const uint4 h = and4(jH, iH);
      bcnt_acc4(acc[a][0], h);
      bcnt_acc4(acc[a][1], and4(h, xor4(jR, iR)));
      if constexpr (GENERAL) {
        const uint4 jN = or4(jH, jR);
        const uint4 jP = and4(jH, jR);
        const uint4 iN = or4(iH, iR);
        const uint4 iP = and4(iH, iR);
        bcnt_acc4(acc[a][2], and4(iN, jN));
        bcnt_acc4(acc[a][3], and4(iN, jH));
        bcnt_acc4(acc[a][4], and4(iN, jP));
        bcnt_acc4(acc[a][5], and4(jN, iH));
        bcnt_acc4(acc[a][6], and4(jN, iP));
      }
      __builtin_amdgcn_sched_barrier(0);
      iH = nH;
      iR = nR;
    }
  }
}

// LDS rows of a tile.  Row r holds variant vlo + r for the first-variant (I) rows r < jrow_base; the 32
// second-variant (J) rows follow at jrow_base.  When the block's distance range starts within 32 variants of the
// diagonal the two ranges touch or overlap and the J rows simply ARE the upper I rows (one copy, one DMA):
// a tile over distances 1..8U stages 8U+32 rows instead of 8U+63.
struct TileGeom {
  uint32_t j0, sfirst, send;
  int64_t vlo;    // variant of LDS row 0, before clamping to the subcontig
  int dmax;       // farthest distance staged
  int jrow_base;  // LDS row of variant j0
  int rtot;       // LDS rows in use
};

__device__ __forceinline__ int64_t row_variant(const TileGeom& G, int r) {
  int64_t v = (r < G.jrow_base) ? (G.vlo + r) : (static_cast<int64_t>(G.j0) + (r - G.jrow_base));
  if (v < static_cast<int64_t>(G.sfirst)) {
    v = G.sfirst;
  }
  if (v >= static_cast<int64_t>(G.send)) {
    v = static_cast<int64_t>(G.send) - 1;
  }
  return v;
}

// units_total = how many of the item's 8-distance units (nearest first) are staged
__device__ __forceinline__ TileGeom make_geom(const WorkItem& it, uint32_t units_total) {
  TileGeom G;
  G.j0 = it.j0;
  G.sfirst = it.sfirst;
  G.send = it.send;
  G.dmax = static_cast<int>(it.d0 + 8 * units_total - 1);
  const int n_irows = static_cast<int>(8 * units_total) + 31;
  G.jrow_base = (G.dmax < n_irows) ? G.dmax : n_irows;
  G.rtot = G.jrow_base + kTileJ;
  G.vlo = static_cast<int64_t>(it.j0) - G.dmax;
  return G;
}

// ---- global -> LDS staging by LDS-DMA (global_load_lds_dwordx4), a ring of 2..4 stages ------------------
// A stage is a linear array of 16-byte slots, kLdsRowSlots (9) per row: 8 data slots (4 hom + 4 ref2het) + 1 pad
// slot that keeps the row stride odd.  One DMA wave-instruction fills 64 consecutive slots (1 KiB, lane l ->
// slot 64*T + l); the per-lane GLOBAL address is free, so each lane simply fetches the 16 bytes that belong in its
// slot (pad slots re-fetch slot 0 of their row and are never read).  The lane->source mapping does not depend on
// the k-chunk: it is computed when the tile is planned and again whenever early termination shrinks the tile.
// Stage count S = how many stages of the current tile fit the block's LDS (>= 2): chunk kc lives in stage kc % S
// and up to S-1 chunks are in flight, so a tile that has shrunk to its near units hides the DMA latency that a
// double buffer cannot.
constexpr int kMaxTileRows = kTileJ + 8 * kMaxUnitsPerBlock + 31;
constexpr int kMaxDmaPerWave = ((kMaxTileRows * kLdsRowSlots + 63) / 64 + kWavesPerBlock - 1) / kWavesPerBlock;  // 6 at 12 units
constexpr uint32_t kMaxStages = 4;
constexpr uint32_t kChunkBytes = kRowChunkDwords * sizeof(uint32_t);

struct Stager {
  uint32_t* src_off;                 // [kMaxDmaPerWave][kBlockThreads] in LDS: byte offset of a lane's 16 B relative to tile_base
  const uint8_t* tile_base;          // chunk 0 of the lowest variant the tile touches
  uint32_t n_instr;                  // DMA wave-instructions per k-chunk for the whole block
  uint32_t mine;                     // ... of which this wave issues
  uint32_t stage_dwords;
  uint32_t stages;
};

// NOTE on registers: nothing that lives across the k-loop may be spilled.  A spill comes back through a scratch
// load, and the vmcnt wait in front of its first use drains the very DMA queue the ring keeps full (measured:
// 2x on the latency-bound tail of a tile).  Hence the per-lane source offsets live in LDS (6 KiB, read back with
// one ds_read per DMA instruction) instead of six registers.
__device__ __forceinline__ void plan_stager(Stager& st, const TileGeom& G, const PairKernelArgs& A, uint32_t wave, uint32_t lane) {
  const uint32_t row_bytes = static_cast<uint32_t>(A.row_dwords * sizeof(uint32_t));
  const int64_t vmin = row_variant(G, 0);
  const uint32_t total_slots = static_cast<uint32_t>(G.rtot) * kLdsRowSlots;
  st.n_instr = (total_slots + 63) / 64;
  st.mine = (st.n_instr > wave) ? (st.n_instr - wave + kWavesPerBlock - 1) / kWavesPerBlock : 0;
  st.stage_dwords = st.n_instr * 256;
  const uint32_t fit = A.lds_dwords / st.stage_dwords;
  st.stages = (fit < kMaxStages) ? fit : kMaxStages;
  st.tile_base = reinterpret_cast<const uint8_t*>(A.planes) + static_cast<uint64_t>(vmin) * row_bytes;
#pragma unroll
  for (int t = 0; t < kMaxDmaPerWave; ++t) {
    const uint32_t L = (wave + kWavesPerBlock * t) * 64 + lane;
    int row = static_cast<int>(L / kLdsRowSlots);
    uint32_t sl = L % kLdsRowSlots;
    if (row >= G.rtot) {
      row = G.rtot - 1;
    }
    if (sl == kLdsRowSlots - 1) {
      sl = 0;
    }
    st.src_off[t * kBlockThreads + wave * 64 + lane] = static_cast<uint32_t>(row_variant(G, row) - vmin) * row_bytes + sl * 16;
  }
}

// queue k-chunk kc into ring stage `stage`
__device__ __forceinline__ void dma_chunk(const Stager& st, uint32_t* lds, uint32_t kc, uint32_t stage, uint32_t wave, uint32_t lane) {
  const uint8_t* chunk_base = st.tile_base + static_cast<uint64_t>(kc) * kChunkBytes;
  uint32_t* dst = lds + stage * st.stage_dwords;
#pragma unroll
  for (int t = 0; t < kMaxDmaPerWave; ++t) {
    const uint32_t T = wave + kWavesPerBlock * t;
    if (T < st.n_instr) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(chunk_base + st.src_off[t * kBlockThreads + wave * 64 + lane]),
                                       (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
    }
  }
}


// Position in the ring: next chunk to consume / to queue and the stages they map to.
struct Ring {
  uint32_t kc, read_stage;
  uint32_t issued, issue_stage;
};

__device__ __forceinline__ void ring_start(Ring& R, const Stager& st, uint32_t* lds, uint32_t kc, uint32_t chunks, uint32_t wave, uint32_t lane) {
  R.kc = kc;
  R.read_stage = 0;
  R.issued = kc;
  R.issue_stage = 0;
  while ((R.issued < chunks) && (R.issued + 1 < kc + st.stages)) {
    dma_chunk(st, lds, R.issued, R.issue_stage, wave, lane);
    ++R.issued;
    R.issue_stage = (R.issue_stage + 1 == st.stages) ? 0 : R.issue_stage + 1;
  }
}

// chunk R.kc is ready in stage R.read_stage for every wave on return; the chunk that reuses the stage consumed in
// the previous iteration is queued (every wave is past the barrier, so nobody still reads it)
__device__ __forceinline__ const uint4* ring_acquire(Ring& R, const Stager& st, uint32_t* lds, uint32_t chunks, uint32_t wave, uint32_t lane) {
  wait_dma_then_barrier(st.mine * (R.issued - R.kc - 1));
  if (R.issued < chunks) {
    dma_chunk(st, lds, R.issued, R.issue_stage, wave, lane);
    ++R.issued;
    R.issue_stage = (R.issue_stage + 1 == st.stages) ? 0 : R.issue_stage + 1;
  }
  return reinterpret_cast<const uint4*>(lds + R.read_stage * st.stage_dwords);
}

__device__ __forceinline__ void ring_release(Ring& R, const Stager& st) {
  ++R.kc;
  R.read_stage = (R.read_stage + 1 == st.stages) ? 0 : R.read_stage + 1;
}


// The same question for tiles with missing calls (7 partial counts per pair).  The final statistics run over the
// pairwise-complete samples C = C_P + C_R; everything over the visited part C_P is known exactly, and C_R is what is
// left of each variant's own called samples in the remainder R after dropping the (unknown) ones where the partner
// is missing -- at most min(partner's missing calls in R, own calls in R) of them.  In z = 1 - x (all quantities
// >= 0; r^2 is invariant under the recoding):
//   n   in [n_P + max(nR_i - dmax_i, nR_j - dmax_j), n_P + min(nR_i, nR_j)]
//   Zs  in [Zs_P + a_i - min(2 dmax_i, a_i), Zs_P + a_i]       Zq >= Zq_P + b_i - min(4 dmax_i, b_i)     (same for w)
//   ZW  in ZW_P + a_i a_j / |R| +- sqrt((b_i - a_i^2/|R|)(b_j - a_j^2/|R|))   (Cauchy-Schwarz on R, missing = 0), ZW >= ZW_P
// and interval arithmetic gives |n ZW - Zs Ws| <= cmax, n Zq - Zs^2 >= var_lo.  The pair is hopeless when
// (cmax + 1)^2 (1 + 1e-6) < thresh var1_lo var2_lo (1 - 1e-6).  Worst case (a rare variant whose remaining carriers
// could all coincide with the partner's missing calls) the variance bound falls back to what the visited samples
// alone guarantee, so such pairs terminate later, never wrongly.
__device__ __forceinline__ bool pair_hopeless(const PairKernelArgs& A, const uint32_t (&c)[7], const cp_gen_slot& gi, const cp_gen_slot& gj, double rs) {
  // visited part, exact
  const double n_p = c[2];
  const double s1 = 2.0 * c[6] - static_cast<double>(c[5]), q1 = c[5];
  const double s2 = 2.0 * c[4] - static_cast<double>(c[3]), q2 = c[3];
  const double dot = static_cast<double>(c[0]) - 2.0 * c[1];
  const double zs_p = n_p - s1, zq_p = n_p - 2.0 * s1 + q1;
  const double ws_p = n_p - s2, wq_p = n_p - 2.0 * s2 + q2;
  const double zw_p = n_p - s1 - s2 + dot;
  // remainder
  const double nr_i = gi.nm_r, a_i = gi.zs_r, b_i = gi.zq_r;
  const double nr_j = gj.nm_r, a_j = gj.zs_r, b_j = gj.zq_r;
  const double dmax_i = fmin(rs - nr_j, nr_i);
  const double dmax_j = fmin(rs - nr_i, nr_j);
  const double n_lo = n_p + fmax(nr_i - dmax_i, nr_j - dmax_j);
  const double n_hi = n_p + fmin(nr_i, nr_j);
  const double zs_hi = zs_p + a_i, zs_lo = zs_hi - fmin(2.0 * dmax_i, a_i);
  const double ws_hi = ws_p + a_j, ws_lo = ws_hi - fmin(2.0 * dmax_j, a_j);
  const double zq_lo = zq_p + b_i - fmin(4.0 * dmax_i, b_i);
  const double wq_lo = wq_p + b_j - fmin(4.0 * dmax_j, b_j);
  const double centre = a_i * a_j / rs;
  const double spread = sqrt(fmax(b_i - a_i * a_i / rs, 0.0) * fmax(b_j - a_j * a_j / rs, 0.0)) * (1.0 + 1e-9);
  const double zw_hi = zw_p + centre + spread;
  const double zw_lo = fmax(zw_p, zw_p + centre - spread);
  const double cov_hi = n_hi * zw_hi - zs_lo * ws_lo;
  const double cov_lo = n_lo * zw_lo - zs_hi * ws_hi;
  const double cmax = fmax(fabs(cov_hi), fabs(cov_lo)) + 1.0;
  const double var1_lo = n_lo * zq_lo - zs_hi * zs_hi;
  const double var2_lo = n_lo * wq_lo - ws_hi * ws_hi;
  return (var1_lo > 0.0) && (var2_lo > 0.0) && (cmax * cmax * (1.0 + 1e-6) < A.thresh * (1.0 - 1e-6) * var1_lo * var2_lo);
}

// How many of the wave's NA units (nearest first) must stay live after checkpoint cp: everything up to the farthest
// unit that still has a pair that could reach the threshold (LD decays with distance: the far end goes first).
template <bool GENERAL, int NA>
__device__ __forceinline__ uint32_t live_units(const PairKernelArgs& A, uint32_t j0, uint32_t jend, uint32_t d_first, int tx, int ty, uint32_t b,
                                               const uint32_t (&acc)[kMaxUnitsPerBlock][Counts<GENERAL>::n], uint32_t cp) {
  bool hopeless[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    hopeless[a] = true;
  }
  uint32_t j = j0 + tx + 8 * b;
  asm volatile("" : "+v"(j));  // (keeps the units' address arithmetic inside the checkpoint instead of hoisted into long-lived registers)
  if (j < jend) {
    const uint32_t span_j = j - A.lo[j];
    // four units' records per round trip to memory (one latency per group, few registers)
    if constexpr (GENERAL) {
      const uint64_t seen = static_cast<uint64_t>(A.checkpoint_chunk[cp]) * (kChunkDwords * 32);
      const double rs = static_cast<double>(A.founder_ct - static_cast<uint32_t>(seen));  // |R| (checkpoints lie inside the row)
      const cp_gen_slot gj = A.cp_gen[static_cast<uint64_t>(j) * kCheckpoints + cp];
#pragma unroll
      for (int a0 = 0; a0 < NA; a0 += 4) {
        cp_gen_slot gi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t d = d_first + ty + 8 * (a0 + q);
          gi[q] = A.cp_gen[static_cast<uint64_t>((d <= j) ? (j - d) : 0) * kCheckpoints + cp];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((a0 + q < NA) && (d_first + ty + 8 * (a0 + q) <= span_j)) {
            hopeless[(a0 + q < NA) ? a0 + q : 0] = hopeless[(a0 + q < NA) ? a0 + q : 0] && pair_hopeless(A, acc[(a0 + q < NA) ? a0 + q : 0], gi[q], gj, rs);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const cp_slot cj = A.cp_stats[static_cast<uint64_t>(j) * kCpStride + cp];
      const cp_slot gj = A.cp_stats[static_cast<uint64_t>(j) * kCpStride + kCheckpoints];
#pragma unroll
      for (int a0 = 0; a0 < NA; a0 += 4) {
        cp_slot ci[4], gi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t d = d_first + ty + 8 * (a0 + q);
          const uint64_t i = (d <= j) ? (j - d) : 0;
          ci[q] = A.cp_stats[i * kCpStride + cp];
          gi[q] = A.cp_stats[i * kCpStride + kCheckpoints];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if ((a0 + q < NA) && (d_first + ty + 8 * (a0 + q) <= span_j)) {
            hopeless[(a0 + q < NA) ? a0 + q : 0] = hopeless[(a0 + q < NA) ? a0 + q : 0] && pair_hopeless(A, acc[(a0 + q < NA) ? a0 + q : 0], ci[q], gi[q], cj, gj);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  uint32_t live = 0;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (!__all(hopeless[a])) {
      live = a + 1;
    }
  }
  return live;
}

// One wave per work item: does any LDS row of the tile carry missing calls?  Decides which of the two
// pair_tiles_kernel instantiations owns the item (the other one exits at once).
__global__ __launch_bounds__(256) void classify_items_kernel(PairKernelArgs A) {
  const uint32_t item_idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (item_idx >= A.n_items) {
    return;
  }
  const WorkItem it = A.items[item_idx];
  const TileGeom G = make_geom(it, it.units);
  int miss = 0;
  uint32_t missing_calls = 0;
  for (int r = lane; r < G.rtot; r += 64) {
    const ldp_variant_rec rec = A.recs[row_variant(G, r)];
    miss |= (rec.flags & 4u) ? 1 : 0;
    missing_calls += A.founder_ct - rec.nm_ct;
  }
  const unsigned long long any = __ballot(miss);
  missing_calls = wave_reduce_add(missing_calls);
  if (lane == 0) {
    // 0: complete data.  1: missing calls, sparse enough (< 1 % of the tile's genotypes) for the interval bound of
    // general_live_units to pay; 2: missing calls, checkpoints off (measured: no tile terminates at 5 %).
    const bool sparse = (static_cast<uint64_t>(missing_calls) * 100 < static_cast<uint64_t>(G.rtot) * A.founder_ct);
    A.item_general[item_idx] = any ? (sparse ? 1 : 2) : 0;
  }
}

// Epilogue staging: accumulators go through LDS ([counter][thread]) so the per-pair decision code runs as a rolled
// loop with a handful of live registers instead of one unrolled copy per unit; four units at a time.
constexpr int kEpilogueLdsDwords = 4 * 7 * kBlockThreads;

// The k-loop between two checkpoints, one instantiation per unit count.  The dispatch sits OUTSIDE the chunk loop on
// purpose: with it inside, the register allocator gives every case its own homes for the accumulators and pays for
// that with ~100 v_mov per chunk (measured: 19 % of all VALU instructions).
template <bool GENERAL, int NA>
__device__ __forceinline__ void run_chunks(Ring& R, const Stager& st, uint32_t* lds, uint32_t kc_end, uint32_t chunks, uint32_t wave, uint32_t lane,
                                           int jrow, int irow, uint32_t (&acc)[kMaxUnitsPerBlock][Counts<GENERAL>::n]) {
  while (R.kc < kc_end) {
    const uint4* l4 = ring_acquire(R, st, lds, chunks, wave, lane);
    if constexpr (NA > 0) {
      tile_chunk<GENERAL, NA>(l4, jrow, irow, acc);
    }
    ring_release(R, st);
  }
}

// (a switch over NA = 1 .. kMaxUnitsPerBlock)
template <bool GENERAL, int NA>
__device__ __forceinline__ void dispatch_run(uint32_t live, Ring& R, const Stager& st, uint32_t* lds, uint32_t kc_end, uint32_t chunks, uint32_t wave,
                                             uint32_t lane, int jrow, int irow, uint32_t (&acc)[kMaxUnitsPerBlock][Counts<GENERAL>::n]) {
  if constexpr (NA <= kMaxUnitsPerBlock) {
    if (live == static_cast<uint32_t>(NA)) {
      run_chunks<GENERAL, NA>(R, st, lds, kc_end, chunks, wave, lane, jrow, irow, acc);
    } else {
      dispatch_run<GENERAL, NA + 1>(live, R, st, lds, kc_end, chunks, wave, lane, jrow, irow, acc);
    }
  }
}

template <bool GENERAL, int NA>
__device__ __forceinline__ uint32_t dispatch_live(uint32_t live, const PairKernelArgs& A, uint32_t j0, uint32_t jend, uint32_t d_first, int tx, int ty,
                                                  uint32_t b, const uint32_t (&acc)[kMaxUnitsPerBlock][Counts<GENERAL>::n], uint32_t cp) {
  if constexpr (NA <= kMaxUnitsPerBlock) {
    if (live == static_cast<uint32_t>(NA)) {
      return live_units<GENERAL, NA>(A, j0, jend, d_first, tx, ty, b, acc, cp);
    }
    return dispatch_live<GENERAL, NA + 1>(live, A, j0, jend, d_first, tx, ty, b, acc, cp);
  } else {
    return live;
  }
}

template <bool GENERAL>
__global__ __launch_bounds__(kBlockThreads, GENERAL ? 2 : 4) void pair_tiles_kernel(PairKernelArgs A) {
  constexpr int kCounts = Counts<GENERAL>::n;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_live[kWavesPerBlock];
  __shared__ uint32_t s_src_off[kMaxDmaPerWave * kBlockThreads];
  // XCD-aware order: hardware places block b on XCD b % 8; give each XCD a contiguous run of work
  // items so neighbouring J-blocks (which share most of their window rows) hit the same L2.
  const uint32_t per_xcd = (A.n_items + 7) / 8;
  const uint32_t item_idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (item_idx >= A.n_items) {
    return;
  }
  if ((A.mf_active == 2) || (A.mf_active && (*A.route == kRouteComplete))) {
    return;  // the matrix-pipe kernels own this launch (ldp_pair_mfma.hip)
  }
  const uint32_t item_class = A.item_general[item_idx];  // classify_items_kernel: 0 complete, 1 / 2 missing calls
  if ((item_class != 0) != GENERAL) {
    return;
  }
  const WorkItem it = A.items[item_idx];
  const uint32_t tid = threadIdx.x;
  // (readfirstlane: tells the compiler the wave index, and everything derived from it, is wave-uniform -> SGPRs)
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const int tx = lane & 7;
  const int ty = lane >> 3;
  const int col = tx + 8 * static_cast<int>(wave);  // this lane's second variant within the J-tile

  uint32_t acc[kMaxUnitsPerBlock][kCounts];
#pragma unroll
  for (int a = 0; a < kMaxUnitsPerBlock; ++a) {
#pragma unroll
    for (int q = 0; q < kCounts; ++q) {
      acc[a][q] = 0;
    }
  }
  uint32_t live = it.units;    // units this wave still accumulates (nearest first; wave-uniform)
  uint32_t staged = it.units;  // units the block still stages (block-uniform)
  TileGeom G = make_geom(it, staged);
  Stager st;
  st.src_off = s_src_off;
  plan_stager(st, G, A, wave, lane);
  int jrow = G.jrow_base + col;
  int irow = G.dmax + col - static_cast<int>(it.d0) - ty;  // I-row of unit 0: variant j - (d0 + ty)
  Ring R;
  uint32_t next_cp = 0;
  // checkpoints: off for inspection / r^2 runs (cp_stats null) and for tiles where missing calls are too dense for
  // the interval bound to ever fire (class 2)
  const uint32_t n_cp = (A.cp_stats && (item_class != 2)) ? A.n_checkpoints : 0;
  ring_start(R, st, lds, 0, A.chunks, wave, lane);
  while (R.kc < A.chunks) {
    // run to the next checkpoint (or to the end of the rows)
    const bool cp_ahead = (next_cp < n_cp);  // block-uniform
    const uint32_t kc_end = cp_ahead ? A.checkpoint_chunk[next_cp] : A.chunks;  // checkpoints are < chunks
    if (live) {
      dispatch_run<GENERAL, 1>(live, R, st, lds, kc_end, A.chunks, wave, lane, jrow, irow, acc);
    } else {
      run_chunks<GENERAL, 0>(R, st, lds, kc_end, A.chunks, wave, lane, jrow, irow, acc);  // (still stages its share)
    }
    if (cp_ahead) {
      if (live) {
        const uint32_t keep = dispatch_live<GENERAL, 1>(live, A, it.j0, it.jend, it.d0, tx, ty, wave, acc, next_cp);
        if (keep != live) {
          // every pair of the dropped units is provably below the threshold
          if (lane == 0) {
            atomicAdd(A.counters + 1, static_cast<unsigned long long>(A.chunks - R.kc) * (live - keep));
          }
          live = __builtin_amdgcn_readfirstlane(keep);
        }
      }
      ++next_cp;
      // how much of the distance range does the block still need?
      if (lane == 0) {
        s_live[wave] = live;
      }
      __syncthreads();  // (a full fence: this wave's queued chunks have landed, too)
      uint32_t need = 0;
#pragma unroll
      for (uint32_t w = 0; w < kWavesPerBlock; ++w) {
        need = (s_live[w] > need) ? s_live[w] : need;
      }
      __syncthreads();  // (s_live is rewritten at the next checkpoint)
      if (!need) {
        break;  // nothing left that could reach the threshold
      }
      if (need < staged) {
        // shrink the tile to the units still live: fewer rows per chunk, more chunks in flight.  Every wave is past
        // the barrier and none has outstanding DMA, so the ring can simply be restarted at the next chunk.
        staged = need;
        G = make_geom(it, staged);
        plan_stager(st, G, A, wave, lane);
        jrow = G.jrow_base + col;
        irow = G.dmax + col - static_cast<int>(it.d0) - ty;
        ring_start(R, st, lds, R.kc, A.chunks, wave, lane);
      }
    }
  }
  __syncthreads();  // staging is over: LDS becomes the epilogue's scratch
  uint32_t n_true = 0;  // predicates this thread found true
  const uint32_t j = it.j0 + col;
  const uint32_t lo_j = (j < it.jend) ? A.lo[j] : 0;
#pragma unroll
  for (int a0 = 0; a0 < kMaxUnitsPerBlock; a0 += 4) {
    if (static_cast<uint32_t>(a0) < live) {  // (pairs of dropped units are all below the threshold)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int q = 0; q < kCounts; ++q) {
          lds[(a * kCounts + q) * kBlockThreads + tid] = acc[(a0 + a < kMaxUnitsPerBlock) ? a0 + a : 0][q];
        }
      }
      if (j < it.jend) {
#pragma unroll 1
        for (uint32_t a = 0; (a < 4) && (a0 + a < live); ++a) {
          const uint32_t d = it.d0 + ty + 8 * (a0 + a);
          if (d > j - lo_j) {
            continue;
          }
          const uint32_t i = j - d;
          const uint32_t* c = lds + (a * kCounts) * kBlockThreads + tid;
          ldp_pair_stats_t ps;
          ps.dot = static_cast<int32_t>(c[0]) - 2 * static_cast<int32_t>(c[kBlockThreads]);
          if constexpr (GENERAL) {
            const uint32_t c2 = c[2 * kBlockThreads], c3 = c[3 * kBlockThreads], c4 = c[4 * kBlockThreads];
            const uint32_t c5 = c[5 * kBlockThreads], c6 = c[6 * kBlockThreads];
            ps.nm = c2;
            ps.ssq2 = c3;
            ps.sum2 = static_cast<int32_t>(2 * c4 - c3);
            ps.ssq1 = c5;
            ps.sum1 = static_cast<int32_t>(2 * c6 - c5);
          } else {
            ps.nm = A.founder_ct;
            ps.sum1 = A.recs[i].sum;
            ps.ssq1 = A.recs[i].ssq;
            ps.sum2 = A.recs[j].sum;
            ps.ssq2 = A.recs[j].ssq;
          }
          n_true += emit_pair(A, i, j, lo_j, ps) ? 1 : 0;
        }
      }
    }
  }
  n_true = wave_reduce_add(n_true);
  if ((lane == 0) && n_true) {
    atomicAdd(A.counters, static_cast<unsigned long long>(n_true));
  }
}

// max_rows = the largest TileGeom::rtot over the work items (ldp_tile_rows() in the engine mirrors make_geom)
size_t pair_tiles_lds_bytes(uint32_t max_rows) {
  const size_t stage = ((static_cast<size_t>(max_rows) * kLdsRowSlots + 63) / 64) * 1024;  // whole DMA instructions
  const size_t epi = static_cast<size_t>(kEpilogueLdsDwords) * sizeof(uint32_t);
  return (2 * stage > epi) ? 2 * stage : epi;
}

// ---- chrX pairs of the r^2 outputs (ComputeXR2, plink2_ld.cc:7122-7190) ----------------------------------------------------------
// A pair with a chrX variant weighs the male founders down in all six sums -- by 1/2 when both variants are on chrX, by
// 1 - sqrt(2)/2 when one is -- before the usual quotient (clamped at 1; undefined when a weighted variance is not positive).
// The engines' tuples (+-1 coding, each engine's own major-allele orientation) become the reference's counts of the target
// orientation's allele exactly, in integers; then the reference's doubles, fma for fma (its documented AVX2 build defines
// FP_FAST_FMA; v_fma_f64 is the same operation, the build has -ffp-contract=off).
namespace {
struct XCounts {
  int64_t n, g1, q1, g2, q2, d;
};
__device__ __forceinline__ XCounts x_counts(const ldp_pair_stats_t& t, bool flip1, bool flip2) {
  XCounts c;
  c.n = t.nm;
  c.g1 = c.n - t.sum1;
  c.q1 = c.n - 2 * static_cast<int64_t>(t.sum1) + t.ssq1;
  c.g2 = c.n - t.sum2;
  c.q2 = c.n - 2 * static_cast<int64_t>(t.sum2) + t.ssq2;
  c.d = c.n - t.sum1 - t.sum2 + t.dot;
  if (flip1) {  // g -> 2 - g
    c.q1 = 4 * c.n - 4 * c.g1 + c.q1;
    c.g1 = 2 * c.n - c.g1;
    c.d = 2 * c.g2 - c.d;
  }
  if (flip2) {
    c.q2 = 4 * c.n - 4 * c.g2 + c.q2;
    c.g2 = 2 * c.n - c.g2;
    c.d = 2 * c.g1 - c.d;
  }
  return c;
}
__global__ __launch_bounds__(256) void x_weighted_kernel(XWeightedArgs A) {
  const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= static_cast<uint64_t>(A.rows) * A.cols) {
    return;
  }
  const uint32_t q = static_cast<uint32_t>(idx / A.cols), c = static_cast<uint32_t>(idx % A.cols);
  const uint32_t j = A.row_first + q, i = A.col_first + c;
  if (i >= j) {
    return;
  }
  const bool xi = A.is_x[i] != 0, xj = A.is_x[j] != 0;
  if (!(xi || xj)) {
    return;
  }
  const double nan_ref = __longlong_as_double(static_cast<long long>(0xfff8000000000000ull));  // (the reference's 0.0 / 0.0 on x86)
  const XCounts a = x_counts(A.all[idx], A.flip_all && A.flip_all[i], A.flip_all && A.flip_all[j]);
  XCounts m = {0, 0, 0, 0, 0, 0};
  if (A.male) {
    m = x_counts(A.male[idx], A.flip_male && A.flip_male[i], A.flip_male && A.flip_male[j]);
  }
  double r = nan_ref;
  if (a.n) {
    const double male_downwt = (xi && xj) ? 0.5 : (1.0 - 0.5 * 1.4142135623730951);
    const double w_obs = fma(-male_downwt, static_cast<double>(m.n), static_cast<double>(a.n));
    const double w_g1 = fma(-male_downwt, static_cast<double>(m.g1), static_cast<double>(a.g1));
    const double w_g2 = fma(-male_downwt, static_cast<double>(m.g2), static_cast<double>(a.g2));
    const double w_q1 = fma(-male_downwt, static_cast<double>(m.q1), static_cast<double>(a.q1));
    const double w_q2 = fma(-male_downwt, static_cast<double>(m.q2), static_cast<double>(a.q2));
    const double w_d = fma(-male_downwt, static_cast<double>(m.d), static_cast<double>(a.d));
    const double var1 = fma(w_q1, w_obs, -__dmul_rn(w_g1, w_g1));
    const double var2 = fma(w_q2, w_obs, -__dmul_rn(w_g2, w_g2));
    if ((var1 > 0.0) && (var2 > 0.0)) {
      const double var_prod = __dmul_rn(var1, var2);
      const double cov = fma(w_d, w_obs, -__dmul_rn(w_g1, w_g2));
      const double quot = __ddiv_rn(__dmul_rn(cov, cov), var_prod);
      r = (1.0 < quot) ? 1.0 : quot;
      if (A.unsquared) {
        r = __dsqrt_rn(r);
        if (cov < 0.0) {
          r = -r;
        }
      }
    }
  }
  if (A.hits) {
    if (fabs(r) >= A.min_r2) {  // (false for NaN)
      const unsigned long long slot = atomicAdd(A.hit_count, 1ull);
      if (slot < A.hit_capacity) {
        ldp_r2_hit h;
        h.first = i;
        h.second = j;
        h.r2 = r;
        A.hits[slot] = h;
      }
    }
    return;
  }
  const uint64_t o = static_cast<uint64_t>(q) * A.out_ld + c;
  if (A.as_float) {
    static_cast<float*>(A.out)[o] = (r != r) ? __uint_as_float(0xffc00000u) : static_cast<float>(r);
  } else {
    static_cast<double*>(A.out)[o] = r;
  }
}
}  // namespace

hipError_t launch_x_weighted(const XWeightedArgs& a, hipStream_t stream) {
  const uint64_t n = static_cast<uint64_t>(a.rows) * a.cols;
  if (!n) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(x_weighted_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pair_tiles(const PairKernelArgs& a_in, uint32_t max_rows, hipStream_t stream, hipEvent_t* ev) {
  if (!a_in.n_items) {
    return hipSuccess;
  }
  PairKernelArgs a = a_in;
  const uint32_t per_xcd = (a.n_items + 7) / 8;
  size_t lds = pair_tiles_lds_bytes(max_rows);
  if (const char* pad = LDP_ENV("LDP_DEBUG_LDS_KB")) {  // tuning aid: force fewer resident blocks per CU
    lds = std::max<size_t>(lds, static_cast<size_t>(atoi(pad)) * 1024);
  }
  a.lds_dwords = static_cast<uint32_t>(lds / sizeof(uint32_t));
  hipLaunchKernelGGL(classify_items_kernel, dim3((a.n_items + 3) / 4), dim3(256), 0, stream, a);
  if (ev) {
    (void)hipEventRecord(ev[0], stream);
  }
  hipLaunchKernelGGL(pair_tiles_kernel<false>, dim3(per_xcd * 8), dim3(kBlockThreads), lds, stream, a);
  if (ev) {
    (void)hipEventRecord(ev[1], stream);
    (void)hipEventRecord(ev[2], stream);
  }
  hipLaunchKernelGGL(pair_tiles_kernel<true>, dim3(per_xcd * 8), dim3(kBlockThreads), lds, stream, a);
  if (ev) {
    (void)hipEventRecord(ev[3], stream);
  }
  return hipGetLastError();
}

// ================================================================================================
// reference pair kernel: one wave per pair, lanes across plane dwords
// ================================================================================================
__global__ __launch_bounds__(256) void pair_stats_ref_kernel(const uint32_t* planes, uint64_t row_dwords, uint32_t chunks, uint32_t base_variant,
                                                            const uint32_t* first, const uint32_t* second, uint32_t n_pairs, ldp_pair_stats_t* out) {
  const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (pair >= n_pairs) {
    return;
  }
  const uint32_t* r1 = planes + static_cast<uint64_t>(first[pair] - base_variant) * row_dwords;
  const uint32_t* r2 = planes + static_cast<uint64_t>(second[pair] - base_variant) * row_dwords;
  uint32_t c[7] = {0, 0, 0, 0, 0, 0, 0};
  const uint32_t plane_dwords = chunks * kChunkDwords;
  for (uint32_t p = lane; p < plane_dwords; p += 64) {
    const uint32_t off = (p / kChunkDwords) * kRowChunkDwords + (p % kChunkDwords);
    const uint32_t h1 = r1[off], q1 = r1[off + kChunkDwords];
    const uint32_t h2 = r2[off], q2 = r2[off + kChunkDwords];
    const uint32_t n1 = h1 | q1, n2 = h2 | q2;
    const uint32_t h = h1 & h2;
    c[0] += __popc(h);
    c[1] += __popc(h & (q1 ^ q2));
    c[2] += __popc(n1 & n2);
    c[3] += __popc(n1 & h2);
    c[4] += __popc(n1 & h2 & q2);
    c[5] += __popc(n2 & h1);
    c[6] += __popc(n2 & h1 & q1);
  }
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    c[q] = wave_reduce_add(c[q]);
  }
  if (lane == 0) {
    ldp_pair_stats_t st;
    st.nm = c[2];
    st.ssq2 = c[3];
    st.sum2 = static_cast<int32_t>(2 * c[4] - c[3]);
    st.ssq1 = c[5];
    st.sum1 = static_cast<int32_t>(2 * c[6] - c[5]);
    st.dot = static_cast<int32_t>(c[0]) - 2 * static_cast<int32_t>(c[1]);
    out[pair] = st;
  }
}

hipError_t launch_pair_stats_ref(const uint32_t* planes, uint64_t row_dwords, uint32_t chunks, uint32_t plane_base_variant,
                                 const uint32_t* first, const uint32_t* second, uint32_t n_pairs,
                                 ldp_pair_stats_t* out, hipStream_t stream) {
  if (!n_pairs) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(pair_stats_ref_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, planes, row_dwords, chunks, plane_base_variant, first, second, n_pairs, out);
  return hipGetLastError();
}

}  // namespace ldp
