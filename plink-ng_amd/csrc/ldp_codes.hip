// ldp_codes.hip -- the 2-bit code image (ldp_device.h) that the matrix-pipe pair kernels read: the pass that counts a row
// (and, for input that is not already a REF- / INVERSE-coded row of the image, writes it), and the one-wave-per-pair reference
// kernel on the image.
//
//   codes_kernel   what prepare_kernel is for the bit-planes (FillVaggs plink2_ld.cc:725-738; allele counts / major allele
//                  plink2_data.cc:2421-2443, plink2_filter.cc:2113-2153, plink2_common.h:559-567; PgrPlink1ToPlink2InplaceUnsafe
//                  pgenlib_read.cc:2157; HapsplitMustPhased pgenlib_misc.cc:1887 for --indep-pairphase rows) WITHOUT the
//                  SplitHomRef2het step (pgenlib_misc.cc:1797-1885) and WITHOUT GenovecInvertUnsafe (pgenlib_misc.cc:1090): the
//                  pair kernels expand the codes themselves and take the orientation from the records.  A REF-coded row that
//                  already sits in the image is read once and not written: the pass is a read of N/4 bytes per variant.
#include "ldp_device.h"
#include "ldp_pair_device.h"

#include <algorithm>
#include <cstdlib>
#include <string>

namespace ldp {

namespace {

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t src_dword(const uint8_t* row, uint32_t nbytes, uint32_t didx, bool aligned4) {
  const uint32_t off = didx * 4;
  if (off >= nbytes) {
    return 0;
  }
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

// .bed -> .pgen codes, sixteen at a time: 00 -> 10, 01 -> 11, 10 -> 01, 11 -> 00 (pgenlib_read.cc:2157)
__device__ __forceinline__ uint32_t pgen_of_bed(uint32_t w) {
  const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
  return ((hi ^ 0x55555555u) << 1) | (lo ^ hi);
}

// bits at positions 2 s (s < 8) of the low half of x -> positions 4 s
__device__ __forceinline__ uint32_t spread_to_nibbles(uint32_t x) {
  uint32_t t = x & 0x5555u;
  t = (t | (t << 8)) & 0x00ff00ffu;
  t = (t | (t << 4)) & 0x0f0f0f0fu;
  t = (t | (t << 2)) & 0x33333333u;
  return t & 0x11111111u;
}

// --indep-pairphase rows (LDP_GENO_PHASED, include/ldprune_hip.h): 16 samples (one dword of 2-bit codes + 16 phaseinfo bits) ->
// 32 haplotypes = two dwords of codes.  Haplotype h (1 = carries the counted allele) is carried as the genotype code 2 h, a
// missing call as 11 on both; which haplotype gets a het's counted allele follows HapsplitMustPhased (pgenlib_misc.cc:1917:
// het + phaseinfo -> the second one).  The statistics are 4 x the reference's (ldp_kernels.hip: hap_planes_of_16).
__device__ __forceinline__ void hap_codes_of_16(uint32_t w, uint32_t phase16, uint32_t* out0, uint32_t* out1) {
  const uint32_t lo = w & 0x55555555u;
  const uint32_t hi = (w >> 1) & 0x55555555u;
  const uint32_t miss = lo & hi;
  const uint32_t het = lo & ~hi;
  const uint32_t two = hi & ~lo;
  uint32_t ph = phase16 & 0xffffu;  // bit s -> bit 2 s
  ph = (ph | (ph << 8)) & 0x00ff00ffu;
  ph = (ph | (ph << 4)) & 0x0f0f0f0fu;
  ph = (ph | (ph << 2)) & 0x33333333u;
  ph = (ph | (ph << 1)) & 0x55555555u;
  const uint32_t b1_first = two | (het & ~ph) | miss;
  const uint32_t b1_second = two | (het & ph) | miss;
  // sample s -> nibble s: [b0 first, b1 first, b0 second, b1 second], b0 = missing
  *out0 = spread_to_nibbles(miss) * 5u | (spread_to_nibbles(b1_first) << 1) | (spread_to_nibbles(b1_second) << 3);
  *out1 = spread_to_nibbles(miss >> 16) * 5u | (spread_to_nibbles(b1_first >> 16) << 1) | (spread_to_nibbles(b1_second >> 16) << 3);
}

// codes 00 <-> 10 of sixteen samples (01 and 11 stay): GenovecInvertUnsafe on the 2-bit codes
__device__ __forceinline__ uint32_t invert_codes(uint32_t w) { return w ^ ((~w & 0x55555555u) << 1); }
__device__ __forceinline__ u32x4 invert_codes(u32x4 w) {
  w.x = invert_codes(w.x);
  w.y = invert_codes(w.y);
  w.z = invert_codes(w.z);
  w.w = invert_codes(w.w);
  return w;
}

// One block per variant, one 16-byte unit of the image row (64 samples) per thread and iteration.  ITERS > 0: the iterations
// are unrolled with all of a thread's loads issued up front (few threads with many 16-byte loads in flight each beat many
// threads with few, as in prepare_kernel); ITERS == 0: a rolled loop for rows beyond the register budget.
template <int THREADS, int ITERS, bool NT = false>
__global__ __launch_bounds__(THREADS) void codes_kernel(PrepareArgs A) {
  constexpr int kWaves = THREADS / 64;
  __shared__ uint32_t red[kWaves][3 + 2 * kCheckpoints + kGenCheckpoints];
  __shared__ uint32_t s_alt_major;
  __shared__ uint32_t s_flip;     // the row changes orientation in this pass: what was read is stored with codes 00 <-> 10 swapped
  __shared__ uint32_t s_differs;  // the FINAL image row's orientation differs from the record's (major allele)
  __shared__ int32_t s_sum;
  const uint32_t v = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint8_t* row = A.geno + static_cast<uint64_t>(v) * A.stride_bytes;
  uint8_t* out_row = A.codes_out + static_cast<uint64_t>(v) * A.code_row_bytes;
  const bool in_place = (row == out_row);  // a row of the image itself (ldp_map_rows): counted where it is
  // ... and such a row may be stored inverted relative to the input by an earlier pass (major-allele-oriented image, ldp_device.h)
  const bool inv_in = in_place && A.stored_inv && (A.stored_inv[v] != 0);
  const bool phased = (A.encoding & LDP_GENO_PHASED) != 0;
  const bool bed = ((A.encoding & 3) == LDP_GENO_BED);
  const uint32_t n_units = static_cast<uint32_t>(A.code_row_bytes / 16);
  const bool aligned4 = ((reinterpret_cast<uintptr_t>(row) & 3) == 0);
  // plain rows: ceil(founder_ct / 4) bytes of codes.  Phased rows: ceil(S / 4) bytes of codes, padding, ceil(S / 8) bytes of phase bits
  const uint32_t samples = phased ? (A.founder_ct >> 1) : A.founder_ct;
  const uint32_t code_bytes = (samples + 3) >> 2;
  const uint32_t phase_off = (code_bytes + 3) & ~3u;
  const uint32_t phase_bytes = (samples + 7) >> 3;
  // units that are whole 16-byte pieces of a plain source row: one vector load, no tail, no padding
  const uint32_t n_fast = (phased || !aligned4) ? 0u : (A.founder_ct / 64);

  uint32_t hom_ct = 0, r2h_ct = 0, both_ct = 0;
  uint32_t rest[kCheckpoints];  // hom calls | code-0 calls << 16 in k-chunks >= checkpoint k (a thread's share stays below 2^16)
  uint32_t rest_nm[kGenCheckpoints];  // missing calls (+ padding) there, for the checkpoints the six-product kernel uses (its bound: cp_gen)
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    rest[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < kGenCheckpoints; ++k) {
    rest_nm[k] = 0;
  }
  // everything a unit needs besides a plain vector load: phased rows, unaligned rows, the tail of the row and its padding
  auto slow_unit = [&](uint32_t u, u32x4& w) {
    const uint32_t s0 = u * 64;  // first sample (haplotype) of the unit
    if (s0 >= A.founder_ct) {
      w.x = w.y = w.z = w.w = 0xffffffffu;  // padding up to the stage boundary: "missing"
      return;
    }
    if (phased) {
      const uint32_t ph = src_dword(row + phase_off, phase_bytes, u, aligned4);
      uint32_t o0, o1, o2, o3;
      hap_codes_of_16(src_dword(row, code_bytes, 2 * u, aligned4), ph, &o0, &o1);
      hap_codes_of_16(src_dword(row, code_bytes, 2 * u + 1, aligned4), ph >> 16, &o2, &o3);
      w.x = o0;
      w.y = o1;
      w.z = o2;
      w.w = o3;
    } else {
      w.x = src_dword(row, code_bytes, 4 * u, aligned4);
      w.y = src_dword(row, code_bytes, 4 * u + 1, aligned4);
      w.z = src_dword(row, code_bytes, 4 * u + 2, aligned4);
      w.w = src_dword(row, code_bytes, 4 * u + 3, aligned4);
      if (bed) {
        w.x = pgen_of_bed(w.x);
        w.y = pgen_of_bed(w.y);
        w.z = pgen_of_bed(w.z);
        w.w = pgen_of_bed(w.w);
      }
    }
    if (s0 + 64 > A.founder_ct) {
      // samples >= founder_ct: coded missing, whatever the caller's trailing bits were
      const uint32_t left = A.founder_ct - s0;  // 1..63
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t first = 16u * d;
        uint32_t m = 0;
        if (left <= first) {
          m = 0xffffffffu;
        } else if (left - first < 16) {
          m = ~((1u << (2 * (left - first))) - 1u);
        }
        w[d] |= m;
      }
    }
  };
  auto count_unit = [&](uint32_t u, const u32x4& w) {
    // 32 samples per operation: lo / hi = the low / high code bits of two dwords interleaved (any sample order will do)
    const uint32_t lo0 = (w.x & 0x55555555u) | ((w.y & 0x55555555u) << 1), hi0 = ((w.x >> 1) & 0x55555555u) | (w.y & 0xaaaaaaaau);
    const uint32_t lo1 = (w.z & 0x55555555u) | ((w.w & 0x55555555u) << 1), hi1 = ((w.z >> 1) & 0x55555555u) | (w.w & 0xaaaaaaaau);
    const uint32_t hc = __popc(~lo0) + __popc(~lo1);                  // homozygous calls (codes 00, 10)
    const uint32_t bc = __popc(~(lo0 | hi0)) + __popc(~(lo1 | hi1));  // code 00
    hom_ct += hc;
    r2h_ct += __popc(~hi0) + __popc(~hi1);                            // codes 00, 01
    both_ct += bc;
    const uint32_t chunk = u / (kChunkDwords * 32 / 64);              // 512-sample k-chunk
    const uint32_t packed = hc | (bc << 16);
#pragma unroll
    for (int k = 0; k < kCheckpoints; ++k) {
      rest[k] += (chunk >= A.checkpoint_chunk[k]) ? packed : 0;
    }
  };
  // missing calls (and the padding behind the last sample, coded the same) per remainder, for the six-product kernel's bound: only
  // threads that met one come back for this
  auto count_missing = [&](uint32_t u, const u32x4& w) {
    const uint32_t mc = __popc(w.x & (w.x >> 1) & 0x55555555u) + __popc(w.y & (w.y >> 1) & 0x55555555u) + __popc(w.z & (w.z >> 1) & 0x55555555u) +
                        __popc(w.w & (w.w >> 1) & 0x55555555u);
    const uint32_t chunk = u / (kChunkDwords * 32 / 64);
#pragma unroll
    for (int k = 0; k < kGenCheckpoints; ++k) {
      rest_nm[k] += (chunk >= A.checkpoint_chunk[kGenCheckpointFirst + k]) ? mc : 0;
    }
  };
  uint32_t slow_first = 0;  // units from here on go through slow_unit()
  [[maybe_unused]] u32x4 w_held[(ITERS > 0) ? ITERS : 1];  // the row's whole units, kept until the row's orientation in the image is known
  if constexpr (ITERS > 0) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const uint32_t u = tid + it * THREADS;
      if (u < n_fast) {
        const u32x4_a4 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4*>(row + 16ull * u)) : *reinterpret_cast<const u32x4_a4*>(row + 16ull * u);
        w_held[it].x = t.x;
        w_held[it].y = t.y;
        w_held[it].z = t.z;
        w_held[it].w = t.w;
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const uint32_t u = tid + it * THREADS;
      if (u < n_fast) {
        if (bed) {
          w_held[it].x = pgen_of_bed(w_held[it].x);
          w_held[it].y = pgen_of_bed(w_held[it].y);
          w_held[it].z = pgen_of_bed(w_held[it].z);
          w_held[it].w = pgen_of_bed(w_held[it].w);
        }
        count_unit(u, w_held[it]);  // (stored below, once the row's major allele -- and with it the row's orientation in the image -- is known)
      }
    }
    // (calls present = codes 0, 1, 2 = hom + (codes 0 or 1) - code 0: a thread whose units are all calls skips this)
    uint32_t my_units = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      my_units += (tid + it * THREADS < n_fast) ? 1u : 0u;
    }
    if (hom_ct + r2h_ct - both_ct != 64 * my_units) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const uint32_t u = tid + it * THREADS;
        if (u < n_fast) {
          count_missing(u, w_held[it]);
        }
      }
    }
    slow_first = (n_fast < static_cast<uint32_t>(ITERS * THREADS)) ? n_fast : static_cast<uint32_t>(ITERS * THREADS);
  }
  // what is left: all of a phased / unaligned / very long row; otherwise the tail unit and the padding
  for (uint32_t u = slow_first + tid; u < n_units; u += THREADS) {
    u32x4 w;
    bool store = !in_place;
    if (u < n_fast) {
      const u32x4_a4 t = *reinterpret_cast<const u32x4_a4*>(row + 16ull * u);
      w.x = bed ? pgen_of_bed(t.x) : t.x;
      w.y = bed ? pgen_of_bed(t.y) : t.y;
      w.z = bed ? pgen_of_bed(t.z) : t.z;
      w.w = bed ? pgen_of_bed(t.w) : t.w;
    } else {
      slow_unit(u, w);
      store = store || (u * 64 + 64 > A.founder_ct);  // (the tail and the padding are the engine's to write, in place too)
    }
    if (store) {
      __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(out_row + 16ull * u));
    }
    if (u * 64 < A.founder_ct) {
      const uint32_t calls_before = hom_ct + r2h_ct - both_ct;
      count_unit(u, w);
      if (hom_ct + r2h_ct - both_ct != calls_before + 64) {
        count_missing(u, w);
      }
    }
  }
  hom_ct = wave_reduce_add(hom_ct);
  r2h_ct = wave_reduce_add(r2h_ct);
  both_ct = wave_reduce_add(both_ct);
  uint32_t rest_both[kCheckpoints];
#pragma unroll
  for (int k = 0; k < kCheckpoints; ++k) {
    rest_both[k] = wave_reduce_add(rest[k] >> 16);
    rest[k] = wave_reduce_add(rest[k] & 0xffffu);
  }
  if (__any((rest_nm[0] | rest_nm[kGenCheckpoints - 1]) != 0)) {  // (a later remainder's count is part of an earlier one's)
#pragma unroll
    for (int k = 0; k < kGenCheckpoints; ++k) {
      rest_nm[k] = wave_reduce_add(rest_nm[k]);
    }
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
#pragma unroll
    for (int k = 0; k < kGenCheckpoints; ++k) {
      red[tid >> 6][3 + 2 * kCheckpoints + k] = rest_nm[k];
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
    // genotype counts of the row in the INPUT's orientation: code 0, 1, 2 (a row read inverted: 0 and 2 change places)
    const uint32_t n0 = inv_in ? (hom_ct - both_ct) : both_ct;
    const uint32_t n1 = r2h_ct - both_ct;
    const uint32_t n2 = inv_in ? both_ct : (hom_ct - both_ct);
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
    // the record is in major-allele orientation (what GenovecInvertUnsafe would have made of the row); the image is not
    const uint32_t plus_ct = alt_major ? n2 : n0;
    const uint32_t minus_ct = alt_major ? n0 : n2;
    const uint32_t nm_ct = plus_ct + minus_ct + n1;
    rec.nm_ct = nm_ct;
    rec.sum = static_cast<int32_t>(plus_ct - minus_ct);
    rec.ssq = hom_ct;
    const uint32_t mono = ((!plus_ct) && (!minus_ct)) || (plus_ct == nm_ct) || (minus_ct == nm_ct);  // plink2_ld.cc:902
    // the row's orientation in the image: inverted iff ALT is the major allele (and the engine orients its rows); whatever it was read as
    const uint32_t want_inv = (A.orient && alt_major) ? 1u : 0u;
    rec.flags = alt_major | (mono << 1) | ((nm_ct != A.founder_ct) ? 4u : 0u) | (want_inv ? kRecStoredInverted : 0u);
    A.recs[v] = rec;
    if (A.stored_inv) {
      A.stored_inv[v] = static_cast<uint8_t>(want_inv);
    }
    s_alt_major = alt_major;
    s_flip = (want_inv != (inv_in ? 1u : 0u)) ? 1u : 0u;
    s_differs = alt_major ^ want_inv;
    // the sum in the FINAL image's orientation: what the checkpoint bound pairs with the kernel's partial dot products
    s_sum = (alt_major ^ want_inv) ? -rec.sum : rec.sum;
  }
  __syncthreads();
  const bool flip_now = (s_flip != 0);
  // ---- the stores: units held in registers go out now, in the row's final orientation; what the loop above stored (tail, padding, all of a very
  // long row) is inverted where it lies if the row changes orientation ----
  if constexpr (ITERS > 0) {
    if ((!in_place) || flip_now) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const uint32_t u = tid + it * THREADS;
        if (u < n_fast) {
          __builtin_nontemporal_store(flip_now ? invert_codes(w_held[it]) : w_held[it], reinterpret_cast<u32x4*>(out_row + 16ull * u));
        }
      }
    }
  }
  if (flip_now) {
    for (uint32_t u = slow_first + tid; u < n_units; u += THREADS) {  // (the thread that stored a unit reads it back: program order)
      u32x4* at = reinterpret_cast<u32x4*>(out_row + 16ull * u);
      const u32x4 t = *at;
      *at = invert_codes(t);
    }
  }
  // (the six-product kernel's slots by the LAST wave, beside the first one's: the tail of a block is a chain of latencies)
  constexpr uint32_t kGenTid0 = (kWaves > 1) ? (kWaves - 1) * 64 : kCpSlots;
  const bool cp_thread = tid < kCpSlots, gen_thread = (tid >= kGenTid0) && (tid < kGenTid0 + 1 + kGenCheckpoints);
  if (A.cp_stats && (cp_thread || gen_thread)) {
    // early-termination statistics (layout: ldp_device.h), in the orientation of the image: one 16-byte slot per thread, one
    // contiguous 144-byte record per variant
    const double N = static_cast<double>(A.founder_ct);
    union {
      cp_slot cp;
      cp_gen_slot gen;
      u32x4 raw;
    } slot;
    if (tid < kCheckpoints) {
      uint32_t hom_r = 0, both_r = 0;
      for (int w = 0; w < kWaves; ++w) {
        hom_r += red[w][3 + tid];
        both_r += red[w][3 + kCheckpoints + tid];
      }
      if (flip_now) {
        both_r = hom_r - both_r;  // (code 00 of the FINAL image row: the row is stored inverted relative to what was read)
      }
      const double s_r = static_cast<double>(static_cast<int32_t>(2 * both_r - hom_r));
      const uint64_t seen = static_cast<uint64_t>(A.checkpoint_chunk[tid]) * (kChunkDwords * 32);
      const double n_r = (seen < A.founder_ct) ? static_cast<double>(A.founder_ct - seen) : 1.0;
      const double v_r = fmax(static_cast<double>(hom_r) - s_r * s_r / n_r, 0.0);
      slot.cp.a = s_r * sqrt(N / n_r);
      slot.cp.b = sqrt(N * v_r);
    } else if (cp_thread) {
      uint32_t hom_all = 0;
      for (int w = 0; w < kWaves; ++w) {
        hom_all += red[w][0];
      }
      const double S = static_cast<double>(s_sum);
      slot.cp.a = S;
      slot.cp.b = sqrt(fmax(N * static_cast<double>(hom_all) - S * S, 0.0)) * A.cp_tv_scale;
    } else {
      // the six-product kernel's slots, in z = 1 - x (0 = code 0, 1 = het, 2 = code 2) over the variant's own calls:
      // sum z = hets + 2 code-2 calls, sum z^2 = hets + 4 code-2 calls
      uint32_t hom = 0, both = 0, nm = 0;
      uint32_t flag = 0;
      if (tid == kGenTid0) {  // the whole row (calls = codes 0, 1, 2 = hom + (codes 0 or 1) - code 0), with the ALT-major flag
        uint32_t r2h = 0;
        for (int w = 0; w < kWaves; ++w) {
          hom += red[w][0];
          r2h += red[w][1];
          both += red[w][2];
        }
        nm = hom + r2h - both;
        flag = s_differs;  // (gen_row turns z to the minor allele's count where the image counts the major allele's)
      } else {
        const uint32_t k = tid - kGenTid0 - 1, cp = kGenCheckpointFirst + k;
        uint32_t miss_r = 0;
        for (int w = 0; w < kWaves; ++w) {
          hom += red[w][3 + cp];
          both += red[w][3 + kCheckpoints + cp];
          miss_r += red[w][3 + 2 * kCheckpoints + k];
        }
        // calls = the counted samples of the remainder (whole 64-sample units: the last one's padding is coded missing) - missing
        const uint64_t seen64 = static_cast<uint64_t>(A.checkpoint_chunk[cp]) * (kChunkDwords * 32);
        const uint64_t counted = (static_cast<uint64_t>(A.founder_ct) + 63) & ~static_cast<uint64_t>(63);
        nm = (seen64 < counted) ? static_cast<uint32_t>(counted - seen64) - miss_r : 0u;
      }
      if (flip_now) {
        both = hom - both;  // (as above: z is the code of the FINAL image row)
      }
      slot.gen.nm_r = nm;
      slot.gen.zs_r = nm + hom - 2 * both;
      slot.gen.zq_r = nm + 3 * hom - 4 * both;
      slot.gen.pad = flag;
    }
    reinterpret_cast<u32x4*>(A.cp_stats)[static_cast<uint64_t>(v) * kCpStride + (cp_thread ? tid : (kCpSlots + tid - kGenTid0))] = slot.raw;
  }
}

// reference pair kernel on the image: one wave per pair, lanes across 16-byte units
__global__ __launch_bounds__(256) void pair_stats_ref_codes_kernel(const uint8_t* codes, uint64_t code_row_bytes, const ldp_variant_rec* recs, const uint32_t* first,
                                                                  const uint32_t* second, uint32_t n_pairs, ldp_pair_stats_t* out) {
  const uint32_t pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (pair >= n_pairs) {
    return;
  }
  const uint32_t i = first[pair], j = second[pair];
  const uint32_t* r1 = reinterpret_cast<const uint32_t*>(codes + static_cast<uint64_t>(i) * code_row_bytes);
  const uint32_t* r2 = reinterpret_cast<const uint32_t*>(codes + static_cast<uint64_t>(j) * code_row_bytes);
  uint32_t c[7] = {0, 0, 0, 0, 0, 0, 0};
  const uint32_t n_dwords = static_cast<uint32_t>(code_row_bytes / 4);
  for (uint32_t p = lane; p < n_dwords; p += 64) {
    const uint32_t w1 = r1[p], w2 = r2[p];
    // sixteen samples at the even bit positions: h = homozygous, q = code 00 or 01 ("ref2het"), n = call present
    const uint32_t h1 = ~w1 & 0x55555555u, q1 = ~(w1 >> 1) & 0x55555555u, h2 = ~w2 & 0x55555555u, q2 = ~(w2 >> 1) & 0x55555555u;
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
    // the image is in the rows' own orientation; the integers are reported against the major allele
    const int32_t s1 = img_differs(recs[i].flags) ? -1 : 1, s2 = img_differs(recs[j].flags) ? -1 : 1;
    ldp_pair_stats_t st;
    st.nm = c[2];
    st.ssq2 = c[3];
    st.sum2 = s2 * static_cast<int32_t>(2 * c[4] - c[3]);
    st.ssq1 = c[5];
    st.sum1 = s1 * static_cast<int32_t>(2 * c[6] - c[5]);
    st.dot = s1 * s2 * (static_cast<int32_t>(c[0]) - 2 * static_cast<int32_t>(c[1]));
    out[pair] = st;
  }
}

// ldp_map_rows: the rows a caller is handed are in the INPUT's orientation
__global__ __launch_bounds__(256) void unflip_rows_kernel(uint8_t* codes, uint64_t code_row_bytes, uint8_t* stored_inv, uint32_t n) {
  const uint32_t v = blockIdx.x;
  if ((v >= n) || !stored_inv[v]) {
    return;
  }
  u32x4* row = reinterpret_cast<u32x4*>(codes + static_cast<uint64_t>(v) * code_row_bytes);
  const uint32_t n_units = static_cast<uint32_t>(code_row_bytes / 16);
  for (uint32_t u = threadIdx.x; u < n_units; u += 256) {
    row[u] = invert_codes(row[u]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    stored_inv[v] = 0;
  }
}

}  // namespace

hipError_t launch_unflip_rows(uint8_t* codes, uint64_t code_row_bytes, uint8_t* stored_inv, uint32_t n, hipStream_t stream) {
  if (!n || !stored_inv) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(unflip_rows_kernel, dim3(n), dim3(256), 0, stream, codes, code_row_bytes, stored_inv, n);
  return hipGetLastError();
}

hipError_t launch_codes(const PrepareArgs& a, hipStream_t stream) {
  if (!a.n_variants) {
    return hipSuccess;
  }
  const uint64_t units = a.code_row_bytes / 16;
  // the geometry prepare_kernel was tuned to (config 2: 128 x 7 best; N = 500,000: 512 x 16), a unit = 64 samples in both
#define LDP_CODES(T, I) hipLaunchKernelGGL((codes_kernel<T, I>), dim3(a.n_variants), dim3(T), 0, stream, a)
  if (const char* shape = LDP_ENV("LDP_DEBUG_CODES_SHAPE")) {  // tuning aid: "64x13", "128x7nt", "256x4", "256x4nt"
    const std::string sh(shape);
    if ((sh == "64x13") && (units <= 64 * 13)) {
      hipLaunchKernelGGL((codes_kernel<64, 13>), dim3(a.n_variants), dim3(64), 0, stream, a);
      return hipGetLastError();
    }
    if ((sh == "128x7nt") && (units <= 128 * 7)) {
      hipLaunchKernelGGL((codes_kernel<128, 7, true>), dim3(a.n_variants), dim3(128), 0, stream, a);
      return hipGetLastError();
    }
    if ((sh == "256x4") && (units <= 256 * 4)) {
      hipLaunchKernelGGL((codes_kernel<256, 4>), dim3(a.n_variants), dim3(256), 0, stream, a);
      return hipGetLastError();
    }
    if ((sh == "256x4nt") && (units <= 256 * 4)) {
      hipLaunchKernelGGL((codes_kernel<256, 4, true>), dim3(a.n_variants), dim3(256), 0, stream, a);
      return hipGetLastError();
    }
  }
  if (units <= 128 * 2) {
    LDP_CODES(128, 2);
  } else if (units <= 128 * 4) {
    LDP_CODES(128, 4);
  } else if (units <= 128 * 7) {
    LDP_CODES(128, 7);
  } else if (units <= 128 * 8) {
    LDP_CODES(128, 8);
  } else if (units <= 128 * 16) {
    LDP_CODES(128, 16);
  } else if (units <= 256 * 16) {
    LDP_CODES(256, 16);
  } else if (units <= 512 * 16) {
    LDP_CODES(512, 16);
  } else if (units <= 1024 * 16) {
    LDP_CODES(1024, 16);
  } else {
    LDP_CODES(1024, 0);
  }
#undef LDP_CODES
  return hipGetLastError();
}

hipError_t launch_pair_stats_ref_codes(const uint8_t* codes, uint64_t code_row_bytes, const ldp_variant_rec* recs, const uint32_t* first, const uint32_t* second,
                                       uint32_t n_pairs, ldp_pair_stats_t* out, hipStream_t stream) {
  if (!n_pairs) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(pair_stats_ref_codes_kernel, dim3((n_pairs + 3) / 4), dim3(256), 0, stream, codes, code_row_bytes, recs, first, second, n_pairs, out);
  return hipGetLastError();
}

}  // namespace ldp
