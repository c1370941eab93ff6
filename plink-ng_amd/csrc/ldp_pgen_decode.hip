// ldp_pgen_decode.hip -- variant records of a variable-width .pgen file decoded ON THE DEVICE, straight from the file's bytes
// into 2-bit genotype rows (what ReadGenovecSubsetUnsafe produces, 2.0/include/pgenlib_read.cc:2849-2912, and -- for variants
// with more than one ALT allele -- what PgrGetInv1 produces for the major allele, pgenlib_read.cc:5417-5563).
//
//   pgen_main_kernel   one workgroup per record, main track: type 0 (plain 2-bit), 1 (one bit per sample + exceptions,
//                      pgenlib_read.cc:2186-2303 Parse1or2bitGenovec / ParseOnebitUnsafe), 4 / 6 / 7 (difflist against an all-0 /
//                      all-2 / all-missing row, :2436-2532 ParseAndApplyDifflist), 2 / 3 (difflist against the most recent
//                      non-LD record, type 3 then inverted 0 <-> 2, :2687-2760 LdLoadAndCopyGenovecSubsetIfNecessary).  Two
//                      launches: the records that stand alone, then the LD-compressed ones on top of their finished bases.
//   pgen_aux1_kernel   one workgroup per variant with more than one ALT allele: auxiliary track 1 (pgen_spec.tex:469-540;
//                      Get1Multiallelic pgenlib_read.cc:5417-5563: aux1a = REF/ALTx with x >= 2, aux1b = ALTx/ALTy other than
//                      ALT1/ALT1, each as a bit array over the main track's category or as a sample-id list, then packed
//                      allele codes), allele counts over the samples, the major allele in the reference's arithmetic
//                      (plink2_filter.cc:2113-2153 freq = count * (1 / total); GetMajIdxMulti plink2_common.cc:1042-1070;
//                      GetAlleleFreq plink2_common.h:584-593), and the row rewritten as copies of NON-major alleles.
//
// A difflist (pgen_spec.tex:367-430) is entries in groups of 64: per group the first sample id, per group but the last the
// byte size of its delta stream, [2-bit values for all entries,] then the varint deltas.  The groups are independent once the
// sizes are summed, so a workgroup's threads each take a run of groups; entries touch distinct samples, so their 2-bit fields
// go in with atomics on the row's dwords.  Malformed input never reads outside the record: every byte access is checked
// against the record's end, and a violation raises the launch's error word (1 + record index) instead of a row.
#include "ldp_device.h"
#include "ldp_pair_device.h"

#include <cstdlib>

namespace ldp {

namespace {

constexpr int kThreads = 256;
constexpr int kAuxThreads = 1024;  // the auxiliary-track kernel: a thread's share of a row is a chain of latencies, so more, shorter ones
constexpr uint32_t kChunk = 8;  // dwords of a row a thread handles together in the auxiliary-track kernel: their bit-array windows are loaded together

struct ByteCursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok;
  __device__ __forceinline__ uint32_t u8() {
    if (p >= end) {
      ok = false;
      return 0;
    }
    return *p++;
  }
  __device__ __forceinline__ uint32_t varint() {
    uint32_t v = 0;
    for (int shift = 0; shift < 35; shift += 7) {
      const uint32_t b = u8();
      if (!ok) {
        return 0;
      }
      v |= (b & 0x7fu) << shift;
      if (!(b & 0x80u)) {
        return v;
      }
    }
    ok = false;
    return 0;
  }
  __device__ __forceinline__ bool skip(uint64_t n) {
    if (static_cast<uint64_t>(end - p) < n) {
      ok = false;
      return false;
    }
    p += n;
    return true;
  }
};

__device__ __forceinline__ uint32_t id_width(uint32_t sample_ct) { return (sample_ct <= 256) ? 1 : ((sample_ct <= 65536) ? 2 : ((sample_ct <= 16777216) ? 3 : 4)); }

__device__ __forceinline__ uint32_t read_le(const uint8_t* p, uint32_t nbytes) {
  uint32_t v = 0;
  for (uint32_t k = 0; k < nbytes; ++k) {
    v |= static_cast<uint32_t>(p[k]) << (8 * k);
  }
  return v;
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));

// dword d of a byte string of nbytes (bytes past the end read as zero)
__device__ __forceinline__ uint32_t dword_of_bytes(const uint8_t* p, uint64_t nbytes, uint64_t d) {
  const uint64_t off = d * 4;
  if (off + 4 <= nbytes) {
    return *reinterpret_cast<const u32_unaligned*>(p + off);  // (records start at any byte: one unaligned load, not four byte loads)
  }
  uint32_t w = 0;
  for (uint32_t k = 0; k < 4; ++k) {
    if (off + k < nbytes) {
      w |= static_cast<uint32_t>(p[off + k]) << (8 * k);
    }
  }
  return w;
}

// the 16 low bits of x -> the even bit positions of a dword
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  uint32_t t = x & 0xffffu;
  t = (t | (t << 8)) & 0x00ff00ffu;
  t = (t | (t << 4)) & 0x0f0f0f0fu;
  t = (t | (t << 2)) & 0x33333333u;
  t = (t | (t << 1)) & 0x55555555u;
  return t;
}

// a load that sees what the atomics of other waves (performed at the L2) have done to a row: not through this CU's L1
__device__ __forceinline__ uint32_t load_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The 2-bit field of `sample` becomes `code`.  Entries of a list touch distinct samples, so the field still holds what the row was
// filled with whatever the neighbours in its dword are doing: one read, and ONE atomic that flips the bits that differ (none when
// the list repeats the row's value) -- the lists of a reference-written file are a few hundred million entries per million
// variants, and two atomics each was what the kernel spent its time on.
__device__ __forceinline__ void set_field(uint32_t* row, uint32_t sample, uint32_t code) {
  const uint32_t sh = 2 * (sample & 15);
  uint32_t* w = row + (sample >> 4);
  const uint32_t old = (load_l2(w) >> sh) & 3u;
  if (old != code) {
    atomicXor(w, (old ^ code) << sh);
  }
}

// exclusive prefix of one number per thread (and the total), through LDS
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive(uint32_t mine, uint32_t* s_tmp, uint32_t tid, uint32_t* total) {
  // inclusive scan inside the wave (shuffles), the waves' totals through LDS
  const uint32_t lane = tid & 63, wave = tid >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (uint32_t off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(incl, off, 64);
    incl += (lane >= off) ? y : 0u;
  }
  __syncthreads();  // (s_tmp may still be read from the previous use)
  if (lane == 63) {
    s_tmp[wave] = incl;
  }
  __syncthreads();
  uint32_t before = incl - mine, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < NT / 64; ++w) {
    const uint32_t x = s_tmp[w];
    before += (w < wave) ? x : 0u;
    all += x;
  }
  *total = all;
  return before;
}

// The header of a difflist, parsed by every thread; the thread's share of the groups and where its delta bytes start.
struct Difflist {
  uint32_t L, G, idw;
  const uint8_t* first_ids;
  const uint8_t* sizes;
  const uint8_t* vals;  // nullptr: ids only
  const uint8_t* deltas;
  uint32_t g0, g1;      // this thread's groups
  const uint8_t* mine;  // its first delta byte
};

// false: malformed (or L == 0: then c.p is the list's end and D.L == 0)
template <int NT>
__device__ __forceinline__ bool difflist_open(ByteCursor& c, uint32_t sample_ct, bool with_values, uint32_t* s_tmp, uint32_t tid, Difflist* D) {
  D->L = c.varint();
  D->G = 0;
  D->g0 = D->g1 = 0;
  if ((!c.ok) || (D->L > sample_ct)) {
    c.ok = false;
    return false;
  }
  if (!D->L) {
    return true;
  }
  D->G = (D->L + 63) / 64;
  D->idw = id_width(sample_ct);
  D->first_ids = c.p;
  if (!c.skip(static_cast<uint64_t>(D->G) * D->idw)) {
    return false;
  }
  D->sizes = c.p;
  if (!c.skip(D->G - 1)) {
    return false;
  }
  D->vals = nullptr;
  if (with_values) {
    D->vals = c.p;
    if (!c.skip((D->L + 3) / 4)) {
      return false;
    }
  }
  D->deltas = c.p;
  const uint32_t gpt = (D->G + NT - 1) / NT;
  D->g0 = (tid * gpt < D->G) ? tid * gpt : D->G;
  D->g1 = (D->g0 + gpt < D->G) ? D->g0 + gpt : D->G;
  uint32_t bytes = 0;
  for (uint32_t g = D->g0; g < D->g1; ++g) {
    if (g + 1 < D->G) {
      bytes += static_cast<uint32_t>(D->sizes[g]) + 63u;
    }
  }
  uint32_t total;
  const uint32_t before = block_exclusive<NT>(bytes, s_tmp, tid, &total);
  if (static_cast<uint64_t>(c.end - D->deltas) < total) {
    c.ok = false;
    return false;
  }
  D->mine = D->deltas + before;
  return true;
}

// f(sample id, entry index) for the entries of this thread's groups.  Returns false on malformed input.  *list_end (written by
// the thread that owns the last group) = first byte behind the list.
template <class F>
__device__ __forceinline__ bool difflist_walk(const Difflist& D, const uint8_t* rec_end, uint32_t sample_ct, const uint8_t** list_end, F f) {
  ByteCursor q{D.mine, rec_end, true};
  for (uint32_t g = D.g0; g < D.g1; ++g) {
    uint32_t id = read_le(D.first_ids + static_cast<uint64_t>(g) * D.idw, D.idw);
    const uint32_t k0 = g * 64;
    const uint32_t kend = (D.L < k0 + 64) ? D.L : (k0 + 64);
    const uint8_t* gstart = q.p;
    for (uint32_t k = k0; k < kend; ++k) {
      if (k != k0) {
        id += q.varint();
        if (!q.ok) {
          return false;
        }
      }
      if (id >= sample_ct) {
        return false;
      }
      f(id, k);
    }
    if (g + 1 < D.G) {
      if (static_cast<uint32_t>(q.p - gstart) != static_cast<uint32_t>(D.sizes[g]) + 63u) {
        return false;  // (the size bytes are what lets the groups be read independently: they have to be right)
      }
    } else {
      *list_end = q.p;
    }
  }
  return true;
}

__device__ __forceinline__ uint32_t packed_get(const uint8_t* base, uint64_t idx, uint32_t width_bits) {
  if (!width_bits) {
    return 0;
  }
  if (width_bits == 8) {
    return base[idx];
  }
  const uint64_t bit = idx * width_bits;
  return (static_cast<uint32_t>(base[bit >> 3]) >> (bit & 7)) & ((1u << width_bits) - 1u);
}

// ... the same on a row staged in LDS (rows of up to kPgenLdsRowBytes: the record is assembled there, its list goes in with LDS
// atomics -- two orders of magnitude cheaper than atomics at the L2 -- and the finished row is written out once, coalesced)
__device__ __forceinline__ void set_field_lds(uint32_t* row, uint32_t sample, uint32_t code) {
  const uint32_t sh = 2 * (sample & 15);
  uint32_t* w = row + (sample >> 4);
  const uint32_t old = (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 3u;
  if (old != code) {
    __hip_atomic_fetch_xor(w, (old ^ code) << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <bool LDS>
__global__ __launch_bounds__(kThreads) void pgen_main_kernel(PgenDecodeArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t row_lds[];
  __shared__ uint32_t s_tmp[kThreads];
  __shared__ int s_bad;
  const uint32_t v = blockIdx.x;
  const PgenRecDesc R = A.recs[v];
  const uint32_t type = R.vrtype & 7u;
  const bool is_ld = (type == 2) || (type == 3);
  if (is_ld != (A.pass == 1)) {
    return;
  }
  const uint32_t tid = threadIdx.x;
  if (tid == 0) {
    s_bad = 0;
  }
  uint32_t* gout = reinterpret_cast<uint32_t*>(A.rows + static_cast<uint64_t>(v) * A.stride);
  uint32_t* out = LDS ? row_lds : gout;  // where the row is assembled
  const uint32_t row_dwords = static_cast<uint32_t>(A.stride / 4);
  const uint32_t n = A.sample_ct;
  const uint64_t nb = (static_cast<uint64_t>(n) + 3) / 4;
  ByteCursor c{A.bytes + R.off, A.bytes + R.off + R.len, true};
  bool bad = false;
  bool has_list = false;
  switch (type) {
    case 0:
      if (R.len < nb) {
        bad = true;
        break;
      }
#pragma unroll 4
      for (uint32_t d = tid; d < row_dwords; d += kThreads) {
        out[d] = dword_of_bytes(c.p, nb, d);
      }
      c.p += nb;
      break;
    case 1: {
      // one bit per sample: byte 0 = 4 * low code + (high - low); a set bit = the high code (pgen_spec.tex:437-452)
      const uint32_t code = c.u8();
      const uint32_t low = code >> 2;
      const uint32_t high = low + (code & 3u);
      const uint64_t bit_bytes = (static_cast<uint64_t>(n) + 7) / 8;
      if ((!c.ok) || (high > 3) || (high == low) || (static_cast<uint64_t>(c.end - c.p) < bit_bytes)) {
        bad = true;
        break;
      }
      for (uint32_t d = tid; d < row_dwords; d += kThreads) {
        uint32_t bits = 0;
        if (2ull * d < bit_bytes) {
          bits = c.p[2ull * d];
          if (2ull * d + 1 < bit_bytes) {
            bits |= static_cast<uint32_t>(c.p[2ull * d + 1]) << 8;
          }
        }
        out[d] = low * 0x55555555u + spread16(bits) * (high - low);
      }
      c.p += bit_bytes;
      has_list = true;
      break;
    }
    case 2:
    case 3: {
      const uint32_t* base = nullptr;
      if (R.base == kPgenBaseCarried) {
        base = reinterpret_cast<const uint32_t*>(A.carried_base);
      } else if (R.base < A.n) {
        base = reinterpret_cast<const uint32_t*>(A.rows + static_cast<uint64_t>(R.base) * A.stride);
      }
      if (!base) {
        bad = true;
        break;
      }
      for (uint32_t d = tid; d < row_dwords; d += kThreads) {
        out[d] = base[d];
      }
      has_list = true;
      break;
    }
    case 4:
    case 6:
    case 7: {
      const uint32_t fill = ((type == 4) ? 0u : ((type == 6) ? 2u : 3u)) * 0x55555555u;
      for (uint32_t d = tid; d < row_dwords; d += kThreads) {
        out[d] = fill;
      }
      has_list = true;
      break;
    }
    default:  // 5: reserved; the reference decodes it as all hom-REF (pgenlib_read.cc:2740-2742)
      for (uint32_t d = tid; d < row_dwords; d += kThreads) {
        out[d] = 0;
      }
      break;
  }
  __syncthreads();  // (the row's plain stores are visible to the workgroup's atomics; s_bad is initialised)
  const uint8_t* main_end = c.p;
  bool owns_end = (tid == 0);
  if (has_list && !bad) {
    Difflist D;
    if (!difflist_open<kThreads>(c, n, true, s_tmp, tid, &D)) {
      bad = true;
    } else if (D.L) {
      owns_end = (D.g0 < D.g1) && (D.g1 == D.G);
      const uint8_t* vals = D.vals;
      if (!difflist_walk(D, c.end, n, &main_end, [&](uint32_t id, uint32_t k) {
            const uint32_t code = (static_cast<uint32_t>(vals[k >> 2]) >> (2 * (k & 3))) & 3u;
            if constexpr (LDS) {
              set_field_lds(row_lds, id, code);
            } else {
              set_field(gout, id, code);
            }
          })) {
        bad = true;
      }
    } else {
      main_end = c.p;
    }
  }
  if (bad) {
    s_bad = 1;
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      atomicCAS(A.error, 0, static_cast<int>(v) + 1);
    }
    return;
  }
  if (owns_end) {
    A.main_end[v] = static_cast<uint64_t>(main_end - A.bytes);
  }
  // 0 <-> 2 for type 3 (GenovecInvertUnsafe), then the bits behind the last sample are cleared whatever the record said
  for (uint32_t d = tid; d < row_dwords; d += kThreads) {
    uint32_t w = LDS ? row_lds[d] : load_l2(gout + d);
    if (type == 3) {
      w ^= ((~w) << 1) & 0xaaaaaaaau;
    }
    const uint64_t s0 = 16ull * d;
    if (s0 >= n) {
      w = 0;
    } else if (s0 + 16 > n) {
      w &= (1u << (2 * (n - static_cast<uint32_t>(s0)))) - 1u;
    }
    gout[d] = w;
  }
}

// ---- auxiliary track 1 ------------------------------------------------------------------------------------------
// One patch set of the track (category 1: the main track's code-1 samples, REF/ALT1 -> REF/ALTx; category 2: its code-2
// samples, ALT1/ALT1 -> ALTx/ALTy), located and ready to be walked by the workgroup.
struct PatchSet {
  uint32_t fmt;           // 0 bit array over the category's samples, 1 sample-id list, 15 none
  const uint8_t* bits;    // fmt 0
  uint32_t k_first;       // fmt 0: index (within the category) of this thread's first category sample ...
  uint32_t rank_first;    // ... and how many patched samples precede it
  Difflist D;             // fmt 1
  uint32_t patched;       // entries
  const uint8_t* vals;    // the packed allele codes behind the set
};

struct Alleles {
  uint32_t lo, hi;
};

__device__ __forceinline__ Alleles patch_alleles(const PatchSet& S, uint32_t cat, uint32_t allele_ct, uint32_t idx) {
  const uint32_t alt_ct = allele_ct - 1;
  Alleles a;
  if (cat == 1) {
    const uint32_t w = (alt_ct == 2) ? 0u : ((alt_ct == 3) ? 1u : ((alt_ct <= 5) ? 2u : ((alt_ct <= 17) ? 4u : 8u)));
    a.lo = 0;
    a.hi = 2 + packed_get(S.vals, idx, w);
  } else if (alt_ct == 2) {
    const uint32_t both = (static_cast<uint32_t>(S.vals[idx >> 3]) >> (idx & 7)) & 1u;
    a.lo = both ? 2u : 1u;
    a.hi = 2;
  } else {
    const uint32_t w = (alt_ct <= 4) ? 2u : ((alt_ct <= 16) ? 4u : 8u);
    a.lo = 1 + packed_get(S.vals, 2ull * idx, w);
    a.hi = 1 + packed_get(S.vals, 2ull * idx + 1, w);
  }
  return a;
}

__device__ __forceinline__ uint64_t patch_value_bytes(uint32_t cat, uint32_t allele_ct, uint32_t patched) {
  const uint32_t alt_ct = allele_ct - 1;
  if (cat == 1) {
    const uint32_t w = (alt_ct == 2) ? 0u : ((alt_ct == 3) ? 1u : ((alt_ct <= 5) ? 2u : ((alt_ct <= 17) ? 4u : 8u)));
    return (static_cast<uint64_t>(patched) * w + 7) / 8;
  }
  if (alt_ct == 2) {
    return (static_cast<uint64_t>(patched) + 7) / 8;
  }
  const uint32_t w = (alt_ct <= 4) ? 2u : ((alt_ct <= 16) ? 4u : 8u);
  return (static_cast<uint64_t>(patched) * 2 * w + 7) / 8;
}

// bits [k, k + nb) of a bit array (nb <= 16), read as ONE (unaligned) dword where the record has four bytes left, byte by byte
// at its very end: the category samples of a dword of the row own consecutive bits of the patch set's array
__device__ __forceinline__ uint32_t bit_window(const uint8_t* bits, const uint8_t* rec_end, uint32_t k, uint32_t nb) {
  const uint8_t* p = bits + (k >> 3);
  uint32_t w = 0;
  if (p + 4 <= rec_end) {
    w = *reinterpret_cast<const u32_unaligned*>(p);
  } else {
    for (uint32_t b = 0; (b < 4) && (p + b < rec_end); ++b) {
      w |= static_cast<uint32_t>(p[b]) << (8 * b);
    }
  }
  return (w >> (k & 7)) & ((1u << nb) - 1u);
}

// category mask of a dword of main-track codes: bit 2 s set iff sample s has code `cat` (1 or 2)
__device__ __forceinline__ uint32_t cat_mask(uint32_t w, uint32_t cat) {
  const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
  return (cat == 1) ? (lo & ~hi) : (hi & ~lo);
}

// bit `idx` of a track that ends at `end` (*bad: the record is too short for it)
__device__ __forceinline__ uint32_t track_bit(const uint8_t* p, const uint8_t* end, uint64_t idx, bool* bad) {
  const uint8_t* q = p + (idx >> 3);
  if (q >= end) {
    *bad = true;
    return 0;
  }
  return (static_cast<uint32_t>(*q) >> (idx & 7)) & 1u;
}
// number of set bits among bits [b0, b0 + n) of the track
__device__ __forceinline__ uint32_t track_popcount(const uint8_t* p, const uint8_t* end, uint64_t b0, uint32_t n, bool* bad) {
  uint32_t ct = 0;
  uint64_t b = b0;
  const uint64_t b1 = b0 + n;
  while ((b < b1) && (b & 7)) {
    ct += track_bit(p, end, b++, bad);
  }
  while (b + 8 <= b1) {
    const uint8_t* q = p + (b >> 3);
    if (q >= end) {
      *bad = true;
      return ct;
    }
    ct += __builtin_popcount(static_cast<uint32_t>(*q));
    b += 8;
  }
  while (b < b1) {
    ct += track_bit(p, end, b++, bad);
  }
  return ct;
}

template <bool LDS>
__global__ __launch_bounds__(kAuxThreads) void pgen_aux1_kernel(PgenDecodeArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t main_lds[];  // LDS: the main track, read five times below, staged once
  __shared__ uint32_t s_tmp[kAuxThreads];
  __shared__ int s_cnt[256];
  __shared__ int s_bad;
  __shared__ const uint8_t* s_ptr;
  __shared__ uint32_t s_maj;
  const uint32_t m = blockIdx.x;
  const uint32_t v = A.multi_rec[m];
  const PgenRecDesc R = A.recs[v];
  const uint32_t tid = threadIdx.x;
  const uint32_t n = A.sample_ct;
  const uint32_t allele_ct = R.allele_ct;
  uint32_t* row = reinterpret_cast<uint32_t*>(A.rows + static_cast<uint64_t>(v) * A.stride);
  const uint32_t n_dwords = (n + 15) / 16;
  const uint32_t dpt = (n_dwords + kAuxThreads - 1) / kAuxThreads;
  const uint32_t d0 = (tid * dpt < n_dwords) ? tid * dpt : n_dwords;
  const uint32_t d1 = (d0 + dpt < n_dwords) ? d0 + dpt : n_dwords;
  if (tid == 0) {
    s_bad = 0;
  }
  if (tid < 256) {
    s_cnt[tid] = 0;
  }
  // Every thread owns a contiguous run of the row's dwords (the rank of a sample within its category runs along the row), which
  // makes its reads of the row from global memory a chain of latencies; from LDS they are not.
  const uint32_t* mtrack = row;
  if constexpr (LDS) {
    // (one wave per SIMD: nothing hides a load's latency but the thread's own other loads -- eight in flight)
    for (uint32_t d = tid; d < n_dwords; d += 8 * kAuxThreads) {
      uint32_t v8[8];
#pragma unroll
      for (uint32_t i = 0; i < 8; ++i) {
        v8[i] = (d + i * kAuxThreads < n_dwords) ? row[d + i * kAuxThreads] : 0u;
      }
#pragma unroll
      for (uint32_t i = 0; i < 8; ++i) {
        if (d + i * kAuxThreads < n_dwords) {
          main_lds[d + i * kAuxThreads] = v8[i];
        }
      }
    }
    __syncthreads();
    mtrack = main_lds;
  }
  // ---- the main track's categories: counts, and each thread's rank among them
  // (A.sample_mask, optional: the engine's samples are a SUBSET of the file's -- the founders --, and the allele counts that choose the
  // major allele run over them alone (the allele-frequency pass, plink2_filter.cc:2113-2153, counts founders); the ranks that locate
  // a sample's patch run over all of the file's samples either way)
  const uint32_t* __restrict__ smask = A.sample_mask;
  auto mask_of_dword = [&](uint32_t d) -> uint32_t {  // the 16 samples of code dword d, one bit each in the even positions
    return smask ? spread16(smask[d >> 1] >> (16 * (d & 1))) : 0x55555555u;
  };
  auto in_mask = [&](uint32_t sample) -> bool { return (!smask) || ((smask[sample >> 5] >> (sample & 31)) & 1u); };
  uint32_t my1 = 0, my2 = 0, my3 = 0, mk1 = 0, mk2 = 0, mk3 = 0;
  for (uint32_t d = d0; d < d1; ++d) {
    const uint32_t w = mtrack[d];
    const uint32_t c1 = cat_mask(w, 1), c2 = cat_mask(w, 2), c3 = w & (w >> 1) & 0x55555555u;
    my1 += __popc(c1);
    my2 += __popc(c2);
    my3 += __popc(c3);
    if (smask) {
      const uint32_t mm = mask_of_dword(d);
      mk1 += __popc(c1 & mm);
      mk2 += __popc(c2 & mm);
      mk3 += __popc(c3 & mm);
    }
  }
  uint32_t n1, n2, n3;
  const uint32_t pre1 = block_exclusive<kAuxThreads>(my1, s_tmp, tid, &n1);
  const uint32_t pre2 = block_exclusive<kAuxThreads>(my2, s_tmp, tid, &n2);
  (void)block_exclusive<kAuxThreads>(my3, s_tmp, tid, &n3);
  // genotype counts over the samples that count: f0 hom-REF, f1 REF/ALT1 (or a patched het), f2 ALT1/ALT1 (or patched)
  uint32_t f1 = n1, f2 = n2, f3 = n3, f_all = n;
  if (smask) {
    (void)block_exclusive<kAuxThreads>(mk1, s_tmp, tid, &f1);
    (void)block_exclusive<kAuxThreads>(mk2, s_tmp, tid, &f2);
    (void)block_exclusive<kAuxThreads>(mk3, s_tmp, tid, &f3);
    f_all = A.mask_ct;
  }
  const uint32_t f0 = f_all - f1 - f2 - f3;
  bool bad = (allele_ct < 3) || (allele_ct > 255);
  // ---- locate the two patch sets
  PatchSet S[2];
  S[0].fmt = S[1].fmt = 15;
  S[0].patched = S[1].patched = 0;
  const uint8_t* rec_end = A.bytes + R.off + R.len;
  const uint8_t* aux2 = A.bytes + A.main_end[v];  // where the track behind this one begins (the hardcall-phase track of --indep-pairphase records)
  if ((R.vrtype & 8u) && !bad) {
    ByteCursor c{A.bytes + A.main_end[v], rec_end, true};
    const uint32_t fmt = c.u8();
    bad = !c.ok;
#pragma unroll
    for (uint32_t cat = 1; cat <= 2; ++cat) {
      if (bad) {
        break;
      }
      PatchSet& P = S[cat - 1];
      P.fmt = (cat == 1) ? (fmt & 15u) : (fmt >> 4);
      const uint32_t ncat = (cat == 1) ? n1 : n2;
      const uint32_t pre = (cat == 1) ? pre1 : pre2;
      const uint32_t mine = (cat == 1) ? my1 : my2;
      if (P.fmt == 15) {
        continue;
      }
      if (P.fmt == 0) {
        P.bits = c.p;
        if (!c.skip((static_cast<uint64_t>(ncat) + 7) / 8)) {
          bad = true;
          break;
        }
        // how many of this thread's category samples are patched: per dword of the row one window of the bit array
        uint32_t set = 0;
        {
          uint32_t k = pre;
          for (uint32_t d = d0; d < d1; d += kChunk) {
            uint32_t kk[kChunk], nbits[kChunk];
#pragma unroll
            for (uint32_t i = 0; i < kChunk; ++i) {
              nbits[i] = (d + i < d1) ? __popc(cat_mask(mtrack[d + i], cat)) : 0u;
              kk[i] = k;
              k += nbits[i];
            }
#pragma unroll
            for (uint32_t i = 0; i < kChunk; ++i) {  // (independent loads: in flight together)
              set += nbits[i] ? __popc(bit_window(P.bits, rec_end, kk[i], nbits[i])) : 0u;
            }
          }
        }
        P.k_first = pre;
        P.rank_first = block_exclusive<kAuxThreads>(set, s_tmp, tid, &P.patched);
      } else if (P.fmt == 1) {
        if (!difflist_open<kAuxThreads>(c, n, false, s_tmp, tid, &P.D)) {
          bad = true;
          break;
        }
        P.patched = P.D.L;
        if (P.D.L) {
          // where the list ends only the walk of its last group tells
          const uint8_t* list_end = nullptr;
          const bool last = (P.D.g0 < P.D.g1) && (P.D.g1 == P.D.G);
          Difflist tail = P.D;
          bool walked = true;
          if (last) {
            // this thread's earlier groups have known sizes: skip to the last one
            for (uint32_t g = tail.g0; g + 1 < tail.g1; ++g) {
              tail.mine += static_cast<uint32_t>(tail.sizes[g]) + 63u;
            }
            tail.g0 = tail.g1 - 1;
            walked = difflist_walk(tail, rec_end, n, &list_end, [](uint32_t, uint32_t) {});
          }
          __syncthreads();
          if (last) {
            s_ptr = walked ? list_end : nullptr;
          }
          __syncthreads();
          if (!s_ptr) {
            bad = true;
            break;
          }
          c.p = s_ptr;
        }
      } else {
        bad = true;  // reserved format
        break;
      }
      P.vals = c.p;
      if (!c.skip(patch_value_bytes(cat, allele_ct, P.patched))) {
        bad = true;
      }
    }
    aux2 = c.p;
  }
  if (bad) {
    s_bad = 1;
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      atomicCAS(A.error, 0, static_cast<int>(v) + 1);
    }
    return;
  }
  // f(sample, category, alleles) over this thread's share of the patches of both sets; `codes` = the main-track dword of the
  // sample for the bit-array form (its own), looked up for the list form
  auto for_each_patch = [&](bool lists, auto f) -> bool {
    bool ok = true;
#pragma unroll
    for (uint32_t cat = 1; cat <= 2; ++cat) {
      const PatchSet& P = S[cat - 1];
      if ((P.fmt == 1) && lists && P.D.L) {
        const uint8_t* unused = nullptr;
        ok = difflist_walk(P.D, rec_end, n, &unused, [&](uint32_t id, uint32_t k) { f(id, cat, patch_alleles(P, cat, allele_ct, k)); }) && ok;
      }
    }
    return ok;
  };
  // ---- pass 1: allele counts.  Main track: 2 n0 + n1 REF copies, n1 + 2 n2 ALT1 copies; a category-1 patch turns one ALT1
  // copy into ALTx, a category-2 patch turns two into ALTx + ALTy.
  // (the first four alleles are counted in registers and added to the workgroup's histogram once per thread: tens of thousands of
  // LDS atomics on the one or two addresses nearly every patch names were most of this kernel's time)
  int local_cnt0 = 0, local_cnt1 = 0, local_cnt2 = 0, local_cnt3 = 0;
  auto bump = [&](uint32_t a, int delta) {
    if (a == 1) {
      local_cnt1 += delta;
    } else if (a == 2) {
      local_cnt2 += delta;
    } else if (a == 3) {
      local_cnt3 += delta;
    } else if (a == 0) {
      local_cnt0 += delta;
    } else {
      atomicAdd(&s_cnt[a], delta);
    }
  };
  auto count_patch = [&](uint32_t sample, uint32_t cat, Alleles a) {
    if ((a.hi >= allele_ct) || (a.lo >= allele_ct)) {
      s_bad = 1;
      return;
    }
    if (!in_mask(sample)) {
      return;  // (checked, not counted)
    }
    if (cat == 1) {
      bump(1, -1);
      bump(a.hi, 1);
    } else {
      bump(1, -2);
      bump(a.lo, 1);
      bump(a.hi, 1);
    }
  };
  // bit-array sets: the thread's own dwords, kChunk at a time (w[i] = main-track dword d + i, cnt of them valid), samples in order;
  // f(i, sample, category, alleles)
  auto walk_chunk = [&](uint32_t d, const uint32_t (&w)[kChunk], uint32_t cnt, uint32_t (&k)[2], uint32_t (&rank)[2], auto f) {
    uint32_t msk[2][kChunk], win[2][kChunk];
#pragma unroll
    for (uint32_t cat = 1; cat <= 2; ++cat) {
      const PatchSet& P = S[cat - 1];
      uint32_t kk[kChunk], nbits[kChunk];
#pragma unroll
      for (uint32_t i = 0; i < kChunk; ++i) {
        msk[cat - 1][i] = ((P.fmt == 0) && (i < cnt)) ? cat_mask(w[i], cat) : 0u;
        nbits[i] = __popc(msk[cat - 1][i]);
        kk[i] = k[cat - 1];
        k[cat - 1] += nbits[i];
      }
#pragma unroll
      for (uint32_t i = 0; i < kChunk; ++i) {  // bit t of a window: the dword's t-th sample of the category is patched
        win[cat - 1][i] = nbits[i] ? bit_window(P.bits, rec_end, kk[i], nbits[i]) : 0u;
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < kChunk; ++i) {
#pragma unroll
      for (uint32_t cat = 1; cat <= 2; ++cat) {
        const PatchSet& P = S[cat - 1];
        uint32_t wv = win[cat - 1][i];
        while (wv) {
          const uint32_t t = __builtin_ctz(wv);
          wv &= wv - 1;
          uint32_t mm = msk[cat - 1][i];
          for (uint32_t j = 0; j < t; ++j) {
            mm &= mm - 1;
          }
          f(i, 16 * (d + i) + (__builtin_ctz(mm) >> 1), cat, patch_alleles(P, cat, allele_ct, rank[cat - 1]++));
        }
      }
    }
  };
  {
    uint32_t k[2] = {S[0].k_first, S[1].k_first}, rank[2] = {S[0].rank_first, S[1].rank_first};
    for (uint32_t d = d0; d < d1; d += kChunk) {
      uint32_t w[kChunk];
      const uint32_t cnt = (d1 - d < kChunk) ? (d1 - d) : kChunk;
#pragma unroll
      for (uint32_t i = 0; i < kChunk; ++i) {
        w[i] = (i < cnt) ? mtrack[d + i] : 0u;
      }
      walk_chunk(d, w, cnt, k, rank, [&](uint32_t /*i*/, uint32_t sample, uint32_t cat, Alleles a) { count_patch(sample, cat, a); });
    }
    // list sets: the sample must be in the set's category (checked while the row still holds the main track)
    const bool ok = for_each_patch(true, [&](uint32_t id, uint32_t cat, Alleles a) {
      const uint32_t code = (mtrack[id >> 4] >> (2 * (id & 15))) & 3u;
      if (code != cat) {
        s_bad = 1;
      }
      count_patch(id, cat, a);
    });
    if (!ok) {
      s_bad = 1;
    }
    if (local_cnt0) {
      atomicAdd(&s_cnt[0], local_cnt0);
    }
    if (local_cnt1) {
      atomicAdd(&s_cnt[1], local_cnt1);
    }
    if (local_cnt2) {
      atomicAdd(&s_cnt[2], local_cnt2);
    }
    if (local_cnt3) {
      atomicAdd(&s_cnt[3], local_cnt3);
    }
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) {
      atomicCAS(A.error, 0, static_cast<int>(v) + 1);
    }
    return;
  }
  if (tid == 0) {
    // ComputeAlleleFreqs (plink2_filter.cc:2113-2153): freq[a] = count[a] * (1 / total) for all alleles but the last, 1 / k each
    // when nothing is observed; GetMajIdxMulti (plink2_common.cc:1042-1070); GetAlleleFreq (plink2_common.h:584-593)
    const uint64_t c_ref = 2ull * f0 + f1;
    const int64_t c_alt1 = static_cast<int64_t>(f1) + 2ll * f2 + s_cnt[1];
    auto count_of = [&](uint32_t a) -> uint64_t { return (a == 0) ? c_ref : ((a == 1) ? static_cast<uint64_t>(c_alt1) : static_cast<uint64_t>(s_cnt[a])); };
    const uint64_t tot = 2ull * (static_cast<uint64_t>(f0) + f1 + f2);
    const double tot_recip = tot ? __ddiv_rn(1.0, static_cast<double>(tot)) : 0.0;
    const double none = __ddiv_rn(1.0, static_cast<double>(allele_ct));
    auto freq_of = [&](uint32_t a) -> double { return tot ? __dmul_rn(static_cast<double>(count_of(a)), tot_recip) : none; };
    uint32_t maj;
    const double ref_freq = freq_of(0);
    if (ref_freq >= 0.5) {
      maj = 0;
    } else {
      const double alt1_freq = freq_of(1);
      if (alt1_freq >= 0.5) {
        maj = 1;
      } else {
        maj = 1;
        double max_freq = alt1_freq;
        if (ref_freq >= alt1_freq) {
          maj = 0;
          max_freq = ref_freq;
        }
        double tot_nonlast = __dadd_rn(ref_freq, alt1_freq);
        for (uint32_t a = 2; a + 1 < allele_ct; ++a) {
          const double f = freq_of(a);
          if (f > max_freq) {
            maj = a;
            max_freq = f;
          }
          tot_nonlast = __dadd_rn(tot_nonlast, f);
        }
        if (__dadd_rn(max_freq, tot_nonlast) < 1.0 - kSmallEpsilon) {
          maj = allele_ct - 1;
        }
      }
    }
    double mf;
    if (maj + 1 < allele_ct) {
      mf = freq_of(maj);
    } else {
      double last = __dsub_rn(1.0, freq_of(0));
      for (uint32_t a = 1; a + 1 < allele_ct; ++a) {
        last = __dsub_rn(last, freq_of(a));
      }
      mf = (last > 0.0) ? last : 0.0;
    }
    s_maj = maj;
    A.maj_idx[m] = maj;
    A.maj_freq[m] = mf;
    A.row_inverse[v] = 1;
  }
  __syncthreads();
  const uint32_t maj = s_maj;
  const bool phased = A.phase_off != 0;
  if ((maj == 0) && !phased) {
    return;  // REF is the major allele: the main track already counts the copies of the others
  }
  // ---- --indep-pairphase (PgrGetInv1P -> Get1MP, pgenlib_read.cc:7016, :6962): the hardcall-phase track behind this one holds one
  // phasepresent / phaseinfo bit per heterozygous call OF THE FILE, in sample order -- the main track's code-1 calls (REF/ALTx) and the
  // category-2 patches with two different alleles (ALTx/ALTy; ReadGenovecHphaseSubsetUnsafe: all_hets |= aux1b hets, :5497-5510) --, while
  // the collapsed row has a het where exactly one allele of the call is the major one.  Step 1, while the main track is still what the file
  // holds: the file's het calls as a bitmap in the row's phase area (16 bits per code dword: each thread its own dwords; the list-form
  // patches land where they fall, hence the atomics).
  uint16_t* phase16 = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(row) + A.phase_off);
  if (phased) {
    for (uint32_t d = d0; d < d1; ++d) {
      phase16[d] = 0;
    }
    __syncthreads();
    uint32_t* het_words = reinterpret_cast<uint32_t*>(phase16);
    auto mark_het = [&](uint32_t sample, uint32_t cat, Alleles a) {
      if ((cat == 2) && (a.lo != a.hi)) {
        atomicOr(het_words + (sample >> 5), 1u << (sample & 31));
      }
    };
    {
      uint32_t k[2] = {S[0].k_first, S[1].k_first}, rank[2] = {S[0].rank_first, S[1].rank_first};
      for (uint32_t d = d0; d < d1; d += kChunk) {
        uint32_t w[kChunk];
        const uint32_t cnt = (d1 - d < kChunk) ? (d1 - d) : kChunk;
#pragma unroll
        for (uint32_t i = 0; i < kChunk; ++i) {
          w[i] = (i < cnt) ? mtrack[d + i] : 0u;
        }
        walk_chunk(d, w, cnt, k, rank, [&](uint32_t /*i*/, uint32_t sample, uint32_t cat, Alleles a) { mark_het(sample, cat, a); });
      }
      (void)for_each_patch(true, mark_het);
    }
    __syncthreads();
    for (uint32_t d = d0; d < d1; ++d) {
      uint32_t c1 = cat_mask(mtrack[d], 1);  // bit 2 s -> bit s
      c1 = (c1 | (c1 >> 1)) & 0x33333333u;
      c1 = (c1 | (c1 >> 2)) & 0x0f0f0f0fu;
      c1 = (c1 | (c1 >> 4)) & 0x00ff00ffu;
      c1 = (c1 | (c1 >> 8)) & 0x0000ffffu;
      phase16[d] = static_cast<uint16_t>(phase16[d] | c1);
    }
    __syncthreads();  // (in the form without the LDS copy the collapse below rewrites the main track other threads have just read)
  }
  // ---- pass 2: the row as copies of non-major alleles.  Main-track codes first (major = ALT1: 2 - code; a later ALT: every
  // call is two non-major copies), then the patched samples.
  auto code_of = [&](Alleles a) -> uint32_t { return 2u - ((a.lo == maj) ? 1u : 0u) - ((a.hi == maj) ? 1u : 0u); };
  if (maj != 0) {
    uint32_t k[2] = {S[0].k_first, S[1].k_first}, rank[2] = {S[0].rank_first, S[1].rank_first};
    for (uint32_t d = d0; d < d1; d += kChunk) {
      uint32_t w[kChunk], t[kChunk];
      const uint32_t cnt = (d1 - d < kChunk) ? (d1 - d) : kChunk;
#pragma unroll
      for (uint32_t i = 0; i < kChunk; ++i) {
        w[i] = (i < cnt) ? mtrack[d + i] : 0u;
        t[i] = (maj == 1) ? (w[i] ^ (((~w[i]) << 1) & 0xaaaaaaaau)) : (0xaaaaaaaau | (w[i] & (w[i] >> 1) & 0x55555555u));
      }
      walk_chunk(d, w, cnt, k, rank, [&](uint32_t i, uint32_t sample, uint32_t /*cat*/, Alleles a) {
        const uint32_t sh = 2 * (sample & 15);
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) {  // (i is not a compile-time index)
          if (j == i) {
            t[j] = (t[j] & ~(3u << sh)) | (code_of(a) << sh);
          }
        }
      });
#pragma unroll
      for (uint32_t i = 0; i < kChunk; ++i) {
        if (i < cnt) {
          const uint64_t s0 = 16ull * (d + i);
          uint32_t tv = t[i];
          if (s0 + 16 > n) {
            tv &= (1u << (2 * (n - static_cast<uint32_t>(s0)))) - 1u;
          }
          row[d + i] = tv;
        }
      }
    }
  }
  if (maj != 0) {
    __syncthreads();
    (void)for_each_patch(true, [&](uint32_t id, uint32_t /*cat*/, Alleles a) { set_field(row, id, code_of(a)); });
  }
  if (!phased) {
    return;
  }
  // ---- step 2: the phase bits of the collapsed row.  A het call of the file consumes its phasepresent bit (and, when that is set, its
  // phaseinfo bit) whether or not it is still a het after the collapse; a het of the collapsed row takes "the counted (non-major) allele is
  // on the first haplotype" from it.  Get1MP hands the file's phaseinfo through unchanged -- "the HIGHER allele of the call is on the first
  // haplotype" -- and HapsplitMustPhased reads it as "the counted allele is": right when the major allele is REF (the counted one is then
  // the higher), and for a later major allele the reference's reading, which plink2-hip reproduces (p2h_tables.cpp,
  // multiallelic_inverse_row; tests/test_pairphase.py), comes to the complement: bit ^ (major != REF).  A het of the collapsed row
  // without phase reports the record (the reference's "not fully phased", plink2_ld.cc:2045-2049).
  __syncthreads();  // (the list-form patches above wrote other threads' dwords)
  uint32_t hets = 0;
  for (uint32_t d = d0; d < d1; ++d) {
    hets += __popc(static_cast<uint32_t>(phase16[d]));
  }
  uint32_t het_ct = 0;
  const uint32_t het_before = block_exclusive<kAuxThreads>(hets, s_tmp, tid, &het_ct);
  const bool has_track = (R.vrtype & 0x10u) != 0;
  bool pbad = false, unphased = false, explicit_present = false;
  const uint8_t* info = aux2;
  uint64_t info_bit = 1;
  uint32_t present_before = het_before;
  if (has_track && het_ct) {
    if ((aux2 < A.bytes + R.off) || (aux2 >= rec_end)) {
      pbad = true;
    } else {
      explicit_present = (aux2[0] & 1u) != 0;
    }
    if (explicit_present) {
      const uint32_t mine = track_popcount(aux2, rec_end, 1ull + het_before, hets, &pbad);
      uint32_t present_ct = 0;
      present_before = block_exclusive<kAuxThreads>(mine, s_tmp, tid, &present_ct);
      info = aux2 + 1 + het_ct / 8;
      info_bit = 0;
      if ((!present_ct) || (info + (present_ct + 7) / 8 > rec_end)) {
        pbad = true;
      }
    } else if (aux2 + 1 + het_ct / 8 > rec_end) {
      pbad = true;
    }
  }
  const uint32_t flip = (maj != 0) ? 1u : 0u;
  uint32_t k = het_before, r = present_before;
  for (uint32_t d = d0; d < d1; ++d) {
    uint32_t fh = phase16[d];
    const uint32_t w = row[d];
    uint32_t bits16 = 0;
    while (fh && !pbad) {
      const uint32_t pos = static_cast<uint32_t>(__builtin_ctz(fh));
      fh &= fh - 1;
      const bool het_now = ((w >> (2 * pos)) & 3u) == 1u;
      const bool present = has_track && (explicit_present ? (track_bit(aux2, rec_end, 1ull + k, &pbad) != 0) : true);
      ++k;
      if (present) {
        const uint32_t b = track_bit(info, rec_end, info_bit + r, &pbad);
        ++r;
        if (het_now) {
          bits16 |= (b ^ flip) << pos;
        }
      } else if (het_now) {
        unphased = true;
      }
    }
    phase16[d] = static_cast<uint16_t>(bits16);
  }
  if (pbad) {
    atomicCAS(A.error, 0, static_cast<int>(v) + 1);
  } else if (unphased && A.unphased) {
    atomicMin(A.unphased, v);
  }
}

// ---- auxiliary track 2: phased heterozygous hard-calls (--indep-pairphase) ------------------------------------------------
// pgen_spec.tex "Phased heterozygous hard-calls"; ReadGenovecHphaseSubsetUnsafe, pgenlib_read.cc:6704; what PgrGetInv1P hands
// HapsplitMustPhased (pgenlib_read.cc:7016, pgenlib_misc.cc:1887).  Behind the main track of a record whose type byte has bit 4
// set: bit 0 of the first byte says whether a "phasepresent" bit per heterozygous call follows (bits 1 .. H of the same bytes;
// otherwise every het call is phased); then one "phaseinfo" bit per PHASED het call (set = the ALT allele on the first haplotype,
// "1|0") -- from the next byte boundary when phasepresent is stored, from bit 1 of the first byte when not.  One workgroup per
// record: count the het calls of the decoded row (a block scan gives every thread the rank of its first one), the phased ones
// among them likewise, then every thread sets the phase bits of its own samples: the row's second part (LDP_GENO_PHASED layout,
// ldprune_hip.h: phase bit of sample s = bit s % 8 of byte phase_off + s / 8), all of it written (zero where there is no het).
// A het call without phase -- no track at all, or a clear phasepresent bit -- reports its record: A.unphased <- the lowest index.
__global__ __launch_bounds__(kThreads) void pgen_phase_kernel(PgenDecodeArgs A) {
  __shared__ uint32_t s_tmp[kThreads];
  const uint32_t v = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const PgenRecDesc R = A.recs[v];
  if (R.allele_ct > 2) {
    return;  // several ALT alleles: the phase bits of the collapsed row are pgen_aux1_kernel's (its het calls are not the main track's)
  }
  const uint32_t n = A.sample_ct;
  uint8_t* rowb = A.rows + static_cast<uint64_t>(v) * A.stride;
  const uint32_t* row = reinterpret_cast<const uint32_t*>(rowb);
  uint16_t* phase = reinterpret_cast<uint16_t*>(rowb + A.phase_off);  // 16 samples (one code dword) per entry
  const uint32_t n_dw = (n + 15) / 16;
  const uint32_t per = (n_dw + kThreads - 1) / kThreads;
  const uint32_t d0 = min(tid * per, n_dw), d1 = min(d0 + per, n_dw);
  uint32_t hets = 0;
  for (uint32_t d = d0; d < d1; ++d) {
    const uint32_t w = row[d];
    hets += __builtin_popcount(w & ~(w >> 1) & 0x55555555u);
  }
  uint32_t het_ct = 0;
  const uint32_t het_before = block_exclusive<kThreads>(hets, s_tmp, tid, &het_ct);
  const bool has_track = (R.vrtype & 0x10u) != 0;
  const uint8_t* rec_end = A.bytes + R.off + R.len;
  const uint8_t* aux2 = A.bytes + A.main_end[v];
  bool bad = false, unphased = false;
  bool explicit_present = false;
  const uint8_t* info = aux2;
  uint64_t info_bit = 1;
  uint32_t present_before = het_before;
  if (has_track && het_ct) {
    if ((aux2 < A.bytes + R.off) || (aux2 >= rec_end)) {
      bad = true;
    } else {
      explicit_present = (aux2[0] & 1u) != 0;
    }
    if (explicit_present) {
      const uint32_t mine = track_popcount(aux2, rec_end, 1ull + het_before, hets, &bad);
      uint32_t present_ct = 0;
      present_before = block_exclusive<kThreads>(mine, s_tmp, tid, &present_ct);
      info = aux2 + 1 + het_ct / 8;
      info_bit = 0;
      if ((!present_ct) || (info + (present_ct + 7) / 8 > rec_end)) {
        bad = true;
      }
    } else if (aux2 + 1 + het_ct / 8 > rec_end) {
      bad = true;
    }
  } else if (het_ct && !has_track) {
    unphased = true;  // het calls and no phase track at all
  }
  uint32_t k = het_before, r = present_before;
  for (uint32_t d = d0; d < d1; ++d) {
    const uint32_t w = row[d];
    uint32_t h = w & ~(w >> 1) & 0x55555555u;
    uint32_t bits16 = 0;
    while (h && has_track && !bad) {
      const uint32_t pos = static_cast<uint32_t>(__builtin_ctz(h)) >> 1;
      h &= h - 1;
      const bool present = explicit_present ? (track_bit(aux2, rec_end, 1ull + k, &bad) != 0) : true;
      ++k;
      if (present) {
        bits16 |= track_bit(info, rec_end, info_bit + r, &bad) << pos;
        ++r;
      } else {
        unphased = true;
      }
    }
    phase[d] = static_cast<uint16_t>(bits16);
  }
  // (the padding between the codes and the phase bits, and the phase bytes behind the last code dword's two, stay as the main
  // track's kernel left them: zero)
  if (bad) {
    atomicCAS(A.error, 0, static_cast<int>(v) + 1);
  } else if (unphased && A.unphased) {
    atomicMin(A.unphased, v);
  }
}

}  // namespace

hipError_t launch_pgen_phase(const PgenDecodeArgs& a, uint32_t n_records, hipStream_t stream) {
  if (!n_records || !a.phase_off) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(pgen_phase_kernel, dim3(n_records), dim3(kThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_pgen_main(const PgenDecodeArgs& a, hipStream_t stream) {
  if (!a.n) {
    return hipSuccess;
  }
  PgenDecodeArgs p = a;
  // rows that fit the LDS beside a few other workgroups are assembled there
  static const bool lds_ok = []() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&pgen_main_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPgenLdsRowBytes)) == hipSuccess;
  }();
  const bool in_lds = lds_ok && (a.stride <= kPgenLdsRowBytes) && !a.no_lds;
  for (int pass = 0; pass < (a.any_ld ? 2 : 1); ++pass) {
    p.pass = pass;
    if (in_lds) {
      hipLaunchKernelGGL(pgen_main_kernel<true>, dim3(a.n), dim3(kThreads), static_cast<size_t>(a.stride), stream, p);
    } else {
      hipLaunchKernelGGL(pgen_main_kernel<false>, dim3(a.n), dim3(kThreads), 0, stream, p);
    }
  }
  return hipGetLastError();
}

hipError_t launch_pgen_aux1(const PgenDecodeArgs& a, hipStream_t stream) {
  if (!a.n_multi) {
    return hipSuccess;
  }
  static const bool lds_ok = []() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&pgen_aux1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPgenLdsRowBytes)) == hipSuccess;
  }();
  if (lds_ok && (a.stride <= kPgenLdsRowBytes) && !a.no_lds) {
    hipLaunchKernelGGL(pgen_aux1_kernel<true>, dim3(a.n_multi), dim3(kAuxThreads), static_cast<size_t>(a.stride), stream, a);
  } else {
    hipLaunchKernelGGL(pgen_aux1_kernel<false>, dim3(a.n_multi), dim3(kAuxThreads), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace ldp
