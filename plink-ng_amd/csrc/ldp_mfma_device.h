// ldp_mfma_device.h -- device-side pieces shared by the matrix-pipe pair kernels (ldp_pair_mfma.hip: parallelogram wave items for
// narrow bands, ldp_pair_wide.hip: 8 x 8 block tiles for wide ones): the FP4 expansion of the 2-bit genotype codes, the MFMA
// wrapper and the geometry of a staged 256-sample slice.  Device code only.
#ifndef LDP_MFMA_DEVICE_H
#define LDP_MFMA_DEVICE_H

#include "ldp_device.h"

namespace ldp {

typedef int mf_v8i __attribute__((ext_vector_type(8)));
typedef float mf_v16f __attribute__((ext_vector_type(16)));

struct Frag {
  uint32_t d[4];  // 32 E2M1 values: one lane's share (one row, 32 samples) of a 32 x 64 operand
};

// 16 samples of 2-bit codes (00 hom-REF, 01 het, 10 hom-ALT, 11 missing; sample s at bits 2s, 2s+1) -> the E2M1 nibbles of
// the odd samples of X: magnitude at nibble bit 2 (value 2.0) = !b0, sign at bit 3 = b1, i.e. x = +2 / 0 / -2 / -0.
// f(X) = (X ^ 0x44444444) & 0xCCCCCCCC; the even samples are the odd ones of X << 2.  One v_bitop3_b32 each.
__device__ __forceinline__ uint32_t fp4_x(uint32_t X) { return __builtin_amdgcn_bitop3_b32(X, 0x44444444u, 0xccccccccu, 0x28); }  // (a ^ b) & c
// call present (n = !(b0 & b1)) and homozygous (h = !b0 = |x|), both as 2.0 at nibble bit 2
__device__ __forceinline__ uint32_t fp4_n(uint32_t X) { return __builtin_amdgcn_bitop3_b32(X, X >> 1, 0x44444444u, 0x2a); }  // !(a & b) & c

// 32 samples (two code dwords) of one variant -> one lane's share of a 32 x 64 operand
__device__ __forceinline__ void fp4_of_codes(uint32_t c0, uint32_t c1, Frag& f) {
  f.d[0] = fp4_x(c0);
  f.d[1] = fp4_x(c0 << 2);
  f.d[2] = fp4_x(c1);
  f.d[3] = fp4_x(c1 << 2);
}

// C[row of a][column of b] += sum over 64 samples.  E8M0 scale 0x7e = 1/2 for both operands: (+-2 / 2) (+-2 / 2) = +-1.
constexpr int kFp4Scale = 0x7e7e7e7e;
__device__ __forceinline__ mf_v16f mfma_fp4(const Frag& a, const Frag& b, mf_v16f c) {
  const mf_v8i A = {static_cast<int>(a.d[0]), static_cast<int>(a.d[1]), static_cast<int>(a.d[2]), static_cast<int>(a.d[3]), 0, 0, 0, 0};
  const mf_v8i B = {static_cast<int>(b.d[0]), static_cast<int>(b.d[1]), static_cast<int>(b.d[2]), static_cast<int>(b.d[3]), 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, kFp4Scale, 0, kFp4Scale);
}

// ---- complete data: the ALLELE-COUNT coding ---------------------------------------------------------------------------
// The pair kernels run at the socket's power cap (profiles/r04_power_during_step.txt: 1,330-1,380 W of 1,400 W, the shader clock
// pulled from 2.4 to ~2.1 GHz), so what an MFMA costs is its energy, and that depends on the operand VALUES: on genotype data
// tools/energy_probe.hip measures the instruction 8-11 % cheaper with hom-REF = 0 (most products 0 x 0) than with hom-REF = +2
// (profiles/r04_energy_probe.txt).  A 2-bit code IS an E2M1 value when it sits in the low half of a nibble: 00 -> 0, 01 -> 0.5,
// 10 -> 1.0 (11 -> 1.5), i.e. g / 2 with g = the count of the coded allele; E8M0 block scale 0x80 = 2 on both operands makes every
// product g_i g_j.  Even samples: X & 0x33333333; odd samples: (X >> 2) & 0x33333333 -- the same three VALU per 16 samples as the
// +-2 coding.  For COMPLETE rows x = 1 - g, so the statistic the reference wants (DotprodWords, plink2_ld.cc:235-251) follows
// exactly from G = sum g_i g_j and the rows' own sums: dot = N - sum g_i - sum g_j + G = G - N + S_i + S_j (S = the row's sum of x
// in the image's orientation).  Padding samples are coded 11 and add 9 each to every G: a constant of the launch (g_bias below).
// Rows with missing calls never come here: code 11 would count as 3 (the SPARSE instantiation and the missing-call kernels keep
// the +-2 coding, where a missing call is 0).  Integer-exact while 4 N + 9 * 511 < 2^24 (kMfMaxFounders).
constexpr int kFp4ScaleG = static_cast<int>(0x80808080u);
__device__ __forceinline__ void fp4_g_of_codes(uint32_t c0, uint32_t c1, Frag& f) {
  f.d[0] = c0 & 0x33333333u;
  f.d[1] = (c0 >> 2) & 0x33333333u;
  f.d[2] = c1 & 0x33333333u;
  f.d[3] = (c1 >> 2) & 0x33333333u;
}
__device__ __forceinline__ mf_v16f mfma_fp4g(const Frag& a, const Frag& b, mf_v16f c) {
  const mf_v8i A = {static_cast<int>(a.d[0]), static_cast<int>(a.d[1]), static_cast<int>(a.d[2]), static_cast<int>(a.d[3]), 0, 0, 0, 0};
  const mf_v8i B = {static_cast<int>(b.d[0]), static_cast<int>(b.d[1]), static_cast<int>(b.d[2]), static_cast<int>(b.d[3]), 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, kFp4ScaleG, 0, kFp4ScaleG);
}
// GC = true: allele-count coding (complete data); false: the +-2 coding (a missing call is 0)
template <bool GC>
__device__ __forceinline__ void fp4_expand(uint32_t c0, uint32_t c1, Frag& f) {
  if constexpr (GC) {
    fp4_g_of_codes(c0, c1, f);
  } else {
    fp4_of_codes(c0, c1, f);
  }
}
template <bool GC>
__device__ __forceinline__ mf_v16f mfma_pair(const Frag& a, const Frag& b, mf_v16f c) {
  if constexpr (GC) {
    return mfma_fp4g(a, b, c);
  } else {
    return mfma_fp4(a, b, c);
  }
}
// ---- rows with missing calls, prune launches: allele counts g' and the MISSING flag u ----------------------------------------------
// The four-product form of the missing-call kernels (ldp_pair_mfma.hip) multiplied x (+-2 coding, 0 at a missing call) and n (call
// present: 2.0 almost everywhere).  At the power cap the operand values are what an MFMA costs (above), so since round 4 the vectors are
//   g' = the 2-bit code itself as an allele count (0, 1, 2 -- and 3 at a missing call: the code is not touched, three VALU per dword)
//   u  = 1 at a missing call (5 % of the genotypes at config 5, none elsewhere: four VALU per dword; seven in all, as x and n cost)
// both as E2M1 in the low half of a nibble (g' / 2, u / 2 = 0.5) with block scale 2.  The four products
//   P1 = g'_i.g'_j   P4 = u_i.u_j   P3 = u_i.g'_j   P2 = g'_i.u_j
// give, with z = g' - 3 u (the allele count with a missing call as 0): z.z = P1 - 3 P2 - 3 P3 + 9 P4, z_i.u_j = P2 - 3 P4,
// u_i.z_j = P3 - 3 P4, u.u = P4, and with each row's own U = its missing calls and Z = its sum of z over the same samples (from the
// records / the checkpoint slots) everything ComputeIndepPairwiseR2Components wants over the pairwise-complete samples, exactly, in
// integers (x = 1 - z on called samples): nm = n - U_i - U_j + u.u,  S1 = Z_i - z_i.u_j,  S2 = Z_j - u_i.z_j,
//   sum1 = nm - S1   sum2 = nm - S2   dot = nm - S1 - S2 + z.z         (image orientation, as before).
// Padding samples are coded 11 in every row: they drop out of z.z, z.u and u.z by themselves and add n_pad to u.u.
// Exact while 9 N < 2^24: kMfGuMaxFounders (ldp_device.h); engines with more founders keep the x / n operands.
__device__ __forceinline__ void fp4_gu_of_dword(uint32_t X, uint32_t* g_even, uint32_t* g_odd, uint32_t* u_even, uint32_t* u_odd) {
  const uint32_t Xs = X >> 2;
  *g_even = X & 0x33333333u;
  *g_odd = Xs & 0x33333333u;
  *u_even = __builtin_amdgcn_bitop3_b32(X, X >> 1, 0x11111111u, 0x80);   // a & b & c
  *u_odd = __builtin_amdgcn_bitop3_b32(Xs, Xs >> 1, 0x11111111u, 0x80);
}
__device__ __forceinline__ void fp4_gu_of_codes(uint32_t c0, uint32_t c1, Frag& fg, Frag& fu) {
  fp4_gu_of_dword(c0, &fg.d[0], &fg.d[1], &fu.d[0], &fu.d[1]);
  fp4_gu_of_dword(c1, &fg.d[2], &fg.d[3], &fu.d[2], &fu.d[3]);
}
// accumulators (P1, P4, P3, P2) of a pair over n_vis real samples (+ n_pad padding samples) and the two rows' (U, Z) over the same
// samples -> dot, nm, sum2, sum1
__device__ __forceinline__ void x_from_gu(int32_t P1, int32_t P4, int32_t P3, int32_t P2, int32_t Ui, int32_t Zi, int32_t Uj, int32_t Zj, int32_t n_vis, int32_t n_pad,
                                          int32_t* dot, int32_t* nm, int32_t* sum2, int32_t* sum1) {
  const int32_t uu = P4 - n_pad;
  const int32_t zu = P2 - 3 * P4, uz = P3 - 3 * P4;
  const int32_t zz = P1 - 3 * P2 - 3 * P3 + 9 * P4;
  const int32_t n = n_vis - Ui - Uj + uu;
  const int32_t S1 = Zi - zu, S2 = Zj - uz;
  *nm = n;
  *sum1 = n - S1;
  *sum2 = n - S2;
  *dot = n - S1 - S2 + zz;
}
// what a launch subtracts from G: N for the shift x = 1 - g, 9 per padding sample the kernel walks over (it visits whole stages of
// stage_samples; the image's rows are whole 512-sample chunks with every sample beyond founder_ct coded 11)
__device__ __forceinline__ int32_t g_bias_of(uint32_t founder_ct, uint32_t stage_samples) {
  const uint32_t visited = ((founder_ct + stage_samples - 1u) / stage_samples) * stage_samples;
  return static_cast<int32_t>(founder_ct + 9u * (visited - founder_ct));
}
// a record's sum of x in the IMAGE's orientation (the record itself is in major-allele orientation; img_differs(flags): they differ -- ALT is major and
// the row is stored as the input had it; a row the count pass stored inverted is major-oriented like its record, ldp_device.h)
__device__ __forceinline__ int32_t sum_img_of(const ldp_variant_rec& r) { return img_differs(r.flags) ? -r.sum : r.sum; }

// a record's (U, Z) over the whole row: missing calls, and the sum of the allele count over its calls (image orientation)
__device__ __forceinline__ void uz_of_rec(const ldp_variant_rec& r, uint32_t founder_ct, int32_t* U, int32_t* Z) {
  *U = static_cast<int32_t>(founder_ct - r.nm_ct);
  *Z = static_cast<int32_t>(r.nm_ct) - sum_img_of(r);
}

typedef uint32_t mf_u4 __attribute__((ext_vector_type(4)));  // (a native vector: usable as an inline-asm operand)

// ---- geometry of a stage: four k-steps = 256 samples = kCodeStageBytes contiguous bytes of a row ---------------------
// A k-step is one MFMA per product: 64 samples, lane half h supplying 32 of them (two code dwords).
// LDS image of a stage: row-block slot b, row r, four 16-byte pieces per row (piece c = bytes 16 c .. of the row's stage);
// piece c sits at slot (32 b + r) * 4 + (c ^ ((r >> 2) & 3)); lane half h reads pieces h and 2 + h: its k-step ks is dword ks of
// each.  The XOR makes the 16 lanes of every ds_read_b128 group hit 16 distinct 4-bank groups without padding, and the DMA
// (lane-linear in LDS, free per-lane global address) simply fetches the piece that belongs in its slot.
// (128-sample stages -- twice the ring depth in the same LDS -- were measured slower everywhere in round 2 and are gone.)
template <int KS>
struct StageGeom {
  static_assert(KS == 4, "256-sample stages");
  static constexpr uint32_t kRowSlots = KS;                       // 16-byte slots per row
  static constexpr uint32_t kBlockSlots = kMfBlock * KS;          // per row-block (uint4 units)
  static constexpr uint32_t kBlockDwords = kBlockSlots * 4;
  static constexpr uint32_t kInstrPerBlock2 = KS;                 // DMA instructions per TWO row-blocks (64 slots each)
  static constexpr uint32_t kStageSamples = 64 * KS;
  static constexpr uint32_t kStagesPerChunk = (kChunkDwords * 32) / kStageSamples;
  __device__ static uint32_t n_instr(uint32_t n_rb) { return (n_rb * KS + 1) / 2; }
  __device__ static uint32_t block_of_instr(uint32_t T) { return T >> 1; }
  __device__ static uint32_t swizzle(uint32_t rr) { return (rr >> 2) & 3; }
  // byte offset of piece `col` inside a row's stage, and of stage s inside the row
  __device__ static uint32_t piece_byte(uint32_t col) { return col * 16; }
  __device__ static uint32_t stage_byte(uint32_t s) { return s * kCodeStageBytes; }
};

// (Keeps hipcc from folding what follows into the LDS reads that produced a and b: a select between two dwords of a
// loaded vector is otherwise re-written into loads that have lost the __restrict__ information, and each of them then
// waits for the whole DMA ring, see mfma_stage.)
__device__ __forceinline__ void opaque(mf_u4& a, mf_u4& b) { asm("" : "+v"(a), "+v"(b)); }

}  // namespace ldp
#endif
