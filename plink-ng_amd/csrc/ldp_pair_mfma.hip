// ldp_pair_mfma.hip -- the pair statistics on the matrix pipe (gfx950 / CDNA4).
//
// DotprodWords (plink2_ld.cc:235-251) computes dot = sum_s x_i[s] * x_j[s] with x in {-1, 0, +1} (0 = het or missing)
// as popcnt(hom_i & hom_j) - 2 popcnt(hom_i & hom_j & (r2h_i ^ r2h_j)).  The same integer is a matrix product over the
// samples, and CDNA4's v_mfma_scale_f32_32x32x64_f8f6f4 contracts 64 samples of a 32 x 32 block of pairs per
// instruction with FP4 (E2M1) operands: +-2 and 0 are exact codes, the E8M0 block scale 1/2 on both operands makes every
// product +-1 / 0, and the f32 accumulators hold integers exactly below 2^24 (kMfMaxFounders).
//
// Operands are expanded straight from the 2-bit genotype codes of the resident image (ldp_device.h): per 16 samples (one
// dword) two v_bitop3_b32 and a shift give sixteen E2M1 nibbles -- magnitude bit 2 = !b0 (homozygous), sign bit 3 = b1
// (hom-ALT) -- once per row-block and k-step, shared by the 32 x 32 pairs of every product the block takes part in
// (tools/mfma_probe.hip checks the instruction facts this relies on, fact 6 the expansion itself).  There is no bit-plane
// image and no conversion pass in front of these kernels: the count pass (ldp_codes.hip) only reads.
//
// The sample order inside a fragment is free as long as both operands use the same one (a dot product is
// order-invariant); every row goes through the same fp4_of_codes(), so it is.  The image keeps the rows' own orientation
// (REF-based): the prune predicate does not depend on it, and the reported integers / --r-unphased take the sign from the
// records' ALT-major flags in the epilogue.
//
// Results are the integers the popcount kernels produce (the parity tests compare every candidate pair's 6-tuple);
// the per-pair epilogue (FP64 predicate, r^2, predicate bits) is shared (ldp_pair_device.h).
#include "ldp_device.h"
#include "ldp_pair_device.h"
#include "ldp_mfma_device.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ldp {

constexpr uint32_t kMfEpiProducts = 4;                                  // products per epilogue round
constexpr uint32_t kMfEpiWaveDwords = kMfEpiProducts * 16 * 64;
constexpr uint32_t kMfLdsDwords = kMfWaves * kMfEpiWaveDwords;          // 64 KiB: epilogue scratch == staging ring
constexpr uint32_t kMfMaxStages = 6;
// checkpoint scratch: two products' accumulators per wave in the lower half, the staged rows' cp_slot pairs behind them
constexpr uint32_t kMfCpWaveDwords = 2 * 16 * 64;
constexpr uint32_t kMfCpScratchDwords = kMfWaves * kMfCpWaveDwords;  // 32 KiB in; kMfMaxRowBlocks * 32 rows * 32 B = 16 KiB follow

// One lane's two code dwords of k-step ks from the raw LDS reads
template <int KS>
__device__ __forceinline__ void kstep_dwords(const mf_u4& H, const mf_u4& R, int ks, uint32_t h, uint32_t* hd, uint32_t* rd) {
  (void)h;
  *hd = H[ks];
  *rd = R[ks];
}

// One stage of a wave's parallelogram.  (st4 is __restrict__ on purpose: without it hipcc assumes the LDS-DMA in flight
// may alias these reads and drains it with s_waitcnt vmcnt(0) in front of every one.)
// J fragments of all k-steps of the stage stay in registers; the V blocks stream past them one at a time (two b128 LDS
// reads -> KS fragments -> KS or 2 KS MFMAs), the next block's reads in flight while the current one is multiplied, so
// only two V blocks' raw dwords are live next to the 128 accumulators.  On the diagonal (V3 = J0, V4 = J1) the last
// two blocks take the J fragments instead of reading and expanding the same rows again; the MFMAs themselves are the
// same instructions either way (a branch around them costs a register copy of every accumulator it touches).
template <int KS, bool ALL, bool GC>
__device__ __forceinline__ void mfma_stage(const mf_u4* __restrict__ st4, const uint32_t (&slot_off)[7], uint32_t oH, uint32_t oR, uint32_t h, uint32_t need,
                                           uint32_t live, bool diag, mf_v16f (&acc)[8]) {
  // ALL: every product is live and no block is aliased (the common state before the first checkpoint of a wide band):
  // the same code without a single test or branch -- 7 SALU instructions per MFMA in the masked form (profiles/)
  if constexpr (ALL) {
    need = 0x7fu;
    live = 0xffu;
    diag = false;
  }
  mf_u4 vH[2], vR[2];
  if (need & 4u) {
    vH[0] = st4[slot_off[2] + oH];
    vR[0] = st4[slot_off[2] + oR];
  }
  Frag fj0[KS], fj1[KS];
  if (need & 1u) {
    mf_u4 H = st4[slot_off[0] + oH], R = st4[slot_off[0] + oR];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t hd, rd;
      kstep_dwords<KS>(H, R, ks, h, &hd, &rd);
      fp4_expand<GC>(hd, rd, fj0[ks]);
    }
  }
  if (need & 2u) {
    mf_u4 H = st4[slot_off[1] + oH], R = st4[slot_off[1] + oR];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t hd, rd;
      kstep_dwords<KS>(H, R, ks, h, &hd, &rd);
      fp4_expand<GC>(hd, rd, fj1[ks]);
    }
  }
  // rows of C = first variant (A operand: a V block), columns = second variant (B operand: a J block)
  // u: row-block (2..6 = V0..V4), B: raw buffer, PM: the products that read it, P0 / P1: its product with J0 / J1 (-1: none),
  // ALIAS: on the diagonal the block IS J0 (0) / J1 (1) (-1: never)
#define LDP_MF_VBLOCK(u, B, PM, P0, P1, ALIAS)                                             \
  if (((u) < 6) && (need & (2u << (u)))) {                                                  \
    vH[(B) ^ 1] = st4[slot_off[((u) < 6) ? (u) + 1 : 6] + oH];                              \
    vR[(B) ^ 1] = st4[slot_off[((u) < 6) ? (u) + 1 : 6] + oR];                              \
  }                                                                                        \
  if (live & (PM)) {                                                                       \
    opaque(vH[B], vR[B]);                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                    \
      Frag fv;                                                                             \
      if (((ALIAS) >= 0) && diag) {                                                        \
        fv = ((ALIAS) == 1) ? fj1[ks] : fj0[ks];                                           \
      } else {                                                                             \
        uint32_t hd, rd;                                                                   \
        kstep_dwords<KS>(vH[B], vR[B], ks, h, &hd, &rd);                                   \
        fp4_expand<GC>(hd, rd, fv);                                                          \
      }                                                                                    \
      if ((P0 >= 0) && (live & (1u << (P0 & 7)))) acc[P0 & 7] = mfma_pair<GC>(fv, fj0[ks], acc[P0 & 7]); \
      if ((P1 >= 0) && (live & (1u << (P1 & 7)))) acc[P1 & 7] = mfma_pair<GC>(fv, fj1[ks], acc[P1 & 7]); \
    }                                                                                      \
  }
  LDP_MF_VBLOCK(2, 0, 0x01u, 0, -1, -1)
  LDP_MF_VBLOCK(3, 1, 0x12u, 1, 4, -1)
  LDP_MF_VBLOCK(4, 0, 0x24u, 2, 5, -1)
  LDP_MF_VBLOCK(5, 1, 0x48u, 3, 6, 0)
  LDP_MF_VBLOCK(6, 0, 0x80u, -1, 7, 1)
#undef LDP_MF_VBLOCK
}

// The stage of a DIAGONAL wave item (V3 = J0, V4 = J1: every wave item of a band narrower than four row-blocks, i.e. all of
// BASELINE config 2) in ONE form: five row-block reads issued together, five expansions, all eight products, no test, no branch.
// The masked form above reads a block, waits, expands and multiplies, block after block, behind a branch each -- with two waves
// per SIMD the LDS latencies and the accumulator copies at the branches add up to most of a stage (profiles/r03_experiments.md).
// Here a wave computes all eight products while one of them is live and nothing once none is; products that hold no candidate
// pair accumulate numbers nobody reads (their row-blocks may not even be staged: the read then hits some other block's rows).
// MFMAs on one accumulator never follow each other directly: V0's single product alternates with V4's.
template <int KS, bool GC>
__device__ __forceinline__ void mfma_stage_diag(const mf_u4* __restrict__ st4, const uint32_t (&slot_off)[7], uint32_t oH, uint32_t oR, mf_v16f (&acc)[8]) {
  mf_u4 jH0 = st4[slot_off[0] + oH], jR0 = st4[slot_off[0] + oR];
  mf_u4 jH1 = st4[slot_off[1] + oH], jR1 = st4[slot_off[1] + oR];
  mf_u4 vH0 = st4[slot_off[2] + oH], vR0 = st4[slot_off[2] + oR];
  mf_u4 vH1 = st4[slot_off[3] + oH], vR1 = st4[slot_off[3] + oR];
  mf_u4 vH2 = st4[slot_off[4] + oH], vR2 = st4[slot_off[4] + oR];
  opaque(jH0, jR0);
  opaque(jH1, jR1);
  Frag fj0[KS], fj1[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    fp4_expand<GC>(jH0[ks], jR0[ks], fj0[ks]);
    fp4_expand<GC>(jH1[ks], jR1[ks], fj1[ks]);
  }
  // rows of C = first variant (A operand: a V block), columns = second variant (B operand: a J block)
  opaque(vH0, vR0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH0[ks], vR0[ks], fv);
    acc[0] = mfma_pair<GC>(fv, fj0[ks], acc[0]);           // (J0, V0)
    acc[7] = mfma_pair<GC>(fj1[ks], fj1[ks], acc[7]);      // (J1, V4 = J1)
  }
  opaque(vH1, vR1);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH1[ks], vR1[ks], fv);
    acc[1] = mfma_pair<GC>(fv, fj0[ks], acc[1]);           // (J0, V1)
    acc[4] = mfma_pair<GC>(fv, fj1[ks], acc[4]);           // (J1, V1)
  }
  opaque(vH2, vR2);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH2[ks], vR2[ks], fv);
    acc[2] = mfma_pair<GC>(fv, fj0[ks], acc[2]);           // (J0, V2)
    acc[5] = mfma_pair<GC>(fv, fj1[ks], acc[5]);           // (J1, V2)
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    acc[3] = mfma_pair<GC>(fj0[ks], fj0[ks], acc[3]);      // (J0, V3 = J0)
    acc[6] = mfma_pair<GC>(fj0[ks], fj1[ks], acc[6]);      // (J1, V3 = J0)
  }
}

#ifdef LDP_MEASURE
// MEASUREMENT ONLY (profiles/r06_experiments.md: what fusing the count pass into the narrow-band pair kernel would cost in its stage loop).
// The stage above plus what a diagonal owner would have to do to count its own J rows while it streams them:
//   * sum g of the 32 rows of each J block: ONE v_mfma_scale_f32_16x16x128_f8f6f4 per J block and k-step against a constant selector operand
//     (the 16 x 128 A operand read from this kernel's 32-row / 64-sample fragment puts rows r and r + 16 into different k groups; a selector
//     whose column 0 has ones in k groups 0 and 2 and column 1 in groups 1 and 3 separates them again): 4 accumulator registers per J block;
//     sum g^2 is already on the diagonal of the (J, J) products;
//   * a missing call anywhere in a row: d & (d >> 1), OR-accumulated over the J blocks' code dwords, two VALU per dword.
// The sums are folded into one number per lane that the caller keeps alive; nothing else reads them (the prototype times the loop, it does not
// replace the count pass).
typedef float mf_v4f __attribute__((ext_vector_type(4)));
template <int KS, bool GC>
__device__ __forceinline__ void mfma_stage_diag_fused(const mf_u4* __restrict__ st4, const uint32_t (&slot_off)[7], uint32_t oH, uint32_t oR, mf_v16f (&acc)[8],
                                                      mf_v4f (&fsum)[2], const Frag& sel, uint32_t& miss) {
  mf_u4 jH0 = st4[slot_off[0] + oH], jR0 = st4[slot_off[0] + oR];
  mf_u4 jH1 = st4[slot_off[1] + oH], jR1 = st4[slot_off[1] + oR];
  mf_u4 vH0 = st4[slot_off[2] + oH], vR0 = st4[slot_off[2] + oR];
  mf_u4 vH1 = st4[slot_off[3] + oH], vR1 = st4[slot_off[3] + oR];
  mf_u4 vH2 = st4[slot_off[4] + oH], vR2 = st4[slot_off[4] + oR];
  opaque(jH0, jR0);
  opaque(jH1, jR1);
  Frag fj0[KS], fj1[KS];
  const mf_v8i S = {static_cast<int>(sel.d[0]), static_cast<int>(sel.d[1]), static_cast<int>(sel.d[2]), static_cast<int>(sel.d[3]), 0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    fp4_expand<GC>(jH0[ks], jR0[ks], fj0[ks]);
    fp4_expand<GC>(jH1[ks], jR1[ks], fj1[ks]);
    miss |= (jH0[ks] & (jH0[ks] >> 1)) | (jR0[ks] & (jR0[ks] >> 1)) | (jH1[ks] & (jH1[ks] >> 1)) | (jR1[ks] & (jR1[ks] >> 1));
    const mf_v8i A0 = {static_cast<int>(fj0[ks].d[0]), static_cast<int>(fj0[ks].d[1]), static_cast<int>(fj0[ks].d[2]), static_cast<int>(fj0[ks].d[3]), 0, 0, 0, 0};
    const mf_v8i A1 = {static_cast<int>(fj1[ks].d[0]), static_cast<int>(fj1[ks].d[1]), static_cast<int>(fj1[ks].d[2]), static_cast<int>(fj1[ks].d[3]), 0, 0, 0, 0};
    fsum[0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A0, S, fsum[0], 4, 4, 0, kFp4ScaleG, 0, kFp4ScaleG);
    fsum[1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A1, S, fsum[1], 4, 4, 0, kFp4ScaleG, 0, kFp4ScaleG);
  }
  opaque(vH0, vR0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH0[ks], vR0[ks], fv);
    acc[0] = mfma_pair<GC>(fv, fj0[ks], acc[0]);
    acc[7] = mfma_pair<GC>(fj1[ks], fj1[ks], acc[7]);
  }
  opaque(vH1, vR1);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH1[ks], vR1[ks], fv);
    acc[1] = mfma_pair<GC>(fv, fj0[ks], acc[1]);
    acc[4] = mfma_pair<GC>(fv, fj1[ks], acc[4]);
  }
  opaque(vH2, vR2);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag fv;
    fp4_expand<GC>(vH2[ks], vR2[ks], fv);
    acc[2] = mfma_pair<GC>(fv, fj0[ks], acc[2]);
    acc[5] = mfma_pair<GC>(fv, fj1[ks], acc[5]);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    acc[3] = mfma_pair<GC>(fj0[ks], fj0[ks], acc[3]);
    acc[6] = mfma_pair<GC>(fj0[ks], fj1[ks], acc[6]);
  }
}
#endif

// ... and once the four FAR products of a diagonal wave item -- (J0, V0) (J0, V1) (J1, V1) (J1, V2): block distances 2 and 3 -- are
// provably below the threshold (on real data they are the first to go: LD decays with distance), the NEAR half alone:
// (J0, V2) (J0, J0) (J1, J0) (J1, J1), three row-block reads and sixteen MFMAs per stage instead of five and thirty-two, and the
// workgroup stops fetching V0 / V1 blocks nobody reads any more.
constexpr uint32_t kDiagFar = 0x33u, kDiagNear = 0xccu;
template <int KS, bool GC>
__device__ __forceinline__ void mfma_stage_near(const mf_u4* __restrict__ st4, const uint32_t (&slot_off)[7], uint32_t oH, uint32_t oR, mf_v16f (&acc)[8]) {
  mf_u4 jH0 = st4[slot_off[0] + oH], jR0 = st4[slot_off[0] + oR];
  mf_u4 jH1 = st4[slot_off[1] + oH], jR1 = st4[slot_off[1] + oR];
  mf_u4 vH2 = st4[slot_off[4] + oH], vR2 = st4[slot_off[4] + oR];
  opaque(jH0, jR0);
  opaque(jH1, jR1);
  opaque(vH2, vR2);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    Frag f0, f1, fv;
    fp4_expand<GC>(jH0[ks], jR0[ks], f0);
    fp4_expand<GC>(jH1[ks], jR1[ks], f1);
    fp4_expand<GC>(vH2[ks], vR2[ks], fv);
    acc[2] = mfma_pair<GC>(fv, f0, acc[2]);   // (J0, V2)
    acc[3] = mfma_pair<GC>(f0, f0, acc[3]);   // (J0, V3 = J0)
    acc[6] = mfma_pair<GC>(f0, f1, acc[6]);   // (J1, V3 = J0)
    acc[7] = mfma_pair<GC>(f1, f1, acc[7]);   // (J1, V4 = J1)
  }
}

// Row-blocks whose LDS rows the live products of a wave read: bit u of the result = J0, J1, V0..V4 (on the diagonal V3 /
// V4 are J0 / J1 and are not read a second time)
__device__ __forceinline__ uint32_t blocks_needed(uint32_t live, bool diag) {
  uint32_t need = ((live & 0x0fu) ? 1u : 0u) | ((live & 0xf0u) ? 2u : 0u);
  need |= (live & 0x01u) ? 4u : 0u;
  need |= (live & 0x12u) ? 8u : 0u;
  need |= (live & 0x24u) ? 16u : 0u;
  if (diag) {
    need |= (live & 0x48u) ? 1u : 0u;
    need |= (live & 0x80u) ? 2u : 0u;
  } else {
    need |= (live & 0x48u) ? 32u : 0u;
    need |= (live & 0x80u) ? 64u : 0u;
  }
  return need;
}

// ---- a few missing calls per row (4.1d) -------------------------------------------------------------------------
// x = 0 where a call is missing, so the dot product of the complete-data kernel is already the exact `dot` of
// ComputeIndepPairwiseR2Components; what the missing calls leave open are the five pairwise-complete counts.  With M_i, M_j
// missing calls in the two rows they are confined to intervals that per-variant numbers pin down:
//   nm   in [N - M_i - M_j, N - max(M_i, M_j)]
//   sum1 in [S_i - k1, S_i + k1],  ssq1 in [Q_i - k1, Q_i],  k1 = min(M_j, Q_i)   (the calls of i that j's missing calls remove)
//   sum2, ssq2 the same with k2 = min(M_i, Q_j)
// and interval arithmetic on cov = dot nm - sum1 sum2, var = ssq nm - sum^2 decides the predicate cov^2 > thresh var1 var2
// for every pair that is not close to the threshold (1e-9 relative slack covers the FP64 rounding of the bounds).  The
// few pairs left open are resolved exactly on the spot: the whole wave counts the five statistics of one such pair from
// the two rows' bit-planes (lanes across plane dwords, as pair_stats_ref_kernel does) and the exact predicate decides.
// Returns 0 below, 1 above, 2 open.
__device__ __forceinline__ int classify_sparse(const PairKernelArgs& A, double dot, const ldp_variant_rec& ri, const ldp_variant_rec& rj) {
  const double N = static_cast<double>(A.founder_ct);
  const double Mi = N - static_cast<double>(ri.nm_ct), Mj = N - static_cast<double>(rj.nm_ct);
  const double nm_lo = fmax(N - Mi - Mj, 0.0), nm_hi = N - fmax(Mi, Mj);
  const double Si = ri.sum, Sj = rj.sum, Qi = ri.ssq, Qj = rj.ssq;
  const double k1 = fmin(Mj, Qi), k2 = fmin(Mi, Qj);
  const double s1_lo = Si - k1, s1_hi = Si + k1, s2_lo = Sj - k2, s2_hi = Sj + k2;
  // cov = dot nm - sum1 sum2
  const double a0 = dot * nm_lo, a1 = dot * nm_hi;
  const double a_lo = fmin(a0, a1), a_hi = fmax(a0, a1);
  const double p0 = s1_lo * s2_lo, p1 = s1_lo * s2_hi, p2 = s1_hi * s2_lo, p3 = s1_hi * s2_hi;
  const double b_lo = fmin(fmin(p0, p1), fmin(p2, p3)), b_hi = fmax(fmax(p0, p1), fmax(p2, p3));
  const double c_lo = a_lo - b_hi, c_hi = a_hi - b_lo;
  const double c2_hi = fmax(c_lo * c_lo, c_hi * c_hi);
  const double c2_lo = ((c_lo <= 0.0) && (c_hi >= 0.0)) ? 0.0 : fmin(c_lo * c_lo, c_hi * c_hi);
  // var = ssq nm - sum^2 (never negative for the true values)
  const double q1_hi = fmax(s1_lo * s1_lo, s1_hi * s1_hi), q1_lo = ((s1_lo <= 0.0) && (s1_hi >= 0.0)) ? 0.0 : fmin(s1_lo * s1_lo, s1_hi * s1_hi);
  const double q2_hi = fmax(s2_lo * s2_lo, s2_hi * s2_hi), q2_lo = ((s2_lo <= 0.0) && (s2_hi >= 0.0)) ? 0.0 : fmin(s2_lo * s2_lo, s2_hi * s2_hi);
  const double v1_lo = fmax((Qi - k1) * nm_lo - q1_hi, 0.0), v1_hi = fmax(Qi * nm_hi - q1_lo, 0.0);
  const double v2_lo = fmax((Qj - k2) * nm_lo - q2_hi, 0.0), v2_hi = fmax(Qj * nm_hi - q2_lo, 0.0);
  const double rhs_lo = A.thresh * v1_lo * v2_lo, rhs_hi = A.thresh * v1_hi * v2_hi;
  if (c2_lo > rhs_hi * (1.0 + 1e-9) + 1.0) {
    return 1;
  }
  if (c2_hi * (1.0 + 1e-9) + 1.0 < rhs_lo) {
    return 0;
  }
  return 2;
}

// (wave_pair_counts: ldp_pair_device.h -- the tile kernel's interval epilogue recounts its open pairs the same way)

// One round (the four products of one J block, already dumped to this wave's LDS scratch).
__device__ __forceinline__ uint32_t sparse_round(const PairKernelArgs& A, const uint32_t* epi, uint32_t lane, int32_t jv, int32_t vv, uint32_t jend,
                                                          uint32_t live, uint32_t lo_j, int round) {
  const uint32_t r = lane & 31, h = lane >> 5;
  uint32_t n_true = 0, n_open = 0;
  const int64_t j64 = static_cast<int64_t>(jv) + kMfBlock * round + r;
  const bool j_ok = (j64 < static_cast<int64_t>(jend)) && (static_cast<int64_t>(lo_j) < j64);
  const uint32_t j = j_ok ? static_cast<uint32_t>(j64) : 0u;
  ldp_variant_rec rj;
  rj.nm_ct = 0;
  rj.sum = 0;
  rj.ssq = 0;
  rj.flags = 0;
  if (j_ok) {
    rj = A.recs[j];
  }
#pragma unroll 1
  for (uint32_t pl = 0; pl < 4; ++pl) {
    if (!(live & (1u << (4 * round + pl)))) {
      continue;  // (wave-uniform)
    }
    const int64_t vfirst = static_cast<int64_t>(vv) + kMfBlock * (pl + round) + 4 * h;
#pragma unroll 1
    for (uint32_t g = 0; g < 16; ++g) {
      const int64_t i64 = vfirst + (g & 3) + 8 * (g >> 2);
      const bool valid = j_ok && (i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64);
      const uint32_t i = valid ? static_cast<uint32_t>(i64) : 0u;
      int32_t dot = static_cast<int32_t>(epi[(pl * 16 + g) * 64 + lane]);
      int cls = 0;
      uint32_t alt_ij = 0;  // bit 0: the image's row i is not major-oriented (img_differs, ldp_device.h), bit 1: row j
      if (valid) {
        const ldp_variant_rec ri = A.recs[i];
        alt_ij = img_differs(ri.flags) | (img_differs(rj.flags) << 1);
        dot = ((alt_ij == 1u) || (alt_ij == 2u)) ? -dot : dot;  // the image's orientation -> the records' (major allele)
        cls = classify_sparse(A, static_cast<double>(dot), ri, rj);
        if (cls == 1) {
          atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
          ++n_true;
        }
      }
      // the pairs left open, one after the other, each by the whole wave
      unsigned long long open = __ballot(cls == 2);
      while (open) {
        const int l = __builtin_ctzll(open);
        open &= open - 1;
        const uint32_t ii = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i), l));
        const uint32_t jj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
        const int32_t dd = __builtin_amdgcn_readlane(dot, l);
        const uint32_t aa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(alt_ij), l));
        const ldp_pair_stats_t st = wave_pair_counts(A, ii, jj, dd, lane, (aa & 1u) != 0, (aa & 2u) != 0);
        if ((static_cast<int>(lane) == l) && exceeds(st, A.thresh)) {
          atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
          ++n_true;
        }
        n_open += (lane == 0) ? 1u : 0u;
      }
    }
  }
  if ((lane == 0) && n_open) {
    atomicAdd(A.counters + 3, static_cast<unsigned long long>(n_open));
  }
  return n_true;
}

// SPARSE: the instantiation that takes launches whose rows have a few missing calls (4.1d): the same stage loops without
// checkpoints, and the interval epilogue.  A kernel of its own rather than a branch in the epilogue: with the interval code
// inside it hipcc re-allocates the complete-data kernel's registers and spills inside the stage loop (inlined), or keeps
// half of them in scratch across the call (out of line).  Both instantiations are launched; the one the data does not
// call for leaves at once.
// DIAGFORM: the instantiation that takes the workgroups whose wave items are all diagonal (MfmaWG::pad bit 1), with the single
// branch-free stage loop of mfma_stage_diag and early termination by whole waves; the other instantiation keeps the general
// forms for everything else.  Separate kernels because a third loop form in one kernel sends hipcc into hundreds of spills.
// FUSE != 0 (measurement build only, DIAGFORM, complete data): mfma_stage_diag_fused instead of mfma_stage_diag and no checkpoints -- what the
// stage loop of a count-fused narrow-band kernel would cost (LDP_DEBUG_DIAG_FUSE=1; the results are those of the exhaustive kernel).
template <int KS, bool SPARSE, bool DIAGFORM, int FUSE = 0>
__global__ __launch_bounds__(kMfWaves * 64, 2) void pair_mfma_kernel(PairKernelArgs A) {
  using G = StageGeom<KS>;
  // complete data: the allele-count coding (ldp_mfma_device.h), the accumulators hold G = sum g_i g_j; the SPARSE instantiation (rows
  // with a few missing calls) keeps the +-2 coding, where a missing call contributes 0 and the accumulators hold the dot product
  // (The masked instantiation <., false, false> -- bands of 4-11 row-blocks, ragged workgroups, the windowed r^2 plans -- keeps the
  // +-2 coding as well: it sits at 254 registers, and with the checkpoint's conversion beside its branches hipcc parks an accumulator
  // in scratch INSIDE the stage loop, whose reload drains the DMA ring every k-step; tests/test_kernel_isa.py.)
  constexpr bool GC = (!SPARSE) && DIAGFORM;
  const int32_t g_bias = g_bias_of(A.founder_ct, G::kStageSamples);
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_src_off[kMfMaxDmaPerWave * kMfWaves * 64];
  __shared__ uint32_t s_need[kMfWaves];
  // route_kernel decided which of the three matrix-pipe kernels owns the launch (ldp_device.h); the others leave at once
  if (*A.route != (SPARSE ? kRouteSparse : kRouteComplete)) {
    return;
  }
  const uint32_t per_xcd = (A.n_mf_wgs + 7) / 8;
  const uint32_t item_idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // XCD-aware, as pair_tiles_kernel
  if (item_idx >= A.n_mf_wgs) {
    return;
  }
  const MfmaWG* __restrict__ wg = A.mf_wgs + item_idx;
  {
    const uint32_t kind = wg->pad;
    if (A.wd_active && (kind & 1u) && ((!SPARSE) || A.wd_sparse)) {
      return;  // a subcontig with a wide band: its complete-data launches belong to pair_mfma_wide_kernel's tiles (ldp_pair_wide.hip)
    }
    if (((kind & 2u) != 0) != DIAGFORM) {
      return;  // the other instantiation's workgroup
    }
  }
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const uint32_t r = lane & 31;
  const uint32_t h = lane >> 5;

  const uint32_t n_rb = __builtin_amdgcn_readfirstlane(wg->n_rb);
  const uint32_t n_instr = G::n_instr(n_rb);  // DMA wave-instructions per stage (64 slots of 16 B each)
  uint32_t mine = (n_instr > wave) ? (n_instr - wave + kMfWaves - 1) / kMfWaves : 0;  // ... of which this wave issues
  const uint32_t stage_dwords = n_instr * 256;
  uint32_t stages = A.lds_dwords / stage_dwords;
  stages = (stages > kMfMaxStages) ? kMfMaxStages : stages;
  const uint32_t row_bytes = static_cast<uint32_t>(A.code_row_bytes);
  const uint32_t n_stages = (A.founder_ct + G::kStageSamples - 1) / G::kStageSamples;

  // ---- DMA plan: per-lane source offsets (LDS) and per-instruction row-block bases (uniform) ----
  const uint8_t* base_t[kMfMaxDmaPerWave];
#pragma unroll
  for (int t = 0; t < kMfMaxDmaPerWave; ++t) {
    const uint32_t T = wave + kMfWaves * t;
    base_t[t] = A.codes;
    if (T < n_instr) {
      const uint32_t blk = G::block_of_instr(T);
      const uint32_t first = __builtin_amdgcn_readfirstlane(wg->rb[(blk < n_rb) ? blk : (n_rb - 1)]);
      base_t[t] += static_cast<uint64_t>(first) * row_bytes;
      const uint32_t L = T * 64 + lane;
      const uint32_t rr = (L / KS) & 31;
      const uint32_t col = (L % KS) ^ G::swizzle(rr);
      uint32_t var = first + rr;
      var = (var < A.n_local) ? var : (A.n_local - 1);
      s_src_off[t * (kMfWaves * 64) + tid] = (var - first) * row_bytes + G::piece_byte(col);
    }
  }

  // ---- this wave's parallelogram ----
  const MfmaWaveItem* __restrict__ wi = wg->w + wave;
  const int32_t jv = __builtin_amdgcn_readfirstlane(wi->jv);
  const int32_t vv = __builtin_amdgcn_readfirstlane(wi->vv);
  const uint32_t jend = __builtin_amdgcn_readfirstlane(wi->jend);
  uint32_t live = (jv >= 0) ? __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(wi->prod_mask)) : 0u;
  const bool diag = (vv + 3 * kMfBlock == jv);
  uint32_t slot_off[7];  // uint4 index of the row-block's first slot
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    slot_off[u] = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(wi->slot[u])) * G::kBlockSlots;
  }
  uint32_t need = blocks_needed(live, diag);
  const uint32_t live0 = live, need0 = need;  // (DIAGFORM: the plan's products; a wave runs all of them until none is live)
  uint32_t stop_stage = 0xffffffffu;         // (DIAGFORM: the stage at which the wave stopped computing ...
  uint32_t far_stop = 0xffffffffu;           //  ... and the one at which it dropped its FAR products)
  // window starts of this lane's two second variants (J0 + r, J1 + r), fetched here: the k-loop must not hold ordinary
  // global loads (see the checkpoint)
  uint32_t lo_j2[2] = {0xffffffffu, 0xffffffffu};  // (lo >= j: no candidate pair)
  if (jv >= 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t j = static_cast<uint32_t>(jv) + kMfBlock * q + r;
      if (j < jend) {
        lo_j2[q] = A.lo[j];
      }
    }
  }
  uint32_t slots_packed = 0;  // LDS row-block slot of J0, J1, V0..V4, 4 bits each (for the checkpoint's rolled loops)
#pragma unroll
  for (int u = 0; u < 7; ++u) {
    slots_packed |= (slot_off[u] / G::kBlockSlots) << (4 * u);
  }
  const uint32_t sw = G::swizzle(r);
  const uint32_t oH = (KS == 4) ? (r * 4 + (h ^ sw)) : (r * 2 + sw);
  const uint32_t oR = (KS == 4) ? (r * 4 + ((2 + h) ^ sw)) : (r * 2 + (1 ^ sw));

  mf_v16f acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }

  uint32_t wg_need = (1u << n_rb) - 1;  // row-block slots some wave still reads (all of them until a checkpoint says otherwise)
  uint32_t next_cp = 0;
  // (the checkpoint bound assumes complete rows.  DIAGFORM: ONE checkpoint, the second of the list (a fraction 1 - sqrt(r2) + 0.04
  // of the samples): from about 0.02 behind 1 - sqrt(r2) the four FAR products of nearly every wave item are provably hopeless
  // -- a third of the plan's MFMAs at config 2 -- and nothing earlier is (r2 0.5: at 0.30 a quarter, at 0.31 a third).  Each
  // further checkpoint costs a drained ring, three barriers and a restart for nothing: config 2 with 1 / 2 / 5 checkpoints
  // 3.50 / 3.73 / 4.32 ms, none 3.91; profiles/r03_experiments.md)
  const uint32_t cp_all = (A.cp_stats && (!SPARSE) && (FUSE == 0)) ? A.n_checkpoints : 0;
#ifdef LDP_MEASURE
  [[maybe_unused]] mf_v4f fsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  [[maybe_unused]] uint32_t fmiss = 0;
  [[maybe_unused]] Frag fsel;
  if constexpr (FUSE != 0) {
    // the selector: E2M1 0.5 (block scale 2 -> 1.0) in k groups 0 and 2 for column 0, groups 1 and 3 for column 1, zero elsewhere
    const uint32_t col = lane & 15, grp = lane >> 4;
    const uint32_t on = (col < 2) && ((grp & 1u) == col);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fsel.d[q] = on ? 0x11111111u : 0u;
    }
  }
#endif
  const uint32_t first_cp = (DIAGFORM && (cp_all >= 2)) ? 1u : 0u;
  const uint32_t n_cp = DIAGFORM ? ((cp_all > first_cp) ? first_cp + 1 : cp_all) : cp_all;
  next_cp = first_cp;
  auto dma_stage = [&](uint32_t s, uint32_t buf) {
    const uint32_t kbyte = G::stage_byte(s);
    uint32_t* dst = lds + buf * stage_dwords;
#pragma unroll
    for (int t = 0; t < kMfMaxDmaPerWave; ++t) {
      const uint32_t T = wave + kMfWaves * t;
      if ((T < n_instr) && ((wg_need >> G::block_of_instr(T)) & 1u)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_t[t] + kbyte + s_src_off[t * (kMfWaves * 64) + tid]),
                                         (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
      }
    }
  };
  auto count_mine = [&]() {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < kMfMaxDmaPerWave; ++t) {
      const uint32_t T = wave + kMfWaves * t;
      m += ((T < n_instr) && ((wg_need >> G::block_of_instr(T)) & 1u)) ? 1u : 0u;
    }
    return m;
  };
  auto checkpoint_stage = [&](uint32_t cp) {
    const uint32_t s = A.checkpoint_chunk[cp] * G::kStagesPerChunk;
    return (s < n_stages) ? s : n_stages;
  };

  __syncthreads();  // (s_src_off is complete)
  uint32_t* epi = lds + wave * kMfEpiWaveDwords;  // this wave's scratch whenever the ring is empty (checkpoints, epilogue)
  // accumulators of the live products among 4 * round .. 4 * round + 3 -> epi[(pl * 16 + g) * 64 + lane], as integers
  auto dump_round = [&](int round) {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      if (live & (1u << (4 * round + pl))) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          epi[(pl * 16 + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>((round ? acc[4 + pl] : acc[pl])[g]));
        }
      }
    }
  };
  // ---- k-loop over stages, ring of `stages` LDS buffers ----
  // The ring never runs past the next checkpoint: when one fires every queued chunk has been consumed and the whole
  // LDS is free for the checkpoint's scratch (the accumulators go through it so the bound is a rolled loop).
  uint32_t issued = 0, issue_buf = 0, read_buf = 0, issued_base = 0;
  uint32_t issue_limit = (next_cp < n_cp) ? checkpoint_stage(next_cp) : n_stages;
  auto ring_fill = [&]() {
    issue_buf = 0;
    read_buf = 0;
    while ((issued < issue_limit) && (issued + 1 < issued_base + stages)) {
      dma_stage(issued, issue_buf);
      ++issued;
      issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
    }
  };
  ring_fill();
  // The stages between two checkpoints, with the product / block tests either compiled in (masked) or not (ALL); the
  // choice is made per segment, outside the loop, so that neither form pays register copies for the other's branches.
  auto run_segment = [&](auto all_tag, uint32_t kc, uint32_t kc_end) {
    constexpr bool ALL = decltype(all_tag)::value;
    for (; kc < kc_end; ++kc) {
      wait_dma_then_barrier(mine * (issued - kc - 1));
      if (issued < issue_limit) {
        dma_stage(issued, issue_buf);  // (reuses the buffer every wave finished reading before the barrier)
        ++issued;
        issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
      }
      const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * stage_dwords);
      read_buf = (read_buf + 1 == stages) ? 0 : read_buf + 1;
      if (ALL || live) {
        mfma_stage<KS, ALL, GC>(st4, slot_off, oH, oR, h, need, live, diag, acc);
      }
    }
  };
  for (uint32_t kc = 0; kc < n_stages;) {
    const uint32_t kc_end = issue_limit;  // the next checkpoint (or the end of the rows)
    if constexpr (DIAGFORM) {
      // one of three branch-free forms per segment (between two checkpoints), chosen out here: all eight products while a FAR one
      // is live, the NEAR four afterwards, nothing once none is (the wave still issues its share of the DMA and meets the barriers)
      auto advance = [&](uint32_t k2) {
        wait_dma_then_barrier(mine * (issued - k2 - 1));
        if (issued < issue_limit) {
          dma_stage(issued, issue_buf);  // (reuses the buffer every wave finished reading before the barrier)
          ++issued;
          issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
        }
        const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * stage_dwords);
        read_buf = (read_buf + 1 == stages) ? 0 : read_buf + 1;
        return st4;
      };
      if (live & kDiagFar) {
        for (uint32_t k2 = kc; k2 < kc_end; ++k2) {
          const mf_u4* __restrict__ st4 = advance(k2);
#ifdef LDP_MEASURE
          if constexpr (FUSE != 0) {
            mfma_stage_diag_fused<KS, GC>(st4, slot_off, oH, oR, acc, fsum, fsel, fmiss);
            continue;
          }
#endif
          mfma_stage_diag<KS, GC>(st4, slot_off, oH, oR, acc);
        }
      } else if (live) {
        for (uint32_t k2 = kc; k2 < kc_end; ++k2) {
          const mf_u4* __restrict__ st4 = advance(k2);
          mfma_stage_near<KS, GC>(st4, slot_off, oH, oR, acc);
        }
      } else {
        for (uint32_t k2 = kc; k2 < kc_end; ++k2) {
          (void)advance(k2);
        }
      }
    } else if ((live == 0xffu) && !diag) {
      run_segment(std::true_type(), kc, kc_end);
    } else {
      run_segment(std::false_type(), kc, kc_end);
    }
    kc = kc_end;
    if (kc >= n_stages) {
      break;
    }
    {  // (block-uniform; issued == kc here)
    // ---- checkpoint (ldp_device.h): drop the products whose candidate pairs are all provably below the threshold ----
    __syncthreads();  // every wave is done with the last stage: LDS is scratch now
    // The checkpoint statistics (ldp_device.h: cp_slot) of every staged row come in by LDS-DMA as well, slot next_cp and
    // the whole-row slot, 32 bytes per row behind the accumulator scratch.  Ordinary global loads inside the k-loop
    // would make hipcc drain the ring (s_waitcnt vmcnt(0)) in front of EVERY LDS read of the loop, not just here.
    {
      const uint8_t* cps = reinterpret_cast<const uint8_t*>(A.cp_stats);
#pragma unroll
      for (int t = 0; t < kMfMaxDmaPerWave; ++t) {
        const uint32_t T = wave + kMfWaves * t;  // row-block slot T: 32 rows x 2 pieces
        if (T < n_rb) {
          const uint32_t first = __builtin_amdgcn_readfirstlane(wg->rb[T]);
          uint32_t var = first + (lane >> 1);
          var = (var < A.n_local) ? var : (A.n_local - 1);
          const uint64_t off = static_cast<uint64_t>(var) * (kCpStride * sizeof(cp_slot)) + ((lane & 1) ? kCheckpoints : next_cp) * sizeof(cp_slot);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cps + off),
                                           (__attribute__((address_space(3))) void*)(lds + kMfCpScratchDwords + T * 256), 16, 0, 0);
        }
      }
    }
    __syncthreads();  // (drains the DMA: the slots are in LDS)
    const cp_slot* __restrict__ cpl = reinterpret_cast<const cp_slot*>(lds + kMfCpScratchDwords);  // [row-block slot][row][2]
    // GC: the accumulators hold G_P = sum over the samples visited of g_i g_j; the partial dot product of x = 1 - g is
    // G_P - n_P + sP_i + sP_j with sP = the row's sum of x over the samples visited, an INTEGER: S - s_R, where s_R is
    // a * sqrt(n_R / N) to the nearest integer (cp_slot: a = s_R * sqrt(N / n_R) as the count pass rounded it; n_P samples visited,
    // no padding among them: a checkpoint sits in front of the last k-chunk).  One integer per staged row, behind the slots.
    int32_t* __restrict__ sp = reinterpret_cast<int32_t*>(lds + kMfCpScratchDwords + kMfMaxRowBlocks * kMfBlock * 8);
    const int32_t cp_seen = static_cast<int32_t>(kc * G::kStageSamples);
    if constexpr (GC) {
      const double n_all = static_cast<double>(A.founder_ct);
      const double kappa = sqrt(((static_cast<double>(cp_seen) < n_all) ? (n_all - static_cast<double>(cp_seen)) : 1.0) / n_all);
      for (uint32_t q = tid; q < n_rb * kMfBlock; q += kMfWaves * 64) {
        sp[q] = static_cast<int32_t>(cpl[2 * q + 1].a) - static_cast<int32_t>(rint(cpl[2 * q].a * kappa));
      }
      __syncthreads();
    }
    if (live) {
      uint32_t keep = 0;
      uint32_t* cp_epi = lds + wave * kMfCpWaveDwords;  // two products per round
#pragma unroll
      for (int round = 0; round < 4; ++round) {
        if (!(live & (0x3u << (2 * round)))) {
          continue;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          if (live & (1u << (2 * round + pl))) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              cp_epi[(pl * 16 + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[2 * round + pl][g]));
            }
          }
        }
        const int q = (round >= 2) ? 1 : 0;
        const int64_t j64 = static_cast<int64_t>(jv) + kMfBlock * q + r;
        const int64_t lo_j = lo_j2[q];
        const uint32_t jslot = (slots_packed >> (4 * q)) & 15u;
        const cp_slot cj = cpl[(jslot * kMfBlock + r) * 2];
        const cp_slot gj = cpl[(jslot * kMfBlock + r) * 2 + 1];
        const int32_t tj = GC ? (sp[jslot * kMfBlock + r] - cp_seen) : 0;
#pragma unroll 1
        for (uint32_t pl = 0; pl < 2; ++pl) {
          const uint32_t p = 2 * round + pl;
          if (!(live & (1u << p))) {
            continue;
          }
          const uint32_t k = (p & 3) + q;  // V block of the product
          const uint32_t vslot = (slots_packed >> (4 * (2 + k))) & 15u;
          bool hopeless = true;
          const int64_t vfirst = static_cast<int64_t>(vv) + kMfBlock * k + 4 * h;
#pragma unroll 2
          for (uint32_t g = 0; g < 16; ++g) {
            const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * h;
            const int64_t i64 = static_cast<int64_t>(vv) + kMfBlock * k + row;
            if ((i64 >= lo_j) && (i64 < j64)) {
              const cp_slot ci = cpl[(vslot * kMfBlock + row) * 2];
              const cp_slot gi = cpl[(vslot * kMfBlock + row) * 2 + 1];
              // |N dot - S_i S_j| <= |c0| + B, see pair_hopeless() in ldp_pair_device.h (dot_p is the partial dot product)
              const double dot_p = static_cast<double>(static_cast<int32_t>(cp_epi[(pl * 16 + g) * 64 + lane]) + (GC ? (tj + sp[vslot * kMfBlock + row]) : 0));
              const double c0 = fma(static_cast<double>(A.founder_ct), dot_p, fma(ci.a, cj.a, -(gi.a * gj.a)));
              const double bound = fabs(c0) + fma(ci.b, cj.b, 1.0);
              hopeless = hopeless && (bound < gi.b * gj.b);
            }
          }
          (void)vfirst;
          if (!__all(hopeless)) {
            keep |= 1u << p;
          }
        }
      }
      keep = __builtin_amdgcn_readfirstlane(keep);
      if (keep != live) {
        if constexpr (DIAGFORM) {
          if ((live & kDiagFar) && !(keep & kDiagFar)) {
            far_stop = kc;   // from here on the NEAR form (or nothing)
          }
          live = keep;
          // (a form reads all of its planned blocks while it runs at all: J0, J1, V0..V2 -- or J0, J1, V2 alone)
          need = (live & kDiagFar) ? need0 : (live ? (need0 & 0x13u) : 0u);
          if (!live) {
            stop_stage = kc;
          }
        } else {
          if (lane == 0) {
            atomicAdd(A.counters + 2, static_cast<unsigned long long>(n_stages - kc) * KS * __builtin_popcount(live & ~keep));
          }
          live = keep;
          need = blocks_needed(live, diag);
        }
      }
    }
    ++next_cp;
    // which staged row-blocks does the workgroup still read?  Dead ones are no longer fetched.
    if (lane == 0) {
      uint32_t m = 0;
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        if (need & (1u << u)) {
          m |= 1u << (slot_off[u] / G::kBlockSlots);
        }
      }
      s_need[wave] = m;
    }
    __syncthreads();
    const uint32_t all_need = s_need[0] | s_need[1] | s_need[2] | s_need[3];
    __syncthreads();  // (s_need is rewritten at the next checkpoint; the scratch reads above are over as well)
    if (!all_need) {
      break;  // nothing left that could reach the threshold
    }
    if (all_need != wg_need) {
      wg_need = __builtin_amdgcn_readfirstlane(all_need);
      mine = count_mine();
    }
    issue_limit = (next_cp < n_cp) ? checkpoint_stage(next_cp) : n_stages;
    issued_base = kc;
    ring_fill();  // restart the ring at this stage
    }
  }
  __syncthreads();  // staging is over: LDS becomes the epilogue's scratch (a private region per wave)
#ifdef LDP_MEASURE
  if constexpr (FUSE != 0) {
    // (keeps the fused sums alive; never true on genotype data)
    if ((fsum[0][0] + fsum[0][1] + fsum[0][2] + fsum[0][3] + fsum[1][0] + fsum[1][1] + fsum[1][2] + fsum[1][3] == -12345.f) && (fmiss == 0x5a5a5a5au)) {
      atomicAdd(A.counters + 3, 1ull);
    }
  }
#endif
  if constexpr (DIAGFORM) {
    if ((lane == 0) && live0) {
      // bookkeeping in product x k-step units (one MFMA each): what early termination saved of the plan's products, and what
      // the branch-free forms computed beyond the plan (products of the wave item that hold no candidate pair)
      const uint32_t ran = (stop_stage < n_stages) ? stop_stage : n_stages;          // stages with any form
      const uint32_t ran_far = (live0 & kDiagFar) ? ((far_stop < ran) ? far_stop : ran) : 0u;  // stages with the full form
      const uint32_t plan_far = __builtin_popcount(live0 & kDiagFar), plan_near = __builtin_popcount(live0 & kDiagNear);
      const unsigned long long saved = static_cast<unsigned long long>(n_stages - ran_far) * plan_far + static_cast<unsigned long long>(n_stages - ran) * plan_near;
      const unsigned long long extra = static_cast<unsigned long long>(ran_far) * (4 - plan_far) + static_cast<unsigned long long>(ran) * (4 - plan_near);
      if (saved) {
        atomicAdd(A.counters + 2, saved * KS);
      }
      if (extra) {
        atomicAdd(A.counters + 1, extra * KS);
      }
    }
  }

  // ---- epilogue: accumulators through LDS so the per-pair code is a rolled loop ----
  // lane l, register g of a product holds first variant (g & 3) + 8 (g >> 2) + 4 (l >> 5) of the V block, second variant
  // l & 31 of the J block (tools/mfma_probe.hip, fact 1)
  uint32_t n_true = 0;
  if constexpr (SPARSE) {
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      if (live & (0xfu << (4 * round))) {
        dump_round(round);
        n_true += sparse_round(A, epi, lane, jv, vv, jend, live, lo_j2[round], round);
      }
    }
    n_true = wave_reduce_add(n_true);
    if ((lane == 0) && n_true) {
      atomicAdd(A.counters, static_cast<unsigned long long>(n_true));
    }
    return;
  }
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!(live & (0xfu << (4 * round)))) {
      continue;
    }
    dump_round(round);
    const int64_t j64 = static_cast<int64_t>(jv) + kMfBlock * round + r;
    if (j64 < static_cast<int64_t>(jend)) {
      const uint32_t j = static_cast<uint32_t>(j64);
      const uint32_t lo_j = lo_j2[round];
      if (lo_j < j) {
        const int32_t sum_j = A.recs[j].sum;
        const uint32_t ssq_j = A.recs[j].ssq;
        const uint32_t flags_j = A.recs[j].flags;
        const int32_t sum_img_j = img_differs(flags_j) ? -sum_j : sum_j;
#pragma unroll 1
        for (uint32_t pl = 0; pl < 4; ++pl) {
          if (!(live & (1u << (4 * round + pl)))) {
            continue;
          }
          const int64_t vfirst = static_cast<int64_t>(vv) + kMfBlock * (pl + round) + 4 * h;
#pragma unroll 1
          for (uint32_t g = 0; g < 16; ++g) {
            const int64_t i64 = vfirst + (g & 3) + 8 * (g >> 2);
            if ((i64 < static_cast<int64_t>(lo_j)) || (i64 >= j64)) {
              continue;
            }
            const uint32_t i = static_cast<uint32_t>(i64);
            const ldp_variant_rec ri = A.recs[i];
            ldp_pair_stats_t ps;
            int32_t dot_img = static_cast<int32_t>(epi[(pl * 16 + g) * 64 + lane]);
            if constexpr (GC) {
              dot_img += sum_img_of(ri) + sum_img_j - g_bias;  // G -> the dot product of x = 1 - g (ldp_mfma_device.h)
            }
            ps.dot = (img_differs(ri.flags) ^ img_differs(flags_j)) ? -dot_img : dot_img;  // the image's orientation -> the records' (major allele)
            ps.nm = A.founder_ct;
            ps.sum1 = ri.sum;
            ps.ssq1 = ri.ssq;
            ps.sum2 = sum_j;
            ps.ssq2 = ssq_j;
            n_true += emit_pair(A, i, j, lo_j, ps) ? 1 : 0;
          }
        }
      }
    }
  }
  n_true = wave_reduce_add(n_true);
  if ((lane == 0) && n_true) {
    atomicAdd(A.counters, static_cast<unsigned long long>(n_true));
  }
}

// ================================================================================================
// tiles with missing calls: all six statistics of ComputeIndepPairwiseR2Components on the matrix pipe
// ================================================================================================
// With missing calls the statistics run over the pairwise-complete samples (plink2_ld.cc:699-723, SumSsqWords :317-335,
// SumSsqNmWords :578-602).  Per variant three vectors over the samples: x in {-1, 0, +1} as above, n = 1 where the call
// is present (hom | ref2het), h = x^2 (hom).  For first variant i and second variant j:
//   dot = x_i.x_j   nm = n_i.n_j   ssq2 = n_i.h_j   sum2 = n_i.x_j   ssq1 = h_i.n_j   sum1 = x_i.n_j
// six integer matrix products instead of one, against seven popcounts instead of two on the VALU path.  All three vectors are
// coded 2.0 at nibble bit 2 (x with its sign at bit 3) and every product carries the 1/2 x 1/2 block scale.  x is in the
// image's orientation: dot and the two sums take their signs from the records' ALT-major flags in the epilogue.
// A workgroup owns ONE second-variant block of a wave item and its (up to) four first-variant blocks, one product per
// wave: 6 x 16 accumulator registers, the J block's three fragment sets of the stage kept in registers.
__device__ __forceinline__ void fp4_nh_of_codes(uint32_t c0, uint32_t c1, const Frag& fx, Frag& fn, Frag& fh) {
  fn.d[0] = fp4_n(c0);
  fn.d[1] = fp4_n(c0 << 2);
  fn.d[2] = fp4_n(c1);
  fn.d[3] = fp4_n(c1 << 2);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fh.d[q] = fx.d[q] & 0x44444444u;  // |x|
  }
}

constexpr uint32_t kMfGenRowBlocks = 5;  // J0, J1 and the three V blocks of the workgroup's half
constexpr uint32_t kMfGenInstr = kMfGenRowBlocks * 2;
constexpr uint32_t kMfGenDmaPerWave = (kMfGenInstr + kMfWaves - 1) / kMfWaves;
constexpr uint32_t kMfGenCpRound = 4;                                      // accumulator rows per checkpoint round
constexpr uint32_t kMfGenCpWaveDwords = 6 * kMfGenCpRound * 64;            // this wave's scratch for one round (6 KiB)
constexpr uint32_t kMfGenCpRowDwords = kMfBlock * (10 * 8 / 4);              // ... and its V block's rows, prepared (2 KiB)
constexpr uint32_t kMfGenCpStatDwords = kMfWaves * (kMfGenCpWaveDwords + kMfGenCpRowDwords);  // the rows' cp_gen slots follow (5 x 1 KiB)

// ---- can a pair still reach the threshold? (the six-product kernel's checkpoints) ----------------------------------------
// With missing calls the statistics run over the pairwise-complete samples C = C_P (visited, known exactly from the six partial
// sums) + C_R (the rest), and C_R is not pinned down by per-variant numbers: of variant i's own calls in the remainder R at most
// dmax_i = min(partner's missing calls in R, own calls in R) drop out.  The popcount kernel's bound (ldp_kernels.hip:
// pair_hopeless) puts every sum into an interval and multiplies the intervals, which at 1 % missing calls already loses the
// 25 % of headroom the complete-data bound has when it fires.  This one removes the dependence first:
//   * each variant is turned to its MINOR allele (z = copies of it: 0, 1, 2; r^2 is invariant) and CENTRED at its own mean
//     alpha over all its calls: z' = z - alpha.  cov = n sum z'w' - sum z' sum w' and var = n sum z'^2 - (sum z')^2 hold for any
//     shift, and with this one the sums of z' are small numbers (sampling noise plus what drops out), so the products of sums
//     -- where interval arithmetic loses most -- hardly matter;
//   * |sum over C_R of z'w'| <= sqrt(sum_{R, i called} z'^2 . sum_{R, j called} w'^2) (Cauchy-Schwarz; a sum of squares only grows
//     with its index set), whatever drops out;
//   * what drops out of a sum is at most its dmax largest (for the lower end: most negative) terms, and the remainder's calls
//     are known by value: c0, c1, c2 of them have z = 0, 1, 2 (from its calls, sum z and sum z^2).
// The pair is hopeless when (n_hi (|ZW'_P| + CS) + |Zs'|max |Ws'|max + 1)^2 (1 + 1e-6) < thresh (1 - 1e-6) var1_lo var2_lo with
// var_lo = n_lo Q'_lo - |Zs'|max^2.  On the bench generator (50,000 samples, r^2 0.5, 1 % missing calls) 97 % of unrelated pairs
// are hopeless 0.08 behind 1 - sqrt(r2) of the samples and all of them at 0.16; the interval product: 11 % and about half.
struct GenRow {
  double sgn;         // -1: the image's x counts the major allele's copies with +1
  double alpha;       // mean of z over the variant's calls
  double nr;          // its calls in the remainder
  double a, b;        // sum of z', of z'^2 over them
  double c0, c1, c2;  // how many of them have z = 0, 1, 2
  double hom_p;       // its homozygous calls in the visited part
  double miss_p;      // its missing calls there
};
__device__ __forceinline__ GenRow gen_row(const cp_gen_slot& cp, const cp_gen_slot& whole, double seen) {
  GenRow g;
  const bool flip = whole.pad != 0;
  const double nr = cp.nm_r, nt = whole.nm_r;
  double a = cp.zs_r, b = cp.zq_r, zt = whole.zs_r;
  // homozygous calls = sum z^2 - 2 sum z + calls (z in {0, 2} <-> (z - 1)^2 = 1), whichever allele z counts
  g.hom_p = (static_cast<double>(whole.zq_r) - 2.0 * whole.zs_r + nt) - (b - 2.0 * a + nr);
  g.miss_p = fmax(seen - (nt - nr), 0.0);
  if (flip) {  // z -> 2 - z
    b = 4.0 * nr - 4.0 * a + b;
    a = 2.0 * nr - a;
    zt = 2.0 * nt - zt;
  }
  g.sgn = flip ? -1.0 : 1.0;
  g.alpha = (nt > 0.0) ? zt / nt : 0.0;
  g.nr = nr;
  g.c2 = 0.5 * (b - a);
  g.c1 = a - 2.0 * g.c2;
  g.c0 = nr - g.c1 - g.c2;
  g.a = a - g.alpha * nr;
  g.b = fmax(b - 2.0 * g.alpha * a + g.alpha * g.alpha * nr, 0.0);
  return g;
}
// the most that dropping at most d of the remainder's calls can take from sum z' (hi), add to it (lo), take from sum z'^2 (q);
// alpha in [0, 1] (checked by the caller)
__device__ __forceinline__ void gen_drops(const GenRow& g, double d, double* hi, double* lo, double* q) {
  const double al = g.alpha;
  const double k2 = fmin(d, g.c2), k1 = fmin(d - k2, g.c1);
  *hi = (2.0 - al) * k2 + (1.0 - al) * k1;  // (the terms z' = 2 - alpha, then 1 - alpha >= 0)
  *lo = al * fmin(d, g.c0);                  // (the terms z' = -alpha)
  // squares: (2 - alpha)^2 first, then the larger of (1 - alpha)^2 and alpha^2
  const bool one_first = (1.0 - al) >= al;
  const double va = one_first ? (1.0 - al) * (1.0 - al) : al * al, ca = one_first ? g.c1 : g.c0;
  const double vb = one_first ? al * al : (1.0 - al) * (1.0 - al), cb = one_first ? g.c0 : g.c1;
  const double ka = fmin(d - k2, ca), kb = fmin(d - k2 - ka, cb);
  *q = (2.0 - al) * (2.0 - al) * k2 + va * ka + vb * kb;
}
// SIX = false (the four-product form has no partial sums of squares): q1 >= max(|s1|, i's homozygous calls in the visited part - j's
// missing calls there), which only lowers the variance bound.
template <bool SIX>
__device__ __forceinline__ bool pair_hopeless_centred(double thresh, double n_p, double s1, double q1, double s2, double q2, double dot, const GenRow& I,
                                                      const GenRow& J, double rs) {
  if constexpr (!SIX) {
    q1 = fmax(fabs(s1), I.hom_p - J.miss_p);
    q2 = fmax(fabs(s2), J.hom_p - I.miss_p);
  }
  s1 *= I.sgn;
  s2 *= J.sgn;
  dot *= I.sgn * J.sgn;
  const double al = I.alpha, be = J.alpha;
  // visited part, exact, in z = 1 - x and centred
  const double zs_p = n_p - s1, zq_p = n_p - 2.0 * s1 + q1;
  const double ws_p = n_p - s2, wq_p = n_p - 2.0 * s2 + q2;
  const double zw_p = n_p - s1 - s2 + dot;
  const double Zs = zs_p - al * n_p, Ws = ws_p - be * n_p;
  const double Zq = fmax(zq_p - 2.0 * al * zs_p + al * al * n_p, 0.0), Wq = fmax(wq_p - 2.0 * be * ws_p + be * be * n_p, 0.0);
  const double ZW = zw_p - al * ws_p - be * zs_p + al * be * n_p;
  const double dmax_i = fmin(rs - J.nr, I.nr), dmax_j = fmin(rs - I.nr, J.nr);
  const double n_lo = n_p + fmax(I.nr - dmax_i, J.nr - dmax_j);
  const double n_hi = n_p + fmin(I.nr, J.nr);
  double hi_i, lo_i, dq_i, hi_j, lo_j, dq_j;
  gen_drops(I, dmax_i, &hi_i, &lo_i, &dq_i);
  gen_drops(J, dmax_j, &hi_j, &lo_j, &dq_j);
  const double zsm = fmax(fabs(Zs + I.a + lo_i), fabs(Zs + I.a - hi_i));
  const double wsm = fmax(fabs(Ws + J.a + lo_j), fabs(Ws + J.a - hi_j));
  const double q1lo = fmax(Zq + I.b - dq_i, Zq), q2lo = fmax(Wq + J.b - dq_j, Wq);
  const double cmax = n_hi * (fabs(ZW) + sqrt(I.b * J.b) * (1.0 + 1e-9)) + zsm * wsm + 1.0;
  const double var1_lo = n_lo * q1lo - zsm * zsm;
  const double var2_lo = n_lo * q2lo - wsm * wsm;
  return (al <= 1.0) && (be <= 1.0) && (var1_lo > 0.0) && (var2_lo > 0.0) && (cmax * cmax * (1.0 + 1e-6) < thresh * (1.0 - 1e-6) * var1_lo * var2_lo);
}

// A workgroup owns one HALF of a wave item -- its four products at block distances {0, 1} (half 1: on the diagonal the blocks
// next to it, which hold the pairs in LD) or {2, 3} (half 0: the far end of the window) -- one product per wave:
// wave w = (J_q, V_{q + t}), q = w >> 1, t = 2 half + (w & 1).  Both halves stage five row-blocks (J0, J1, three V blocks).
// Early termination is by whole workgroups: at a checkpoint every wave bounds its 1,024 pairs with pair_hopeless_centred(); a wave
// whose pairs are all hopeless stops multiplying, and when that is all four the workgroup leaves and the CU takes the next
// one.  (Round 2's layout -- one J block against its four distances per workgroup -- retired the far waves of EVERY workgroup,
// which emptied two SIMDs of each CU and gained no time; here the far products of a window sit in workgroups of their own.)
// The four-product forms' decision for one pair from dot, nm, sum1, sum2 (in ps, major-allele orientation) and the two variants'
// records: 1 = the predicate holds, 0 = it does not, 2 = open (recount).  See the comment on pair_mfma_general_kernel.
__device__ __forceinline__ int classify_four(const ldp_pair_stats_t& ps, const ldp_variant_rec& ri, const ldp_variant_rec& rj, int64_t N, double thresh) {
  const int64_t Mi = N - ri.nm_ct, Mj = N - rj.nm_ct;
  const int64_t mm = static_cast<int64_t>(ps.nm) + Mi + Mj - N;  // missing in both
  const int64_t R1 = Mj - mm, R2 = Mi - mm;                      // j missing and i called; i missing and j called
  int64_t xm1 = static_cast<int64_t>(ri.sum) - ps.sum1, xm2 = static_cast<int64_t>(rj.sum) - ps.sum2;
  xm1 = (xm1 < 0) ? -xm1 : xm1;
  xm2 = (xm2 < 0) ? -xm2 : xm2;
  const int64_t hm1_hi = R1 - ((R1 - xm1) & 1), hm2_hi = R2 - ((R2 - xm2) & 1);
  ldp_pair_stats_t hi = ps, lo = ps;
  hi.ssq1 = static_cast<uint32_t>(static_cast<int64_t>(ri.ssq) - xm1);
  hi.ssq2 = static_cast<uint32_t>(static_cast<int64_t>(rj.ssq) - xm2);
  const int64_t l1 = static_cast<int64_t>(ri.ssq) - hm1_hi, l2 = static_cast<int64_t>(rj.ssq) - hm2_hi;
  lo.ssq1 = static_cast<uint32_t>((l1 > 0) ? l1 : 0);
  lo.ssq2 = static_cast<uint32_t>((l2 > 0) ? l2 : 0);
  // (a consistent pair of inputs has 0 <= xm <= R; anything else is left to the exact route)
  const bool sane = (mm >= 0) && (xm1 <= R1) && (xm2 <= R2) && (xm1 <= static_cast<int64_t>(ri.ssq)) && (xm2 <= static_cast<int64_t>(rj.ssq));
  if (sane && exceeds(hi, thresh)) {
    return 1;
  }
  if (sane && !exceeds_clamped(lo, thresh)) {
    return 0;
  }
  return 2;
}

// SIX = false, prune launches (only the predicate is wanted): FOUR products -- dot, nm, sum1, sum2 -- and the two sums of squares
// from per-variant numbers: ssq1 = (i's homozygous calls) - hm, hm = those of them where j is missing.  There are
// R = M_j - (calls missing in both) = M_j - (nm + M_i + M_j - N) samples where j is missing and i is not, the x_i over them sum to
// S_i - sum1 =: xm, so hm lies in [|xm|, R] with the parity of xm.  The predicate cov^2 > thresh var1 var2 is monotone in ssq1 and
// ssq2 (the reference's own rounding included: products of non-negative doubles), so it is decided whenever both ends of the
// intervals agree; the few pairs left open are counted exactly on the spot by the whole wave (wave_pair_counts, as in 4.1d).
template <bool SIX, bool GU>
__global__ __launch_bounds__(kMfWaves * 64, 3) void pair_mfma_general_kernel(PairKernelArgs A) {
  using G = StageGeom<4>;
  constexpr int NP = SIX ? 6 : 4;
  // GU (four-product form, founder_ct <= kMfGuMaxFounders): the operands are the allele counts g' and the missing flags u
  // (ldp_mfma_device.h): accumulators (P1, P4, P3, P2), turned into (dot, nm, sum2, sum1) where they are read; otherwise x, n (, h)
  static_assert(!(SIX && GU), "the six-product form keeps x, n, h");
  constexpr bool ZC = GU;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_src_off[kMfGenDmaPerWave * kMfWaves * 64];
  __shared__ uint32_t s_live_waves;
  {
    if (*A.route != kRouteGeneral) {
      return;  // complete data, or few enough missing calls for its interval epilogue: pair_mfma_kernel owns this launch
    }
  }
  const uint32_t n_units = A.n_mf_wgs * 8;  // workgroup x wave item x half
  const uint32_t per_xcd = (n_units + 7) / 8;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (idx >= n_units) {
    return;
  }
  const MfmaWG* __restrict__ wg = A.mf_wgs + (idx >> 3);
  if (A.wd_general && (wg->pad & 1u)) {
    return;  // the subcontig has the tile plan: pair_mfma_tile4_kernel owns its pairs in this launch
  }
  const MfmaWaveItem* __restrict__ wi = wg->w + ((idx >> 1) & 3);
  const uint32_t half = idx & 1;
  const int32_t jv0 = wi->jv;
  if (jv0 < 0) {
    return;
  }
  const uint32_t mask8 = wi->prod_mask;
  // bit w = the product of wave w
  const uint32_t mask4 = ((mask8 >> (2 * half)) & 3u) | (((mask8 >> (4 + 2 * half)) & 3u) << 2);
  if (!mask4) {
    return;
  }
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const uint32_t r = lane & 31;
  const uint32_t h = lane >> 5;
  const uint32_t q = wave >> 1;
  const uint32_t vk = q + (wave & 1);  // V block of this wave's product among the half's three: V_{2 half + vk}
  const int32_t jfirst = jv0 + static_cast<int32_t>(kMfBlock * q);
  const int32_t vv = wi->vv;
  const uint32_t jend = wi->jend;
  bool live = (mask4 >> wave) & 1u;
  const int32_t vfirst_blk = vv + static_cast<int32_t>(kMfBlock * (2 * half + vk));
  const uint32_t row_bytes = static_cast<uint32_t>(A.code_row_bytes);
  const uint32_t n_stages = (A.founder_ct + G::kStageSamples - 1) / G::kStageSamples;
  const uint32_t stage_dwords = kMfGenInstr * 256;
  uint32_t stages = A.lds_dwords / stage_dwords;
  stages = (stages > kMfMaxStages) ? kMfMaxStages : stages;
  const uint32_t mine = (kMfGenInstr > wave) ? (kMfGenInstr - wave + kMfWaves - 1) / kMfWaves : 0;

  // ---- DMA plan: row-block slots 0, 1 = J0, J1; 2 + k = V_{2 half + k}.  A block no live product reads is not fetched: its slot
  // simply fetches the rows of the first block that is.
  const uint32_t need_j = ((mask4 & 3u) ? 1u : 0u) | ((mask4 & 12u) ? 2u : 0u);
  const uint32_t need_v = ((mask4 & 1u) ? 1u : 0u) | ((mask4 & 6u) ? 2u : 0u) | ((mask4 & 8u) ? 4u : 0u);  // (V_k: wave 0 reads 0, waves 1 and 2 read 1, wave 3 reads 2)
  const int32_t first_needed = jv0 + static_cast<int32_t>((need_j & 1u) ? 0 : kMfBlock);
  auto slot_first = [&](uint32_t blk) -> int32_t {
    if (blk < 2) {
      return ((need_j >> blk) & 1u) ? (jv0 + static_cast<int32_t>(kMfBlock * blk)) : first_needed;
    }
    return ((need_v >> (blk - 2)) & 1u) ? (vv + static_cast<int32_t>(kMfBlock * (2 * half + blk - 2))) : first_needed;
  };
  const uint8_t* base_t[kMfGenDmaPerWave];
#pragma unroll
  for (int t = 0; t < static_cast<int>(kMfGenDmaPerWave); ++t) {
    const uint32_t T = wave + kMfWaves * t;
    base_t[t] = A.codes;
    if (T < kMfGenInstr) {
      const int32_t first = __builtin_amdgcn_readfirstlane(slot_first(T >> 1));
      base_t[t] += static_cast<uint64_t>(static_cast<uint32_t>(first)) * row_bytes;
      const uint32_t L = T * 64 + lane;
      const uint32_t rr = (L >> 2) & 31;
      const uint32_t col = (L & 3) ^ G::swizzle(rr);
      uint32_t var = static_cast<uint32_t>(first) + rr;
      var = (var < A.n_local) ? var : (A.n_local - 1);
      s_src_off[t * (kMfWaves * 64) + tid] = (var - static_cast<uint32_t>(first)) * row_bytes + G::piece_byte(col);
    }
  }
  auto dma_stage = [&](uint32_t s, uint32_t buf) {
    const uint32_t kbyte = G::stage_byte(s);
    uint32_t* dst = lds + buf * stage_dwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kMfGenDmaPerWave); ++t) {
      const uint32_t T = wave + kMfWaves * t;
      if (T < kMfGenInstr) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_t[t] + kbyte + s_src_off[t * (kMfWaves * 64) + tid]),
                                         (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
      }
    }
  };

  // window start, subcontig end of this lane's second variant
  const int64_t j64 = static_cast<int64_t>(jfirst) + r;
  uint32_t lo_j = 0xffffffffu;
  if (j64 < static_cast<int64_t>(jend)) {
    lo_j = A.lo[static_cast<uint32_t>(j64)];
  }
  const uint32_t sw = G::swizzle(r);
  const uint32_t j_slot = q * G::kBlockSlots;
  const uint32_t oH = r * 4 + (h ^ sw);
  const uint32_t oR = r * 4 + ((2 + h) ^ sw);
  const uint32_t v_slot = (2 + vk) * G::kBlockSlots;

  mf_v16f acc[NP];  // [0] x.x  [1] n.n  [2] n_i.x_j  [3] x_i.n_j  [4] n_i.h_j  [5] h_i.n_j
#pragma unroll
  for (int c = 0; c < NP; ++c) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[c][g] = 0.f;
    }
  }
  // ---- checkpoints: the far half of every wave item, and both halves away from the diagonal.  Two of the list's five (the
  // second and third: 0.04 and 0.08 behind 1 - sqrt(r2)): each costs a drained ring and ~1,000 FP64 operations per lane.
  const bool diag = (vv + 3 * static_cast<int32_t>(kMfBlock) == jv0);
  const uint32_t cp_all = (A.cp_stats && ((half == 0) || !diag)) ? A.n_checkpoints : 0;
  const uint32_t n_cp = (cp_all >= kGenCheckpointFirst + kGenCheckpoints) ? kGenCheckpoints : ((cp_all > kGenCheckpointFirst) ? (cp_all - kGenCheckpointFirst) : 0u);
  auto cp_of = [&](uint32_t k) { return kGenCheckpointFirst + k; };
  auto checkpoint_stage = [&](uint32_t cp) {
    const uint32_t s = A.checkpoint_chunk[cp] * G::kStagesPerChunk;
    return (s < n_stages) ? s : n_stages;
  };
  __syncthreads();  // (s_src_off is complete)
  uint32_t issued = 0, issue_buf = 0, read_buf = 0, issued_base = 0;
  uint32_t next_cp = 0;
  uint32_t issue_limit = (next_cp < n_cp) ? checkpoint_stage(cp_of(next_cp)) : n_stages;
  auto ring_fill = [&]() {
    issue_buf = 0;
    read_buf = 0;
    while ((issued < issue_limit) && (issued + 1 < issued_base + stages)) {
      dma_stage(issued, issue_buf);
      ++issued;
      issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
    }
  };
  ring_fill();
  uint32_t stop_stage = n_stages;  // where this wave's product was retired (bookkeeping)
  for (uint32_t kc = 0; kc < n_stages;) {
    const uint32_t kc_end = issue_limit;
    for (; kc < kc_end; ++kc) {
      wait_dma_then_barrier(mine * (issued - kc - 1));
      if (issued < issue_limit) {
        dma_stage(issued, issue_buf);
        ++issued;
        issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
      }
      const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * stage_dwords);
      read_buf = (read_buf + 1 == stages) ? 0 : read_buf + 1;
      if (!live) {
        continue;
      }
      mf_u4 jH = st4[j_slot + oH], jR = st4[j_slot + oR];
      mf_u4 vH = st4[v_slot + oH], vR = st4[v_slot + oR];
      opaque(jH, jR);
      Frag jx[4], jn[4], jh[4];  // (ZC: jx holds z)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if constexpr (ZC) {
          fp4_gu_of_codes(jH[ks], jR[ks], jx[ks], jn[ks]);  // (jx = g', jn = u)
        } else {
          fp4_of_codes(jH[ks], jR[ks], jx[ks]);
          fp4_nh_of_codes(jH[ks], jR[ks], jx[ks], jn[ks], jh[ks]);
        }
      }
      opaque(vH, vR);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag vx, vn, vh;
        if constexpr (ZC) {
          fp4_gu_of_codes(vH[ks], vR[ks], vx, vn);
        } else {
          fp4_of_codes(vH[ks], vR[ks], vx);
          fp4_nh_of_codes(vH[ks], vR[ks], vx, vn, vh);
        }
        // rows of C = first variant i (A operand: the V block), columns = second variant j (B operand: the J block)
        acc[0] = mfma_pair<ZC>(vx, jx[ks], acc[0]);
        acc[1] = mfma_pair<ZC>(vn, jn[ks], acc[1]);
        acc[2] = mfma_pair<ZC>(vn, jx[ks], acc[2]);
        acc[3] = mfma_pair<ZC>(vx, jn[ks], acc[3]);
        if constexpr (SIX) {
          acc[4] = mfma_fp4(vn, jh[ks], acc[4]);
          acc[5] = mfma_fp4(vh, jn[ks], acc[5]);
        }
      }
    }
    if (kc >= n_stages) {
      break;
    }
    // ---- checkpoint (block-uniform; issued == kc: the ring is empty and the LDS is scratch) ----
    __syncthreads();
    if (tid == 0) {
      s_live_waves = 0;
    }
    {
      // the rows' remainder statistics (cp_gen_slot, 16 bytes) come in by LDS-DMA like the rows themselves: row-block slot T by
      // wave T & 3, lanes 0..31 = its rows' slots of this checkpoint, lanes 32..63 = their whole-row slots
      const uint8_t* cpg = reinterpret_cast<const uint8_t*>(A.cp_stats);
      const uint32_t gen_slot = kCpSlots + 1 + next_cp;  // (the remainder behind checkpoint kGenCheckpointFirst + next_cp)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t T = wave + kMfWaves * t;
        if (T < kMfGenRowBlocks) {
          const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(slot_first(T)));
          uint32_t var = first + r;
          var = (var < A.n_local) ? var : (A.n_local - 1);
          const uint64_t off = (static_cast<uint64_t>(var) * kCpStride + (h ? static_cast<uint32_t>(kCpSlots) : gen_slot)) * sizeof(cp_gen_slot);  // lanes 32..63: the whole-row slot
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cpg + off),
                                           (__attribute__((address_space(3))) void*)(lds + kMfGenCpStatDwords + T * 256), 16, 0, 0);
        }
      }
    }
    __syncthreads();  // (drains the DMA: the slots are in LDS; s_live_waves is zero)
    if (live) {
      const cp_gen_slot* __restrict__ cpl = reinterpret_cast<const cp_gen_slot*>(lds + kMfGenCpStatDwords);  // [row-block slot][64 lanes]
      const uint64_t seen = static_cast<uint64_t>(kc) * G::kStageSamples;
      const double seen_d = static_cast<double>((seen < A.founder_ct) ? seen : A.founder_ct);
      const GenRow gj = gen_row(cpl[q * 64 + r], cpl[q * 64 + 32 + r], seen_d);
      const int32_t uj_p = static_cast<int32_t>(seen_d) - static_cast<int32_t>(cpl[q * 64 + 32 + r].nm_r - cpl[q * 64 + r].nm_r);
      const int32_t zj_p = static_cast<int32_t>(cpl[q * 64 + 32 + r].zs_r - cpl[q * 64 + r].zs_r);
      uint32_t* mine_epi = lds + wave * (kMfGenCpWaveDwords + kMfGenCpRowDwords);
      GenRow* rows_i = reinterpret_cast<GenRow*>(mine_epi + kMfGenCpWaveDwords);
      if (h == 0) {
        rows_i[r] = gen_row(cpl[(2 + vk) * 64 + r], cpl[(2 + vk) * 64 + 32 + r], seen_d);  // (read back by other lanes of this wave only)
      }
      const double rs = (seen < A.founder_ct) ? static_cast<double>(A.founder_ct - seen) : 1.0;
      bool hopeless = true;
#pragma unroll
      for (int round = 0; round < 16 / static_cast<int>(kMfGenCpRound); ++round) {
        if (__all(hopeless)) {  // (one pair that may still reach the threshold keeps the product: no need to look at the rest)
#pragma unroll
          for (int c = 0; c < NP; ++c) {
#pragma unroll
            for (int g = 0; g < static_cast<int>(kMfGenCpRound); ++g) {
              mine_epi[(c * kMfGenCpRound + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[c][round * kMfGenCpRound + g]));
            }
          }
#pragma unroll 1
          for (uint32_t gg = 0; gg < kMfGenCpRound; ++gg) {
            const uint32_t g = round * kMfGenCpRound + gg;
            const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * h;
            const int64_t i64 = static_cast<int64_t>(vfirst_blk) + row;
            if ((lo_j != 0xffffffffu) && (i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64)) {
              const GenRow gi = rows_i[row];
              auto raw = [&](uint32_t c) { return static_cast<int32_t>(mine_epi[(c * kMfGenCpRound + gg) * 64 + lane]); };
              int32_t dot_p = raw(0), nm_p = raw(1), s2_p = raw(2), s1_p = raw(3);
              if constexpr (ZC) {
                // the rows' missing calls and allele-count sums over the samples visited (cp_gen_slot: calls and sum z of the whole row
                // and of the remainder, image orientation; no padding among the samples visited)
                const cp_gen_slot ci = cpl[(2 + vk) * 64 + row], wi = cpl[(2 + vk) * 64 + 32 + row];
                const int32_t n_vis = static_cast<int32_t>(seen_d);
                x_from_gu(raw(0), raw(1), raw(2), raw(3), n_vis - static_cast<int32_t>(wi.nm_r - ci.nm_r), static_cast<int32_t>(wi.zs_r - ci.zs_r), uj_p, zj_p, n_vis, 0, &dot_p, &nm_p,
                          &s2_p, &s1_p);
              }
              auto val = [&](uint32_t c) { return static_cast<double>(raw(c)); };
              hopeless = hopeless && pair_hopeless_centred<SIX>(A.thresh, static_cast<double>(nm_p), static_cast<double>(s1_p), SIX ? val(5) : 0.0, static_cast<double>(s2_p),
                                                                SIX ? val(4) : 0.0, static_cast<double>(dot_p), gi, gj, rs);
            }
          }
        }
      }
      if (__all(hopeless)) {
        live = false;
        stop_stage = kc;
      } else if (lane == 0) {
        atomicAdd(&s_live_waves, 1u);
      }
    }
    __syncthreads();
    if (!s_live_waves) {
      break;  // nothing left for this workgroup
    }
    ++next_cp;
    issued_base = kc;
    issue_limit = (next_cp < n_cp) ? checkpoint_stage(cp_of(next_cp)) : n_stages;
    ring_fill();
  }
  if ((lane == 0) && (stop_stage < n_stages) && ((mask4 >> wave) & 1u)) {
    atomicAdd(A.counters + 2, static_cast<unsigned long long>(n_stages - stop_stage) * 4ull);  // product x 64-sample k-steps not multiplied
  }
  // ---- epilogue: straight from the registers (one product per wave) ----
  uint32_t n_true = 0;
  if constexpr (SIX) {
    if (live && (lo_j != 0xffffffffu) && (static_cast<int64_t>(lo_j) < j64)) {
      const uint32_t j = static_cast<uint32_t>(j64);
      const bool alt_j = img_differs(A.recs[j].flags) != 0;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int64_t i64 = static_cast<int64_t>(vfirst_blk) + (g & 3) + 8 * (g >> 2) + 4 * h;
        if ((i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64)) {
          // the image's orientation -> the records' (major allele): x_i, x_j change sign with their row's ALT-major flag
          const bool alt_i = img_differs(A.recs[static_cast<uint32_t>(i64)].flags) != 0;
          ldp_pair_stats_t ps;
          const int32_t d = static_cast<int32_t>(acc[0][g]), s2 = static_cast<int32_t>(acc[2][g]), s1 = static_cast<int32_t>(acc[3][g]);
          ps.dot = (alt_i != alt_j) ? -d : d;
          ps.nm = static_cast<uint32_t>(static_cast<int32_t>(acc[1][g]));
          ps.ssq2 = static_cast<uint32_t>(static_cast<int32_t>(acc[SIX ? 4 : 0][g]));
          ps.sum2 = alt_j ? -s2 : s2;
          ps.ssq1 = static_cast<uint32_t>(static_cast<int32_t>(acc[SIX ? 5 : 0][g]));
          ps.sum1 = alt_i ? -s1 : s1;
          n_true += emit_pair(A, static_cast<uint32_t>(i64), j, lo_j, ps) ? 1 : 0;
        }
      }
    }
  } else {
    __syncthreads();  // (every wave is past its last stage: the ring is this epilogue's scratch)
  }
  if constexpr (!SIX) if (live) {  // (wave-uniform: the exact route below needs the whole wave)
    // the accumulators go through LDS, eight rows at a time, so that the classification is ONE rolled loop: unrolled sixteen
    // times next to 64 live accumulators it made hipcc spill three of them inside the stage loop
    uint32_t* epi4 = lds + wave * (4 * 8 * 64);
    const bool j_ok = (lo_j != 0xffffffffu) && (static_cast<int64_t>(lo_j) < j64);
    const uint32_t j = j_ok ? static_cast<uint32_t>(j64) : 0u;
    ldp_variant_rec rj;
    rj.nm_ct = 0;
    rj.sum = 0;
    rj.ssq = 0;
    rj.flags = 0;
    if (j_ok) {
      rj = A.recs[j];
    }
    const bool alt_j = img_differs(rj.flags) != 0;
    const int64_t N = static_cast<int64_t>(A.founder_ct);
    uint32_t n_open = 0;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
          epi4[(c * 8 + g8) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[c][round * 8 + g8]));
        }
      }
#pragma unroll 1
    for (uint32_t g8 = 0; g8 < 8; ++g8) {
      const uint32_t g = round * 8 + g8;
      const int64_t i64 = static_cast<int64_t>(vfirst_blk) + (g & 3) + 8 * (g >> 2) + 4 * h;
      const bool valid = j_ok && (i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64);
      const uint32_t i = valid ? static_cast<uint32_t>(i64) : 0u;
      ldp_pair_stats_t ps;
      ps.dot = 0;
      int cls = 0;
      uint32_t alt_ij = 0;
      if (valid) {
        const ldp_variant_rec ri = A.recs[i];
        const bool alt_i = img_differs(ri.flags) != 0;
        alt_ij = (alt_i ? 1u : 0u) | (alt_j ? 2u : 0u);
        int32_t d = static_cast<int32_t>(epi4[(0 * 8 + g8) * 64 + lane]), s2 = static_cast<int32_t>(epi4[(2 * 8 + g8) * 64 + lane]),
                s1 = static_cast<int32_t>(epi4[(3 * 8 + g8) * 64 + lane]), nm = static_cast<int32_t>(epi4[(1 * 8 + g8) * 64 + lane]);
        if constexpr (ZC) {
          int32_t Ui, Zi, Uj, Zj;
          uz_of_rec(ri, A.founder_ct, &Ui, &Zi);
          uz_of_rec(rj, A.founder_ct, &Uj, &Zj);
          x_from_gu(d, nm, s2, s1, Ui, Zi, Uj, Zj, static_cast<int32_t>(A.founder_ct), static_cast<int32_t>(n_stages * G::kStageSamples - A.founder_ct), &d, &nm, &s2, &s1);
        }
        ps.nm = static_cast<uint32_t>(nm);
        ps.dot = (alt_i != alt_j) ? -d : d;
        ps.sum2 = alt_j ? -s2 : s2;
        ps.sum1 = alt_i ? -s1 : s1;
        cls = classify_four(ps, ri, rj, N, A.thresh);
        if (cls == 1) {
          atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
          ++n_true;
        }
      }
      unsigned long long open = __ballot(cls == 2);
      while (open) {
        const int l = __builtin_ctzll(open);
        open &= open - 1;
        const uint32_t ii = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i), l));
        const uint32_t jj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
        const int32_t dd = __builtin_amdgcn_readlane(ps.dot, l);
        const uint32_t aa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(alt_ij), l));
        const ldp_pair_stats_t st = wave_pair_counts(A, ii, jj, dd, lane, (aa & 1u) != 0, (aa & 2u) != 0);
        if ((static_cast<int>(lane) == l) && exceeds(st, A.thresh)) {
          atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
          ++n_true;
        }
        n_open += (lane == 0) ? 1u : 0u;
      }
    }
    }
    if ((lane == 0) && n_open) {
      atomicAdd(A.counters + 3, static_cast<unsigned long long>(n_open));
    }
  }
  n_true = wave_reduce_add(n_true);
  if ((lane == 0) && n_true) {
    atomicAdd(A.counters, static_cast<unsigned long long>(n_true));
  }
}


// ---- the four-product form on the TILE plan of wide bands ------------------------------------------------------------------
// At config 5's density the kernel above fetches 37 x its compulsory bytes (profiles/r03_c5shape_pmc_traffic.json): half a wave
// item stages five row-blocks for four block products, and a row-block is 4 MB at 500,000 samples.  Subcontigs that have the
// wide plan (MfmaTile: 8 x 8 row-blocks, ldp_pair_wide.hip) are therefore taken in QUARTER tiles here: a workgroup of eight
// waves stages 4 J + 4 V row-blocks (0.5 per product), wave w owns J block w & 3 against V blocks 2 (w >> 2), + 1 -- two
// products that share the expanded J operand, 2 x 4 x 16 accumulators, two waves per SIMD.  Stages, checkpoints (whole waves
// retire, the workgroup leaves when all have) and the interval epilogue are those of pair_mfma_general_kernel<false>.
template <int JB>
struct T4 {  // JB J row-blocks x 4 V row-blocks per workgroup, 2 JB waves
  static_assert((JB == 2) || (JB == 4), "half or quarter of a tile's J blocks");
  static constexpr uint32_t kWaves = 2 * JB;
  static constexpr uint32_t kBlocks = JB + 4;                   // staged row-blocks: slots 0 .. JB - 1 = J, the next four = V
  static constexpr uint32_t kInstr = kBlocks * 2;               // DMA instructions per 256-sample stage
  static constexpr uint32_t kDmaPerWave = kInstr / kWaves;      // 2 (JB = 4), 3 (JB = 2)
  static constexpr uint32_t kStageDwords = kInstr * 256;        // 16 / 12 KiB
  static constexpr uint32_t kCpWaveDwords = 4 * kMfGenCpRound * 64 + kMfGenCpRowDwords;  // a wave's checkpoint scratch: four sums x rows, its V block's rows prepared
  static constexpr uint32_t kCpStatDwords = kWaves * kCpWaveDwords;                      // the rows' slots follow (1 KiB per row-block)
  static constexpr uint32_t kLdsDwords = 6 * kStageDwords;      // a ring of six stages: 96 KiB, one workgroup per CU (JB = 4) / 72 KiB, two (JB = 2)
  static constexpr uint32_t kUnits = 2 * (8 / JB);              // workgroups per tile
  static_assert(kInstr % kWaves == 0, "whole instructions per wave");
  static_assert(kCpStatDwords + kBlocks * 256 <= kLdsDwords, "checkpoint scratch inside the ring");
  static_assert(kWaves * 4 * 8 * 64 <= kLdsDwords, "epilogue scratch inside the ring");
};

template <int JB, bool GU>
__global__ __launch_bounds__(T4<JB>::kWaves * 64, 2) void pair_mfma_tile4_kernel(PairKernelArgs A) {
  using G = StageGeom<4>;
  using C = T4<JB>;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_src_off[C::kInstr * 64];
  __shared__ uint32_t s_live_waves;
  __shared__ uint32_t s_need;
  {
    if (*A.route != kRouteGeneral) {
      return;
    }
  }
  const MfmaTile* __restrict__ tiles = A.wd_tiles_plain ? A.wd_tiles_plain : A.wd_tiles;  // (J order: ldp_device.h)
  const uint32_t n_units = (A.wd_tiles_plain ? A.n_wd_tiles_plain : A.n_wd_tiles) * C::kUnits;
  const uint32_t per_xcd = (n_units + 7) / 8;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (idx >= n_units) {
    return;
  }
  const MfmaTile* __restrict__ tile = tiles + (idx / C::kUnits);
  const uint32_t qj = (idx % C::kUnits) >> 1, qv = idx & 1u;  // J blocks JB qj .., V blocks 4 qv ..
  const int32_t jv0 = __builtin_amdgcn_readfirstlane(tile->jv) + static_cast<int32_t>(kMfBlock * JB * qj);
  const int32_t vv0 = __builtin_amdgcn_readfirstlane(tile->vv) + static_cast<int32_t>(kMfBlock * 4 * qv);
  const uint32_t jend = __builtin_amdgcn_readfirstlane(tile->jend);
  const unsigned long long tmask = tile->mask;
  // bit 4 a + b: product (J block a, V block b) of this part of the tile holds candidate pairs
  uint32_t mask16 = 0;
#pragma unroll
  for (uint32_t a = 0; a < static_cast<uint32_t>(JB); ++a) {
    mask16 |= (static_cast<uint32_t>(tmask >> (8 * (JB * qj + a) + 4 * qv)) & 0xfu) << (4 * a);
  }
  mask16 = __builtin_amdgcn_readfirstlane(mask16);
  if (!mask16) {
    return;
  }
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const uint32_t r = lane & 31;
  const uint32_t h = lane >> 5;
  const uint32_t ja = wave % JB, vb0 = 2 * (wave / JB);
  uint32_t live = (mask16 >> (4 * ja + vb0)) & 3u;  // bit p: (J block ja, V block vb0 + p)
  const uint32_t mine_products = live;
  const int32_t jfirst = jv0 + static_cast<int32_t>(kMfBlock * ja);
  const int32_t vfirst0 = vv0 + static_cast<int32_t>(kMfBlock * vb0);
  const uint32_t row_bytes = static_cast<uint32_t>(A.code_row_bytes);
  // stages are taken in PAIRS (one barrier per 512 samples, as in the wide-band kernel: the image's rows are whole 512-sample
  // k-chunks long, padded with missing calls, and the checkpoints sit on chunk boundaries)
  const uint32_t n_stages = 2 * ((A.founder_ct + 2 * G::kStageSamples - 1) / (2 * G::kStageSamples));
  uint32_t stages = A.lds_dwords / C::kStageDwords;
  stages = (stages > kMfMaxStages) ? kMfMaxStages : stages;
  stages &= ~1u;  // (launch_pair_mfma gives six)

  // ---- DMA plan: row-block slot s (J first, then V) is fetched while bit s of `need` is set: some live product reads it.  The
  // set shrinks at the checkpoints (s_need); a wave issues instructions wave, wave + kWaves, .. of a stage (instruction T = half of
  // row-block slot T >> 1).
  uint32_t need = 0;
#pragma unroll
  for (uint32_t a = 0; a < static_cast<uint32_t>(JB); ++a) {
    const uint32_t rowm = (mask16 >> (4 * a)) & 0xfu;
    need |= (rowm ? (1u << a) : 0u) | (rowm << JB);
  }
  auto slot_first = [&](uint32_t slot) -> int32_t {
    return (slot < static_cast<uint32_t>(JB)) ? (jv0 + static_cast<int32_t>(kMfBlock * slot)) : (vv0 + static_cast<int32_t>(kMfBlock * (slot - JB)));
  };
  auto my_bit = [&](int t) { return 1u << ((wave + C::kWaves * static_cast<uint32_t>(t)) >> 1); };
  auto count_mine = [&]() {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < static_cast<int>(C::kDmaPerWave); ++t) {
      m += (need & my_bit(t)) ? 1u : 0u;
    }
    return m;
  };
  uint32_t mine = count_mine();  // DMA instructions per stage this wave issues
  const uint8_t* base_t[C::kDmaPerWave];
#pragma unroll
  for (int t = 0; t < static_cast<int>(C::kDmaPerWave); ++t) {
    const uint32_t T = wave + C::kWaves * t;
    uint32_t first = static_cast<uint32_t>(slot_first(T >> 1));
    first = (first < A.n_local) ? first : (A.n_local - 1);
    first = __builtin_amdgcn_readfirstlane(first);
    base_t[t] = A.codes + static_cast<uint64_t>(first) * row_bytes;
    const uint32_t L = T * 64 + lane;
    const uint32_t rr = (L >> 2) & 31;
    const uint32_t col = (L & 3) ^ G::swizzle(rr);
    uint32_t var = first + rr;
    var = (var < A.n_local) ? var : (A.n_local - 1);
    s_src_off[t * (C::kWaves * 64) + tid] = (var - first) * row_bytes + G::piece_byte(col);
  }
  auto dma_stage = [&](uint32_t s, uint32_t buf) {
    const uint32_t kbyte = G::stage_byte(s);
    uint32_t* dst = lds + buf * C::kStageDwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(C::kDmaPerWave); ++t) {
      const uint32_t T = wave + C::kWaves * t;
      if (need & my_bit(t)) {  // (wave-uniform)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_t[t] + kbyte + s_src_off[t * (C::kWaves * 64) + tid]),
                                         (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
      }
    }
  };

  const int64_t j64 = static_cast<int64_t>(jfirst) + r;
  uint32_t lo_j = 0xffffffffu;
  if ((j64 >= 0) && (j64 < static_cast<int64_t>(jend))) {
    lo_j = A.lo[static_cast<uint32_t>(j64)];
  }
  const uint32_t sw = G::swizzle(r);
  const uint32_t j_slot = ja * G::kBlockSlots;
  const uint32_t oH = r * 4 + (h ^ sw);
  const uint32_t oR = r * 4 + ((2 + h) ^ sw);
  const uint32_t v_slot0 = (JB + vb0) * G::kBlockSlots, v_slot1 = (JB + 1 + vb0) * G::kBlockSlots;

  mf_v16f acc[2][4];  // [product][0 x.x  1 n.n  2 n_i.x_j  3 x_i.n_j]; GU: [0 g'.g'  1 u.u  2 u_i.g'_j  3 g'_i.u_j], turned into dot, nm, sum2, sum1 where they are read
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        acc[p][c][g] = 0.f;
      }
    }
  }
  const uint32_t cp_all = A.cp_stats ? A.n_checkpoints : 0;
  const uint32_t n_cp = (cp_all >= kGenCheckpointFirst + kGenCheckpoints) ? kGenCheckpoints : ((cp_all > kGenCheckpointFirst) ? (cp_all - kGenCheckpointFirst) : 0u);
  auto checkpoint_stage = [&](uint32_t cp) {
    const uint32_t s = A.checkpoint_chunk[cp] * G::kStagesPerChunk;
    return (s < n_stages) ? s : n_stages;
  };
  __syncthreads();  // (s_src_off is complete)
  uint32_t issued = 0, issue_buf = 0, read_buf = 0, issued_base = 0;
  uint32_t next_cp = 0;
  uint32_t issue_limit = (next_cp < n_cp) ? checkpoint_stage(kGenCheckpointFirst + next_cp) : n_stages;
  auto ring_fill = [&]() {
    issue_buf = 0;
    read_buf = 0;
    while ((issued < issue_limit) && (issued + 2 < issued_base + stages)) {
      dma_stage(issued, issue_buf);
      ++issued;
      issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
    }
  };
  ring_fill();
  // both stages of pair kc, kc + 1 have landed and every wave is done with the pair before: two more stages go into its buffers
  auto advance = [&](uint32_t kc) {
    wait_dma_then_barrier(mine * (issued - kc - 2));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (issued < issue_limit) {
        dma_stage(issued, issue_buf);
        ++issued;
        issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
      }
    }
    const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * C::kStageDwords);
    read_buf = (read_buf + 2 == stages) ? 0 : read_buf + 2;
    return st4;
  };
  // ONE form of the stage (both products whenever the wave has a live one: a retired or never-wanted product accumulates numbers
  // nobody reads -- several forms chosen per segment made hipcc keep a copy of the accumulators per form and spill)
  uint32_t stop_stage[2] = {n_stages, n_stages};
  for (uint32_t kc = 0; kc < n_stages;) {
    const uint32_t kc_end = issue_limit;
    for (; kc < kc_end; kc += 2) {
      const mf_u4* __restrict__ st4 = advance(kc);
      if (!live) {
        continue;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const mf_u4* __restrict__ sq = st4 + q * (C::kStageDwords / 4);
        mf_u4 jH = sq[j_slot + oH], jR = sq[j_slot + oR];
        mf_u4 aH = sq[v_slot0 + oH], aR = sq[v_slot0 + oR];
        mf_u4 bH = sq[v_slot1 + oH], bR = sq[v_slot1 + oR];
        opaque(jH, jR);
        opaque(aH, aR);
        opaque(bH, bR);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          // the J operand of the k-step, expanded once for both products
          // the J operand of the k-step, expanded once for both products (GU: jx = g', jn = u, ldp_mfma_device.h; otherwise x, n)
          Frag jx, jn, unused;
          if constexpr (GU) {
            fp4_gu_of_codes(jH[ks], jR[ks], jx, jn);
          } else {
            fp4_of_codes(jH[ks], jR[ks], jx);
            fp4_nh_of_codes(jH[ks], jR[ks], jx, jn, unused);
          }
          {
            Frag vx, vn;
            if constexpr (GU) {
              fp4_gu_of_codes(aH[ks], aR[ks], vx, vn);
            } else {
              fp4_of_codes(aH[ks], aR[ks], vx);
              fp4_nh_of_codes(aH[ks], aR[ks], vx, vn, unused);
            }
            acc[0][0] = mfma_pair<GU>(vx, jx, acc[0][0]);
            acc[0][1] = mfma_pair<GU>(vn, jn, acc[0][1]);
            acc[0][2] = mfma_pair<GU>(vn, jx, acc[0][2]);
            acc[0][3] = mfma_pair<GU>(vx, jn, acc[0][3]);
          }
          {
            Frag vx, vn;
            if constexpr (GU) {
              fp4_gu_of_codes(bH[ks], bR[ks], vx, vn);
            } else {
              fp4_of_codes(bH[ks], bR[ks], vx);
              fp4_nh_of_codes(bH[ks], bR[ks], vx, vn, unused);
            }
            acc[1][0] = mfma_pair<GU>(vx, jx, acc[1][0]);
            acc[1][1] = mfma_pair<GU>(vn, jn, acc[1][1]);
            acc[1][2] = mfma_pair<GU>(vn, jx, acc[1][2]);
            acc[1][3] = mfma_pair<GU>(vx, jn, acc[1][3]);
          }
        }
      }
    }
    if (kc >= n_stages) {
      break;
    }
    // ---- checkpoint (block-uniform; the ring is empty and the LDS is scratch) ----
    __syncthreads();
    if (tid == 0) {
      s_live_waves = 0;
      s_need = 0;
    }
    {
      // row-block slot T by wave T % kWaves: lanes 0..31 = its rows' remainder slots, lanes 32..63 = their whole-row slots
      const uint8_t* cpg = reinterpret_cast<const uint8_t*>(A.cp_stats);
      const uint32_t gen_slot = kCpSlots + 1 + next_cp;
#pragma unroll
      for (uint32_t t = 0; t < (C::kBlocks + C::kWaves - 1) / C::kWaves; ++t) {
        const uint32_t T = wave + C::kWaves * t;
        if (T < C::kBlocks) {
          uint32_t first = static_cast<uint32_t>(slot_first(T));
          first = (first < A.n_local) ? first : (A.n_local - 1);
          first = __builtin_amdgcn_readfirstlane(first);
          uint32_t var = first + r;
          var = (var < A.n_local) ? var : (A.n_local - 1);
          const uint64_t off = (static_cast<uint64_t>(var) * kCpStride + (h ? static_cast<uint32_t>(kCpSlots) : gen_slot)) * sizeof(cp_gen_slot);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cpg + off),
                                           (__attribute__((address_space(3))) void*)(lds + C::kCpStatDwords + T * 256), 16, 0, 0);
        }
      }
    }
    __syncthreads();  // (drains the DMA; s_live_waves is zero)
    if (live) {
      const cp_gen_slot* __restrict__ cpl = reinterpret_cast<const cp_gen_slot*>(lds + C::kCpStatDwords);
      const uint64_t seen = static_cast<uint64_t>(kc) * G::kStageSamples;
      const double seen_d = static_cast<double>((seen < A.founder_ct) ? seen : A.founder_ct);
      const double rs = (seen < A.founder_ct) ? static_cast<double>(A.founder_ct - seen) : 1.0;
      const GenRow gj = gen_row(cpl[ja * 64 + r], cpl[ja * 64 + 32 + r], seen_d);
      const int32_t uj_p = static_cast<int32_t>(seen_d) - static_cast<int32_t>(cpl[ja * 64 + 32 + r].nm_r - cpl[ja * 64 + r].nm_r);
      const int32_t zj_p = static_cast<int32_t>(cpl[ja * 64 + 32 + r].zs_r - cpl[ja * 64 + r].zs_r);
      uint32_t* mine_epi = lds + wave * C::kCpWaveDwords;
      GenRow* rows_i = reinterpret_cast<GenRow*>(mine_epi + 4 * kMfGenCpRound * 64);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (!(live & (1u << p))) {
          continue;
        }
        if (h == 0) {
          rows_i[r] = gen_row(cpl[(JB + vb0 + p) * 64 + r], cpl[(JB + vb0 + p) * 64 + 32 + r], seen_d);
        }
        bool hopeless = true;
#pragma unroll
        for (int round = 0; round < 16 / static_cast<int>(kMfGenCpRound); ++round) {
          if (__all(hopeless)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
              for (int g = 0; g < static_cast<int>(kMfGenCpRound); ++g) {
                mine_epi[(c * kMfGenCpRound + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[p][c][round * kMfGenCpRound + g]));
              }
            }
#pragma unroll 1
            for (uint32_t gg = 0; gg < kMfGenCpRound; ++gg) {
              const uint32_t g = round * kMfGenCpRound + gg;
              const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * h;
              const int64_t i64 = static_cast<int64_t>(vfirst0) + kMfBlock * p + row;
              if ((lo_j != 0xffffffffu) && (i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64)) {
                const GenRow gi = rows_i[row];
                auto raw = [&](uint32_t c) { return static_cast<int32_t>(mine_epi[(c * kMfGenCpRound + gg) * 64 + lane]); };
                int32_t dot_p = raw(0), nm_p = raw(1), s2_p = raw(2), s1_p = raw(3);
                if constexpr (GU) {
                  const cp_gen_slot ci = cpl[(JB + vb0 + p) * 64 + row], wi = cpl[(JB + vb0 + p) * 64 + 32 + row];
                  const int32_t n_vis = static_cast<int32_t>(seen_d);
                  x_from_gu(raw(0), raw(1), raw(2), raw(3), n_vis - static_cast<int32_t>(wi.nm_r - ci.nm_r), static_cast<int32_t>(wi.zs_r - ci.zs_r), uj_p, zj_p, n_vis, 0, &dot_p,
                            &nm_p, &s2_p, &s1_p);
                }
                hopeless = hopeless && pair_hopeless_centred<false>(A.thresh, static_cast<double>(nm_p), static_cast<double>(s1_p), 0.0, static_cast<double>(s2_p), 0.0,
                                                                    static_cast<double>(dot_p), gi, gj, rs);
              }
            }
          }
        }
        if (__all(hopeless)) {
          live &= ~(1u << p);
          stop_stage[p] = kc;
        }
      }
      if (live && (lane == 0)) {
        atomicAdd(&s_live_waves, 1u);
        atomicOr(&s_need, (1u << ja) | (live << (JB + vb0)));
      }
    }
    __syncthreads();
    if (!s_live_waves) {
      break;
    }
    need = __builtin_amdgcn_readfirstlane(s_need);
    mine = count_mine();
    ++next_cp;
    issued_base = kc;
    issue_limit = (next_cp < n_cp) ? checkpoint_stage(kGenCheckpointFirst + next_cp) : n_stages;
    ring_fill();
  }
  if (lane == 0) {
    unsigned long long skipped = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if ((mine_products >> p) & 1u) {
        skipped += static_cast<unsigned long long>(n_stages - stop_stage[p]) * 4ull;
      }
    }
    if (skipped) {
      atomicAdd(A.counters + 2, skipped);
    }
  }
  // ---- epilogue: intervals for the two sums of squares, exact recount where they leave the predicate open ----
  __syncthreads();  // (every wave is past its last stage: the ring is scratch)
  uint32_t n_true = 0, n_open = 0;
  uint32_t* epi4 = lds + wave * (4 * 8 * 64);
  // (what the epilogue needs per lane is derived again here rather than kept in registers across the stage loops, where hipcc
  // parks such values in scratch and reloads them on every trip)
  {
  uint32_t tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const uint32_t lane = tid_e & 63, r = lane & 31, h = lane >> 5;
  const int64_t j64 = static_cast<int64_t>(jfirst) + r;
  uint32_t lo_j = 0xffffffffu;
  if ((j64 >= 0) && (j64 < static_cast<int64_t>(jend))) {
    lo_j = A.lo[static_cast<uint32_t>(j64)];
  }
  const bool j_ok = (lo_j != 0xffffffffu) && (static_cast<int64_t>(lo_j) < j64);
  const uint32_t j = j_ok ? static_cast<uint32_t>(j64) : 0u;
  ldp_variant_rec rj;
  rj.nm_ct = 0;
  rj.sum = 0;
  rj.ssq = 0;
  rj.flags = 0;
  if (j_ok && live) {
    rj = A.recs[j];
  }
  const bool alt_j = img_differs(rj.flags) != 0;
  const int64_t N = static_cast<int64_t>(A.founder_ct);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!(live & (1u << p))) {  // (wave-uniform)
      continue;
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) {
          epi4[(c * 8 + g8) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[p][c][round * 8 + g8]));
        }
      }
#pragma unroll 1
      for (uint32_t g8 = 0; g8 < 8; ++g8) {
        const uint32_t g = round * 8 + g8;
        const int64_t i64 = static_cast<int64_t>(vfirst0) + kMfBlock * p + (g & 3) + 8 * (g >> 2) + 4 * h;
        const bool valid = j_ok && (i64 >= static_cast<int64_t>(lo_j)) && (i64 < j64);
        const uint32_t i = valid ? static_cast<uint32_t>(i64) : 0u;
        int32_t dot_major = 0;
        uint32_t alt_ij = 0;
        int cls = 0;
        if (valid) {
          const ldp_variant_rec ri = A.recs[i];
          const bool alt_i = img_differs(ri.flags) != 0;
          alt_ij = (alt_i ? 1u : 0u) | (alt_j ? 2u : 0u);
          ldp_pair_stats_t ps;
          int32_t d = static_cast<int32_t>(epi4[(0 * 8 + g8) * 64 + lane]), s2 = static_cast<int32_t>(epi4[(2 * 8 + g8) * 64 + lane]),
                  s1 = static_cast<int32_t>(epi4[(3 * 8 + g8) * 64 + lane]), nm = static_cast<int32_t>(epi4[(1 * 8 + g8) * 64 + lane]);
          if constexpr (GU) {
            int32_t Ui, Zi, Uj, Zj;
            uz_of_rec(ri, A.founder_ct, &Ui, &Zi);
            uz_of_rec(rj, A.founder_ct, &Uj, &Zj);
            x_from_gu(d, nm, s2, s1, Ui, Zi, Uj, Zj, static_cast<int32_t>(A.founder_ct), static_cast<int32_t>(n_stages * G::kStageSamples - A.founder_ct), &d, &nm, &s2, &s1);
          }
          ps.nm = static_cast<uint32_t>(nm);
          ps.dot = (alt_i != alt_j) ? -d : d;
          ps.sum2 = alt_j ? -s2 : s2;
          ps.sum1 = alt_i ? -s1 : s1;
          dot_major = ps.dot;
          cls = classify_four(ps, ri, rj, N, A.thresh);
          if (cls == 1) {
            atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
            ++n_true;
          }
        }
        unsigned long long open = __ballot(cls == 2);
        while (open) {
          const int l = __builtin_ctzll(open);
          open &= open - 1;
          const uint32_t ii = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i), l));
          const uint32_t jj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(j), l));
          const int32_t dd = __builtin_amdgcn_readlane(dot_major, l);
          const uint32_t aa = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(alt_ij), l));
          const ldp_pair_stats_t st = wave_pair_counts(A, ii, jj, dd, lane, (aa & 1u) != 0, (aa & 2u) != 0);
          if ((static_cast<int>(lane) == l) && exceeds(st, A.thresh)) {
            atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
            ++n_true;
          }
          n_open += (lane == 0) ? 1u : 0u;
        }
      }
    }
  }
  if ((lane == 0) && n_open) {
    atomicAdd(A.counters + 3, static_cast<unsigned long long>(n_open));
  }
  n_true = wave_reduce_add(n_true);
  if ((lane == 0) && n_true) {
    atomicAdd(A.counters, static_cast<unsigned long long>(n_true));
  }
  }
}


hipError_t launch_pair_mfma(const PairKernelArgs& a_in, hipStream_t stream, hipEvent_t* ev) {
  if (!a_in.n_mf_wgs) {
    return hipSuccess;
  }
  PairKernelArgs a = a_in;
  // 64 KiB (+ 8 KiB static) lets two workgroups share a CU; LDP_DEBUG_MFMA_LDS_KB trades that for a deeper ring (tuning aid)
  static const size_t lds = []() {
    size_t bytes = static_cast<size_t>(kMfLdsDwords) * sizeof(uint32_t);
    if (const char* kb = LDP_ENV("LDP_DEBUG_MFMA_LDS_KB")) {
      bytes = std::max<size_t>(bytes, std::min<size_t>(static_cast<size_t>(atoi(kb)), 150) * 1024);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_kernel<4, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_kernel<4, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_kernel<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_kernel<4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    return bytes;
  }();
  a.lds_dwords = static_cast<uint32_t>(lds / sizeof(uint32_t));
  if (ev) {
    (void)hipEventRecord(ev[0], stream);
  }
  // the workgroups of the launch in two runs: [0, mf_diag_ct) all-diagonal ones (single-form kernel), the rest (general forms)
  const uint32_t n_diag = a_in.mf_diag_ct, n_rest = a_in.n_mf_wgs - a_in.mf_diag_ct;
  PairKernelArgs ad = a, ar = a;
  ad.n_mf_wgs = n_diag;
  ar.mf_wgs = a.mf_wgs + n_diag;
  ar.n_mf_wgs = n_rest;
  if (n_diag) {
#ifdef LDP_MEASURE
    static const bool fuse = []() {
      const char* v = LDP_ENV("LDP_DEBUG_DIAG_FUSE");
      const bool on = v && (atoi(v) != 0);
      if (on) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_kernel<4, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      }
      return on;
    }();
    if (fuse) {
      hipLaunchKernelGGL((pair_mfma_kernel<4, false, true, 1>), dim3(((n_diag + 7) / 8) * 8), dim3(kMfWaves * 64), lds, stream, ad);
    } else
#endif
    hipLaunchKernelGGL((pair_mfma_kernel<4, false, true>), dim3(((n_diag + 7) / 8) * 8), dim3(kMfWaves * 64), lds, stream, ad);
  }
  if (n_rest) {
    hipLaunchKernelGGL((pair_mfma_kernel<4, false, false>), dim3(((n_rest + 7) / 8) * 8), dim3(kMfWaves * 64), lds, stream, ar);
  }
  if (a.wd_active) {
    const hipError_t wrc = launch_pair_wide(a_in, stream);  // the wide-band subcontigs of the same launch (complete data)
    if (wrc != hipSuccess) {
      return wrc;
    }
  }
  if (a.sparse_ok) {
    if (n_diag) {
      hipLaunchKernelGGL((pair_mfma_kernel<4, true, true>), dim3(((n_diag + 7) / 8) * 8), dim3(kMfWaves * 64), lds, stream, ad);
    }
    if (n_rest) {
      hipLaunchKernelGGL((pair_mfma_kernel<4, true, false>), dim3(((n_rest + 7) / 8) * 8), dim3(kMfWaves * 64), lds, stream, ar);
    }
    if (a.wd_active && a.wd_sparse) {
      const hipError_t wrc = launch_pair_wide(a_in, stream, true);  // ... and the wide-band subcontigs' tiles on that route
      if (wrc != hipSuccess) {
        return wrc;
      }
    }
  }
  if (ev) {
    (void)hipEventRecord(ev[1], stream);
  }
  // the same plan for launches whose rows have missing calls: one workgroup per (wave item, second-variant block)
  if (a_in.mf_active == 2) {  // (1: rows with missing calls stay on the popcount kernel, EngineOptions::mfma_general)
    static const size_t glds = []() {
      // 5 row-blocks = 10 KiB per stage: a ring of five in 50 KiB (+ 3 KiB static) lets THREE workgroups share a CU (138 VGPRs
      // allow three waves per SIMD); LDP_DEBUG_MFMA_GEN_LDS_KB overrides (tuning aid)
      size_t bytes = 50 * 1024;
      if (const char* kb = LDP_ENV("LDP_DEBUG_MFMA_GEN_LDS_KB")) {
        bytes = std::min<size_t>(std::max<size_t>(static_cast<size_t>(atoi(kb)), 20), 150) * 1024;
      }
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_general_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_general_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_general_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
      return bytes;
    }();
    PairKernelArgs g = a_in;
    g.lds_dwords = static_cast<uint32_t>(glds / sizeof(uint32_t));
    // the four-product forms' operands: allele counts and missing flags while 9 N stays exact in f32 (ldp_mfma_device.h), x and n otherwise
    const bool gu = (a_in.mf_gu != 0) && (a_in.founder_ct <= kMfGuMaxFounders);
    // prune launches over subcontigs with the tile plan: quarter tiles (four products); the parallelogram workgroups of those
    // subcontigs then stay out (wd_general)
    if (a_in.wd_general) {
      // quarter tiles: JB = 4 (half tiles of J, two workgroups of four waves per CU, measured the same: profiles/r03_experiments.md)
      static const size_t t4lds = []() {
        const size_t bytes = static_cast<size_t>(T4<4>::kLdsDwords) * sizeof(uint32_t);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_tile4_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_tile4_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        return bytes;
      }();
      PairKernelArgs t4 = a_in;
      t4.lds_dwords = T4<4>::kLdsDwords;
      const uint32_t per_xcd = ((a_in.wd_tiles_plain ? a_in.n_wd_tiles_plain : a_in.n_wd_tiles) * T4<4>::kUnits + 7) / 8;
      if (gu) {
        hipLaunchKernelGGL((pair_mfma_tile4_kernel<4, true>), dim3(per_xcd * 8), dim3(T4<4>::kWaves * 64), t4lds, stream, t4);
      } else {
        hipLaunchKernelGGL((pair_mfma_tile4_kernel<4, false>), dim3(per_xcd * 8), dim3(T4<4>::kWaves * 64), t4lds, stream, t4);
      }
    }
    const uint32_t gper_xcd = (g.n_mf_wgs * 8 + 7) / 8;
    // only the predicate is wanted (no integers, no r^2 values): the four-product form
    if (a_in.mf_four && !a_in.stats && !a_in.r2_out && !a_in.r2_hits) {
      if (gu) {
        hipLaunchKernelGGL((pair_mfma_general_kernel<false, true>), dim3(gper_xcd * 8), dim3(kMfWaves * 64), glds, stream, g);
      } else {
        hipLaunchKernelGGL((pair_mfma_general_kernel<false, false>), dim3(gper_xcd * 8), dim3(kMfWaves * 64), glds, stream, g);
      }
    } else {
      hipLaunchKernelGGL((pair_mfma_general_kernel<true, false>), dim3(gper_xcd * 8), dim3(kMfWaves * 64), glds, stream, g);
    }
  }
  if (ev) {
    (void)hipEventRecord(ev[2], stream);
  }
  return hipGetLastError();
}

// 64-sample k-steps per row as the kernel counts them (counters[2] is in product x k-step units)
uint32_t pair_mfma_ksteps(uint32_t founder_ct) { return ((founder_ct + 511) / 512) * 8; }  // (whole 512-sample k-chunks: the wide-band kernel's stages; the others stop at most four k-steps of padding earlier)

}  // namespace ldp
