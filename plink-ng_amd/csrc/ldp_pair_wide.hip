// ldp_pair_wide.hip -- the complete-data pair statistics of WIDE bands on the matrix pipe: 8 x 8 block tiles, eight waves.
//
// Same arithmetic as pair_mfma_kernel (ldp_pair_mfma.hip: DotprodWords, plink2_ld.cc:235-251, as FP4 matrix products over the
// samples, expanded from the resident 2-bit codes), a different work decomposition.  At BASELINE config 3's density a window
// holds ~1,700 variants = 54 row-blocks, and the parallelogram plan's 32 block products per 15 staged row-blocks made the
// kernel wait for HBM: every row was fetched ~12 times (profiles/r02_c3shape_pmc_traffic.json).  Here a workgroup owns a
// SQUARE of 8 second-variant blocks x 8 first-variant blocks: 64 products for 16 staged row-blocks (32 KiB per 256-sample
// stage), i.e. half the bytes per product, and one workgroup per CU with a two-stage ring of 512-sample stages in 128 KiB of LDS.  Wave w owns
// the 2 x 4 sub-rectangle J blocks 2 (w & 3), + 1 x V blocks 4 ((w ^ (w >> 2)) & 1) .. + 3 (wide_tile: the map, and how it was chosen): eight accumulator sets, six row-block reads
// and expansions per stage (the parallelogram needs seven).  Tiles are aligned to the subcontig start in both directions;
// on the diagonal the V tile is the J tile (8 row-blocks staged) and the products above it are simply not live.
// The launch is eight streams of tiles of equal length, one per XCD (ldp_engine.cpp build_shard): the far tiles of a stream first,
// J tile by J tile, its tiles next to the diagonal (the long ones) at the end; workgroup b runs tile (b & 7) * per_xcd + (b >> 3), so
// the tiles that run together on an XCD are neighbours and share row-blocks through that XCD's L2.
// Early termination as in pair_mfma_kernel: at a checkpoint a wave drops the products that provably hold no pair above the
// threshold, row-blocks nobody reads any more are no longer fetched, a workgroup with nothing left leaves.
#include "ldp_device.h"
#include "ldp_pair_device.h"
#include "ldp_mfma_device.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace ldp {

namespace {

// ---- geometry: a stage is 512 samples = 128 contiguous bytes of a row (one cache line), two half-stages of four k-steps ----
// LDS image of a stage: row-block slot b, row r, eight 16-byte pieces per row (piece c = bytes 16 c .. of the row's stage) at
// unit (32 b + r) * 8 + (c ^ ((r >> 1) & 7)); in half-stage hs lane half h reads pieces 4 hs + h and 4 hs + 2 + h: its k-step
// ks is dword ks of each.  The XOR spreads the 16 lanes of every ds_read_b128 group over the 16 four-bank groups (rows two apart
// share banks at this row pitch), and the DMA (lane-linear in LDS, free per-lane global address) fetches the piece that belongs
// in its unit.  Twice the samples per barrier of pair_mfma_kernel's stages: with ONE workgroup per CU nothing else fills the
// bubble behind a barrier (all eight waves read and expand before the first MFMA), so there should be few of them.
constexpr uint32_t kWdStageSamples = 512;
constexpr uint32_t kWdRowStageBytes = kWdStageSamples / 4;             // 128
constexpr uint32_t kWdPieces = kWdRowStageBytes / 16;                  // 8
constexpr uint32_t kWdBlockUnits = kMfBlock * kWdPieces;               // 16-byte units per row-block and stage: 256
constexpr uint32_t kWdStageDwords = kWdRowBlocks * kWdBlockUnits * 4;  // 16,384 dwords = 64 KiB
constexpr uint32_t kWdMaxStages = 2;
constexpr uint32_t kWdInstrPerBlock = kWdBlockUnits / 64;              // DMA wave-instructions per row-block and stage: 4
constexpr uint32_t kWdDma = (kWdRowBlocks * kWdInstrPerBlock) / kWdWaves;  // per wave and stage: 8
constexpr uint32_t kWdKsteps = kWdStageSamples / 64;                   // MFMAs per product and stage: 8
constexpr uint32_t kWdEpiWaveDwords = 4 * 16 * 64;                    // four products per epilogue round: 16 KiB per wave
constexpr uint32_t kWdLdsDwords = kWdWaves * kWdEpiWaveDwords;        // 128 KiB: epilogue scratch == two-stage ring
constexpr uint32_t kWdCpWaveDwords = 2 * 16 * 64;                     // checkpoint: two products per wave and round
constexpr uint32_t kWdCpScratchDwords = kWdWaves * kWdCpWaveDwords;   // 64 KiB in; 16 row-blocks x 32 rows x 32 B = 16 KiB follow
constexpr uint32_t kWdCpSlotDwords = kWdRowBlocks * kMfBlock * 8;     // two cp_slots per staged row: 16 KiB
constexpr uint32_t kWdCpGenDwords = kWdRowBlocks * kMfBlock * 4;      // SPARSE: + the whole-row cp_gen_slot of every staged row (8 KiB) ...
constexpr uint32_t kWdCpRowDwords = kWdRowBlocks * kMfBlock * 8;      // ... and the SparseRow made of it (16 KiB)
static_assert(kWdStageDwords * kWdMaxStages <= kWdLdsDwords, "ring fits the epilogue scratch");
static_assert(kWdCpScratchDwords + kWdCpSlotDwords + kWdCpGenDwords + kWdCpRowDwords <= kWdLdsDwords, "checkpoint scratch fits");
static_assert(sizeof(SparseRow) == 32 && sizeof(cp_gen_slot) == 16 && sizeof(cp_slot) == 16, "LDS layouts of the checkpoint");
__device__ __forceinline__ uint32_t wd_swizzle(uint32_t row) { return (row >> 1) & 7u; }

// One HALF-stage (256 samples) of a wave's 2 x 4 rectangle: J fragments of all four k-steps in registers, the four V blocks streamed past them
// (two b128 LDS reads -> 4 fragments -> 8 MFMAs, the next block's reads in flight).  There is ONE form of this loop body, without
// a test or a branch: a wave computes all eight of its products as long as one of them is live and stops altogether once none is.
// (A masked form -- one branch per product or per V block -- costs register copies of every accumulator at each branch; with
// eight waves' worth of state hipcc spilled inside the stage loop for it, and a scratch reload's vmcnt wait drains the DMA
// ring.  A wave's eight products sit next to each other in distance, so they mostly die together anyway; the products of a
// partly live wave that hold no candidate pair accumulate numbers nobody reads.)
// ABL != 0 exists in the MEASUREMENT build only (-DLDP_MEASURE, csrc/ldp_env.h; the shipped library instantiates <0> and nothing
// else): ablations for the attribution tables of profiles/r04_experiments.md and r05_experiments.md.  Bits 0-3 make the results
// WRONG by construction.  Bit 0 (value 1) = no DMA after the ring's first fill, bit 1 (2) = expand only the first k-step's operands
// and reuse them (a quarter of the VALU work, the same operand statistics), bit 2 (4) = no LDS reads in the loop (every lane
// multiplies what the first stage left in its registers' place), bit 3 (8) = no s_waitcnt vmcnt / s_barrier in the stage loop (the
// waves run free), bit 4 (16) = no per-pair epilogue, bit 5 (32) = time stamps around the phases of every wave (results right:
// where the cycles go), summed into g_wide_measure, bit 6 (64) = every tile stages the same 512 rows (wrong results; all of the DMA's
// requests hit the L2: what the HBM leg of the traffic costs).
// GC: the allele-count coding of complete rows (ldp_mfma_device.h); false: the +-2 coding, where a missing call is 0 and the
// accumulators hold the exact dot product (the SPARSE instantiation below)
// VC: V blocks of the wave's rectangle -- 4 (the tiles' 2 x 4 rectangles) or 3 (the diagonal tiles' kernel: 2 x 3, see pair_mfma_wide_kernel)
template <int ABL, bool GC = true, int VC = 4>
__device__ __forceinline__ void wide_stage(const mf_u4* __restrict__ st4, const uint32_t (&joff)[2], const uint32_t (&voff)[VC], uint32_t oH, uint32_t oR,
                                           mf_v16f (&acc)[2 * VC]) {
  if constexpr ((ABL & 4) != 0) {
    // (handled by wide_stage_kept below)
    return;
  }
  mf_u4 vH[2], vR[2];
  vH[0] = st4[voff[0] + oH];
  vR[0] = st4[voff[0] + oR];
  Frag fj0[4], fj1[4];
  {
    mf_u4 H = st4[joff[0] + oH], R = st4[joff[0] + oR];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if ((ABL & 2) && ks) {
        fj0[ks] = fj0[0];
      } else {
        fp4_expand<GC>(H[ks], R[ks], fj0[ks]);
      }
    }
  }
  {
    mf_u4 H = st4[joff[1] + oH], R = st4[joff[1] + oR];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if ((ABL & 2) && ks) {
        fj1[ks] = fj1[0];
      } else {
        fp4_expand<GC>(H[ks], R[ks], fj1[ks]);
      }
    }
  }
  // rows of C = first variant (A operand: a V block), columns = second variant (B operand: a J block)
  // b: V block of the wave (0..3), B: its raw buffer; products b (with J0) and 4 + b (with J1)
#define LDP_WD_VBLOCK(b, B)                                    \
  if ((b) < VC - 1) {                                          \
    vH[(B) ^ 1] = st4[voff[((b) < VC - 1) ? (b) + 1 : VC - 1] + oH]; \
    vR[(B) ^ 1] = st4[voff[((b) < VC - 1) ? (b) + 1 : VC - 1] + oR]; \
  }                                                            \
  opaque(vH[B], vR[B]);                                        \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {           \
    Frag fv;                                                   \
    if ((ABL & 2) && ks) {                                     \
      fv = fj1[(ks + (b)) & 3];                                \
    } else {                                                   \
      fp4_expand<GC>(vH[B][ks], vR[B][ks], fv);              \
    }                                                          \
    acc[b] = mfma_pair<GC>(fv, fj0[ks], acc[b]);                    \
    acc[VC + (b)] = mfma_pair<GC>(fv, fj1[ks], acc[VC + (b)]);      \
  }
  LDP_WD_VBLOCK(0, 0)
  LDP_WD_VBLOCK(1, 1)
  LDP_WD_VBLOCK(2, 0)
  if constexpr (VC == 4) {
    LDP_WD_VBLOCK(3, 1)
  }
#undef LDP_WD_VBLOCK
}

// The same half-stage with the NEXT half-stage's J fragments made beside it (round 6, tools/tile_shape_probe.hip shape "2x4pf": the stage loop alone
// 5.74-5.78 -> 5.98 PFLOP/s).  Behind a stage's barrier the two waves of a SIMD are in the same phase: both read and expand their two J blocks (48
// VALU behind two LDS round trips) before either has an MFMA to issue -- twice per stage.  Here fj0 / fj1 arrive ready; with PREF the pieces of the J
// blocks at (oHn, oRn) -- the stage's second half -- are read with V blocks 1 / 2 and expanded between the MFMAs of V blocks 0 / 1 into fn0 / fn1, so
// the second half-stage starts on the matrix pipe at once and only the first one of a stage has the head (the stage has just landed: nothing of
// it could have been read earlier).  32 more registers.
template <bool GC, int VC, bool PREF>
__device__ __forceinline__ void wide_stage_pf(const mf_u4* __restrict__ st4, const uint32_t (&joff)[2], const uint32_t (&voff)[VC], uint32_t oH, uint32_t oR, uint32_t oHn,
                                              uint32_t oRn, const Frag (&fj0)[4], const Frag (&fj1)[4], Frag (&fn0)[4], Frag (&fn1)[4], mf_v16f (&acc)[2 * VC]) {
  mf_u4 vH[2], vR[2], jH[2], jR[2];
  vH[0] = st4[voff[0] + oH];
  vR[0] = st4[voff[0] + oR];
  if constexpr (PREF) {
    jH[0] = st4[joff[0] + oHn];
    jR[0] = st4[joff[0] + oRn];
  }
#pragma unroll
  for (int b = 0; b < VC; ++b) {
    __builtin_amdgcn_sched_barrier(0);  // (one V block's reads, expansions and MFMAs at a time: hipcc otherwise hoists every block's reads to the front)
    if (b + 1 < VC) {
      vH[(b & 1) ^ 1] = st4[voff[(b + 1 < VC) ? b + 1 : b] + oH];
      vR[(b & 1) ^ 1] = st4[voff[(b + 1 < VC) ? b + 1 : b] + oR];
    }
    if constexpr (PREF) {
      if (b == 0) {
        jH[1] = st4[joff[1] + oHn];
        jR[1] = st4[joff[1] + oRn];
      }
      if (b < 2) {
        opaque(jH[b & 1], jR[b & 1]);
      }
    }
    opaque(vH[b & 1], vR[b & 1]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fv;
      fp4_expand<GC>(vH[b & 1][ks], vR[b & 1][ks], fv);
      if constexpr (PREF) {
        if (b == 0) {
          fp4_expand<GC>(jH[0][ks], jR[0][ks], fn0[ks]);
        } else if (b == 1) {
          fp4_expand<GC>(jH[1][ks], jR[1][ks], fn1[ks]);
        }
      }
      acc[b] = mfma_pair<GC>(fv, fj0[ks], acc[b]);
      acc[VC + b] = mfma_pair<GC>(fv, fj1[ks], acc[VC + b]);
    }
  }
}
// a stage (512 samples) of a wave's rectangle in that form
template <bool GC, int VC>
__device__ __forceinline__ void wide_stage_pair(const mf_u4* __restrict__ st4, const uint32_t (&joff)[2], const uint32_t (&voff)[VC], uint32_t oH0, uint32_t oR0, uint32_t oH1,
                                                uint32_t oR1, mf_v16f (&acc)[2 * VC]) {
  Frag fa0[4], fa1[4], fb0[4], fb1[4];
  {
    mf_u4 H = st4[joff[0] + oH0], R = st4[joff[0] + oR0];
    mf_u4 H1 = st4[joff[1] + oH0], R1 = st4[joff[1] + oR0];
    opaque(H, R);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fp4_expand<GC>(H[ks], R[ks], fa0[ks]);
    }
    opaque(H1, R1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fp4_expand<GC>(H1[ks], R1[ks], fa1[ks]);
    }
  }
  wide_stage_pf<GC, VC, true>(st4, joff, voff, oH0, oR0, oH1, oR1, fa0, fa1, fb0, fb1, acc);
  wide_stage_pf<GC, VC, false>(st4, joff, voff, oH1, oR1, oH1, oR1, fb0, fb1, fa0, fa1, acc);
}

// ABL bit 2: the stage without LDS reads -- every row-block's codes are the two pieces this lane read ONCE (real genotypes of the first
// stage, the dwords rotated per block and k-step so that consecutive MFMAs still see different operands)
template <int ABL>
__device__ __forceinline__ void wide_stage_kept(mf_u4 H, mf_u4 R, mf_v16f (&acc)[8]) {
  opaque(H, R);
  Frag fj0[4], fj1[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if ((ABL & 2) && ks) {
      fj0[ks] = fj0[0];
      fj1[ks] = fj1[0];
    } else {
      fp4_expand<true>(H[ks], R[ks], fj0[ks]);
      fp4_expand<true>(R[ks], H[(ks + 1) & 3], fj1[ks]);
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    mf_u4 vH = {H[b & 3], H[(b + 1) & 3], H[(b + 2) & 3], H[(b + 3) & 3]}, vR = {R[(b + 2) & 3], R[(b + 3) & 3], R[b & 3], R[(b + 1) & 3]};
    opaque(vH, vR);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      Frag fv;
      if ((ABL & 2) && ks) {
        fv = fj1[(ks + b) & 3];
      } else {
        fp4_expand<true>(vH[ks], vR[ks], fv);
      }
      acc[b] = mfma_pair<true>(fv, fj0[ks], acc[b]);
      acc[4 + b] = mfma_pair<true>(fv, fj1[ks], acc[4 + b]);
    }
  }
}

#ifdef LDP_MEASURE
// what the waves of the measured launches spent where (shader cycles, summed over waves; ABL bit 5 fills 0-5 and 9-10, every measured
// instantiation 6-8): [0] s_waitcnt vmcnt in front of the stage barrier, [1] the stage barrier itself, [2] stages of waves with a live
// product (DMA issue + LDS reads + expansions + MFMA issue), [3] stages of waves with none, [4] checkpoints, [5] epilogue, [6] entry to
// exit, [7] waves, [8] entry to exit in 100 MHz wall ticks (shader clock = 100 MHz x [6] / [8]), [9] / [10] stage visits live / dead
__device__ unsigned long long g_wide_measure[16];
__device__ __forceinline__ unsigned long long wd_clk() {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

// row-block slots (bit s: slot s of the stage) a wave with a live product reads: its two J blocks and its four V blocks
template <int VC = 4>
__device__ __forceinline__ uint32_t wide_slots_needed(uint32_t live, uint32_t a0, uint32_t vslot0) {
  // (VC == 3: a diagonal tile, eight staged row-blocks -- the last wave's third V block does not exist and is never live)
  return live ? (((3u << a0) | (((1u << VC) - 1u) << vslot0)) & ((VC == 3) ? 0xffu : 0xffffu)) : 0u;  // (live: the rectangle lies inside the tile)
}

// The SPARSE instantiation's epilogue for one J block of a wave's rectangle: its (up to) four products are in this wave's LDS scratch,
// product pl = (the J block, V block pl), `jb` / `vb` = first variant of the J block / of V block 0, live4 = which of them hold
// candidate pairs that are still alive.  Per pair: sparse_decide on the exact dot product and the two records; the pairs it leaves
// open one after the other, each by the whole wave, from the two rows of the image (ldp_counters::sparse_exact_pairs counts them).
__device__ __forceinline__ uint32_t wide_sparse_round(const PairKernelArgs& A, const uint32_t* epi, uint32_t lane, int32_t jb, int32_t vb, uint32_t jend, uint32_t live4,
                                                      uint32_t lo_j) {
  const uint32_t r = lane & 31, h = lane >> 5;
  uint32_t n_true = 0, n_open = 0;
  const int64_t j64 = static_cast<int64_t>(jb) + r;
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
  const SparseRow Jr = sparse_row_of(rj);
  const double n_all = static_cast<double>(A.founder_ct);
#pragma unroll 1
  for (uint32_t pl = 0; pl < 4; ++pl) {
    if (!(live4 & (1u << pl))) {
      continue;  // (wave-uniform)
    }
    const int64_t vfirst = static_cast<int64_t>(vb) + kMfBlock * pl + 4 * h;
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
        const double d = static_cast<double>(dot);
        cls = sparse_decide(A.thresh, n_all, d, d, sparse_row_of(ri), Jr);
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

// SPARSE: the instantiation that owns the tiles of launches whose rows have a FEW missing calls (route kRouteSparse, DESIGN.md 4.1d;
// the reference's per-pair dispatch between DotprodWords / SumSsqWords / SumSsqNmWords, plink2_ld.cc:699-723, seen from the fast plan):
// the +-2 coding, so the one product per block is the exact `dot` with missing calls in it; checkpoints that bound the pair from the
// partial dot product AND what the two rows' missing calls can move (sparse_decide with the dot product's Cauchy-Schwarz interval); an
// epilogue that decides a pair from per-variant counts where the intervals allow and recounts the few pairs they leave open from the
// two rows (wave_pair_counts) -- no approximation reaches the output.  A kernel of its own (template parameter) because hipcc
// re-allocates the complete-data kernel's registers as soon as the interval code shares a function with it.
// VC == 3: the body for DIAGONAL tiles (PairKernelArgs::wd_diag_split; complete data, prune launches; round 6).  A diagonal tile holds 36 live
// products -- those on and below the diagonal -- and runs to the end of the rows (the pairs in LD are there): with 2 x 4 rectangles six waves
// compute 48 products at eight per wave while two have nothing to do, and the tile takes as long as a full one.  Here the same triangle is cut
// into eight 2 x 3 rectangles -- J blocks (0,1) x V 0-2, (2,3) x V 0-2 | 3-5, (4,5) x V 0-2 | 3-5, (6,7) x V 0-2 | 3-5 | 6-7 -- 48 products again,
// but six per wave on all eight waves: three quarters of the matrix-pipe time per stage.  Everything else is the code of the 2 x 4 body; the kernel
// picks one of the two per workgroup (block-uniform), so each body keeps its ONE form of the stage loop.  Measured (profiles/r06_experiments.md
// section 4): as a launch of its own behind the others the diagonal tiles lose the L2 sharing with their neighbours (the share 339 against 273 ms
// of pair kernels); inside the one launch 266.7 against 273.3 ms.  2 x 2 quads for tiles with at most eight live quads (the far tile of a J tile)
// were built too and bought nothing: those tiles are bound by their staging, not by the matrix pipe.
template <int ABL, bool SPARSE, int VC>
__device__ __forceinline__ void wide_tile(const PairKernelArgs& A, uint32_t* __restrict__ lds, uint32_t* __restrict__ s_need, const MfmaTile* __restrict__ tile) {
  static_assert((VC == 4) || ((VC == 3) && !SPARSE && (ABL == 0)), "2 x 4 rectangles, or the diagonal tiles' 2 x 3");
  constexpr uint32_t NP = 2 * VC;              // products per wave
  constexpr uint32_t kColMask = (1u << VC) - 1u;
#ifdef LDP_MEASURE
  unsigned long long m_t[6] = {0, 0, 0, 0, 0, 0}, m_visits[2] = {0, 0};
  const unsigned long long m_clk0 = wd_clk(), m_wall0 = __builtin_amdgcn_s_memrealtime();
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const uint32_t r = lane & 31;
  const uint32_t h = lane >> 5;
  const int32_t jv0 = __builtin_amdgcn_readfirstlane(tile->jv);
  const int32_t vv0 = __builtin_amdgcn_readfirstlane(tile->vv);
  const uint32_t jend = __builtin_amdgcn_readfirstlane(tile->jend);
  const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(tile->mask));
  const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(tile->mask >> 32));
  if (!(mask_lo | mask_hi)) {
    return;  // (padding of an XCD's stream: the launch's streams are of equal length, ldp_engine.cpp build_shard)
  }
  const bool diag = (jv0 == vv0);
  [[maybe_unused]] const int32_t g_bias = g_bias_of(A.founder_ct, kWdStageSamples);
  const uint32_t row_bytes = static_cast<uint32_t>(A.code_row_bytes);
  const uint32_t n_stages = (A.founder_ct + kWdStageSamples - 1) / kWdStageSamples;  // (the image's rows are whole stages long: ldp_device.h)
  const uint32_t stage_dwords = kWdStageDwords;
  uint32_t stages = A.lds_dwords / stage_dwords;
  stages = (stages > kWdMaxStages) ? kWdMaxStages : stages;

  // ---- this wave's rectangle: J blocks a0, a0 + 1, V blocks b0 .. b0 + VC - 1 (product b: J block a0, V block b0 + b; VC + b: J block a0 + 1) ----
  // VC == 3, wave 0 .. 7: (a0, b0) = (0,0) (2,0) (2,3) (4,0) (4,3) (6,0) (6,3) (6,6)
  // VC == 4, wave 0 .. 7: (a0, b0) = (0,0) (2,4) (4,0) (6,4) | (0,4) (2,0) (4,4) (6,0).  Waves w and w + 4 read the same two J blocks (as in rounds 2-5); since
  // round 6 neighbouring waves take different V halves.  Four maps, one library each, alternating on one box (profiles/r06_experiments.md 4b): this one
  // 272.5 ms of pair kernels per step of the share; b0 = 4 (w >> 2), the map of rounds 2-5, 275.0-275.8; waves w, w + 4 sharing their V blocks instead
  // 276.1-276.5; sharing nothing 277.2 against 271.8.
  auto rect_a0 = [](uint32_t w) -> uint32_t { return ((VC == 4) ? (0x64206420u >> (4 * w)) : (0x66644220u >> (4 * w))) & 0xfu; };
  auto rect_b0 = [](uint32_t w) -> uint32_t { return ((VC == 4) ? (0x04044040u >> (4 * w)) : (0x63030300u >> (4 * w))) & 0xfu; };
  const uint32_t a0 = rect_a0(wave), b0 = rect_b0(wave);
  const uint32_t vslot0 = (diag ? 0u : static_cast<uint32_t>(kWdTile)) + b0;
  auto mask_row = [&](uint32_t a) { return ((a < 4) ? (mask_lo >> (8 * a)) : (mask_hi >> (8 * (a - 4)))) & 0xffu; };
  uint32_t live = ((mask_row(a0) >> b0) & kColMask) | (((mask_row(a0 + 1) >> b0) & kColMask) << VC);
  live = __builtin_amdgcn_readfirstlane(live);
  // row-block slots the workgroup reads: the rectangles of the waves that own a live product
  uint32_t wg_need = 0;
#pragma unroll
  for (uint32_t w = 0; w < static_cast<uint32_t>(kWdWaves); ++w) {
    const uint32_t wa = rect_a0(w), wb = rect_b0(w);
    const uint32_t wl = ((mask_row(wa) | mask_row(wa + 1)) >> wb) & kColMask;
    wg_need |= wide_slots_needed<VC>(wl, wa, (diag ? 0u : static_cast<uint32_t>(kWdTile)) + wb);
  }
  wg_need = __builtin_amdgcn_readfirstlane(wg_need);
  auto slot_first = [&](uint32_t s) { return (s < static_cast<uint32_t>(kWdTile)) ? (jv0 + static_cast<int32_t>(kMfBlock * s)) : (vv0 + static_cast<int32_t>(kMfBlock * (s - kWdTile))); };

  // ---- DMA plan: per-lane source offsets (registers) and per-instruction row-block bases (uniform) ----
  // instruction T of a stage = a quarter of row-block slot T >> 2: eight rows x eight 16-byte pieces
  const uint8_t* base_t[kWdDma];
  uint32_t src_off[kWdDma];
#pragma unroll
  for (int t = 0; t < static_cast<int>(kWdDma); ++t) {
    const uint32_t T = wave + kWdWaves * t;
    const uint32_t slot = T / kWdInstrPerBlock;
    uint32_t first = static_cast<uint32_t>(slot_first(slot));
    if constexpr ((ABL & 64) != 0) {
      first = slot * kMfBlock;  // (measurement: every tile of the launch stages the SAME 512 rows -- the DMA's L2 -> LDS leg without its HBM leg)
    }
    first = (first < A.n_local) ? first : (A.n_local - 1);  // (a block beyond the rows is never live; keep its address legal anyway)
    first = __builtin_amdgcn_readfirstlane(first);
    base_t[t] = A.codes + static_cast<uint64_t>(first) * row_bytes;
    const uint32_t rr = (T % kWdInstrPerBlock) * 8 + (lane >> 3);
    const uint32_t col = (lane & 7) ^ wd_swizzle(rr);
    uint32_t var = first + rr;
    var = (var < A.n_local) ? var : (A.n_local - 1);
    src_off[t] = (var - first) * row_bytes + col * 16;
  }
  auto count_mine = [&]() {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kWdDma); ++t) {
      m += ((wg_need >> ((wave + kWdWaves * t) / kWdInstrPerBlock)) & 1u) ? 1u : 0u;
    }
    return m;
  };
  uint32_t mine = (ABL & 1) ? 0u : count_mine();  // DMA wave-instructions per stage this wave issues

  uint32_t joff[2], voff[VC];  // uint4 index of the row-block's first slot
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    joff[q] = (a0 + q) * kWdBlockUnits;
  }
#pragma unroll
  for (int b = 0; b < VC; ++b) {
    const uint32_t slot = vslot0 + b;
    voff[b] = (((VC == 3) && (slot > 7u)) ? 7u : slot) * kWdBlockUnits;  // (VC == 3, last wave: a block that does not exist is read as block 7 and never live)
  }
  uint32_t need = wide_slots_needed<VC>(live, a0, vslot0);
  // window starts of this lane's two second variants (J0 + r, J1 + r), fetched here: the k-loop must not hold ordinary
  // global loads (hipcc would drain the DMA ring in front of every LDS read of the loop)
  uint32_t lo_j2[2] = {0xffffffffu, 0xffffffffu};  // (lo >= j: no candidate pair)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t j = static_cast<uint32_t>(jv0) + kMfBlock * (a0 + q) + r;
    if (j < jend) {
      lo_j2[q] = A.lo[j];
    }
  }
  const uint32_t sw = wd_swizzle(r);
  // unit offsets of this lane's two pieces inside a row-block, for the two half-stages
  const uint32_t oH0 = r * kWdPieces + (h ^ sw), oR0 = r * kWdPieces + ((2 + h) ^ sw);
  const uint32_t oH1 = r * kWdPieces + ((4 + h) ^ sw), oR1 = r * kWdPieces + ((6 + h) ^ sw);

  mf_v16f acc[NP];
#pragma unroll
  for (int p = 0; p < static_cast<int>(NP); ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }

  mf_u4 keptH = {0, 0, 0, 0}, keptR = {0, 0, 0, 0};  // (ABL bit 2 only)
  uint32_t next_cp = 0;
  const uint32_t n_cp = A.cp_stats ? A.n_checkpoints : 0;
  const uint32_t live0 = live;        // the products of the plan
  uint32_t stop_stage = n_stages;     // stages this wave computes (it stops as a whole, at a checkpoint)
  auto dma_stage = [&](uint32_t s, uint32_t buf) {
    const uint32_t kbyte = s * kWdRowStageBytes;
    uint32_t* dst = lds + buf * stage_dwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kWdDma); ++t) {
      const uint32_t T = wave + kWdWaves * t;
      if (((wg_need >> (T / kWdInstrPerBlock)) & 1u) && (!(ABL & 1) || (s < kWdMaxStages))) {  // (ABL bit 0: only the ring's first fill leaves HBM: what the DMA costs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_t[t] + kbyte + src_off[t]),
                                         (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
      }
    }
  };
  auto checkpoint_stage = [&](uint32_t cp) {
    const uint32_t s = A.checkpoint_chunk[cp];  // (a stage is a 512-sample k-chunk here)
    return (s < n_stages) ? s : n_stages;
  };

  uint32_t* epi = lds + wave * kWdEpiWaveDwords;  // this wave's scratch whenever the ring is empty (epilogue)
  // ---- k-loop over stages, ring of `stages` LDS buffers; the ring never runs past the next checkpoint ----
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
  for (uint32_t kc = 0; kc < n_stages;) {
    const uint32_t kc_end = issue_limit;  // the next checkpoint (or the end of the rows)
    for (; kc < kc_end; ++kc) {
#ifdef LDP_MEASURE
      unsigned long long m_s0 = 0;
      if constexpr ((ABL & 32) != 0) {
        // (a two-stage ring: the stage about to be read is the only one in flight, the wait is always vmcnt(0))
        const unsigned long long a0 = wd_clk();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long a1 = wd_clk();
        asm volatile("s_barrier" ::: "memory");
        m_s0 = wd_clk();
        m_t[0] += a1 - a0;
        m_t[1] += m_s0 - a1;
      } else if constexpr ((ABL & 8) == 0) {
        wait_dma_then_barrier(mine * (issued - kc - 1));
      }
#else
      wait_dma_then_barrier(mine * (issued - kc - 1));
#endif
      if (issued < issue_limit) {
        dma_stage(issued, issue_buf);  // (reuses the buffer every wave finished reading before the barrier)
        ++issued;
        issue_buf = (issue_buf + 1 == stages) ? 0 : issue_buf + 1;
      }
      const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + read_buf * stage_dwords);
      read_buf = (read_buf + 1 == stages) ? 0 : read_buf + 1;
      if constexpr (((ABL & 4) != 0) && (VC == 4)) {
        if (kc == 0) {
          keptH = st4[joff[0] + oH0];
          keptR = st4[joff[0] + oR0];
        }
        if (live) {
          wide_stage_kept<ABL>(keptH, keptR, acc);
          wide_stage_kept<ABL>(keptR, keptH, acc);
        }
      } else if (live) {
        if constexpr (((ABL & 6) == 0) && !SPARSE) {  // (measurement build: every ablation that leaves the stage's reads and expansions alone times THIS form)
          // (the SPARSE instantiation keeps the two plain half-stages: measured with this form too -- 14 values parked in scratch, none inside a k-loop --
          // the slice at 0.1 % / 0.3 % missing calls 33.35 / 33.30 against 33.30 / 33.40 ms and 40.27 / 40.23 against 40.43 / 40.39 ms of pair kernels:
          // nothing; its waves wait for their +-2-coded operands' energy and for the interval checkpoints, not for the head of a half-stage)
          wide_stage_pair<true, VC>(st4, joff, voff, oH0, oR0, oH1, oR1, acc);
        } else {
          wide_stage<ABL, !SPARSE, VC>(st4, joff, voff, oH0, oR0, acc);
          wide_stage<ABL, !SPARSE, VC>(st4, joff, voff, oH1, oR1, acc);
        }
      }
#ifdef LDP_MEASURE
      if constexpr ((ABL & 32) != 0) {
        m_t[live ? 2 : 3] += wd_clk() - m_s0;
        m_visits[live ? 0 : 1] += 1;
      }
#endif
    }
    if (kc >= n_stages) {
      break;
    }
#ifdef LDP_MEASURE
    const unsigned long long m_c0 = ((ABL & 32) != 0) ? wd_clk() : 0ull;
#endif
    // ---- checkpoint (ldp_device.h): drop the products whose candidate pairs are all provably below the threshold ----
    __syncthreads();  // every wave is done with the last stage: LDS is scratch now
    {
      // the checkpoint statistics of every staged row by LDS-DMA too: slot next_cp and the whole-row slot, 32 bytes per row
      const uint8_t* cps = reinterpret_cast<const uint8_t*>(A.cp_stats);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t T = wave + kWdWaves * t;  // row-block slot T: 32 rows x 2 pieces
        if ((wg_need >> T) & 1u) {
          uint32_t first = static_cast<uint32_t>(slot_first(T));
          first = (first < A.n_local) ? first : (A.n_local - 1);
          uint32_t var = first + (lane >> 1);
          var = (var < A.n_local) ? var : (A.n_local - 1);
          const uint64_t off = static_cast<uint64_t>(var) * (kCpStride * sizeof(cp_slot)) + ((lane & 1) ? kCheckpoints : next_cp) * sizeof(cp_slot);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cps + off),
                                           (__attribute__((address_space(3))) void*)(lds + kWdCpScratchDwords + T * 256), 16, 0, 0);
        }
      }
      if constexpr (SPARSE) {
        // ... and the whole-row cp_gen_slot (calls, sum z, sum z^2: the row's calls by value) of the rows of row-block slots 2 wave, + 1
        const uint32_t T = 2 * wave + (lane >> 5);
        uint32_t first = static_cast<uint32_t>(slot_first(T));
        first = (first < A.n_local) ? first : (A.n_local - 1);
        uint32_t var = first + (lane & 31);
        var = (var < A.n_local) ? var : (A.n_local - 1);
        const uint64_t off = static_cast<uint64_t>(var) * (kCpStride * sizeof(cp_slot)) + kCpSlots * sizeof(cp_slot);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cps + off),
                                         (__attribute__((address_space(3))) void*)(lds + kWdCpScratchDwords + kWdCpSlotDwords + wave * 256), 16, 0, 0);
      }
    }
    __syncthreads();  // (drains the DMA: the slots are in LDS)
    const cp_slot* __restrict__ cpl = reinterpret_cast<const cp_slot*>(lds + kWdCpScratchDwords);  // [row-block slot][row][2]
    // allele-count coding (ldp_mfma_device.h): the accumulators hold G_P = sum over the samples visited of g_i g_j; the partial dot
    // product of x = 1 - g is G_P - n_P + sP_i + sP_j with sP = the row's sum of x over the samples visited, an INTEGER: S - s_R,
    // where s_R is a * sqrt(n_R / N) to the nearest integer (cp_slot: a = s_R * sqrt(N / n_R) as the count pass rounded it; n_P
    // samples visited, no padding among them: a checkpoint sits in front of the last k-chunk).  One integer per staged row, behind the slots.
    int32_t* __restrict__ sp = reinterpret_cast<int32_t*>(lds + kWdCpScratchDwords + kWdRowBlocks * kMfBlock * 8);
    const int32_t cp_seen = static_cast<int32_t>(kc * kWdStageSamples);
    // SPARSE: one SparseRow per staged row (its calls by value, the image's orientation) instead: the accumulators hold the partial
    // dot product itself
    SparseRow* __restrict__ srow = reinterpret_cast<SparseRow*>(lds + kWdCpScratchDwords + kWdCpSlotDwords + kWdCpGenDwords);
    if constexpr (SPARSE) {
      const cp_gen_slot* __restrict__ genl = reinterpret_cast<const cp_gen_slot*>(lds + kWdCpScratchDwords + kWdCpSlotDwords);
      for (uint32_t q = tid; q < kWdRowBlocks * kMfBlock; q += kWdWaves * 64) {
        srow[q] = sparse_row_of(genl[q]);
      }
      __syncthreads();
    } else {
      const double n_all = static_cast<double>(A.founder_ct);
      const double kappa = sqrt(((static_cast<double>(cp_seen) < n_all) ? (n_all - static_cast<double>(cp_seen)) : 1.0) / n_all);
      for (uint32_t q = tid; q < kWdRowBlocks * kMfBlock; q += kWdWaves * 64) {
        sp[q] = static_cast<int32_t>(cpl[2 * q + 1].a) - static_cast<int32_t>(rint(cpl[2 * q].a * kappa));
      }
      __syncthreads();
    }
    if (live) {
      uint32_t keep = 0;
      uint32_t* cp_epi = lds + wave * kWdCpWaveDwords;  // two products per round
      // (rounds of two products of ONE J block: J0's products 0 .. VC - 1 in pairs, then J1's VC .. 2 VC - 1; VC == 3: the second round of a J
      // block holds one product)
      constexpr int kHalf = (VC + 1) / 2;
#pragma unroll
      for (int round = 0; round < 2 * kHalf; ++round) {
        const int q = round / kHalf;  // J0 / J1
        const int p0 = q * VC + 2 * (round % kHalf);  // the round's first product
        const uint32_t round_bits = ((2 * (round % kHalf) + 1 < VC) ? 0x3u : 0x1u) << p0;
        if (!(live & round_bits)) {
          continue;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          if (round_bits & live & (1u << (p0 + pl))) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              cp_epi[(pl * 16 + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>(acc[(p0 + pl < static_cast<int>(NP)) ? p0 + pl : 0][g]));
            }
          }
        }
        const int64_t j64 = static_cast<int64_t>(jv0) + kMfBlock * (a0 + q) + r;
        const int64_t lo_j = lo_j2[q];
        const uint32_t jslot = a0 + q;
        const cp_slot cj = cpl[(jslot * kMfBlock + r) * 2];
        const cp_slot gj = cpl[(jslot * kMfBlock + r) * 2 + 1];
        const int32_t tj = SPARSE ? 0 : (sp[jslot * kMfBlock + r] - cp_seen);
        SparseRow Jr;
        if constexpr (SPARSE) {
          Jr = srow[jslot * kMfBlock + r];
        }
#pragma unroll 1
        for (uint32_t pl = 0; pl < 2; ++pl) {
          const uint32_t p = static_cast<uint32_t>(p0) + pl;
          if (!(live & round_bits & (1u << p))) {
            continue;
          }
          const uint32_t b = p - static_cast<uint32_t>(q) * VC;  // V block of the product
          const uint32_t vslot = vslot0 + b;
          bool hopeless = true;
#pragma unroll 2
          for (uint32_t g = 0; g < 16; ++g) {
            const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * h;
            const int64_t i64 = static_cast<int64_t>(vv0) + kMfBlock * (b0 + b) + row;
            if ((i64 >= lo_j) && (i64 < j64)) {
              const cp_slot ci = cpl[(vslot * kMfBlock + row) * 2];
              if constexpr (SPARSE) {
                // N dot lies within B = b_i b_j of N dot_p + a_i a_j (Cauchy-Schwarz on the remainders of x with 0 at a missing call: the
                // count pass's slots as they are); dot is an integer, and the +- 1 covers the slots' own rounding (ldp_pair_device.h)
                const double n_all = static_cast<double>(A.founder_ct);
                const double mid = fma(n_all, static_cast<double>(static_cast<int32_t>(cp_epi[(pl * 16 + g) * 64 + lane])), ci.a * cj.a);
                const double wid = fma(ci.b, cj.b, 1.0);
                const double d_lo = floor((mid - wid) / n_all), d_hi = ceil((mid + wid) / n_all);
                hopeless = hopeless && (sparse_decide(A.thresh, n_all, d_lo, d_hi, srow[vslot * kMfBlock + row], Jr) == 0);
              } else {
                const cp_slot gi = cpl[(vslot * kMfBlock + row) * 2 + 1];
                // |N dot - S_i S_j| <= |c0| + B, see pair_hopeless() in ldp_pair_device.h (dot_p is the partial dot product)
                const double dot_p = static_cast<double>(static_cast<int32_t>(cp_epi[(pl * 16 + g) * 64 + lane]) + tj + sp[vslot * kMfBlock + row]);
                const double c0 = fma(static_cast<double>(A.founder_ct), dot_p, fma(ci.a, cj.a, -(gi.a * gj.a)));
                const double bound = fabs(c0) + fma(ci.b, cj.b, 1.0);
                hopeless = hopeless && (bound < gi.b * gj.b);
              }
            }
            if constexpr (SPARSE) {
              if (!__all(hopeless)) {
                break;  // (the product stays: its other pairs need no answer; the interval test is ~70 FP64 operations a pair)
              }
            }
          }
          if (!__all(hopeless)) {
            keep |= 1u << p;
          }
        }
      }
      keep = __builtin_amdgcn_readfirstlane(keep);
      if (keep != live) {
        live = keep;
        need = wide_slots_needed<VC>(live, a0, vslot0);
        if (!live) {
          stop_stage = kc;  // the wave computes nothing from here on
        }
      }
    }
    ++next_cp;
    // which staged row-blocks does the workgroup still read?  Dead ones are no longer fetched.
    if (lane == 0) {
      s_need[wave] = need;
    }
    __syncthreads();
    uint32_t all_need = 0;
#pragma unroll
    for (int w = 0; w < kWdWaves; ++w) {
      all_need |= s_need[w];
    }
    __syncthreads();  // (s_need is rewritten at the next checkpoint; the scratch reads above are over as well)
    if (!all_need) {
      break;  // nothing left that could reach the threshold
    }
    if (all_need != wg_need) {
      wg_need = __builtin_amdgcn_readfirstlane(all_need);
      mine = (ABL & 1) ? 0u : count_mine();
    }
    issue_limit = (next_cp < n_cp) ? checkpoint_stage(next_cp) : n_stages;
    issued_base = kc;
    ring_fill();  // restart the ring at this stage
#ifdef LDP_MEASURE
    if constexpr ((ABL & 32) != 0) {
      m_t[4] += wd_clk() - m_c0;
    }
#endif
  }
  __syncthreads();  // staging is over: LDS becomes the epilogue's scratch (a private region per wave)
  if ((lane == 0) && live0) {
    // bookkeeping in product x k-step units (one MFMA each): what early termination saved of the plan's products, and what the
    // single-form stage loop computed beyond the plan (the products of this wave's rectangle that hold no candidate pair)
    const uint32_t planned = __builtin_popcount(live0);
    if (stop_stage < n_stages) {
      atomicAdd(A.counters + 2, static_cast<unsigned long long>(n_stages - stop_stage) * kWdKsteps * planned);
    }
    if (planned < NP) {
      atomicAdd(A.counters + 1, static_cast<unsigned long long>(stop_stage) * kWdKsteps * (NP - planned));
    }
  }

  // ---- epilogue: accumulators through LDS so the per-pair code is a rolled loop ----
  // lane l, register g of a product holds first variant (g & 3) + 8 (g >> 2) + 4 (l >> 5) of the V block, second variant
  // l & 31 of the J block (tools/mfma_probe.hip, fact 1)
  uint32_t n_true = 0;
#ifdef LDP_MEASURE
  const unsigned long long m_e0 = wd_clk();
  if constexpr ((ABL & 16) != 0) {
    live = 0;  // (no per-pair epilogue: what it costs)
  }
#endif
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!(live & (kColMask << (VC * round)))) {
      continue;
    }
#pragma unroll
    for (int pl = 0; pl < VC; ++pl) {
      if (live & (1u << (VC * round + pl))) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          epi[(pl * 16 + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>((round ? acc[VC + pl] : acc[pl])[g]));
        }
      }
    }
    if constexpr (SPARSE) {
      n_true += wide_sparse_round(A, epi, lane, jv0 + static_cast<int32_t>(kMfBlock * (a0 + round)), vv0 + static_cast<int32_t>(kMfBlock * b0), jend,
                                  (live >> (4 * round)) & 0xfu, lo_j2[round]);
      continue;
    }
    const int64_t j64 = static_cast<int64_t>(jv0) + kMfBlock * (a0 + round) + r;
    if (j64 < static_cast<int64_t>(jend)) {
      const uint32_t j = static_cast<uint32_t>(j64);
      const uint32_t lo_j = lo_j2[round];
      if (lo_j < j) {
        const int32_t sum_j = A.recs[j].sum;
        const uint32_t ssq_j = A.recs[j].ssq;
        const uint32_t flags_j = A.recs[j].flags;
        const int32_t sum_img_j = img_differs(flags_j) ? -sum_j : sum_j;
#pragma unroll 1
        for (uint32_t pl = 0; pl < static_cast<uint32_t>(VC); ++pl) {
          if (!(live & (1u << (VC * round + pl)))) {
            continue;
          }
          const int64_t vfirst = static_cast<int64_t>(vv0) + kMfBlock * (b0 + pl) + 4 * h;
#pragma unroll 1
          for (uint32_t g = 0; g < 16; ++g) {
            const int64_t i64 = vfirst + (g & 3) + 8 * (g >> 2);
            if ((i64 < static_cast<int64_t>(lo_j)) || (i64 >= j64)) {
              continue;
            }
            const uint32_t i = static_cast<uint32_t>(i64);
            const ldp_variant_rec ri = A.recs[i];
            ldp_pair_stats_t ps;
            const int32_t dot_img = static_cast<int32_t>(epi[(pl * 16 + g) * 64 + lane]) + sum_img_of(ri) + sum_img_j - g_bias;  // G -> dot of x = 1 - g
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
#ifdef LDP_MEASURE
  {
    const unsigned long long m_clk1 = wd_clk(), m_wall1 = __builtin_amdgcn_s_memrealtime();
    m_t[5] = m_clk1 - m_e0;
    if (lane == 0) {
      if constexpr ((ABL & 32) != 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          atomicAdd(&g_wide_measure[q], m_t[q]);
        }
        atomicAdd(&g_wide_measure[9], m_visits[0]);
        atomicAdd(&g_wide_measure[10], m_visits[1]);
      }
      atomicAdd(&g_wide_measure[6], m_clk1 - m_clk0);
      atomicAdd(&g_wide_measure[7], 1ull);
      atomicAdd(&g_wide_measure[8], m_wall1 - m_wall0);
    }
  }
#endif
}

template <int ABL, bool SPARSE = false>
__global__ __launch_bounds__(kWdWaves * 64, 2) void pair_mfma_wide_kernel(PairKernelArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_need[kWdWaves];
  if (*A.route != (SPARSE ? kRouteSparse : kRouteComplete)) {
    return;  // complete rows / a few missing calls / many: one kernel family owns a launch (route_kernel); the others leave at once
  }
  const uint32_t per_xcd = (A.n_wd_tiles + 7) / 8;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // consecutive tiles on one XCD
  if (idx >= A.n_wd_tiles) {
    return;
  }
  const MfmaTile* __restrict__ tile = A.wd_tiles + idx;
  if constexpr ((ABL == 0) && !SPARSE) {
    // a diagonal tile in 2 x 3 rectangles (block-uniform: two whole bodies side by side, each with its ONE form of the stage loop -- a second form
    // INSIDE a loop is what made hipcc spill in rounds 3-5)
    if (A.wd_diag_split && (tile->jv == tile->vv)) {
      wide_tile<0, false, 3>(A, lds, s_need, tile);
      return;
    }
  }
  wide_tile<ABL, SPARSE, 4>(A, lds, s_need, tile);
}


// ================================================================================================================================
// pair_mfma_wide_async_kernel -- the same tiles, the same arithmetic, NO workgroup barrier in the stage loop (round 5).
//
// With s_barrier at every stage the eight waves of the one workgroup a CU holds move in lock-step: after a barrier all of them issue
// their DMA share, read LDS and expand before the first MFMA, so the two waves of a SIMD meet their bubbles together and the matrix
// pipe idles (profiles/r05_experiments.md: the attribution).  Here the waves only exchange two counters each through LDS:
//   s_flags[w]      stages whose DMA share of wave w has landed  (wave w: s_waitcnt vmcnt, then the store)
//   s_flags[8 + w]  stages wave w has finished reading
// A stage is 256 samples (32 KiB for the 16 row-blocks), the ring holds FOUR: the one being multiplied, the next one (landed), and
// two in flight -- the same 64 KiB in flight and the same two stage-times (4,096 matrix-pipe cycles per SIMD) for a load to land as
// the two-stage ring of 512-sample stages above.  Wave w, stage s:
//   top     wait until every wave's share of stage s has landed (s_flags[0..7] > s)            -- normally true for a stage already
//   issue   its share of stage s + 3 into the buffer of stage s - 1, once every wave has finished that one (s_flags[8..15] >= s);
//           not yet?  then after the stage (the wave is ahead of the others: nothing is late)
//   multiply stage s (wide_stage, as above)
//   end     store "finished s"; s_waitcnt vmcnt until its share of stage s + 2 has landed (issued two stages ago), store that
// so a wave may run a whole stage ahead of or behind any other without anybody waiting, and the two waves of a SIMD drift out of
// phase: one multiplies while the other reads and expands.  LDS executes a wave's operations in order, so "data read, then flag
// stored" and "flag seen, then data read" need no more than program order (compiler fences only).  A spin that never ends would be a
// bug in this protocol: it traps after ~2^22 polls instead of hanging the device.
// Checkpoints, epilogue, plan and counters are those of the kernel above; the two are interchangeable launch by launch
// (PairKernelArgs::wd_async, engine option "wide_async") and the parity tests run both.
constexpr uint32_t kWaStageSamples = 256;
constexpr uint32_t kWaPieces = 4;                                        // 16-byte pieces per row and stage
constexpr uint32_t kWaBlockUnits = kMfBlock * kWaPieces;                 // 128
constexpr uint32_t kWaStageDwords = kWdRowBlocks * kWaBlockUnits * 4;    // 8,192 dwords = 32 KiB
constexpr uint32_t kWaRing = 4;
constexpr uint32_t kWaInstrPerBlock = kWaBlockUnits / 64;                // 2
constexpr uint32_t kWaDma = (kWdRowBlocks * kWaInstrPerBlock) / kWdWaves;  // DMA wave-instructions per wave and stage: 4
constexpr uint32_t kWaKsteps = kWaStageSamples / 64;                     // 4
static_assert(kWaStageDwords * kWaRing <= kWdLdsDwords, "ring fits the epilogue scratch");
__device__ __forceinline__ uint32_t wa_swizzle(uint32_t row) { return (row >> 2) & 3u; }  // (StageGeom<4>'s: 64-byte row pitch)

// wait until at most `allowed` of this wave's loads are in flight (no barrier)
__device__ __forceinline__ void wait_dma_only(uint32_t allowed) {
#define LDP_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
  switch (allowed) {
    LDP_WAIT_CASE(1) LDP_WAIT_CASE(2) LDP_WAIT_CASE(3) LDP_WAIT_CASE(4) LDP_WAIT_CASE(5) LDP_WAIT_CASE(6) LDP_WAIT_CASE(7) LDP_WAIT_CASE(8)
    LDP_WAIT_CASE(9) LDP_WAIT_CASE(10) LDP_WAIT_CASE(11) LDP_WAIT_CASE(12)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef LDP_WAIT_CASE
}

// The counters are read and written by hand-placed LDS instructions: through a pointer hipcc would either take the generic address
// space (flat_load: it counts against vmcnt AND lgkmcnt) or, seeing an LDS read behind an LDS-DMA it cannot tell apart from it, put
// `s_waitcnt vmcnt(0)` in front of every poll -- and drain the ring.  (An LDS operation the compiler does not know about only makes
// its own lgkmcnt waits more conservative: LDS returns in order.)
__device__ __forceinline__ uint32_t wa_flag_read(uint32_t lds_byte_addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_byte_addr) : "memory");
  return v;
}
__device__ __forceinline__ void wa_flag_write(uint32_t lds_byte_addr, uint32_t value) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_byte_addr), "v"(value) : "memory");
}

template <int ABL>
__global__ __launch_bounds__(kWdWaves * 64, 2) void pair_mfma_wide_async_kernel(PairKernelArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t s_need[kWdWaves];
  __shared__ uint32_t s_flags[2 * kWdWaves];
  if (*A.route != kRouteComplete) {
    return;
  }
  const uint32_t per_xcd = (A.n_wd_tiles + 7) / 8;
  const uint32_t idx = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (idx >= A.n_wd_tiles) {
    return;
  }
  const MfmaTile* __restrict__ tile = A.wd_tiles + idx;
#ifdef LDP_MEASURE
  unsigned long long m_t[3] = {0, 0, 0};
  const unsigned long long m_clk0 = wd_clk(), m_wall0 = __builtin_amdgcn_s_memrealtime();
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane = tid & 63;
  const uint32_t r = lane & 31;
  const uint32_t h = lane >> 5;
  const int32_t jv0 = __builtin_amdgcn_readfirstlane(tile->jv);
  const int32_t vv0 = __builtin_amdgcn_readfirstlane(tile->vv);
  const uint32_t jend = __builtin_amdgcn_readfirstlane(tile->jend);
  const uint32_t mask_lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(tile->mask));
  const uint32_t mask_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(tile->mask >> 32));
  if (!(mask_lo | mask_hi)) {
    return;
  }
  if (tid < 2 * kWdWaves) {
    s_flags[tid] = 0;
  }
  __syncthreads();
  const uint32_t flags_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)s_flags));
  const uint32_t poll_addr = flags_addr + 4 * (lane & 15);
  const bool diag = (jv0 == vv0);
  const int32_t g_bias = g_bias_of(A.founder_ct, kWdStageSamples);   // (whole 512-sample chunks are visited, as above)
  const uint32_t row_bytes = static_cast<uint32_t>(A.code_row_bytes);
  const uint32_t n_stages = 2 * ((A.founder_ct + kWdStageSamples - 1) / kWdStageSamples);

  const uint32_t a0 = 2 * (wave & 3), b0 = 4 * (wave >> 2);
  const uint32_t vslot0 = (diag ? 0u : static_cast<uint32_t>(kWdTile)) + b0;
  auto mask_row = [&](uint32_t a) { return ((a < 4) ? (mask_lo >> (8 * a)) : (mask_hi >> (8 * (a - 4)))) & 0xffu; };
  uint32_t live = ((mask_row(a0) >> b0) & 0xfu) | (((mask_row(a0 + 1) >> b0) & 0xfu) << 4);
  live = __builtin_amdgcn_readfirstlane(live);
  uint32_t wg_need = 0;
#pragma unroll
  for (uint32_t w = 0; w < static_cast<uint32_t>(kWdWaves); ++w) {
    const uint32_t wa = 2 * (w & 3), wb = 4 * (w >> 2);
    const uint32_t wl = ((mask_row(wa) | mask_row(wa + 1)) >> wb) & 0xfu;
    wg_need |= wide_slots_needed(wl, wa, (diag ? 0u : static_cast<uint32_t>(kWdTile)) + wb);
  }
  wg_need = __builtin_amdgcn_readfirstlane(wg_need);
  auto slot_first = [&](uint32_t s) { return (s < static_cast<uint32_t>(kWdTile)) ? (jv0 + static_cast<int32_t>(kMfBlock * s)) : (vv0 + static_cast<int32_t>(kMfBlock * (s - kWdTile))); };

  // ---- DMA plan: instruction T of a stage = half of row-block slot T >> 1: sixteen rows x four 16-byte pieces ----
  const uint8_t* base_t[kWaDma];
  uint32_t src_off[kWaDma];
#pragma unroll
  for (int t = 0; t < static_cast<int>(kWaDma); ++t) {
    const uint32_t T = wave + kWdWaves * t;
    const uint32_t slot = T / kWaInstrPerBlock;
    uint32_t first = static_cast<uint32_t>(slot_first(slot));
    first = (first < A.n_local) ? first : (A.n_local - 1);
    first = __builtin_amdgcn_readfirstlane(first);
    base_t[t] = A.codes + static_cast<uint64_t>(first) * row_bytes;
    const uint32_t rr = (T % kWaInstrPerBlock) * 16 + (lane >> 2);
    const uint32_t col = (lane & 3) ^ wa_swizzle(rr);
    uint32_t var = first + rr;
    var = (var < A.n_local) ? var : (A.n_local - 1);
    src_off[t] = (var - first) * row_bytes + col * 16;
  }
  auto count_mine = [&]() {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kWaDma); ++t) {
      m += ((wg_need >> ((wave + kWdWaves * t) / kWaInstrPerBlock)) & 1u) ? 1u : 0u;
    }
    return m;
  };
  uint32_t mine = count_mine();

  uint32_t joff[2], voff[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    joff[q] = (a0 + q) * kWaBlockUnits;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    voff[b] = (vslot0 + b) * kWaBlockUnits;
  }
  uint32_t need = wide_slots_needed(live, a0, vslot0);
  uint32_t lo_j2[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t j = static_cast<uint32_t>(jv0) + kMfBlock * (a0 + q) + r;
    if (j < jend) {
      lo_j2[q] = A.lo[j];
    }
  }
  const uint32_t sw = wa_swizzle(r);
  const uint32_t oH = r * kWaPieces + (h ^ sw), oR = r * kWaPieces + ((2 + h) ^ sw);

  mf_v16f acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      acc[p][g] = 0.f;
    }
  }

  uint32_t next_cp = 0;
  const uint32_t n_cp = A.cp_stats ? A.n_checkpoints : 0;
  const uint32_t live0 = live;
  uint32_t stop_stage = n_stages;
  auto dma_stage = [&](uint32_t s) {
    const uint32_t kbyte = s * (kWaStageSamples / 4);
    uint32_t* dst = lds + (s & (kWaRing - 1)) * kWaStageDwords;
#pragma unroll
    for (int t = 0; t < static_cast<int>(kWaDma); ++t) {
      const uint32_t T = wave + kWdWaves * t;
      if ((wg_need >> (T / kWaInstrPerBlock)) & 1u) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_t[t] + kbyte + src_off[t]),
                                         (__attribute__((address_space(3))) void*)(dst + T * 256), 16, 0, 0);
      }
    }
  };
  auto checkpoint_stage = [&](uint32_t cp) {
    const uint32_t s = 2 * A.checkpoint_chunk[cp];  // (checkpoints sit on 512-sample chunk boundaries)
    return (s < n_stages) ? s : n_stages;
  };
  // poll the sixteen counters: has every wave's share of stage `s_landed` landed, has every wave finished stage `s_done`?
  // (s_done < 0: nothing to wait for)
  auto poll = [&](uint32_t need_landed, uint32_t need_done, bool* landed_ok, bool* done_ok) {
    const uint32_t v = wa_flag_read(poll_addr);
    *landed_ok = __all((lane & 8) || (v >= need_landed));
    *done_ok = __all(!(lane & 8) || (v >= need_done));
  };
  auto spin_fail = [&]() { __builtin_trap(); };

  uint32_t* epi = lds + wave * kWdEpiWaveDwords;
  uint32_t issued = 0, confirmed = 0;
  uint32_t issue_limit = (next_cp < n_cp) ? checkpoint_stage(next_cp) : n_stages;
  auto ring_fill = [&](uint32_t base) {  // (every buffer is free: the kernel's start, or behind a checkpoint's barriers)
    while ((issued < issue_limit) && (issued < base + kWaRing - 1)) {
      dma_stage(issued);
      ++issued;
    }
  };
  // this wave's share of every stage <= c has landed: tell the others
  auto confirm = [&](uint32_t c) {
    wait_dma_only(mine * (issued - c - 1));
    wa_flag_write(flags_addr + 4 * wave, c + 1);
    confirmed = c + 1;
  };
  ring_fill(0);
  for (uint32_t kc = 0; kc < n_stages;) {
    const uint32_t kc_end = issue_limit;
    for (; kc < kc_end; ++kc) {
      if (confirmed <= kc) {
        confirm(kc);  // (only behind a ring_fill)
      }
#ifdef LDP_MEASURE
      const unsigned long long m_p0 = ((ABL & 32) != 0) ? wd_clk() : 0ull;
#endif
      // buffer of the stage to issue next (stage `issued`, buffer of stage issued - 4): free once every wave has finished that stage
      const bool want_issue = (issued < issue_limit);
      const uint32_t need_done = (issued >= kWaRing) ? (issued - kWaRing + 1) : 0u;
      bool landed_ok, done_ok;
      poll(kc + 1, need_done, &landed_ok, &done_ok);
      if (live) {
        for (uint32_t spin = 0; !landed_ok; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          poll(kc + 1, need_done, &landed_ok, &done_ok);
          if (spin > (1u << 22)) {
            spin_fail();
          }
        }
      }
      bool pending = false;
      if (want_issue) {
        if (done_ok) {
          dma_stage(issued);
          ++issued;
        } else {
          pending = true;
        }
      }
#ifdef LDP_MEASURE
      if constexpr ((ABL & 32) != 0) {
        m_t[0] += wd_clk() - m_p0;
      }
#endif
      if (live) {
        const mf_u4* __restrict__ st4 = reinterpret_cast<const mf_u4*>(lds + (kc & (kWaRing - 1)) * kWaStageDwords);
        wide_stage<0>(st4, joff, voff, oH, oR, acc);
      }
      wa_flag_write(flags_addr + 4 * (kWdWaves + wave), kc + 1);  // finished reading stage kc (LDS runs this wave's reads before this store)
#ifdef LDP_MEASURE
      const unsigned long long m_q0 = ((ABL & 32) != 0) ? wd_clk() : 0ull;
#endif
      if (pending) {
        for (uint32_t spin = 0;; ++spin) {
          poll(0, need_done, &landed_ok, &done_ok);
          if (done_ok) {
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (spin > (1u << 22)) {
            spin_fail();
          }
        }
        dma_stage(issued);
        ++issued;
      }
#ifdef LDP_MEASURE
      const unsigned long long m_q1 = ((ABL & 32) != 0) ? wd_clk() : 0ull;
#endif
      // the share issued two stages ago must have landed by now: the others will want it a stage from here
      {
        const uint32_t c = (kc + 2 < issued) ? (kc + 2) : (issued - 1);
        if (confirmed <= c) {
          confirm(c);
        }
      }
#ifdef LDP_MEASURE
      if constexpr ((ABL & 32) != 0) {
        m_t[1] += m_q1 - m_q0;
        m_t[2] += wd_clk() - m_q1;
      }
#endif
    }
    if (kc >= n_stages) {
      break;
    }
    // ---- checkpoint (as in the kernel above; a stage is 256 samples here) ----
    __syncthreads();
    {
      const uint8_t* cps = reinterpret_cast<const uint8_t*>(A.cp_stats);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t T = wave + kWdWaves * t;
        if ((wg_need >> T) & 1u) {
          uint32_t first = static_cast<uint32_t>(slot_first(T));
          first = (first < A.n_local) ? first : (A.n_local - 1);
          uint32_t var = first + (lane >> 1);
          var = (var < A.n_local) ? var : (A.n_local - 1);
          const uint64_t off = static_cast<uint64_t>(var) * (kCpStride * sizeof(cp_slot)) + ((lane & 1) ? kCheckpoints : next_cp) * sizeof(cp_slot);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cps + off),
                                           (__attribute__((address_space(3))) void*)(lds + kWdCpScratchDwords + T * 256), 16, 0, 0);
        }
      }
    }
    __syncthreads();
    const cp_slot* __restrict__ cpl = reinterpret_cast<const cp_slot*>(lds + kWdCpScratchDwords);
    int32_t* __restrict__ sp = reinterpret_cast<int32_t*>(lds + kWdCpScratchDwords + kWdRowBlocks * kMfBlock * 8);
    const int32_t cp_seen = static_cast<int32_t>(kc * kWaStageSamples);
    {
      const double n_all = static_cast<double>(A.founder_ct);
      const double kappa = sqrt(((static_cast<double>(cp_seen) < n_all) ? (n_all - static_cast<double>(cp_seen)) : 1.0) / n_all);
      for (uint32_t q = tid; q < kWdRowBlocks * kMfBlock; q += kWdWaves * 64) {
        sp[q] = static_cast<int32_t>(cpl[2 * q + 1].a) - static_cast<int32_t>(rint(cpl[2 * q].a * kappa));
      }
      __syncthreads();
    }
    if (live) {
      uint32_t keep = 0;
      uint32_t* cp_epi = lds + wave * kWdCpWaveDwords;
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
        const int64_t j64 = static_cast<int64_t>(jv0) + kMfBlock * (a0 + q) + r;
        const int64_t lo_j = lo_j2[q];
        const uint32_t jslot = a0 + q;
        const cp_slot cj = cpl[(jslot * kMfBlock + r) * 2];
        const cp_slot gj = cpl[(jslot * kMfBlock + r) * 2 + 1];
        const int32_t tj = sp[jslot * kMfBlock + r] - cp_seen;
#pragma unroll 1
        for (uint32_t pl = 0; pl < 2; ++pl) {
          const uint32_t p = 2 * round + pl;
          if (!(live & (1u << p))) {
            continue;
          }
          const uint32_t b = p & 3;
          const uint32_t vslot = vslot0 + b;
          bool hopeless = true;
#pragma unroll 2
          for (uint32_t g = 0; g < 16; ++g) {
            const uint32_t row = (g & 3) + 8 * (g >> 2) + 4 * h;
            const int64_t i64 = static_cast<int64_t>(vv0) + kMfBlock * (b0 + b) + row;
            if ((i64 >= lo_j) && (i64 < j64)) {
              const cp_slot ci = cpl[(vslot * kMfBlock + row) * 2];
              const cp_slot gi = cpl[(vslot * kMfBlock + row) * 2 + 1];
              const double dot_p = static_cast<double>(static_cast<int32_t>(cp_epi[(pl * 16 + g) * 64 + lane]) + tj + sp[vslot * kMfBlock + row]);
              const double c0 = fma(static_cast<double>(A.founder_ct), dot_p, fma(ci.a, cj.a, -(gi.a * gj.a)));
              const double bound = fabs(c0) + fma(ci.b, cj.b, 1.0);
              hopeless = hopeless && (bound < gi.b * gj.b);
            }
          }
          if (!__all(hopeless)) {
            keep |= 1u << p;
          }
        }
      }
      keep = __builtin_amdgcn_readfirstlane(keep);
      if (keep != live) {
        live = keep;
        need = wide_slots_needed(live, a0, vslot0);
        if (!live) {
          stop_stage = kc;
        }
      }
    }
    ++next_cp;
    if (lane == 0) {
      s_need[wave] = need;
    }
    __syncthreads();
    uint32_t all_need = 0;
#pragma unroll
    for (int w = 0; w < kWdWaves; ++w) {
      all_need |= s_need[w];
    }
    __syncthreads();
    if (!all_need) {
      break;
    }
    if (all_need != wg_need) {
      wg_need = __builtin_amdgcn_readfirstlane(all_need);
      mine = count_mine();
    }
    issue_limit = (next_cp < n_cp) ? checkpoint_stage(next_cp) : n_stages;
    ring_fill(kc);
  }
  __syncthreads();
  if ((lane == 0) && live0) {
    const uint32_t planned = __builtin_popcount(live0);
    if (stop_stage < n_stages) {
      atomicAdd(A.counters + 2, static_cast<unsigned long long>(n_stages - stop_stage) * kWaKsteps * planned);
    }
    if (planned < 8) {
      atomicAdd(A.counters + 1, static_cast<unsigned long long>(stop_stage) * kWaKsteps * (8 - planned));
    }
  }

  // ---- epilogue (as above) ----
  uint32_t n_true = 0;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!(live & (0xfu << (4 * round)))) {
      continue;
    }
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      if (live & (1u << (4 * round + pl))) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          epi[(pl * 16 + g) * 64 + lane] = static_cast<uint32_t>(static_cast<int32_t>((round ? acc[4 + pl] : acc[pl])[g]));
        }
      }
    }
    const int64_t j64 = static_cast<int64_t>(jv0) + kMfBlock * (a0 + round) + r;
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
          const int64_t vfirst = static_cast<int64_t>(vv0) + kMfBlock * (b0 + pl) + 4 * h;
#pragma unroll 1
          for (uint32_t g = 0; g < 16; ++g) {
            const int64_t i64 = vfirst + (g & 3) + 8 * (g >> 2);
            if ((i64 < static_cast<int64_t>(lo_j)) || (i64 >= j64)) {
              continue;
            }
            const uint32_t i = static_cast<uint32_t>(i64);
            const ldp_variant_rec ri = A.recs[i];
            ldp_pair_stats_t ps;
            const int32_t dot_img = static_cast<int32_t>(epi[(pl * 16 + g) * 64 + lane]) + sum_img_of(ri) + sum_img_j - g_bias;
            ps.dot = (img_differs(ri.flags) ^ img_differs(flags_j)) ? -dot_img : dot_img;
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
#ifdef LDP_MEASURE
  {
    const unsigned long long m_clk1 = wd_clk(), m_wall1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      if constexpr ((ABL & 32) != 0) {
        atomicAdd(&g_wide_measure[11], m_t[0]);  // top-of-stage polls (+ the DMA issue behind them)
        atomicAdd(&g_wide_measure[12], m_t[1]);  // deferred issues
        atomicAdd(&g_wide_measure[13], m_t[2]);  // s_waitcnt vmcnt of the confirmations
      }
      atomicAdd(&g_wide_measure[6], m_clk1 - m_clk0);
      atomicAdd(&g_wide_measure[7], 1ull);
      atomicAdd(&g_wide_measure[8], m_wall1 - m_wall0);
    }
  }
#endif
}

}  // namespace

hipError_t launch_pair_wide(const PairKernelArgs& a_in, hipStream_t stream, bool sparse) {
  if (!a_in.n_wd_tiles) {
    return hipSuccess;
  }
  PairKernelArgs a = a_in;
  const size_t lds = static_cast<size_t>(kWdLdsDwords) * sizeof(uint32_t);
  a.lds_dwords = static_cast<uint32_t>(lds / sizeof(uint32_t));
  if (a.wd_async || sparse) {
    a.wd_diag_split = 0;  // (the diagonal tiles' kernel exists for the barrier kernel on complete data)
  }
  const uint32_t per_xcd = (a.n_wd_tiles + 7) / 8;
  const dim3 grid(per_xcd * 8), block(kWdWaves * 64);
  if (sparse) {
    // the same tiles for launches whose rows have a few missing calls (route kRouteSparse; always the barrier kernel)
    static const bool sparse_attr_set =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) == hipSuccess;
    (void)sparse_attr_set;
    hipLaunchKernelGGL((pair_mfma_wide_kernel<0, true>), grid, block, lds, stream, a);
    return hipGetLastError();
  }
#ifdef LDP_MEASURE
  // LDP_DEBUG_WIDE_ABLATE (measurement build only; bits 0-3 give WRONG results): see wide_stage.  Read at every launch.
  const char* v = LDP_ENV("LDP_DEBUG_WIDE_ABLATE");
  const int ablate = v ? atoi(v) : 0;
  if (a.wd_async) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_async_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_async_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (ablate & 32) {
      hipLaunchKernelGGL(pair_mfma_wide_async_kernel<32>, grid, block, lds, stream, a);
    } else {
      hipLaunchKernelGGL(pair_mfma_wide_async_kernel<0>, grid, block, lds, stream, a);
    }
    return hipGetLastError();
  }
#define LDP_WD_CASE(n)                                                                                                                                        \
  case n:                                                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_kernel<n>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)); \
    hipLaunchKernelGGL(pair_mfma_wide_kernel<n>, grid, block, lds, stream, a);                                                                                \
    break;
  switch (ablate) {
    LDP_WD_CASE(1) LDP_WD_CASE(2) LDP_WD_CASE(4) LDP_WD_CASE(7) LDP_WD_CASE(8) LDP_WD_CASE(9) LDP_WD_CASE(15) LDP_WD_CASE(16) LDP_WD_CASE(25) LDP_WD_CASE(31) LDP_WD_CASE(32) LDP_WD_CASE(64)
    default:
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      hipLaunchKernelGGL(pair_mfma_wide_kernel<0>, grid, block, lds, stream, a);
      break;
  }
#undef LDP_WD_CASE
#else
  static const bool attr_set = []() {
    const int bytes = static_cast<int>(kWdLdsDwords * sizeof(uint32_t));
    return (hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) &&
           (hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_mfma_wide_async_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess);
  }();
  (void)attr_set;
  if (a.wd_async) {
    hipLaunchKernelGGL(pair_mfma_wide_async_kernel<0>, grid, block, lds, stream, a);
  } else {
    hipLaunchKernelGGL(pair_mfma_wide_kernel<0>, grid, block, lds, stream, a);
  }
#endif
  return hipGetLastError();
}

#ifdef LDP_MEASURE
// (measurement build only) the sums of g_wide_measure since the last reset
extern "C" int ldp_measure_wide_counters(unsigned long long* out16, int reset) {
  if (out16 && (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wide_measure), 16 * sizeof(unsigned long long)) != hipSuccess)) {
    return 1;
  }
  if (reset) {
    const unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wide_measure), zero, sizeof(zero)) != hipSuccess) {
      return 1;
    }
  }
  return 0;
}
#endif

}  // namespace ldp
