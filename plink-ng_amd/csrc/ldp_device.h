// ldp_device.h -- layout constants and launch-side declarations shared by the HIP kernels
// (ldp_kernels.hip) and the host runtime (ldp_engine*.cpp).
#ifndef LDP_DEVICE_H
#define LDP_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ldprune_hip.h"
#include "../../include/ldprune_hip_debug.h"
#include "ldp_env.h"

namespace ldp {

// ---- HBM layout of the bit-planes -------------------------------------------------------------
// One variant = row of `chunks` k-chunks; a k-chunk = kChunkDwords dwords of the `hom` plane followed
// by kChunkDwords dwords of the `ref2het` plane, so one k-chunk of one variant
// (128 contiguous bytes) is exactly what 8 lanes x 16 B DMA into one LDS row.  Bit s%32 of plane dword s/32 = sample s.
// Pad dwords (samples >= founder_ct) are zero in both planes == "missing", which contributes to no count.
constexpr int kChunkDwords = 16;                 // plane dwords per k-chunk (512 samples)
constexpr int kRowChunkDwords = 2 * kChunkDwords; // hom + ref2het = 128 B
constexpr int kLdsRowDwords = kRowChunkDwords + 4; // 36: odd number (9) of 16-B slots -> conflict-free b128 reads

// ---- HBM layout of the 2-bit code image (the resident format of the matrix-pipe kernels) ------------------------
// One variant = one row of 2-bit genotype codes as the .pgen main track / PgrGetInv1 has them (00 hom-REF -- or hom-major for
// LDP_GENO_INVERSE input --, 01 het, 10 hom-ALT, 11 missing; sample s at bits 2 (s % 16) of dword s / 16), kCodeStageBytes
// (= 256 samples) at a time: the row stride is a whole number of stages and the samples beyond founder_ct are coded 11
// ("missing": no contribution to any count), so the kernels never mask.  The image is NOT re-oriented to the major allele: the
// prune predicate cov^2 > thr var1 var2 does not depend on the orientation, and where a sign does (the reported sums and dot
// product, --r-unphased) the epilogue takes it from the records' ALT-major flags.  Same bytes per variant as the input: rows that
// already are REF- / INVERSE-coded in the engine's image (ldp_map_rows) are counted in place, nothing is rewritten.
// Rows of the image are stored MAJOR-allele-oriented since round 6 (EngineOptions::orient_rows): the count pass, which decides the major allele, stores
// an ALT-major row with codes 00 <-> 10 swapped -- what GenovecInvertUnsafe (pgenlib_misc.cc:1090) does to the reference's rows -- and says so in the
// record (flags bit 3) and in a per-row byte the engine keeps (a row counted in place a second time must be read as what it is).  The kernels run at the
// socket's power cap and hom-major = code 00 = operand 0 is the cheapest genotype to multiply: 285.5 -> 273.3 ms of pair kernel on the config-3 share
// when every row is stored that way (profiles/r06_share_alt_minor.json).  Rows that are already inverse-coded (LDP_GENO_INVERSE) are never touched.
constexpr uint32_t kRecAltMajor = 1u, kRecStoredInverted = 8u;
// 1: the image row's orientation differs from the record's (major allele): ALT is major and the row is stored as the input had it
__host__ __device__ inline uint32_t img_differs(uint32_t flags) { return (flags ^ (flags >> 3)) & 1u; }
constexpr uint32_t kCodeStageSamples = 256;
constexpr uint32_t kCodeStageBytes = kCodeStageSamples / 4;  // 64
// (rows are whole 512-sample k-chunks long -- 128 bytes, a cache line -- so that the wide-band kernel's stages never straddle a row end)
inline uint64_t code_row_bytes_of(uint32_t founder_ct) { return static_cast<uint64_t>((founder_ct + 511) / 512) * 128; }

// ---- pair-tile geometry -------------------------------------------------------------------------
// A block owns kTileJ consecutive "second" variants j and a run of distances d = j - i in units of 8.  Wave w owns
// second-variant group w (lane (tx = lane&7, ty = lane>>3): j = j0 + tx + 8w) of ALL the block's units
// (d = d0 + ty + 8a, a < units <= kMaxUnitsPerBlock): one pair per lane and unit ("column layout").
constexpr int kTileJ = 32;
constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = 64 * kWavesPerBlock;
constexpr int kMaxUnitsPerBlock = 12;  // 96 distances; wider windows take several blocks per J-tile

struct WorkItem {
  uint32_t j0;        // first second-variant index
  uint32_t jend;      // exclusive end (<= j0 + kTileJ)
  uint32_t d0;        // first distance handled by the block (>= 1)
  uint32_t units;     // number of 8-distance units (<= kMaxUnitsPerBlock)
  uint32_t sfirst;    // subcontig bounds (row clamp)
  uint32_t send;
};

// ---- early termination of hopeless tiles (complete-data kernel) ----------------------------------------------
// At up to kCheckpoints k-chunk boundaries a wave
// bounds, for each of its pairs, how large |N*dot - S_i*S_j| can still become.  With R = the samples not yet
// visited, s = sum over R, q = sum of squares over R, n = |R|:
//     dot_R = sum_R (x-s_i/n)(y-s_j/n) + s_i*s_j/n,   |first term| <= sqrt(q_i - s_i^2/n) * sqrt(q_j - s_j^2/n)
// (Cauchy-Schwarz on the centred remainders).  prepare_kernel stores per variant, per checkpoint, the two numbers
// this needs, pre-scaled so the pair test is a handful of FP64 ops (cp_stats, kCpSlots x 16 bytes per variant):
//     slot k < kCheckpoints: { s_R * sqrt(N / n_R),  sqrt(N * (q_R - s_R^2 / n_R)) }
//     slot kCheckpoints    : { S (whole-row sum),    sqrt(N*Q - S^2) * sqrt(sqrt(thresh) * (1 - 1e-6)) }
// A distance unit whose pairs all provably stay below the r^2 threshold is dropped (far end of a wave's unit list
// first); the tile is re-planned to the rows the block still needs; a block with nothing left leaves the k-loop.
// Results are unchanged: only pairs whose predicate is provably false are skipped.
constexpr int kCheckpoints = 5;
constexpr int kCpSlots = kCheckpoints + 1;
// ... followed, in the same per-variant record, by what the six-product kernel's bound needs (cp_gen_slot below, written by the
// code image's count pass only): the whole row, then the remainders behind checkpoints kGenCheckpointFirst .. + kGenCheckpoints - 1
// (0.08 and 0.16 behind 1 - sqrt(r2) of the samples: where that kernel stops).  One contiguous 144-byte store per variant.
constexpr int kGenCheckpointFirst = 2;
constexpr int kGenCheckpoints = 2;
constexpr int kCpStride = kCpSlots + 1 + kGenCheckpoints;
struct cp_slot {
  double a, b;
};
// Tiles with missing calls: the pair statistics run over the pairwise-complete samples, which no per-variant number
// pins down, so the bound works on intervals.  Per variant and checkpoint, over the not-yet-visited samples R where
// the variant itself is called, in z = 1 - x (0 = homozygous major, 1 = het, 2 = homozygous minor):
// (in the code image's cp_stats records: slot kCpSlots = the WHOLE row -- its calls, sum z, sum z^2, and in pad its ALT-major
// flag --, slots kCpSlots + 1 + k = the remainder behind checkpoint kGenCheckpointFirst + k)
struct cp_gen_slot {
  uint32_t nm_r;  // calls
  uint32_t zs_r;  // sum z
  uint32_t zq_r;  // sum z^2
  uint32_t pad;
};

// ---- matrix-pipe pair tiles (ldp_pair_mfma.hip) -------------------------------------------------------------
// The complete-data dot product sum x_i x_j, x in {-1, 0, +1}, is an integer matrix product over the samples.  It runs
// on the MFMA pipe as FP4 (E2M1: +-1 and 0 are exact) with f32 accumulation, which is integer-exact below 2^24.
// Work is cut into row-blocks of kMfBlock consecutive variants, aligned to the subcontig start.  A WAVE owns one
// "parallelogram": second-variant blocks J0, J1 = J0 + 32 and first-variant blocks V0..V4 (consecutive), products
//   (J0, V0) (J0, V1) (J0, V2) (J0, V3)   and   (J1, V1) (J1, V2) (J1, V3) (J1, V4),
// i.e. both J blocks against the same four block distances; on the diagonal V3 = J0 and V4 = J1.  A workgroup of four
// waves stages the union of its waves' row-blocks (<= kMfMaxRowBlocks) through LDS, 128 or 256 samples at a time.
constexpr int kMfBlock = 32;
constexpr int kMfMaxRowBlocks = 16;
constexpr int kMfWaves = 4;
constexpr int kMfMaxDmaPerWave = (2 * kMfMaxRowBlocks + kMfWaves - 1) / kMfWaves;  // (256-sample stages: two DMA instructions of 64 slots per row-block)
constexpr uint32_t kMfMaxFounders = 4000000;   // f32 accumulators stay integer-exact: complete rows accumulate sum g_i g_j <= 4 N (ldp_mfma_device.h)
constexpr uint32_t kMfGuMaxFounders = 1800000;  // ... and the four-product form's allele-count / missing-flag operands while 9 N < 2^24 (ldp_mfma_device.h)

struct MfmaWaveItem {
  int32_t jv;        // first variant of J0 (J1 = jv + 32); < 0: the wave has nothing to do
  int32_t vv;        // first variant of V0 (may lie before the subcontig: such rows hold no candidate pair)
  uint32_t jend;     // second variants >= jend belong to the next subcontig (its own blocks own their pairs)
  uint8_t slot[7];   // LDS row-block slot of J0, J1, V0..V4
  uint8_t prod_mask; // products (bit p; p < 4: (J0, V_p), p >= 4: (J1, V_{p-3})) that hold candidate pairs
};
struct MfmaWG {
  uint32_t rb[kMfMaxRowBlocks];  // first variant of each staged row-block
  uint32_t n_rb;
  uint32_t j_lo, j_hi;           // second variants the workgroup's waves own: [j_lo, j_hi)
  uint32_t pad;                  // bit 0: the subcontig also has a wide plan (MfmaTile), which owns it on complete-data launches;
                                 // bit 1: every wave item is diagonal (V3 = J0, V4 = J1): the single-form kernel's workgroup
  MfmaWaveItem w[kMfWaves];
};

// Wide bands (config 3: ~1,700 variants = 54 row-blocks per window): a workgroup of EIGHT waves owns a square tile of 8 second-variant
// blocks x 8 first-variant blocks, 64 block products for 16 staged row-blocks -- twice the products per staged byte of the
// parallelogram plan above (32 for 15), which is what a band this wide is short of (profiles/r02_c3shape_pmc_traffic.json: the
// narrow plan re-fetched every row ~12 times from HBM).  Wave w owns J blocks 2 (w & 3), + 1 and V blocks 4 (w >> 2) .. + 3.
// Tiles are aligned to the subcontig start in both directions, so on the diagonal the V tile IS the J tile (8 row-blocks staged).
constexpr int kWdWaves = 8;
constexpr int kWdTile = 8;                       // row-blocks per tile side
constexpr int kWdRowBlocks = 2 * kWdTile;        // staged row-blocks: slots 0..7 the J blocks, 8..15 the V blocks
constexpr uint32_t kWdMinReach = 12;             // a subcontig whose band reaches this many row-blocks takes the wide plan
struct MfmaTile {
  int32_t jv;        // first variant of J block 0 (J block a = jv + 32 a)
  int32_t vv;        // first variant of V block 0 (V block b = vv + 32 b); == jv on the diagonal
  uint32_t jend;     // second variants >= jend belong to the next subcontig
  uint32_t pad;
  uint64_t mask;     // bit 8 a + b: product (J_a, V_b) holds candidate pairs
};

struct PairKernelArgs {
  const uint32_t* planes;        // [variant][chunk][2][kChunkDwords]  (popcount kernels; nullptr when the code image is resident)
  const uint8_t* codes;          // [variant][code_row_bytes]          (matrix-pipe kernels; nullptr when the bit-planes are)
  uint64_t code_row_bytes;
  uint64_t row_dwords;           // dwords per variant row = chunks * kRowChunkDwords
  uint32_t chunks;
  uint32_t founder_ct;
  const ldp_variant_rec* recs;
  const uint32_t* lo;            // window start per variant
  const uint64_t* row_off;       // predicate row offset (u32 words) per variant
  uint32_t* pred;                // predicate bit rows
  const WorkItem* items;
  uint32_t n_items;
  uint32_t plane_base_variant;   // variant index of planes row 0
  double thresh;                 // r2 * (1 + 2^-44)
  ldp_pair_stats_t* stats;       // optional (parity tests)
  const uint64_t* pair_off;      // optional, with stats
  unsigned long long* counters;  // [0] = predicates true
  uint8_t* item_general;         // per work item: 1 = some row has missing calls -> general kernel
  // --r2-unphased matrix mode: when r2_out != nullptr the epilogue stores r^2 of pair (i<j) at
  // r2_out[(j - r2_row_first) * r2_ld + i] (float or double) instead of predicate bits
  const cp_slot* cp_stats;       // [variant][kCpStride]; nullptr disables early termination
  const cp_gen_slot* cp_gen;     // [variant][kCheckpoints]: the same for tiles with missing calls
  uint32_t checkpoint_chunk[kCheckpoints];  // ascending; a checkpoint fires after chunk (value - 1) is consumed
  uint32_t n_checkpoints;
  uint32_t lds_dwords;           // dynamic LDS of the launch (set by launch_pair_tiles)
  void* r2_out;
  uint64_t r2_ld;
  uint32_t r2_row_first;         // only second variants j in [r2_row_first, r2_row_end) are stored
  uint32_t r2_row_end;
  uint32_t r2_col_first;         // ... and only first variants i in [r2_col_first, r2_col_end) (a column block of the matrix);
  uint32_t r2_col_end;           //     dense rows then start at column r2_col_first: element (j - r2_row_first) * r2_ld + (i - r2_col_first)
  uint64_t r2_band_base;         // r2_ld == 0: band layout, element pair_off[j] - r2_band_base + (i - lo[j])
  uint32_t r2_float;             // 0: doubles, 1: floats, 2: ldp_pair_stats_t (the six integers of every pair instead of their r^2; dense layouts only)
  uint32_t r_signed;             // 0: r^2; 1: r = +-sqrt(r^2), sign of the covariance of the rows as stored (major-allele orientation);
                                 // 2: the same in REF orientation (the sign flips when exactly one row was inverted by prepare_kernel)
  // device-side filter (ldp_r2_unphased_hits): with r2_hits != nullptr a pair with |r^2| >= r2_min is appended at
  // slot atomicAdd(counters[3]) when that is below r2_hit_capacity, and nothing is stored to r2_out
  ldp_r2_hit* r2_hits;
  uint64_t r2_hit_capacity;
  double r2_min;
  // matrix-pipe tiles (launch_pair_mfma); `route` (one word, written by route_kernel from what prepare_kernel saw of the
  // rows converted so far) sends a whole launch to one kernel: kRouteComplete -> pair_mfma_kernel, kRouteSparse -> its
  // interval-epilogue instantiation (4.1d), kRouteGeneral -> pair_mfma_general_kernel; mf_active = 1: the popcount
  // kernels of the launch only run for non-zero (the general matrix-pipe kernel is off), 2: they never run
  const MfmaWG* mf_wgs;
  uint32_t n_mf_wgs;
  uint32_t mf_diag_ct;           // the first mf_diag_ct workgroups of mf_wgs are all-diagonal (MfmaWG::pad bit 1): pair_mfma_kernel's DIAGFORM instantiation
  uint32_t n_local;              // rows in `planes`
  uint32_t mf_active;            // the launch also carries matrix-pipe work items
  const uint32_t* route;         // kRoute*
  uint32_t wd_general;           // (set by the launcher) the tile plan's subcontigs are taken by pair_mfma_tile4_kernel: the missing-call kernel skips their workgroups
  uint32_t mf_four;              // prune launches on the six-product route may use its four-product form (EngineOptions::pair_four)
  uint32_t mf_gu;                // ... with allele counts g' and missing flags u as operands (ldp_mfma_device.h; founder_ct <= kMfGuMaxFounders)
  uint32_t sparse_ok;            // the route may be kRouteSparse (prune launches): launch that instantiation as well
  // wide-band tiles (launch_pair_wide): complete-data launches only; the workgroups of mf_wgs whose MfmaWG::pad is 1 cover the
  // same subcontigs for the other two routes and are skipped by pair_mfma_kernel<., false> when wd_active is set
  const MfmaTile* wd_tiles;      // in launch order (eight XCD streams of equal length, far tiles first; ldp_engine.cpp build_shard)
  uint32_t n_wd_tiles;
  uint32_t wd_active;
  // the same tiles in plain J order, for pair_mfma_tile4_kernel: its quarter tiles stop at their own checkpoints, workgroup by
  // workgroup, and the deferred order costs it 3 % (config 5's density: 115.1 against 118.3 ms); nullptr: wd_tiles serves both
  const MfmaTile* wd_tiles_plain;
  uint32_t n_wd_tiles_plain;
  uint32_t wd_async;             // the tiles run on pair_mfma_wide_async_kernel (no workgroup barrier in the stage loop; EngineOptions::wide_async)
  uint32_t wd_sparse;            // the tiles also own the launch on the kRouteSparse route (pair_mfma_wide_kernel<., SPARSE>; EngineOptions::wide_sparse):
                                 // pair_mfma_kernel<., SPARSE = true> then skips the workgroups of their subcontigs as the complete-data kernel does
  uint32_t wd_diag_split;        // complete-data prune launches: the tiles on the diagonal are left to pair_mfma_wide_kernel<., false, 3> (2 x 3 rectangles on all eight
                                 // waves), queued behind the 2 x 4 kernel (EngineOptions::wide_diag_kernel)
};

constexpr uint32_t kRouteComplete = 0, kRouteSparse = 1, kRouteGeneral = 2;

// The predicate rows [row_first, row_end) of a launch group as CSR (ldp_pred_csr.hip): meta[j] = (first entry, entries) of row j, an entry =
// (word index inside the row, the word's bits), ascending.  meta, ent and overflow are pinned HOST memory the kernel writes directly;
// counter (device) runs on from launch to launch of a run.
struct PredCsrArgs {
  const uint32_t* pred;
  const uint64_t* row_off;
  uint32_t row_first, row_end;
  uint2* meta;
  uint2* ent;
  unsigned long long* counter;
  uint64_t capacity;   // entries `ent` holds
  uint32_t* overflow;  // set to 1 when the run's non-zero words do not fit
};
hipError_t launch_pred_compact(const PredCsrArgs& a, hipStream_t stream);

// What the rows a launch reads miss (miss_stats_kernel over their records, when the launch is queued): the largest count in a
// row, and striped over kMissStripes words the total of missing calls and the number of rows beyond `miss_high` of them.
constexpr uint32_t kMissStripes = 64;
struct MissStats {
  uint32_t max_missing;
  uint32_t pad;
  unsigned long long total[kMissStripes];
  unsigned long long high_rows[kMissStripes];
};
// *route_out = kRouteComplete when no row had a missing call, kRouteSparse when the total is at most total_limit and the
// high rows at most high_limit (both 0: never), else kRouteGeneral
// adds the missing calls of recs[0, n) to *stats (block reductions, a handful of atomics)
hipError_t launch_miss_stats(const ldp_variant_rec* recs, uint32_t n, uint32_t founder_ct, uint32_t miss_high, MissStats* stats, hipStream_t stream);
hipError_t launch_route(const MissStats* stats, unsigned long long total_limit, unsigned long long high_limit, int allow_sparse, uint32_t* route_out,
                        hipStream_t stream);

struct PrepareArgs {
  const uint8_t* geno;           // row 0 = variant `first`
  uint64_t stride_bytes;
  uint32_t n_variants;
  uint32_t founder_ct;
  int encoding;
  uint32_t* planes;              // row 0 = variant `first`
  uint64_t row_dwords;
  uint32_t chunks;
  ldp_variant_rec* recs;         // entry 0 = variant `first`
  cp_slot* cp_stats;             // entry 0 = variant `first`, kCpStride each; may be nullptr
  cp_gen_slot* cp_gen;           // entry 0 = variant `first`, kCheckpoints each (written iff cp_stats is)
  double cp_tv_scale;            // sqrt(sqrt(thresh) * (1 - 1e-6))
  uint32_t checkpoint_chunk[kCheckpoints];
  uint32_t n_checkpoints;
  const uint8_t* row_inverse;    // optional, entry 0 = variant `first`: nonzero = this row is LDP_GENO_INVERSE whatever `encoding` says
  uint8_t* codes_out;            // code image rows (row 0 = variant `first`), code_row_bytes apart; launch_codes() only.  When it
  uint64_t code_row_bytes;       //   equals `geno` (and the strides agree) the rows are counted in place and only their padding is written
  MissStats* miss_stats;         // what the rows' missing calls add up to (may be nullptr)
  uint32_t miss_high;            // rows with more missing calls than this count as high rows
  const uint32_t* extra_het;     // per variant: het calls the sample map turned into missing ones (allele counts only); may be nullptr
  bool fix_cp_gen;               // redo cp_gen for rows with missing calls (cp_gen_fix_kernel)
  uint8_t* stored_inv;           // launch_codes() only, entry 0 = variant `first`, may be nullptr: 1 = the image row is stored inverted relative to the input (read for rows
                                 //   counted in place, written for every row)
  int orient;                    // launch_codes(): store ALT-major rows inverted (major-allele-oriented image)
};

hipError_t launch_prepare(const PrepareArgs& a, hipStream_t stream);
// the same for the code image: rows of any accepted encoding -> REF- (or INVERSE-) coded 2-bit rows + the per-variant records and
// checkpoint statistics (in the image's own orientation); no bit-planes
hipError_t launch_codes(const PrepareArgs& a, hipStream_t stream);
// arbitrary pairs from the code image (the integers in major-allele orientation, like launch_pair_stats_ref)
// image rows [0, n) with stored_inv[v] != 0 back to the input's orientation, flags cleared (ldp_map_rows: the caller is about to see / rewrite them)
hipError_t launch_unflip_rows(uint8_t* codes, uint64_t code_row_bytes, uint8_t* stored_inv, uint32_t n, hipStream_t stream);
hipError_t launch_pair_stats_ref_codes(const uint8_t* codes, uint64_t code_row_bytes, const ldp_variant_rec* recs, const uint32_t* first, const uint32_t* second,
                                       uint32_t n_pairs, ldp_pair_stats_t* out, hipStream_t stream);
constexpr double kSmallEpsilon = 0.00000000000005684341886080801486968994140625;  // 2^-44 (plink2_float.h:119)

// ---- .pgen records decoded on the device (ldp_pgen_decode.hip) ----------------------------------------------------
constexpr uint32_t kPgenBaseCarried = 0xfffffffeu;  // PgenRecDesc::base: the row the engine kept from the previous launch
constexpr uint32_t kPgenNoBase = 0xffffffffu;
constexpr uint64_t kPgenLdsRowBytes = 128 * 1024;   // rows up to this size (524,288 samples) are assembled in LDS
struct PgenRecDesc {
  uint64_t off;        // first byte of the record in the launch's byte buffer
  uint32_t len;        // bytes
  uint32_t base;       // LD-compressed records (types 2, 3): row of this launch that holds their base, or kPgenBaseCarried
  uint32_t allele_ct;  // from the .pvar (2 = one ALT allele)
  uint32_t vrtype;     // the file's variant record type byte
};
struct PgenDecodeArgs {
  const uint8_t* bytes;         // the records' bytes
  const PgenRecDesc* recs;
  uint32_t n;                   // records = rows of this launch
  uint32_t sample_ct;           // samples of the FILE
  uint8_t* rows;                // out: 2-bit rows, ceil(sample_ct / 4) bytes used, the rest of the stride zero
  uint64_t stride;              // multiple of 16
  const uint8_t* carried_base;  // the most recent non-LD row before this launch, or nullptr
  uint64_t* main_end;           // out, per record: offset in `bytes` of the first byte behind the main track
  int* error;                   // out: 0, or 1 + the index of a malformed record
  int pass;                     // (set by the launcher)
  int any_ld;                   // some record is LD-compressed
  // variants with more than one ALT allele: their records, and what the collapse decides
  const uint32_t* multi_rec;    // record indices
  uint32_t n_multi;
  double* maj_freq;             // out, per entry of multi_rec: GetAlleleFreq of the major allele
  uint32_t* maj_idx;            // out: the major allele
  uint8_t* row_inverse;         // out, per RECORD: 1 = the row now counts copies of non-major alleles (LDP_GENO_INVERSE)
  const uint32_t* sample_mask;  // optional bitmap over the file's samples: the ones whose alleles count (a subset sample map: the founders)
  uint32_t mask_ct;             // ... and how many they are
  // --indep-pairphase: the hardcall-phase track decoded into the rows' second part (LDP_GENO_PHASED layout); 0 = not wanted
  uint64_t phase_off;           // byte offset of the phase bits inside a row (a multiple of 4, < stride)
  uint32_t* unphased;           // out: lowest record index with a het call that has no phase (atomicMin; preset to UINT32_MAX)
  int no_lds;                   // (test hook, option "decode_no_lds") assemble rows in global memory whatever their size
};
hipError_t launch_pgen_main(const PgenDecodeArgs& a, hipStream_t stream);
// the phase tracks of records [0, n_records) on top of their decoded main tracks (after launch_pgen_main on the same stream)
hipError_t launch_pgen_phase(const PgenDecodeArgs& a, uint32_t n_records, hipStream_t stream);
hipError_t launch_pgen_aux1(const PgenDecodeArgs& a, hipStream_t stream);

// Sample-mapped rows (ldp_set_sample_map): out row v = 2-bit REF codes of columns map[f] & 0x7fffffff of in row v, hets of
// columns with bit 31 set replaced by missing and counted into extra_het[v]; in rows are .pgen- or .bed-coded.
hipError_t launch_gather_rows(const uint8_t* in, uint64_t in_stride, uint32_t n_variants, int in_is_bed, const uint32_t* map, uint32_t out_ct, uint8_t* out,
                              uint64_t out_stride, uint32_t* extra_het, hipStream_t stream);
// ev[0..3] (optional): recorded before/after the complete-data kernel and before/after the general kernel
hipError_t launch_pair_tiles(const PairKernelArgs& a, uint32_t max_rows, hipStream_t stream, hipEvent_t* ev);
hipError_t launch_pair_stats_ref(const uint32_t* planes, uint64_t row_dwords, uint32_t chunks, uint32_t plane_base_variant,
                                 const uint32_t* first, const uint32_t* second, uint32_t n_pairs,
                                 ldp_pair_stats_t* out, hipStream_t stream);
size_t pair_tiles_lds_bytes(uint32_t max_rows);
// ev[0..2] (optional): recorded before the complete-data kernel, between it and the missing-calls kernel, and after
hipError_t launch_pair_mfma(const PairKernelArgs& a, hipStream_t stream, hipEvent_t* ev);
// chrX pairs of the r^2 outputs (ldp_kernels.hip: x_weighted_kernel): the two engines' six integers of every pair of a dense block
// (row q = second variant row_first + q, column c = first variant col_first + c, c < cols) -> the male-weighted r^2 (or r)
struct XWeightedArgs {
  const ldp_pair_stats_t* all;   // [rows][cols], all founders
  const ldp_pair_stats_t* male;  // the same over the male founders; nullptr: none
  uint32_t rows, cols;
  uint32_t row_first, col_first;
  const uint8_t* is_x;           // per engine row (variant): on chrX
  const uint8_t* flip_all;       // per engine row: the engine's orientation differs from the target's; nullptr: never
  const uint8_t* flip_male;
  uint32_t unsquared;            // r instead of r^2
  uint32_t as_float;
  void* out;                     // dense: element q * out_ld + c, written only where the pair involves chrX (c + col_first < q + row_first)
  uint64_t out_ld;
  ldp_r2_hit* hits;              // != nullptr: pairs with |value| >= min_r2 appended at atomicAdd(hit_count) while below hit_capacity; `out` unused
  uint64_t hit_capacity;
  unsigned long long* hit_count;
  double min_r2;
};
hipError_t launch_x_weighted(const XWeightedArgs& a, hipStream_t stream);
// the wide-band tiles of the launch (complete-data route); queued between ev[0] and ev[1] of launch_pair_mfma by the caller's order
// (sparse: the SPARSE instantiation -- the same tiles on the route of launches whose rows have a few missing calls)
hipError_t launch_pair_wide(const PairKernelArgs& a, hipStream_t stream, bool sparse = false);
uint32_t pair_mfma_ksteps(uint32_t founder_ct);  // 64-sample k-steps per row (the unit of counters[2])

// LDS rows of a work item that stages `units` 8-distance units starting at distance d0 (make_geom in the kernel)
inline uint32_t tile_rows(uint32_t d0, uint32_t units) {
  const uint32_t dmax = d0 + 8 * units - 1;
  const uint32_t n_irows = 8 * units + 31;
  return ((dmax < n_irows) ? dmax : n_irows) + kTileJ;
}

}  // namespace ldp
#endif
