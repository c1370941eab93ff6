/* ldprune_hip_debug.h -- test hooks, kernel-selection switches and the benchmark generator of libldprune_hip.so.
 *
 * Nothing here belongs to the drop-in boundary (include/ldprune_hip.h): a host that prunes never calls any of it.  The entry points
 * exist for the parity tests (host-only replay and plan views that run without a GPU), for measurements (switching kernels on one
 * engine) and for bench.py (a deterministic genotype generator, so that the 1.25 TB matrix of the metric never has to exist on disk).
 */
#ifndef LDPRUNE_HIP_DEBUG_H
#define LDPRUNE_HIP_DEBUG_H

#include "ldprune_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-only replay of the greedy scan (plink2_ld.cc:931-1100) from a caller-supplied list of the candidate
 * pairs whose predicate is TRUE (global variant indices, first < second, each inside the band).  Needs
 * variant records (ldp_debug_set_variant_recs, for the monomorphic flags) and maj_freqs.  No GPU is
 * touched: this is how the host logic is tested on a CPU-only machine. */
int ldp_debug_set_variant_recs(ldp_engine* e, const ldp_variant_rec* recs);
int ldp_debug_replay_pairs(ldp_engine* e, uint64_t n_true, const uint32_t* first, const uint32_t* second, uint64_t* removed);
/* Kernel-selection switches of ONE engine, for tests and measurements (the defaults are what production runs use).  The shipped
 * library reads NO environment variable (csrc/ldp_env.h): this call is the only way to switch anything, engine by engine; only the
 * measurement build (-DLDP_MEASURE, lib/libldprune_hip_measure.so, tools/) presets them from LDP_* variables.  name:
 *   "early_exit"      0/1: checkpoints that drop provably sub-threshold products
 *   "pair_mfma"       0/1: matrix-pipe kernels on the 2-bit code image; 0 = the popcount kernels on bit-planes (before ldp_set_variants*())
 *   "pair_sparse"     0/1: the interval epilogue for rows with a few missing calls
 *   "sparse_frac"     mean missing fraction up to which a launch takes it
 *   "pair_four"       0/1: prune launches over rows with more missing calls than that multiply four products per pair and take the
 *                     two sums of squares from per-variant intervals (exact count for the few pairs they leave open); 0 = all six
 *   "pair_gu"         0/1: the four-product form's operands are allele counts and missing flags (default 1; exact up to 1,800,000
 *                     founders, engines with more use 0 by themselves); 0 = the +-2 coded x and the call flags n of rounds 2-3
 *   "pair_four_tiles" 0/1: ... and in wide bands (subcontigs with the tile plan) that form runs over quarter tiles instead of the
 *                     parallelogram plan (default 1)
 *   "orient_rows"     0/1: the count pass stores a row whose ALT allele is the major one with codes 00 <-> 10 swapped, so that every row of the image is
 *                     major-allele-oriented (default 1: hom-major = operand 0 is the cheapest genotype for the power-capped pair kernels; the record
 *                     says so in flags bit 3); 0 = rows stay as the input had them (rounds 2-5)
 *   "pred_csr"        0/1: prune runs return the predicate rows as their non-zero words, compacted on the device and written straight into pinned host
 *                     memory (default 1); 0 = the dense rows are copied back whole, as in rounds 1-5
 *   "csr_capacity"    k: (test hook, before ldp_set_variants()) the compacted rows' buffer holds k entries; a run with more non-zero words falls back
 *                     to the dense rows (0 = a quarter of all predicate words)
 *   "wide_async"      0/1: the 8 x 8 tiles on the barrier-free experiment of round 5 (pair_mfma_wide_async_kernel; default 0).  For measurements
 *                     and tests ONLY: a wave whose LDS poll does not come true within ~2^22 polls (a fraction of a second) traps, which ends the
 *                     HIP context of the whole process with no LDP_ERR code -- timing alone can do that under a debugger, a single-stepping
 *                     profiler or an oversubscribed device
 *   "wide_sparse"     0/1: launches whose rows have a few missing calls keep the 8 x 8 tiles of wide-band subcontigs (the tile kernel's
 *                     SPARSE instantiation; default 1); 0 = they fall back to the parallelogram plan as in rounds 2-5
 *   "wide_diag_kernel" 0/1: complete-data prune launches run the tiles ON the diagonal (36 live products, to the end of the rows) in eight 2 x 3
 *                     rectangles, a second body of pair_mfma_wide_kernel picked per workgroup (default 1); 0 = 2 x 4 rectangles for every tile
 *   "wide_diag_last"  k: within a launch every XCD's stream of 8 x 8 tiles runs its far tiles first and the tiles fewer than k tile
 *                     distances from the diagonal (the long ones: they hold the pairs in LD) at the end; 0 = plain J order (default 2;
 *                     before ldp_set_variants())
 *   "wide_min_reach"  row-blocks a subcontig's band must reach to take the 8 x 8 tile plan of the wide-band kernel; 0 = always,
 *                     a huge value = never (before ldp_set_variants())
 *   "replay_steps"    k: ldp_debug_replay_pairs() walks every subcontig in k instalments, the way the streaming replay of a run advances
 *   "decode_rows"     k: ldp_load_pgen_records*() decode in launches of k rows (LD chains cut everywhere)
 *   "decode_no_lds"   0/1: decoded rows are assembled in global memory (what rows beyond 128 KiB take) instead of LDS
 *   "x_rows"          k: ldp_r2_unphased_block_x*() work in chunks of k rows
 * Results never depend on these.  Unknown name: LDP_ERR_INVALID. */
int ldp_debug_set_option(ldp_engine* e, const char* name, double value);
/* The .pgen reader's phase / subset routines use pext / pdep where the host has BMI2; on != 0 forces the portable loops for the whole
 * process (a test hook: both give the same bytes). */
int ldp_pgen_debug_force_portable(int on);
/* Host-only view of the matrix-pipe work plan (csrc/ldp_device.h: MfmaWG) in the engine's shard-local variant
 * indices, for the CPU test that every candidate pair is owned by exactly one 32 x 32 block product.  Per workgroup
 * 63 words: n_rb (bit 31: see ldp_debug_wide_plan; bit 30: every wave item is diagonal), j_lo, j_hi, rb[16], then per wave jv, vv, jend, prod_mask, slot[7].  lo_local (optional, *local_ct
 * entries) receives the window starts in the same index space.  words == NULL only counts. */
int ldp_debug_mfma_plan(const ldp_engine* e, uint32_t* wg_count, uint32_t* words, uint64_t capacity_words, uint32_t* lo_local, uint32_t* local_ct);

/* The wide-band plan (csrc/ldp_device.h: MfmaTile; subcontigs whose band reaches the "wide_min_reach" option in row-blocks): per
 * tile 5 words: jv, vv, jend, mask bits 0-31, mask bits 32-63 (bit 8 a + b: the product of J block a and V block b).  The
 * workgroups of ldp_debug_mfma_plan() that belong to such subcontigs carry bit 31 in their first word; complete-data launches
 * leave those to the tiles.  words == NULL only counts. */
int ldp_debug_wide_plan(const ldp_engine* e, uint32_t* tile_count, uint32_t* words, uint64_t capacity_words);


/* ---- synthetic workload (benchmark / test support, not part of the reference seam) ---- */
/* Deterministic genotype generator for the SURVEY.md 8(d) workload: rows [first_variant, +n_variants) of
 * REF-based codes (LDP_GENO_REF) written to `out` (host or device memory), each genotype a pure function of
 * (seed, variant index, sample index).  LD is planted like the reference's --dummy
 * (plink2_import.cc:16387-16432).  `stream` is a hipStream_t (device output only; may be NULL). */
int ldp_synth_genotypes(uint64_t seed, uint64_t first_variant, uint32_t n_variants, uint32_t founder_ct, double missing_rate,
                        void* out, uint64_t stride_bytes, int location, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDPRUNE_HIP_DEBUG_H */
