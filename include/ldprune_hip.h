/* ldprune_hip.h -- C-ABI of the MI355X-native pairwise-LD pruning engine.
 *
 * Drop-in boundary for PLINK 2.0's --indep-pairwise hot path.  The reference has no FFI for this
 * path; the seam replaced here is the hand-off inside LdPrune() (2.0/plink2_ld.cc:2530) between
 * "everything upstream is file decoding" and the worker pool:
 *
 *     LdPruneSubcontigSplitAll()  2.0/plink2_ld.cc:2165   -> ldp_set_variants()
 *     LoadBalance()               2.0/plink2_ld.cc:2341   -> ldp_set_shard()        (subcontig -> GPU)
 *     IndepPairwise() decode loop 2.0/plink2_ld.cc:1345   -> ldp_load_genotypes()   (PgrGetInv1 output,
 *                                                            or raw .pgen/.bed codes + device-side
 *                                                            allele counts: plink2_data.cc:2304,
 *                                                            plink2_filter.cc:2113,3311)
 *     IndepPairwiseThread()       2.0/plink2_ld.cc:801    -> ldp_run()              (HIP kernels + replay)
 *     removed_variants_collapsed  2.0/plink2_ld.cc:2555   <- ldp_run() output bitmap
 *
 * Conventions follow the reference's own GPU seam (2.0/cuda/plink2_matrix_cuda.h:24-104): plain C
 * types, int return codes (0 = success), caller owns every buffer it passes in, the engine owns its
 * device memory, no exceptions cross the boundary.  A nonzero code maps to kPglRetGpuFail /
 * kPglRetNomem (2.0/include/plink2_base.h:351-395) on the reference side; see INTEGRATION.md.
 *
 * There is NO CPU fallback behind this API: without a usable HIP device every compute entry point
 * returns LDP_ERR_GPU.
 */
#ifndef LDPRUNE_HIP_H
#define LDPRUNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ldp_engine ldp_engine;

enum {
  LDP_OK = 0,
  LDP_ERR_INVALID = 1,     /* bad argument / inconsistent input  (kPglRetInconsistentInput) */
  LDP_ERR_NOMEM = 2,       /* host or device allocation failed   (kPglRetNomem)             */
  LDP_ERR_GPU = 3,         /* HIP runtime / kernel failure       (kPglRetGpuFail)           */
  LDP_ERR_STATE = 4,       /* call sequence error                (kPglRetImproperFunctionCall) */
  LDP_ERR_UNSUPPORTED = 5, /* e.g. >= 2^30 founders              (kPglRetNotYetSupported)   */
  LDP_ERR_UNPHASED = 6     /* --indep-pairphase: a het call without phase (kPglRetInconsistentInput, plink2_ld.cc:2045-2049) */
};

/* Genotype encodings accepted by ldp_load_genotypes(); all are 2 bits per sample, sample-minor,
 * little-endian within bytes, ceil(founder_ct/4) meaningful bytes per variant row. */
enum {
  LDP_GENO_INVERSE = 0, /* PgrGetInv1 output (pgenlib_read.cc:5544): 0 hom-major, 1 het, 2 hom-nonmajor,
                           3 missing.  Caller supplies maj_freqs via ldp_set_maj_freqs(). */
  LDP_GENO_REF = 1,     /* .pgen main-track coding: 0 hom-REF, 1 het, 2 hom-ALT, 3 missing.  The engine
                           counts alleles, picks the major allele and inverts on the device. */
  LDP_GENO_BED = 2,     /* PLINK 1 .bed coding: 0 hom-A1(ALT), 1 missing, 2 het, 3 hom-A2(REF)
                           (pgenlib_read.cc:2157 PgrPlink1ToPlink2InplaceUnsafe), then as LDP_GENO_REF. */
  /* --indep-pairphase (plink2_ld.cc:1449-2163): OR this into LDP_GENO_INVERSE or LDP_GENO_REF.  The engine's
   * founder_ct is then the HAPLOTYPE count (2 x samples); a row holds, for S = founder_ct/2 samples, ceil(S/4)
   * bytes of 2-bit codes, zero padding to a multiple of 4 bytes, then ceil(S/8) bytes of phaseinfo bits (sample s =
   * bit s%8 of byte s/8; set = the counted allele of a het call sits on the first haplotype, as PgrGetInv1P /
   * HapsplitMustPhased define it, pgenlib_read.cc:7016, pgenlib_misc.cc:1887; ignored for non-het calls) --
   * ldp_phased_row_bytes() bytes in all.  Every het call must be phased (the caller checks, as the reference does at
   * plink2_ld.cc:2045).  The conversion kernel splits each sample into its two haplotypes (hom -> both, het -> the
   * phased one, missing -> both missing); haplotype h in {0,1} is carried as the genotype code 2h, whose
   * pair statistics are exactly 4 x the reference's cov12 / variance1 / variance2 (ldp_kernels.hip), so the prune
   * decision is the same bit. */
  LDP_GENO_PHASED = 4,
  /* OR this into LDP_GENO_REF, LDP_GENO_BED or LDP_GENO_INVERSE: the rows hold the samples of the FILE (raw_sample_ct
     of ldp_set_sample_map()) and the engine gathers its founder_ct columns from them on the device.  With
     LDP_GENO_INVERSE the caller has already counted against the major allele and supplies maj_freq as usual. */
  LDP_GENO_MAPPED = 8
};
/* bytes of one LDP_GENO_PHASED row / offset of its phaseinfo bits, for hap_ct haplotypes (= the engine's founder_ct) */
uint64_t ldp_phased_row_bytes(uint32_t hap_ct);
uint64_t ldp_phased_phase_offset(uint32_t hap_ct);

enum { LDP_MEM_HOST = 0, LDP_MEM_DEVICE = 1 };

/* Mirrors the fields of LdInfo (2.0/plink2_ld.h:113-120) that --indep-pairwise consumes. */
typedef struct {
  uint32_t founder_ct;         /* samples per variant row; 2 <= founder_ct < 2^30 (plink2_ld.cc:1122,2537) */
  uint32_t prune_window_size;  /* bp when window_is_bp (already kb*1000*(1+2^-44), plink2.cc:7266), else variants */
  uint32_t prune_window_incr;  /* step; must be 1 when window_is_bp (plink2.cc:7290) */
  uint32_t window_is_bp;       /* kfLdPruneWindowBp */
  uint32_t plink1_order;       /* kfLdPrunePlink1Order (--indep-order 1) */
  double prune_last_param;     /* raw r^2 in [0,1); the engine applies *(1+2^-44) as plink2_ld.cc:1255 does */
  int32_t device;              /* HIP device ordinal; -1 = current device */
  void* stream;                /* hipStream_t to run on; NULL = engine creates its own */
} ldp_params;

/* Integer 6-tuple of ComputeIndepPairwiseR2Components (plink2_ld.cc:699-723); 1 = first (lower
 * index), 2 = second.  cov12 = dot*nm - sum1*sum2, var_k = ssq_k*nm - sum_k^2 (:1085-1087). */
typedef struct {
  uint32_t nm;
  int32_t sum1;
  uint32_t ssq1;
  int32_t sum2;
  uint32_t ssq2;
  int32_t dot;
} ldp_pair_stats_t;

/* Per-variant aggregates (VariantAggs, plink2_ld.cc:691-695) + what the allele-count pass produced. */
typedef struct {
  uint32_t nm_ct;
  int32_t sum;
  uint32_t ssq;
  uint32_t flags;        /* bit0: ALT is the major allele (the reference inverts such a row), bit1: monomorphic (:902), bit2: has missing calls,
                            bit3 (internal): the engine's resident row is stored inverted relative to the input (major-allele-oriented image) */
  uint32_t n_homref;     /* raw counts before inversion (all zero for LDP_GENO_INVERSE input) */
  uint32_t n_het;
  uint32_t n_homalt;
  uint32_t reserved;
} ldp_variant_rec;

typedef struct {
  uint64_t candidate_pairs;  /* in-window pairs the pair kernel was asked to decide */
  uint64_t computed_pairs;   /* pair slots the kernel actually evaluated (tile padding included) */
  uint64_t replay_pairs;     /* predicate bits the greedy replay consumed */
  uint64_t pred_true;        /* candidate pairs above threshold */
  double ms_prepare;         /* device time of the count pass (codes_kernel; prepare_kernel on bit-plane engines) in the last ldp_load_genotypes() (HIP events) */
  double ms_pair_kernel;     /* device time of all pair-kernel launches in the last ldp_run() (HIP events on the engine stream) */
  double ms_pair_fast;       /* ... of the popcount fallback pair_tiles_kernel<false> (bit-plane engines: > 4,000,000 founders or pair_mfma off) alone */
  double ms_pair_general;    /* ... of pair_tiles_kernel<true> (the same fallback, tiles with missing calls) alone */
  double ms_replay;          /* host wall time of the replay in the last ldp_run() */
  double ms_run_total;       /* host wall time of the last ldp_run() */
  uint32_t pair_kernel_launches;
  uint32_t subcontig_ct;
  uint32_t owned_subcontig_ct;
  uint32_t window_max;       /* as LdPruneSubcontigSplitAll reports it */
  uint64_t tile_unit_chunks;       /* popcount fallback only: its work in the last run in (8-distance unit x k-chunk) steps ... */
  uint64_t early_exit_unit_chunks; /* ... and how many of them early termination skipped (provably sub-threshold tiles) */
  double ms_pair_mfma;             /* device time of the complete-data matrix-pipe kernels (pair_mfma_kernel, pair_mfma_wide_kernel) in the last run */
  uint64_t mfma_block_products;    /* 32 x 32 block products of the matrix-pipe plan (0 when that path is off) */
  uint64_t mfma_product_stages;    /* ... times the 64-sample k-steps of a row: the MFMA instructions of an exhaustive run */
  uint64_t mfma_skipped_product_stages; /* ... and how much of it early termination skipped in the last run */
  double ms_pair_mfma_general;     /* device time of the missing-call matrix-pipe kernels (pair_mfma_general_kernel, pair_mfma_tile4_kernel) */
  uint64_t sparse_exact_pairs;     /* few missing calls: pairs the interval test left open and the kernel resolved exactly */
  /* Which matrix-pipe kernel the device-side route gave the pair launches of the last run (one word per launch group, written by
   * route_kernel from the rows' missing-call totals): complete data -> pair_mfma_kernel, a few missing calls -> its interval
   * epilogue, otherwise the six-product kernel.  All zero when the popcount kernels own the run. */
  uint32_t route_complete_launches;
  uint32_t route_sparse_launches;
  uint32_t route_general_launches;
  uint32_t wide_tiles;             /* 8 x 8 block tiles of the plan (wide-band subcontigs, complete-data launches) */
  uint64_t mfma_extra_product_stages; /* product x k-step units the wide-band kernel computed beyond the plan (its waves run all eight
                                         products of their rectangle or none): executed MFMA = stages - skipped + extra */
  uint32_t four_tile_launches;     /* launches of the last run whose tiles went to pair_mfma_tile4_kernel (wide bands, rows with missing
                                      calls, four-product form) */
  uint32_t decoded_in_place_rows; /* variant records ldp_load_pgen_records() decoded straight into the image (no scratch row, no copy) since ldp_create() */
  uint32_t sparse_tile_launches;  /* launches of the last run whose 8 x 8 tiles ran on the route of rows with a FEW missing calls
                                     (pair_mfma_wide_kernel's SPARSE instantiation: exact dot product, interval epilogue) */
  uint32_t reserved0;
} ldp_counters;

/* ---- lifecycle ---- */
int ldp_create(const ldp_params* params, ldp_engine** out);
void ldp_destroy(ldp_engine* e);
const char* ldp_last_error(const ldp_engine* e);
/* number of usable HIP devices (0 when there is none); never fails */
int ldp_device_count(void);
/* Optional: create the device's context now (first stream, first pinned allocation, first copy), from any thread, so that the first
 * ldp_load_genotypes() of an engine on that device does not pay for it (tens to hundreds of milliseconds).  LDP_ERR_GPU without
 * such a device. */
int ldp_prewarm(int device);
/* NUMA node of the host the device is attached to (its PCI function's numa_node in sysfs), or -1 when the host does not say.  The
 * library never moves its caller's threads; a caller on a multi-socket host that loads from host memory gains from running its loading
 * threads there -- and from first-touching its buffers there -- (30 -> 39 GB/s file -> HBM on the round-5 boxes): plink2-hip does. */
int ldp_device_numa_node(int device);
/* Host copy threads.  The file -> pinned-memory copies of ldp_load_genotypes*() run on a pool of copy threads that all engines of a
 * process share (created by the first load, with that thread's CPU affinity); engines that load at the same time take turns.  A host
 * that feeds SEVERAL engines concurrently -- one feeding thread per engine, each bound next to its engine's device, the way the
 * reference's main thread fills every worker's slot of a batch (plink2_ld.cc:1292-1417) -- calls this once per engine FROM the thread
 * that will feed it: the engine gets copy threads of its own, created here and inheriting the calling thread's affinity, and its
 * pinned staging ring (allocated by its first load) is first touched by them.  Idempotent.  plink2-hip --gpus N does this. */
int ldp_use_private_copy_threads(ldp_engine* e);
/* Largest founder_ct whose pair statistics run on the matrix pipe (FP4 operands, f32 accumulators that hold the integers
 * exactly); larger jobs run on the popcount kernels.  Same results either way. */
uint32_t ldp_matrix_pipe_max_founders(void);

/* ---- planning (host only; usable without a GPU) ---- */
/* variant_ct included variants in file order with chr0/unplaced already stripped (StripUnplacedK,
 * plink2_ld.cc:2542); chr_idx[v] = chromosome order index (nondecreasing); bps[v] = position (may be
 * NULL for count-based windows).  Runs the subcontig split and the window iterator
 * (LdPruneNextSubcontig/LdPruneNextWindow, plink2_ld.cc:605-689) to fix the candidate-pair band. */
int ldp_set_variants(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps);
/* subcontig table: info[2*k] = length, info[2*k+1] = first variant index; returns count via *ct */
int ldp_get_subcontigs(const ldp_engine* e, uint32_t* ct, uint32_t* info, uint32_t info_capacity_pairs);
/* Restrict this engine to the subcontigs LPT assigns to `rank` of `world` (weights = lengths,
 * cf. plink2_ld.cc:2686-2694).  Only owned variants need genotype rows; ldp_run() reports bits for
 * owned variants only.  owner[k] (optional, subcontig_ct entries) receives the rank of subcontig k. */
int ldp_set_shard(ldp_engine* e, uint32_t rank, uint32_t world, uint32_t* owner);
/* The one exchange step of a multi-GPU prune, for a C/C++ host (one process or one process per GPU): every rank contributes the
 * removed bits of ITS variants, packed in shard order, and ONE ncclAllGather over device buffers (RCCL over xGMI; segments padded
 * to the longest: the allgatherv) gives every rank every segment; they are then stitched into global variant order -- what
 * IndepPairwise does with its threads' bit ranges (plink2_ld.cc:1418-1426).  comm: an ncclComm_t whose size and rank equal
 * ldp_set_shard()'s world and rank.  removed_local: the bitmap ldp_run() produced on this rank; removed_global (out):
 * (variant_ct + 63) / 64 words, identical on every rank.  RCCL is bound at run time (librccl.so.1): LDP_ERR_UNSUPPORTED where it
 * is not installed.  ldp_comm_init_all() / ldp_comm_destroy() wrap ncclCommInitAll / ncclCommDestroy for hosts that drive
 * several devices from one process (plink2-hip --gpus N) and do not want RCCL's header.
 * A rank that cannot enter the collective (bad arguments, no device, allocation failure) calls ncclCommAbort on its communicator
 * before it returns the error, so that its peers come back from the all-gather with an error instead of waiting for it forever;
 * after a nonzero return the communicator is gone: do not use it again (ldp_comm_destroy() on it is a harmless no-op).  A host whose rank failed BEFORE the exchange
 * (ldp_run() returned an error) should not enter it at all -- abort or simply drop the communicators, as plink2-hip does. */
int ldp_allgather_removed(ldp_engine* e, void* nccl_comm, const uint64_t* removed_local, uint64_t* removed_global);
int ldp_comm_init_all(int n, const int* devices, void** comms);
void ldp_comm_destroy(void* comm);
/* The same exchange in its three steps, for hosts with another transport (MPI, a host-side copy between engines of one process --
 * plink2-hip where RCCL is absent or refuses the device set): ldp_shard_segment_words() = 64-bit words of one padded segment (the
 * longest shard's bits, at least one word; the same on every rank); ldp_pack_removed_segment() = this rank's removed bits in shard
 * order (its owned subcontigs in file order), `segment` holding that many words; ldp_stitch_removed_segments() = world x words
 * words, rank-major, back into global variant order (removed_global: (variant_ct + 63) / 64 words).  Host-only, no GPU touched. */
int ldp_shard_segment_words(const ldp_engine* e, uint64_t* words);
int ldp_pack_removed_segment(const ldp_engine* e, const uint64_t* removed_local, uint64_t* segment);
int ldp_stitch_removed_segments(const ldp_engine* e, const uint64_t* segments, uint64_t* removed_global);
/* per-variant window start lo[v] (first candidate partner index) and candidate pair total */
int ldp_get_band(const ldp_engine* e, uint32_t* lo, uint64_t* candidate_pairs);

/* ---- data ---- */
/* Rows [first_variant, first_variant+n) of the variant table.  `geno` points at row first_variant;
 * rows are stride_bytes apart.  location: LDP_MEM_HOST or LDP_MEM_DEVICE.  Rows outside this
 * engine's shard are ignored.  The engine keeps them as 2-bit REF- / INVERSE-coded rows resident in HBM (bit-planes only with more
 * founders than ldp_matrix_pipe_max_founders()), computes the per-variant aggregates (FillVaggs, plink2_ld.cc:725) and, for
 * REF/BED encodings, the allele counts and the major allele (rows are not re-oriented: the records carry the flag).  Host buffers may be reused as soon as the call returns (rows travel through a
 * pinned staging ring).  Device buffers are read asynchronously on the engine's stream: the data must be
 * complete before the call (synchronise the producing stream) and must stay valid until the next ldp_run() /
 * ldp_get_* call returns.
 * Load in variant order when you can: with host buffers the engine starts the pair kernel for a group of variants
 * as soon as everything the group needs has been converted, so the pair work overlaps the remaining transfers.
 * Loading a variant again (a new pass over the data) simply starts over. */
int ldp_load_genotypes(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* geno, uint64_t stride_bytes,
                       int location, int encoding);
/* The same for rows that lie in a FILE as fixed-width records (a .bed, a fixed-width .pgen: the main-thread read loop of
 * plink2_ld.cc:1345-1390 for those formats): row of variant first_variant + k at file_offset + k * stride_bytes of the open
 * descriptor fd.  The rows are pread() straight into the pinned staging ring by the engine's copy threads -- the page cache is
 * copied out in large runs, where a memcpy out of a mapping of the file first pays a minor page fault per 4 KiB -- and cross PCIe
 * from there.  The descriptor is only read (pread: its file position is untouched) and may be closed once the call returns. */
int ldp_load_genotypes_fd(ldp_engine* e, uint32_t first_variant, uint32_t n, int fd, uint64_t file_offset, uint64_t stride_bytes, int encoding);
/* Zero-copy loading.  The engine keeps the genotypes as one resident image of 2-bit rows -- the same codes, the same N/4 bytes
 * per variant as the input -- and a producer that runs on the device (a decoder, a generator) can write its rows straight into
 * it: *device_rows receives the device address of the image row of first_variant, *stride_bytes the distance between rows
 * (ceil(founder_ct / 4) rounded up to a multiple of 64: write the ceil(founder_ct / 4) meaningful bytes of each row, the engine
 * fills the padding).  Then call ldp_load_genotypes() with exactly this pointer, stride and range, LDP_MEM_DEVICE and
 * LDP_GENO_REF or LDP_GENO_INVERSE: the rows are counted where they are (one read of the data, no copy, no conversion pass).
 * This is the role of the raw_tgenovecs buffers the reference's main thread decodes into for its workers (plink2_ld.cc:1206,
 * 1357).  The variants must be consecutive owned rows (one subcontig run at a time).  Rows loaded from any other buffer are
 * copied into the image as before.  LDP_ERR_UNSUPPORTED when the engine keeps bit-planes instead (more founders than
 * ldp_matrix_pipe_max_founders()): load from your own buffer then.
 * The image belongs to the engine: the load that counts a row also decides its major allele, and it keeps a row whose ALT allele is the
 * major one INVERTED (codes 00 <-> 10 swapped: what GenovecInvertUnsafe does to the reference's rows, pgenlib_misc.cc:1090).  Loading the same
 * mapped rows again without rewriting them is fine (the engine knows which rows it inverted); a producer that wants to REWRITE rows it has
 * already loaded calls ldp_map_rows() on them again first -- the call puts them back into the input's orientation and returns when that is
 * done. */
int ldp_map_rows(ldp_engine* e, uint32_t first_variant, uint32_t n, void** device_rows, uint64_t* stride_bytes);
/* Variant records of a variable-width .pgen file, decoded ON THE DEVICE from the file's own bytes and loaded: what the reference's
 * reader thread does one variant at a time before LdPrune's pair loop (plink2_ld.cc:1345-1390 PgrGetInv1 ->
 * ReadGenovecSubsetUnsafe, pgenlib_read.cc:2849-2912: plain 2-bit, one-bit + exceptions, difflists, LD-compressed chains,
 * pgenlib_read.cc:2186-2760; and for variants with more than one ALT allele Get1Multiallelic, pgenlib_read.cc:5417-5563, with
 * the major allele chosen as ComputeAlleleFreqs / GetMajIdxMulti do, plink2_filter.cc:2113-2153, plink2_common.cc:1042-1070).
 * Only the main track and auxiliary track 1 are read; phase and dosage tracks behind them are ignored.
 *   recs[q]      record of variant first_variant + q inside `bytes` (host or device memory, `location`): offset, length, the
 *                file's variant record type byte, and the allele count from the .pvar (ldp_pgen_record_index() fills the first
 *                three from a file's index).  Variants this engine does not own are decoded too (LD chains run through them).
 *   ld_base      optional: the record that stands alone (not LD-compressed) on which recs[0], if it is LD-compressed, builds,
 *                when that record is not part of this call.  Without it an LD-compressed first record builds on the last
 *                stand-alone record of the previous call, provided this call starts where that one ended -- in the engine
 *                (first_variant) AND in the file: recs[0].offset must equal the previous call's last offset + length.  Record
 *                offsets of consecutive calls must therefore live in ONE address space (whole-file offsets, as
 *                ldp_pgen_record_index() gives them), whatever `bytes` points at; a caller that hands over every chunk in a
 *                buffer of its own with offsets restarting at 0 passes ld_base (or cuts its chunks at stand-alone records),
 *                otherwise LDP_ERR_INVALID.
 *   raw_sample_ct  samples of the file = the engine's founder_ct, or the raw count of ldp_set_sample_map() (the engine then
 *                gathers its columns as for LDP_GENO_MAPPED rows; the major allele of a variant with allele_ct > 2 is counted
 *                over the map's samples when the map is a plain subset of the file's -- every sample at most once, no het ->
 *                missing: the founders among non-founders, plink2_filter.cc:2113-2153 --, and such variants are refused,
 *                LDP_ERR_UNSUPPORTED, under any other map: the chrX / chrY layouts weigh samples).
 *   major_allele_out  optional, n entries: the major allele of the variants with allele_ct > 2 (their maj_freq is set as by
 *                ldp_set_maj_freqs), UINT32_MAX for the others (decided by the count pass: ldp_get_variant_recs).
 * LDP_ERR_INVALID for a malformed record (no row of the call counts as loaded then). */
typedef struct ldp_pgen_rec {
  uint64_t offset;
  uint32_t length;
  uint8_t vrtype;
  uint8_t allele_ct; /* 2..255 */
  uint16_t reserved;
} ldp_pgen_rec;
int ldp_load_pgen_records(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                          const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* major_allele_out);
/* The same for --indep-pairphase (plink2_ld.cc:1449-2163; loader :2040-2052): the records' main tracks AND their hardcall-phase tracks
 * (auxiliary track 2, pgen_spec "Phased heterozygous hard-calls"; ReadGenovecHphaseSubsetUnsafe pgenlib_read.cc:6704, what
 * PgrGetInv1P :7016 hands HapsplitMustPhased pgenlib_misc.cc:1887) are decoded on the device into LDP_GENO_PHASED rows and loaded;
 * the engine's founder_ct is the haplotype count, 2 x raw_sample_ct, and every sample of the file is used (no sample map).  Every
 * heterozygous call must be phased, as the reference demands of its founders (plink2_ld.cc:2045-2049): otherwise nothing of the
 * offending launch is loaded, the call returns LDP_ERR_UNPHASED and *unphased_variant (optional) receives the lowest variant of the
 * call with such a het call (UINT32_MAX when there is none).  Records with more than one ALT allele (allele_ct > 2) are collapsed on their
 * major allele as in ldp_load_pgen_records(), and the phase of a het of the collapsed row comes from the file's bit of that call the way
 * PgrGetInv1P -> Get1MP hands it on (pgenlib_read.cc:7016, :6962: the track counts every het call of the file, ALTx/ALTy ones included; its
 * phaseinfo is passed through as "the counted allele is on the first haplotype", which for a major allele other than REF is the complement of
 * what the file says -- the reference prunes with that reading, and so does this). */
int ldp_load_pgen_records_phased(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                                 const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* unphased_variant);
/* Give the engine's device memory back (image, records, predicate rows, staging) while keeping its plan: for a caller that works
 * through more data than fits HBM, one engine (chromosome) after the other.  The next ldp_load_genotypes() / ldp_map_rows()
 * allocates again; every row has to be loaded again before the next ldp_run(). */
int ldp_release_device(ldp_engine* e);
/* Sample-mapped rows: the rows the reference assembles per variant on chrX, chrY and MT before LdPrune's pair loop
 * (plink2_ld.cc:1356-1388: founder subset, SetHetMissing on the haploid samples, on chrX the males once and the non-males
 * as two pseudo-samples -- DESIGN.md section 7), built on the device.  Column f of an engine row is sample src_sample[f] of
 * the loaded row; with het_to_missing[f] != 0 a heterozygous call there becomes missing.  The allele counts that decide
 * the major allele (and give maj_freq) are taken over the columns BEFORE that substitution, one allele pair per column,
 * which is the reference's chrX / haploid frequency arithmetic (plink2_filter.cc:2137-2147 with the male / non-male
 * weights of :2096-2112) when the non-male columns appear twice.  src_sample, het_to_missing: founder_ct entries each.
 * Rows loaded with LDP_GENO_MAPPED are then raw_sample_ct samples wide (ceil(raw_sample_ct / 4) bytes). */
int ldp_set_sample_map(ldp_engine* e, uint32_t raw_sample_ct, const uint32_t* src_sample, const uint8_t* het_to_missing);
/* major-allele frequencies (GetAlleleFreq(..., maj_alleles[v]), plink2_ld.cc:915) for LDP_GENO_INVERSE
 * input; for REF/BED input the engine derives them itself and this call overrides them. */
int ldp_set_maj_freqs(ldp_engine* e, uint32_t first_variant, uint32_t n, const double* maj_freqs);
/* --indep-preferred (plink2_ld.cc:916-918): bitmap over variants, bit set = preferred */
int ldp_set_preferred(ldp_engine* e, const uint64_t* preferred_bitmap);

/* ---- compute ---- */
/* removed: bitmap of variant_ct bits (caller-allocated, (variant_ct+63)/64 words), bit v set <=> variant v
 * pruned == removed_variants_collapsed (plink2_ld.cc:2555,1424).  Non-owned variants' bits are 0.
 * Queues whatever pair work ldp_load_genotypes() has not started yet, replays each group of variants on host
 * threads as its predicate rows come back (overlapping the GPU), and returns when the bitmap is complete.
 * Pairs whose predicate is provably false are dropped early inside the kernel; the bitmap is
 * exactly the reference's.  Calling it again recomputes from the resident bit-planes. */
int ldp_run(ldp_engine* e, uint64_t* removed);
/* Same, additionally returning the integer 6-tuple of every candidate pair: stats[pair_off[j] + (i - lo[j])]
 * for lo[j] <= i < j, pair_off = exclusive prefix sum of (j - lo[j]).  Intended for parity tests. */
int ldp_run_with_stats(ldp_engine* e, uint64_t* removed, ldp_pair_stats_t* stats, uint64_t stats_capacity);
/* Arbitrary pairs (first[k] < second[k] not required) through the reference kernel (one wave per pair). */
int ldp_pair_stats(ldp_engine* e, uint32_t n_pairs, const uint32_t* first, const uint32_t* second, ldp_pair_stats_t* out);
/* ---- --r2-unphased matrices (Vcor / VcorMatrix, plink2_ld.cc:12050,9766; ComputeR2 :6654-6682) ---- */
/* All-pairs plan over variant_ct variants (inter-chromosomal pairs included, as the matrix shapes of
 * --r2-unphased do): use instead of ldp_set_variants(), then ldp_load_genotypes() as usual. */
int ldp_set_variants_matrix(ldp_engine* e, uint32_t variant_ct);
/* r^2 for rows [row_first, row_first+row_ct) of the lower triangle incl. the diagonal: out[(j-row_first)*ld_elems
 * + i] for i <= j, as float (bin4) or double (bin); elements with i > j are set to 0.  Undefined pairs are NaN with
 * the reference's bit pattern.  `out` is host memory; ld_elems >= row_first+row_ct.  The caller assembles
 * square / square0 / triangle files from row chunks (VcorMatrixThread :9518-9652 computes the same rows). */
int ldp_r2_unphased_rows(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t ld_elems);

/* The same rows filtered on the device (what the table writers keep, VcorTableWriteThread :10816-10821): every pair
 * first < second with second in [row_first, row_first+row_ct) whose |r^2| >= min_r2 (NaN never passes), in NO
 * particular order -- sort by (first, second) for the .vcor file's order.  *count receives the number found; when it
 * exceeds `capacity` only the first `capacity` stored are valid: call again with a smaller row range or a larger
 * buffer.  On the all-pairs plan (ldp_set_variants_matrix) this is --r2-unphased inter-chr (plink2_ld.cc:11082-11116);
 * on the windowed plan (ldp_set_variants_vcor) the band's pairs, i.e. the default .vcor table. */
typedef struct {
  uint32_t first;
  uint32_t second;
  double r2;
} ldp_r2_hit;
int ldp_r2_unphased_hits(ldp_engine* e, uint32_t row_first, uint32_t row_ct, double min_r2, ldp_r2_hit* out, uint64_t capacity, uint64_t* count);
/* Column blocks of the same all-pairs matrix, for the reference's `--parallel k n` decomposition of the r^2 outputs
 * (VcorMatrix plink2_ld.cc:9800-9824, VcorTable :11157-11168) and for sharding them over devices: only pairs first < second
 * with second in [row_first, row_first+row_ct) AND first in [col_first, col_first+col_ct).  Dense form: out[(second -
 * row_first) * ld_elems + (first - col_first)], the diagonal element included when it lies in the block; hit form as above.
 * Only the block products that touch the column range are computed. */
int ldp_r2_unphased_block(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, int as_float, void* out,
                          uint64_t ld_elems);
int ldp_r2_unphased_block_hits(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, double min_r2, ldp_r2_hit* out,
                               uint64_t capacity, uint64_t* count);
/* chrX pairs of the same block (ComputeXR2, plink2_ld.cc:7122-7190; is_x :9946-9951): a pair with a chrX variant weighs the male
 * founders down in all six sums -- by 1/2 when both variants are on chrX, by 1 - sqrt(2)/2 when one is -- before the quotient
 * (clamped at 1; NaN when a weighted variance is not positive).  `e` holds all founders, `male` (NULL: there are none) the same
 * variants through a sample map of the male founders, on the same device; is_x[variant_ct] marks the chrX rows; flip_all /
 * flip_male[variant_ct] (NULL: none) mark the rows whose engine orientation (ldp_variant_rec.flags bit 0) differs from the one the
 * values are wanted in -- with the irrational weight the rounding depends on it.  Both engines' six integers of every such pair
 * come from the pair kernels and are combined on the device, the reference's doubles fma for fma.
 * Dense form: `out` is the block ldp_r2_unphased_block() filled (same layout, host memory); only the elements of pairs (i < j) with
 * a chrX variant are overwritten, with r^2 or (unsquared != 0) r.  Hit form: those pairs with |value| >= min_r2, as above. */
int ldp_r2_unphased_block_x(ldp_engine* e, ldp_engine* male, const uint8_t* is_x, const uint8_t* flip_all, const uint8_t* flip_male, uint32_t row_first,
                            uint32_t row_ct, uint32_t col_first, uint32_t col_ct, int as_float, int unsquared, void* out, uint64_t ld_elems);
int ldp_r2_unphased_block_x_hits(ldp_engine* e, ldp_engine* male, const uint8_t* is_x, const uint8_t* flip_all, const uint8_t* flip_male, uint32_t row_first,
                                 uint32_t row_ct, uint32_t col_first, uint32_t col_ct, int unsquared, double min_r2, ldp_r2_hit* out, uint64_t capacity,
                                 uint64_t* count);
/* The six integers themselves (ldp_pair_stats_t, the engine's orientation) of the pairs i < j of the block, same layout; zero elsewhere. */
int ldp_pair_stats_block(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, ldp_pair_stats_t* out, uint64_t ld_elems);

/* ---- --r2-unphased table (VcorTable, plink2_ld.cc:11025; window: UpdateVcorWindow :10984-11023) ---- */
/* Windowed plan: variant B is paired with the earlier variants A of its chromosome with bp[B] - bp[A] <= bp_radius
 * and at most var_ct_radius variants between... i.e. B - A <= var_ct_radius in include-order (the reference's
 * --ld-window-kb / --ld-window).  Use instead of ldp_set_variants(), then ldp_load_genotypes();
 * ldp_get_band() returns lo[] (first partner of each second variant). */
int ldp_set_variants_vcor(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps, uint32_t bp_radius,
                          uint32_t var_ct_radius);
/* The same with --ld-window-cm (UpdateVcorWindow :11008-11013): cms[] are the variants' centimorgan positions, nondecreasing
 * inside a chromosome; B > A also has to satisfy cms[B] < cms[A] + cm_radius.  cms == NULL: no centimorgan window. */
int ldp_set_variants_vcor_cm(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps, const double* cms, uint32_t bp_radius,
                             double cm_radius, uint32_t var_ct_radius);
/* r^2 of every candidate pair whose SECOND variant j lies in [row_first, row_first+row_ct), band order: the pairs of j
 * start at element sum_{row_first <= j' < j} (j' - lo[j']) and run over first variants i = lo[j] .. j-1.  Same doubles
 * (bin) / floats (bin4) as ldp_r2_unphased_rows, NaN included; filtering (--ld-window-r2) and the A-major order of the
 * .vcor file are the caller's (VcorTableWriteThread :10680-10950).  `out` is host memory of capacity_elems elements. */
int ldp_r2_unphased_band_rows(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t capacity_elems);
/* --r-unphased (ComputeR2 + the callers' sqrt, plink2_ld.cc:6654-6682, :9633-9641, :10640-10647): mode 1 makes every
 * ldp_r2_unphased_* call return r = +-sqrt(r^2), negative when the covariance of the two variants' major-allele-oriented
 * codes is; mode 2 orients both variants to REF instead ('ref-based'); 0 (default) returns r^2.  The hit filter then
 * compares |r| with min_r2, so pass sqrt(threshold).  NaN where r^2 is undefined, as before. */
int ldp_set_r_signed(ldp_engine* e, int mode);

/* ---- inspection ---- */
int ldp_get_variant_recs(ldp_engine* e, uint32_t first_variant, uint32_t n, ldp_variant_rec* out);
int ldp_get_maj_freqs(ldp_engine* e, uint32_t first_variant, uint32_t n, double* out);
/* bit-planes of one variant as the kernels see them: hom and ref2het, ceil(founder_ct/32) dwords each */
int ldp_get_planes(ldp_engine* e, uint32_t variant, uint32_t* hom, uint32_t* ref2het);
int ldp_get_counters(const ldp_engine* e, ldp_counters* out);


/* ---- genotype file reader (host-side I/O edge; no GPU involved) ---- */
/* Main-track reader for PLINK binary genotype files: .bed (storage mode 0x01), fixed-width .pgen (0x02) and
 * standard variable-width .pgen (0x10: raw / one-bit / LD-compressed / difflist records).  Replaces, for this
 * path, PgfiInitPhase1/2 + PgrInit + ReadGenovecSubsetUnsafe (pgenlib_read.cc:691,1101,2044,2849).  Rows come
 * out as 2-bit codes, ceil(sample_ct/4) bytes each, in the encoding ldp_pgen_info reports (LDP_GENO_BED for
 * .bed, LDP_GENO_REF otherwise), ready for ldp_load_genotypes().  sample_ct_hint/variant_ct_hint: required
 * for .bed (dimensions live in .fam/.bim), cross-checked against the header otherwise (0 = no check). */
typedef struct ldp_pgen ldp_pgen;
int ldp_pgen_open(const char* path, uint32_t sample_ct_hint, uint32_t variant_ct_hint, ldp_pgen** out);
/* The same with the name of the index file of an external-index .pgen (storage mode 0x20: the .pgen holds the records,
 * "<path>.pgi" -- or pgi_path, the reference's --pgi -- the header; PgfiInitPhase1, pgenlib_read.cc:800-840).  pgi_path may
 * be NULL; it is ignored for files that carry their own header. */
int ldp_pgen_open_indexed(const char* path, const char* pgi_path, uint32_t sample_ct_hint, uint32_t variant_ct_hint, ldp_pgen** out);
int ldp_pgen_info(const ldp_pgen* p, uint32_t* variant_ct, uint32_t* sample_ct, int* storage_mode, int* row_encoding, int* has_multiallelic);
/* Which REF alleles are provisional (PgfiInitPhase1 / Phase2, pgenlib_read.cc:790,872-877: control bits 6-7 of the header):
 * returns 0 = the .pgen does not say (the .pvar's INFO/PR does), 1 = none, 2 = all (always for a .bed), 3 = per variant, and
 * then bit v of bits[] (up to bits_bytes bytes, may be NULL) is set when variant v's REF is provisional; -1 on a NULL handle. */
int ldp_pgen_provisional_ref(const ldp_pgen* p, uint8_t* bits, uint64_t bits_bytes);
/* 1 when some variant record carries a dosage track (vrtype bits 5-6).  The reader decodes hardcalls only; the reference
 * derives allele frequencies -- hence the major allele and the prune tie-break -- from dosages when they exist
 * (plink2_data.cc:2424-2566), so a caller that wants the reference's prune list must either refuse such files (plink2-hip
 * does) or supply dosage-based frequencies through ldp_set_maj_freqs(). */
int ldp_pgen_has_dosage(const ldp_pgen* p);
/* ... and whether this variant's record does; its two allele dosage sums over the samples of sample_mask (bit s of byte s >> 3;
 * NULL = all), in the reference's units -- 16384 per ALT allele copy, so a diploid sample adds 32768 to ref + alt -- exactly
 * as GetBasicGenotypeCountsAndDosage16s (pgenlib_read.cc:7917) forms them: a sample's dosage where it has one, its hardcall
 * otherwise.  Also defined for records without a dosage track and for .bed rows (hardcalls only).  From these the reference's
 * allele frequency is ref * (1 / (ref + alt)) (ComputeAlleleFreqs, plink2_filter.cc:2113-2153).  LDP_ERR_UNSUPPORTED for a
 * multiallelic record. */
int ldp_pgen_variant_has_dosage(const ldp_pgen* p, uint32_t variant);
int ldp_pgen_dosage_sums(ldp_pgen* p, uint32_t variant, const uint8_t* sample_mask, uint64_t* ref_dosage, uint64_t* alt_dosage);
/* fixed-width modes only: pointer to row 0 inside the file mapping (zero-copy), NULL for variable-width files */
const void* ldp_pgen_direct_rows(const ldp_pgen* p, uint64_t* stride_bytes);
/* ... and the same rows as (descriptor, offset of row 0, stride) for ldp_load_genotypes_fd(); -1 for variable-width files.  The
 * descriptor stays the reader's (do not close it). */
int ldp_pgen_direct_fd(const ldp_pgen* p, uint64_t* first_row_offset, uint64_t* stride_bytes);
/* For ldp_load_pgen_records(): the file's bytes (the reader's mapping) and the index entries of variants [first_variant, +n) --
 * offset, length, record type; allele_ct is set to 2 (the .pvar knows better).  *ld_base_variant (optional): the variant whose
 * record the first one builds on when it is LD-compressed (GetLdbaseVidx, pgenlib_read.cc:1848), UINT32_MAX otherwise.
 * Fixed-width .pgen files yield plain 2-bit records; LDP_ERR_UNSUPPORTED for a .bed (its rows are not .pgen codes). */
const void* ldp_pgen_file_bytes(const ldp_pgen* p, uint64_t* n_bytes);
int ldp_pgen_record_index(const ldp_pgen* p, uint32_t first_variant, uint32_t n, ldp_pgen_rec* out, uint32_t* ld_base_variant);
/* decode rows [first_variant, first_variant+n) into out_rows; 64k-variant blocks decode on up to `threads` host threads (0 = all) */
int ldp_pgen_read(ldp_pgen* p, uint32_t first_variant, uint32_t n, void* out_rows, uint64_t stride_bytes, uint32_t threads);
/* Multiallelic hard-call track (pgen_spec.tex:469-540; what Get1Multiallelic, pgenlib_read.cc:5417, consumes):
 * per-sample allele index pairs of one variant, allele_lo <= allele_hi, 0 = REF, k = ALTk, 255 = missing.
 * alt_ct = number of ALT alleles in the .pvar (<= 254).  Works for biallelic records too. */
/* --indep-pairphase input (what PgrGetInv1P delivers, minus the major-allele inversion): rows in the
 * LDP_GENO_REF | LDP_GENO_PHASED layout for hap_ct = 2 * sample_ct (2-bit codes, padding, phaseinfo bits from the
 * hardcall-phase track, pgen_spec.tex:541-562; phaseinfo set = ALT on the first haplotype, "1|0").
 * sample_mask (optional, ceil(sample_ct/8) bytes): only these samples' het calls must be phased (the reference
 * checks after founder subsetting).  Returns LDP_ERR_UNPHASED and the lowest offending variant index in
 * *unphased_variant when a het call of a masked sample has no phase.  Records with multiallelic hard-calls get
 * their main-track codes and all-zero phase bits, unchecked (see ldp_pgen_read_alleles_phased). */
int ldp_pgen_read_phased(ldp_pgen* p, uint32_t first_variant, uint32_t n, void* out_rows, uint64_t stride_bytes,
                         const uint8_t* sample_mask, uint32_t threads, uint32_t* unphased_variant);
int ldp_pgen_variant_is_multiallelic(const ldp_pgen* p, uint32_t variant);
int ldp_pgen_read_alleles(ldp_pgen* p, uint32_t variant, uint32_t alt_ct, uint8_t* allele_lo, uint8_t* allele_hi);
/* ... plus the hardcall phase of every heterozygous call (allele_lo != allele_hi), multiallelic ones included, as the
 * file stores it (what ReadGenovecHphaseSubsetUnsafe / Get1MP parse, pgenlib_read.cc:6704,6962): one bit per sample,
 * ceil(sample_ct/8) bytes each; phaseinfo set = the HIGHER-indexed allele sits on the first haplotype ("1|0", "2|1").
 * ldp_pgen_read_phased() leaves the phase bits of multiallelic records zero: this is the reader for them. */
int ldp_pgen_read_alleles_phased(ldp_pgen* p, uint32_t variant, uint32_t alt_ct, uint8_t* allele_lo, uint8_t* allele_hi,
                                 uint8_t* phasepresent, uint8_t* phaseinfo);
/* Keep the samples whose bit is set in sample_mask (ceil(raw_sample_ct/8) bytes), for n_rows rows at once: the 2-bit
 * codes (CopyNyparrNonemptySubset, pgenlib_misc.cc:32,185) and, with phased != 0, the phaseinfo bits of
 * LDP_GENO_PHASED rows (CopyBitarrSubset; input rows laid out for 2*raw_sample_ct haplotypes, output rows for
 * 2*kept).  The founder subsetting PgrGetInv1 does while decoding (plink2_ld.cc:1357).  Host memory, host threads
 * (0 = all). */
int ldp_subset_samples(const void* in_rows, uint64_t in_stride, uint32_t n_rows, uint32_t raw_sample_ct, const uint8_t* sample_mask,
                       void* out_rows, uint64_t out_stride, int phased, uint32_t threads);
const char* ldp_pgen_last_error(const ldp_pgen* p);
void ldp_pgen_close(ldp_pgen* p);

/* Test hooks, kernel-selection switches and the benchmark's synthetic generator are NOT part of this boundary: ldprune_hip_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* LDPRUNE_HIP_H */
