// indep_pairwise_hip.cc -- the reference-side binding of libldprune_hip.so, made real.
//
// This translation unit is OUR code, compiled against the reference's own headers where they lie (-I/root/reference/2.0),
// and linked into a build of the reference (oracle/Makefile, target `ref_hip`) whose plink2_ld.cc has ONE token changed at
// build time: the call `IndepPairwise(...)` inside LdPrune() (plink2_ld.cc:2700, between LoadBalance :2692 and LdPruneWrite
// :2708) becomes `IndepPairwiseHip(...)`, same argument list.  Nothing of the reference is copied into this repo; the patched
// source is a build product under oracle/_ref/.
//
// IndepPairwiseHip() follows the convention of the reference's existing GPU seam (2.0/cuda/plink2_matrix_cuda.h:24-104, failure
// mapping plink2_matrix_calc.cc:9632-9635): plain C calls, int codes, kPglRetGpuFail + "Error: GPU operation failure." on
// any device-side failure.  It keeps the reference's loader (PgrGetInv1 with the founder subset, plink2_ld.cc:1357) and its
// major-allele frequencies (GetAlleleFreq, :915) and hands them to the engine as LDP_GENO_INVERSE rows; the engine returns
// removed_variants_collapsed (:2555, :1424).  Chromosomes the engine's autosomal-diploid path does not cover (chrX, chrY,
// MT, other haploid contigs: the sample-mapped rows of DESIGN.md section 7 are the caller's job) go to the reference's own
// IndepPairwise(), as does everything when PLINK2_HIP_LDPRUNE=0 is set in the environment.
#include "plink2_ld.h"

#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ldprune_hip.h"

namespace plink2 {

// plink2_ld.cc: not static, not in the header
PglErr IndepPairwise(const uintptr_t* variant_include, const ChrInfo* cip, const uint32_t* variant_bps, const uintptr_t* allele_idx_offsets, const AlleleCode* maj_alleles, const double* allele_freqs, const uintptr_t* founder_info, const uint32_t* founder_info_cumulative_popcounts, const uintptr_t* founder_nonmale, const uintptr_t* founder_male, const uintptr_t* founder_nonfemale, const LdInfo* ldip, const uintptr_t* preferred_variants, const uint32_t* subcontig_info, const uint32_t* subcontig_thread_assignments, uint32_t raw_sample_ct, uint32_t founder_ct, uint32_t founder_male_ct, uint32_t founder_nonfemale_ct, uint32_t subcontig_ct, uintptr_t window_max, uint32_t calc_thread_ct, uint32_t max_load, PgenReader* simple_pgrp, uintptr_t* removed_variants_collapsed);

namespace {

PglErr MapLdpError(int rc, const ldp_engine* eng) {
  if (rc == LDP_ERR_NOMEM) {
    return kPglRetNomem;
  }
  if (rc == LDP_ERR_UNSUPPORTED) {
    logerrputs("Error: --indep-pairwise does not support >= 2^30 founders.\n");  // (the reference's own limit, plink2_ld.cc:1122)
    return kPglRetNotYetSupported;
  }
  logputs("\n");
  logerrputs("Error: GPU operation failure.\n");  // plink2_matrix_calc.cc:9634
  if (eng && ldp_last_error(eng)[0]) {
    logerrprintfww("(HIP LD-prune engine: %s)\n", ldp_last_error(eng));
  }
  return kPglRetGpuFail;
}

}  // namespace

PglErr IndepPairwiseHip(const uintptr_t* variant_include, const ChrInfo* cip, const uint32_t* variant_bps, const uintptr_t* allele_idx_offsets, const AlleleCode* maj_alleles, const double* allele_freqs, const uintptr_t* founder_info, const uint32_t* founder_info_cumulative_popcounts, const uintptr_t* founder_nonmale, const uintptr_t* founder_male, const uintptr_t* founder_nonfemale, const LdInfo* ldip, const uintptr_t* preferred_variants, const uint32_t* subcontig_info, const uint32_t* subcontig_thread_assignments, uint32_t raw_sample_ct, uint32_t founder_ct, uint32_t founder_male_ct, uint32_t founder_nonfemale_ct, uint32_t subcontig_ct, uintptr_t window_max, uint32_t calc_thread_ct, uint32_t max_load, PgenReader* simple_pgrp, uintptr_t* removed_variants_collapsed) {
  const uint32_t raw_variant_ct = cip->chr_fo_vidx_start[cip->chr_ct];
  const uint32_t raw_variant_ctl = BitCtToWordCt(raw_variant_ct);
  const uint32_t variant_ct = PopcountWords(variant_include, raw_variant_ctl);
  // ---- who runs it ----
  const char* off = getenv("PLINK2_HIP_LDPRUNE");
  bool use_hip = !(off && (!strcmp(off, "0"))) && (ldp_device_count() > 0) && variant_ct;
  if (use_hip) {
    // any included variant on a haploid / sex chromosome -> the reference's own path (whole job: the result bitmap is one)
    for (uint32_t chr_fo_idx = 0; chr_fo_idx != cip->chr_ct; ++chr_fo_idx) {
      const uint32_t vstart = cip->chr_fo_vidx_start[chr_fo_idx];
      const uint32_t vend = cip->chr_fo_vidx_start[chr_fo_idx + 1];
      if ((vstart != vend) && PopcountBitRange(variant_include, vstart, vend) && IsSet(cip->haploid_mask, cip->chr_file_order[chr_fo_idx])) {
        use_hip = false;
        break;
      }
    }
  }
  if (!use_hip) {
    return IndepPairwise(variant_include, cip, variant_bps, allele_idx_offsets, maj_alleles, allele_freqs, founder_info, founder_info_cumulative_popcounts, founder_nonmale, founder_male, founder_nonfemale, ldip, preferred_variants, subcontig_info, subcontig_thread_assignments, raw_sample_ct, founder_ct, founder_male_ct, founder_nonfemale_ct, subcontig_ct, window_max, calc_thread_ct, max_load, simple_pgrp, removed_variants_collapsed);
  }

  ldp_params p;
  memset(&p, 0, sizeof(p));
  p.founder_ct = founder_ct;
  p.prune_window_size = ldip->prune_window_size;
  p.prune_window_incr = ldip->prune_window_incr;
  p.window_is_bp = (ldip->prune_flags / kfLdPruneWindowBp) & 1;
  p.plink1_order = (ldip->prune_flags / kfLdPrunePlink1Order) & 1;
  p.prune_last_param = ldip->prune_last_param;  // (the engine applies *(1 + kSmallEpsilon) itself, cf. plink2_ld.cc:1255)
  p.device = -1;
  p.stream = nullptr;
  ldp_engine* eng = nullptr;
  int rc = ldp_create(&p, &eng);
  if (rc) {
    return MapLdpError(rc, nullptr);
  }
  PglErr reterr = kPglRetSuccess;
  logprintf("--indep-pairwise (HIP, %d device%s visible): ", ldp_device_count(), (ldp_device_count() == 1)? "" : "s");
  fflush(stdout);
  do {
    // 1. the variant table in include-order: chromosome file-order index and position (what LdPruneSubcontigSplitAll reads)
    std::vector<uint32_t> chr_fo(variant_ct), bps(variant_ct), uidxs(variant_ct);
    {
      uintptr_t variant_uidx_base = 0;
      uintptr_t cur_bits = variant_include[0];
      uint32_t chr_fo_idx = 0;
      uint32_t chr_end = cip->chr_fo_vidx_start[1];
      for (uint32_t variant_idx = 0; variant_idx != variant_ct; ++variant_idx) {
        const uint32_t variant_uidx = BitIter1(variant_include, &variant_uidx_base, &cur_bits);
        while (variant_uidx >= chr_end) {
          ++chr_fo_idx;
          chr_end = cip->chr_fo_vidx_start[chr_fo_idx + 1];
        }
        uidxs[variant_idx] = variant_uidx;
        chr_fo[variant_idx] = chr_fo_idx;
        bps[variant_idx] = variant_bps? variant_bps[variant_uidx] : 0;
      }
    }
    rc = ldp_set_variants(eng, variant_ct, chr_fo.data(), (p.window_is_bp && variant_bps)? bps.data() : nullptr);
    if (rc) {
      reterr = MapLdpError(rc, eng);
      break;
    }
    // 2. genotypes: the reference's own decode (PgrGetInv1 on the founder subset: 2-bit counts of the non-major allele),
    //    a batch of rows at a time, with the frequencies the scan compares (GetAlleleFreq of the major allele)
    PgrSampleSubsetIndex pssi;
    PgrSetSampleSubsetIndex(founder_info_cumulative_popcounts, simple_pgrp, &pssi);
    const uintptr_t row_bytes = NypCtToVecCt(founder_ct) * kBytesPerVec;  // PgrGetInv1 writes whole vectors
    const uint32_t batch = 1 + (256 * 1048576 / row_bytes);
    unsigned char* rows;
    if (cachealigned_malloc(batch * row_bytes, &rows)) {
      reterr = kPglRetNomem;
      break;
    }
    std::vector<double> maj_freqs(batch);
    uint32_t cur_allele_ct = 2;
    for (uint32_t batch_start = 0; (batch_start < variant_ct) && (!reterr); batch_start += batch) {
      const uint32_t n = (variant_ct - batch_start < batch)? (variant_ct - batch_start) : batch;
      for (uint32_t k = 0; k != n; ++k) {
        const uint32_t variant_uidx = uidxs[batch_start + k];
        reterr = PgrGetInv1(founder_info, pssi, founder_ct, variant_uidx, maj_alleles[variant_uidx], simple_pgrp, R_CAST(uintptr_t*, &(rows[k * row_bytes])));
        if (unlikely(reterr)) {
          PgenErrPrintNV(reterr, variant_uidx);
          break;
        }
        uintptr_t allele_idx_base;
        if (!allele_idx_offsets) {
          allele_idx_base = variant_uidx;
        } else {
          allele_idx_base = allele_idx_offsets[variant_uidx];
          cur_allele_ct = allele_idx_offsets[variant_uidx + 1] - allele_idx_base;
          allele_idx_base -= variant_uidx;
        }
        maj_freqs[k] = GetAlleleFreq(&(allele_freqs[allele_idx_base]), maj_alleles[variant_uidx], cur_allele_ct);
      }
      if (reterr) {
        break;
      }
      rc = ldp_load_genotypes(eng, batch_start, n, rows, row_bytes, LDP_MEM_HOST, LDP_GENO_INVERSE);
      if (!rc) {
        rc = ldp_set_maj_freqs(eng, batch_start, n, maj_freqs.data());
      }
      if (rc) {
        reterr = MapLdpError(rc, eng);
      }
    }
    aligned_free(rows);
    if (reterr) {
      break;
    }
    // 3. --indep-preferred: raw-index bitmap -> include-order bitmap
    if (preferred_variants) {
      std::vector<uint64_t> pref((variant_ct + 63) / 64, 0);
      for (uint32_t variant_idx = 0; variant_idx != variant_ct; ++variant_idx) {
        if (IsSet(preferred_variants, uidxs[variant_idx])) {
          pref[variant_idx / 64] |= 1ULL << (variant_idx % 64);
        }
      }
      rc = ldp_set_preferred(eng, pref.data());
      if (rc) {
        reterr = MapLdpError(rc, eng);
        break;
      }
    }
    // 4. bit i set <=> i-th included variant removed
    rc = ldp_run(eng, R_CAST(uint64_t*, removed_variants_collapsed));
    if (rc) {
      reterr = MapLdpError(rc, eng);
      break;
    }
    fputs("done.\n", stdout);
  } while (0);
  ldp_destroy(eng);
  return reterr;
}

}  // namespace plink2
