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
// any device-side failure.  It keeps the reference's decoder (PgrGetInv1, plink2_ld.cc:1357,1363: counts of the non-major
// allele) and its major-allele frequencies (GetAlleleFreq, :915) and hands the rows to the engines as they come out of the
// file -- all samples, LDP_GENO_INVERSE | LDP_GENO_MAPPED -- with one sample map per chromosome class doing what the
// reference's loader does on the host at :1357-1388: the founder subset for the diploid chromosomes; the same with
// SetHetMissing for haploid ones (MT, ...); on chrX the male founders (hets missing) followed by the others, which the
// engine sees twice (their 2x weight); on chrY the non-female founders, hets missing.  Each class runs
// on an engine of its own (different column sets); the bits come back in include-order and are merged into
// removed_variants_collapsed (:2555, :1424).  PLINK2_HIP_LDPRUNE=0 in the environment sends everything to the reference's
// own IndepPairwise().
#include "plink2_ld.h"

#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ldprune_hip.h"

namespace plink2 {

// plink2_ld.cc: not static, not in the header
PglErr IndepPairwise(const uintptr_t* variant_include, const ChrInfo* cip, const uint32_t* variant_bps, const uintptr_t* allele_idx_offsets, const AlleleCode* maj_alleles, const double* allele_freqs, const uintptr_t* founder_info, const uint32_t* founder_info_cumulative_popcounts, const uintptr_t* founder_nonmale, const uintptr_t* founder_male, const uintptr_t* founder_nonfemale, const LdInfo* ldip, const uintptr_t* preferred_variants, const uint32_t* subcontig_info, const uint32_t* subcontig_thread_assignments, uint32_t raw_sample_ct, uint32_t founder_ct, uint32_t founder_male_ct, uint32_t founder_nonfemale_ct, uint32_t subcontig_ct, uintptr_t window_max, uint32_t calc_thread_ct, uint32_t max_load, PgenReader* simple_pgrp, uintptr_t* removed_variants_collapsed);

PglErr IndepPairphase(const uintptr_t* variant_include, const ChrInfo* cip, const uint32_t* variant_bps, const uintptr_t* allele_idx_offsets, const AlleleCode* maj_alleles, const double* allele_freqs, const uintptr_t* founder_info, const uint32_t* founder_info_cumulative_popcounts, const uintptr_t* founder_nonmale, const uintptr_t* founder_male, const uintptr_t* founder_nonfemale, const LdInfo* ldip, const uintptr_t* preferred_variants, const uint32_t* subcontig_info, const uint32_t* subcontig_thread_assignments, uint32_t raw_sample_ct, uint32_t founder_ct, uint32_t founder_male_ct, uint32_t founder_nonfemale_ct, uint32_t subcontig_ct, uintptr_t window_max, uint32_t calc_thread_ct, uint32_t max_load, PgenReader* simple_pgrp, uintptr_t* removed_variants_collapsed);

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
  const uint32_t founder_nonmale_ct = founder_ct - founder_male_ct;
  // ---- who runs it ----
  const char* off = getenv("PLINK2_HIP_LDPRUNE");
  const bool use_hip = !(off && (!strcmp(off, "0"))) && (ldp_device_count() > 0) && variant_ct;
  if (!use_hip) {
    return IndepPairwise(variant_include, cip, variant_bps, allele_idx_offsets, maj_alleles, allele_freqs, founder_info, founder_info_cumulative_popcounts, founder_nonmale, founder_male, founder_nonfemale, ldip, preferred_variants, subcontig_info, subcontig_thread_assignments, raw_sample_ct, founder_ct, founder_male_ct, founder_nonfemale_ct, subcontig_ct, window_max, calc_thread_ct, max_load, simple_pgrp, removed_variants_collapsed);
  }

  // ---- the variant table in include-order, and each variant's chromosome class (plink2_ld.cc:1325-1334) ----
  enum { kDiploid = 0, kHaploid = 1, kChrX = 2, kChrY = 3, kClassCt = 4 };
  const uint32_t all_haploid = IsSet(cip->haploid_mask, 0);  // :1258
  const uint32_t x_code = cip->xymt_codes[kChrOffsetX];
  const uint32_t y_code = cip->xymt_codes[kChrOffsetY];
  std::vector<uint32_t> chr_fo(variant_ct), bps(variant_ct), uidxs(variant_ct);
  std::vector<uint32_t> members[kClassCt];  // include-order indices
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
      const uint32_t chr_idx = cip->chr_file_order[chr_fo_idx];
      const uint32_t is_x = (chr_idx == x_code);
      const uint32_t is_y = (chr_idx == y_code);
      uint32_t cls = kDiploid;
      if (is_x && founder_nonmale_ct) {
        cls = kChrX;
      } else if (is_x || is_y) {
        cls = kChrY;  // (chrX without a single non-male founder takes the chrY branch, :1332,1379-1384)
      } else if (IsSet(cip->haploid_mask, chr_idx)) {
        cls = kHaploid;
      }
      members[cls].push_back(variant_idx);
    }
  }
  // column lists: sample indices of the file, bit 31 = a het call becomes missing (SetHetMissing)
  auto columns_of = [&](const uintptr_t* sample_set, uint32_t het_missing, std::vector<uint32_t>* src, std::vector<unsigned char>* het) {
    for (uint32_t sample_idx = 0; sample_idx != raw_sample_ct; ++sample_idx) {
      if (IsSet(sample_set, sample_idx)) {
        src->push_back(sample_idx);
        het->push_back(het_missing);
      }
    }
  };
  const uint32_t removed_wordct = BitCtToWordCt(variant_ct);
  ZeroWArr(removed_wordct, removed_variants_collapsed);
  logprintf("--indep-pairwise (HIP, %d device%s visible): ", ldp_device_count(), (ldp_device_count() == 1)? "" : "s");
  fflush(stdout);
  PglErr reterr = kPglRetSuccess;
  PgrSampleSubsetIndex pssi;
  PgrClearSampleSubsetIndex(simple_pgrp, &pssi);  // rows of ALL samples: the engines pick their columns
  const uintptr_t row_bytes = NypCtToVecCt(raw_sample_ct) * kBytesPerVec;  // PgrGetInv1 writes whole vectors
  const uint32_t batch = 1 + (256 * 1048576 / row_bytes);
  unsigned char* rows;
  if (cachealigned_malloc(batch * row_bytes, &rows)) {
    return kPglRetNomem;
  }
  std::vector<double> maj_freqs(batch);
  for (uint32_t cls = 0; (cls != kClassCt) && (!reterr); ++cls) {
    const std::vector<uint32_t>& mem = members[cls];
    const uint32_t cls_variant_ct = mem.size();
    if (!cls_variant_ct) {
      continue;
    }
    std::vector<uint32_t> src;
    std::vector<unsigned char> het;
    if (cls == kDiploid) {
      columns_of(founder_info, 0, &src, &het);
    } else if (cls == kHaploid) {
      columns_of(founder_info, 1, &src, &het);
    } else if (cls == kChrX) {
      columns_of(founder_male, 1, &src, &het);
      for (uint32_t rep = 0; rep != 2; ++rep) {
        columns_of(founder_nonmale, all_haploid, &src, &het);
      }
    } else {
      columns_of(founder_nonfemale, 1, &src, &het);
    }
    if (src.size() < 2) {
      continue;  // (nothing to correlate: the reference's loop leaves these variants alone as well)
    }
    ldp_params p;
    memset(&p, 0, sizeof(p));
    p.founder_ct = src.size();
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
      reterr = MapLdpError(rc, nullptr);
      break;
    }
    do {
      std::vector<uint32_t> cls_chr(cls_variant_ct), cls_bps(cls_variant_ct);
      for (uint32_t k = 0; k != cls_variant_ct; ++k) {
        cls_chr[k] = chr_fo[mem[k]];
        cls_bps[k] = bps[mem[k]];
      }
      rc = ldp_set_variants(eng, cls_variant_ct, cls_chr.data(), (p.window_is_bp && variant_bps)? cls_bps.data() : nullptr);
      if (!rc) {
        rc = ldp_set_sample_map(eng, raw_sample_ct, src.data(), het.data());
      }
      if (rc) {
        reterr = MapLdpError(rc, eng);
        break;
      }
      // genotypes: the reference's own decode, a batch of rows at a time, with the frequencies the scan compares
      uint32_t cur_allele_ct = 2;
      for (uint32_t batch_start = 0; (batch_start < cls_variant_ct) && (!reterr); batch_start += batch) {
        const uint32_t n = (cls_variant_ct - batch_start < batch)? (cls_variant_ct - batch_start) : batch;
        for (uint32_t k = 0; k != n; ++k) {
          const uint32_t variant_uidx = uidxs[mem[batch_start + k]];
          reterr = PgrGetInv1(nullptr, pssi, raw_sample_ct, variant_uidx, maj_alleles[variant_uidx], simple_pgrp, R_CAST(uintptr_t*, &(rows[k * row_bytes])));
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
        rc = ldp_load_genotypes(eng, batch_start, n, rows, row_bytes, LDP_MEM_HOST, LDP_GENO_INVERSE | LDP_GENO_MAPPED);
        if (!rc) {
          rc = ldp_set_maj_freqs(eng, batch_start, n, maj_freqs.data());
        }
        if (rc) {
          reterr = MapLdpError(rc, eng);
        }
      }
      if (reterr) {
        break;
      }
      // --indep-preferred: raw-index bitmap -> this engine's bitmap
      if (preferred_variants) {
        std::vector<uint64_t> pref((cls_variant_ct + 63) / 64, 0);
        for (uint32_t k = 0; k != cls_variant_ct; ++k) {
          if (IsSet(preferred_variants, uidxs[mem[k]])) {
            pref[k / 64] |= 1ULL << (k % 64);
          }
        }
        rc = ldp_set_preferred(eng, pref.data());
        if (rc) {
          reterr = MapLdpError(rc, eng);
          break;
        }
      }
      // bit k set <=> the class's k-th variant is removed: into the include-order bitmap
      std::vector<uint64_t> removed((cls_variant_ct + 63) / 64 + 1, 0);
      rc = ldp_run(eng, removed.data());
      if (rc) {
        reterr = MapLdpError(rc, eng);
        break;
      }
      for (uint32_t k = 0; k != cls_variant_ct; ++k) {
        if ((removed[k / 64] >> (k % 64)) & 1) {
          SetBit(mem[k], removed_variants_collapsed);
        }
      }
    } while (0);
    ldp_destroy(eng);
  }
  aligned_free(rows);
  if (!reterr) {
    fputs("done.\n", stdout);
  }
  return reterr;
}

// --indep-pairphase (IndepPairphase, plink2_ld.cc:1802; the call at :2698 is redirected the same way).  Diploid chromosomes:
// PgrGetInv1P on the founder subset (:2039) gives counts of the non-major allele plus phasepresent / phaseinfo; a het call
// without phase is the reference's "not fully phased" error (:2044-2048, HapsplitMustPhased); the rows go to an engine of
// 2 x founder_ct haplotypes as LDP_GENO_INVERSE | LDP_GENO_PHASED, where the conversion kernel does the haplotype split.
// Jobs that include chrX / chrY / MT or another haploid contig go to the reference's own IndepPairphase().
PglErr IndepPairphaseHip(const uintptr_t* variant_include, const ChrInfo* cip, const uint32_t* variant_bps, const uintptr_t* allele_idx_offsets, const AlleleCode* maj_alleles, const double* allele_freqs, const uintptr_t* founder_info, const uint32_t* founder_info_cumulative_popcounts, const uintptr_t* founder_nonmale, const uintptr_t* founder_male, const uintptr_t* founder_nonfemale, const LdInfo* ldip, const uintptr_t* preferred_variants, const uint32_t* subcontig_info, const uint32_t* subcontig_thread_assignments, uint32_t raw_sample_ct, uint32_t founder_ct, uint32_t founder_male_ct, uint32_t founder_nonfemale_ct, uint32_t subcontig_ct, uintptr_t window_max, uint32_t calc_thread_ct, uint32_t max_load, PgenReader* simple_pgrp, uintptr_t* removed_variants_collapsed) {
  const uint32_t raw_variant_ct = cip->chr_fo_vidx_start[cip->chr_ct];
  const uint32_t raw_variant_ctl = BitCtToWordCt(raw_variant_ct);
  const uint32_t variant_ct = PopcountWords(variant_include, raw_variant_ctl);
  const char* off = getenv("PLINK2_HIP_LDPRUNE");
  bool use_hip = !(off && (!strcmp(off, "0"))) && (ldp_device_count() > 0) && variant_ct;
  if (use_hip) {
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
    return IndepPairphase(variant_include, cip, variant_bps, allele_idx_offsets, maj_alleles, allele_freqs, founder_info, founder_info_cumulative_popcounts, founder_nonmale, founder_male, founder_nonfemale, ldip, preferred_variants, subcontig_info, subcontig_thread_assignments, raw_sample_ct, founder_ct, founder_male_ct, founder_nonfemale_ct, subcontig_ct, window_max, calc_thread_ct, max_load, simple_pgrp, removed_variants_collapsed);
  }
  ldp_params p;
  memset(&p, 0, sizeof(p));
  p.founder_ct = 2 * founder_ct;  // haplotypes (plink2_ld.cc:1506)
  p.prune_window_size = ldip->prune_window_size;
  p.prune_window_incr = ldip->prune_window_incr;
  p.window_is_bp = (ldip->prune_flags / kfLdPruneWindowBp) & 1;
  p.plink1_order = (ldip->prune_flags / kfLdPrunePlink1Order) & 1;
  p.prune_last_param = ldip->prune_last_param;
  p.device = -1;
  p.stream = nullptr;
  ldp_engine* eng = nullptr;
  int rc = ldp_create(&p, &eng);
  if (rc) {
    return MapLdpError(rc, nullptr);
  }
  PglErr reterr = kPglRetSuccess;
  logprintf("--indep-pairphase (HIP, %d device%s visible): ", ldp_device_count(), (ldp_device_count() == 1)? "" : "s");
  fflush(stdout);
  uintptr_t* genovec = nullptr;
  uintptr_t* phasepresent = nullptr;
  uintptr_t* phaseinfo = nullptr;
  unsigned char* rows = nullptr;
  do {
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
    PgrSampleSubsetIndex pssi;
    PgrSetSampleSubsetIndex(founder_info_cumulative_popcounts, simple_pgrp, &pssi);
    const uintptr_t code_bytes = NypCtToByteCt(founder_ct);
    const uintptr_t phase_off = ldp_phased_phase_offset(p.founder_ct);
    const uintptr_t phase_bytes = DivUp(founder_ct, 8);
    const uintptr_t row_bytes = (ldp_phased_row_bytes(p.founder_ct) + 3) & (~S_CAST(uintptr_t, 3));
    const uint32_t batch = 1 + (256 * 1048576 / row_bytes);
    if (cachealigned_malloc(NypCtToVecCt(founder_ct) * kBytesPerVec, &genovec) ||
        cachealigned_malloc(BitCtToVecCt(founder_ct) * kBytesPerVec, &phasepresent) ||
        cachealigned_malloc(BitCtToVecCt(founder_ct) * kBytesPerVec, &phaseinfo) ||
        cachealigned_malloc(batch * row_bytes, &rows)) {
      reterr = kPglRetNomem;
      break;
    }
    std::vector<double> maj_freqs(batch);
    uint32_t cur_allele_ct = 2;
    const uint32_t founder_ctl = BitCtToWordCt(founder_ct);
    const uint32_t founder_ctl2 = NypCtToWordCt(founder_ct);
    for (uint32_t batch_start = 0; (batch_start < variant_ct) && (!reterr); batch_start += batch) {
      const uint32_t n = (variant_ct - batch_start < batch)? (variant_ct - batch_start) : batch;
      for (uint32_t k = 0; k != n; ++k) {
        const uint32_t variant_uidx = uidxs[batch_start + k];
        uint32_t phasepresent_ct;
        reterr = PgrGetInv1P(founder_info, pssi, founder_ct, variant_uidx, maj_alleles[variant_uidx], simple_pgrp, genovec, phasepresent, phaseinfo, &phasepresent_ct);
        if (unlikely(reterr)) {
          PgenErrPrintNV(reterr, variant_uidx);
          break;
        }
        ZeroTrailingNyps(founder_ct, genovec);
        if (!phasepresent_ct) {
          ZeroWArr(founder_ctl, phasepresent);
        }
        // every het call must carry phase (HapsplitMustPhased, pgenlib_misc.cc:1887-1921)
        uintptr_t unphased = 0;
        const Halfword* phasepresent_hw = DowncastKWToHW(phasepresent);
        for (uint32_t widx = 0; widx != founder_ctl2; ++widx) {
          const uintptr_t geno_word = genovec[widx];
          const uintptr_t het_word = geno_word & (~(geno_word >> 1)) & kMask5555;
          unphased |= het_word & (~UnpackHalfwordToWord(phasepresent_hw[widx]));
        }
        if (unlikely(unphased)) {
          logputs("\n");
          logerrprintf("Error: --indep-pairphase: 0-based variant #%u is not fully phased.\n", variant_uidx);
          reterr = kPglRetInconsistentInput;
          break;
        }
        unsigned char* row = &(rows[k * row_bytes]);
        memcpy(row, genovec, code_bytes);
        memset(&(row[code_bytes]), 0, phase_off - code_bytes);
        BitvecAnd(phasepresent, founder_ctl, phaseinfo);
        ZeroTrailingBits(founder_ct, phaseinfo);
        memcpy(&(row[phase_off]), phaseinfo, phase_bytes);
        memset(&(row[phase_off + phase_bytes]), 0, row_bytes - phase_off - phase_bytes);
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
      rc = ldp_load_genotypes(eng, batch_start, n, rows, row_bytes, LDP_MEM_HOST, LDP_GENO_INVERSE | LDP_GENO_PHASED);
      if (!rc) {
        rc = ldp_set_maj_freqs(eng, batch_start, n, maj_freqs.data());
      }
      if (rc) {
        reterr = MapLdpError(rc, eng);
      }
    }
    if (reterr) {
      break;
    }
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
    rc = ldp_run(eng, R_CAST(uint64_t*, removed_variants_collapsed));
    if (rc) {
      reterr = MapLdpError(rc, eng);
      break;
    }
    fputs("done.\n", stdout);
  } while (0);
  aligned_free_cond(rows);
  aligned_free_cond(phaseinfo);
  aligned_free_cond(phasepresent);
  aligned_free_cond(genovec);
  ldp_destroy(eng);
  return reterr;
}

}  // namespace plink2
