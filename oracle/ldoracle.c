/* ldoracle.c -- TEST INFRASTRUCTURE ONLY (see ldoracle.h).
 *
 * Scalar restatement of the reference's --indep-pairwise algorithm.  Every function cites the
 * reference lines (under /root/reference/2.0/) whose behaviour it follows.  Deliberately simple:
 * one 64-bit word at a time, no SIMD, one thread, whole subcontig resident.
 */
#include "ldoracle.h"

#include <stdlib.h>
#include <string.h>

static const double kSmallEpsilon = 0.00000000000005684341886080801486968994140625; /* 2^-44, include/plink2_float.h:119 */

static inline uint32_t popcount64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

/* include/plink2_base.h:904-908 PackWordToHalfwordMask5555: gather the even bits of a word. */
static inline uint32_t pack_even_bits(uint64_t w) {
  w &= 0x5555555555555555ULL;
  w = (w | (w >> 1)) & 0x3333333333333333ULL;
  w = (w | (w >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
  w = (w | (w >> 4)) & 0x00ff00ff00ff00ffULL;
  w = (w | (w >> 8)) & 0x0000ffff0000ffffULL;
  w = (w | (w >> 16)) & 0x00000000ffffffffULL;
  return (uint32_t)w;
}

/* pgenlib_misc.cc:1857-1885: hom = NOT low bit, ref2het = NOT high bit, trailing bits zeroed. */
void ldo_split_hom_ref2het(const uint64_t* geno, uint32_t sample_ct, uint64_t* hom, uint64_t* ref2het) {
  const uint32_t out_word_ct = ldo_word_ct(sample_ct);
  const uint32_t in_word_ct = ldo_geno_word_ct(sample_ct);
  for (uint32_t w = 0; w != out_word_ct; ++w) {
    const uint64_t lo = ~geno[2 * w];
    const uint64_t hi = (2 * w + 1 < in_word_ct) ? ~geno[2 * w + 1] : 0;
    uint64_t hom_word = (uint64_t)pack_even_bits(lo) | ((uint64_t)pack_even_bits(hi) << 32);
    uint64_t r2h_word = (uint64_t)pack_even_bits(lo >> 1) | ((uint64_t)pack_even_bits(hi >> 1) << 32);
    if (w == out_word_ct - 1) {
      const uint32_t rem = sample_ct % 64;
      if (rem) {
        const uint64_t mask = (1ULL << rem) - 1;
        hom_word &= mask;
        r2h_word &= mask;
      }
    }
    hom[w] = hom_word;
    ref2het[w] = r2h_word;
  }
}

/* plink2_ld.cc:725-738 */
void ldo_fill_vaggs(const uint64_t* hom, const uint64_t* ref2het, uint32_t word_ct, LdoVaggs* out) {
  uint32_t hom_ct = 0, ref2het_ct = 0, ref2_ct = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    hom_ct += popcount64(hom[w]);
    ref2het_ct += popcount64(ref2het[w]);
    ref2_ct += popcount64(hom[w] & ref2het[w]);
  }
  const uint32_t alt2_ct = hom_ct - ref2_ct;
  out->nm_ct = alt2_ct + ref2het_ct;
  out->sum = (int32_t)(ref2_ct - alt2_ct);
  out->ssq = hom_ct;
  out->plusone_ct = ref2_ct;
  out->minusone_ct = alt2_ct;
}

/* plink2_ld.cc:902 */
int ldo_is_monomorphic(const LdoVaggs* v) {
  return ((!v->plusone_ct) && (!v->minusone_ct)) || (v->plusone_ct == v->nm_ct) || (v->minusone_ct == v->nm_ct);
}

/* plink2_ld.cc:235-251 (scalar tail loop :243-250 is the definition) */
int32_t ldo_dotprod_words(const uint64_t* hom1, const uint64_t* r2h1, const uint64_t* hom2, const uint64_t* r2h2, uint32_t word_ct) {
  int32_t tot_both = 0;
  uint32_t tot_neg = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    const uint64_t hom_word = hom1[w] & hom2[w];
    const uint64_t xor_word = r2h1[w] ^ r2h2[w];
    tot_both += (int32_t)popcount64(hom_word);
    tot_neg += popcount64(hom_word & xor_word);
  }
  return tot_both - 2 * (int32_t)tot_neg;
}

/* plink2_ld.cc:317-335 */
void ldo_sumssq_words(const uint64_t* hom1, const uint64_t* r2h1, const uint64_t* hom2, const uint64_t* r2h2, uint32_t word_ct, int32_t* sum2, uint32_t* ssq2_out) {
  uint32_t ssq2 = 0, plus2 = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    const uint64_t ssq2_word = (hom1[w] | r2h1[w]) & hom2[w];
    ssq2 += popcount64(ssq2_word);
    plus2 += popcount64(ssq2_word & r2h2[w]);
  }
  *sum2 = (int32_t)(2 * plus2 - ssq2);
  *ssq2_out = ssq2;
}

/* plink2_ld.cc:578-602 */
void ldo_sumssqnm_words(const uint64_t* hom1, const uint64_t* r2h1, const uint64_t* hom2, const uint64_t* r2h2, uint32_t word_ct, uint32_t* nm_out, int32_t* sum2, uint32_t* ssq2_out) {
  uint32_t nm = 0, ssq2 = 0, plus2 = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    const uint64_t nm1_word = hom1[w] | r2h1[w];
    nm += popcount64(nm1_word & (hom2[w] | r2h2[w]));
    const uint64_t ssq2_word = nm1_word & hom2[w];
    ssq2 += popcount64(ssq2_word);
    plus2 += popcount64(ssq2_word & r2h2[w]);
  }
  *nm_out = nm;
  *sum2 = (int32_t)(2 * plus2 - ssq2);
  *ssq2_out = ssq2;
}

/* plink2_ld.cc:699-723, with "A" = the side whose aggregates are passed in by value
 * (cur_nm_ct/cur_sum/cur_ssq) and "B" = the side described by a VariantAggs pointer. */
static void r2_components(const uint64_t* a_hom, const uint64_t* a_r2h, const uint64_t* b_hom, const uint64_t* b_r2h,
                          const LdoVaggs* b_vaggs, uint32_t founder_ct, uint32_t* cur_nm_ct, int32_t* a_sum,
                          uint32_t* a_ssq, int32_t* b_sum, uint32_t* b_ssq, int32_t* dot) {
  const uint32_t word_ct = ldo_word_ct(founder_ct);
  *dot = ldo_dotprod_words(a_hom, a_r2h, b_hom, b_r2h, word_ct);
  if (*cur_nm_ct != founder_ct) {
    ldo_sumssq_words(a_hom, a_r2h, b_hom, b_r2h, word_ct, b_sum, b_ssq);
  } else {
    *b_sum = b_vaggs->sum;
    *b_ssq = b_vaggs->ssq;
  }
  const uint32_t b_nm_ct = b_vaggs->nm_ct;
  if (b_nm_ct == founder_ct) {
    return;
  }
  if (*cur_nm_ct != founder_ct) {
    ldo_sumssqnm_words(b_hom, b_r2h, a_hom, a_r2h, word_ct, cur_nm_ct, a_sum, a_ssq);
  } else {
    ldo_sumssq_words(b_hom, b_r2h, a_hom, a_r2h, word_ct, a_sum, a_ssq);
    *cur_nm_ct = b_nm_ct;
  }
}

/* plink2_ld.cc:1059-1065: second is "A", first is "B". */
void ldo_pair_stats(const uint64_t* first_hom, const uint64_t* first_r2h, const LdoVaggs* first_vaggs,
                    const uint64_t* second_hom, const uint64_t* second_r2h, const LdoVaggs* second_vaggs,
                    uint32_t founder_ct, LdoPairStats* out) {
  uint32_t cur_nm_ct = second_vaggs->nm_ct;
  int32_t cur_second_sum = second_vaggs->sum;
  uint32_t cur_second_ssq = second_vaggs->ssq;
  int32_t first_sum;
  uint32_t first_ssq;
  int32_t dot;
  r2_components(second_hom, second_r2h, first_hom, first_r2h, first_vaggs, founder_ct, &cur_nm_ct, &cur_second_sum, &cur_second_ssq, &first_sum, &first_ssq, &dot);
  out->nm = cur_nm_ct;
  out->sum1 = first_sum;
  out->ssq1 = first_ssq;
  out->sum2 = cur_second_sum;
  out->ssq2 = cur_second_ssq;
  out->dot = dot;
}

/* plink2_ld.cc:1085-1087 */
void ldo_cov_vars(const LdoPairStats* s, double* cov12, double* var1, double* var2) {
  *cov12 = (double)(s->dot * (int64_t)s->nm - (int64_t)s->sum1 * s->sum2);
  *var1 = (double)(s->ssq1 * (int64_t)s->nm - (int64_t)s->sum1 * s->sum1);
  *var2 = (double)(s->ssq2 * (int64_t)s->nm - (int64_t)s->sum2 * s->sum2);
}

/* plink2_ld.cc:1090 (built with -ffp-contract=off like the reference, build_dynamic/Makefile:41) */
int ldo_exceeds(const LdoPairStats* s, double prune_ld_thresh) {
  double cov12, var1, var2;
  ldo_cov_vars(s, &cov12, &var1, &var2);
  return cov12 * cov12 > prune_ld_thresh * var1 * var2;
}

/* plink2_ld.cc:1255 */
double ldo_prune_thresh(double r2) { return r2 * (1 + kSmallEpsilon); }

/* Founder allele counts of a biallelic autosomal variant -> REF frequency as a*(1/t)
 * (plink2_filter.cc:2144-2147; the 2^k dosage scaling of plink2_data.cc:2441-2442 cancels exactly),
 * major allele = REF iff freq >= 0.5 (plink2_common.h:559-567), major frequency via GetAlleleFreq
 * (plink2_common.h:584-593).  All-missing variant: freq 1/2 (plink2_filter.cc:2137-2141). */
void ldo_major_allele(const uint64_t* raw_geno, uint32_t sample_ct, uint32_t* alt_is_major, double* maj_freq) {
  uint64_t n_homref = 0, n_het = 0, n_homalt = 0;
  for (uint32_t s = 0; s != sample_ct; ++s) {
    const uint32_t g = (uint32_t)(raw_geno[s / 32] >> (2 * (s % 32))) & 3;
    n_homref += (g == 0);
    n_het += (g == 1);
    n_homalt += (g == 2);
  }
  const uint64_t ref_ct = 2 * n_homref + n_het;
  const uint64_t alt_ct = 2 * n_homalt + n_het;
  const uint64_t tot = ref_ct + alt_ct;
  double ref_freq;
  if (!tot) {
    ref_freq = 1.0 / 2.0;
  } else {
    const double tot_recip = 1.0 / (double)tot;
    ref_freq = (double)ref_ct * tot_recip;
  }
  if (ref_freq >= 0.5) {
    *alt_is_major = 0;
    *maj_freq = ref_freq;
  } else {
    *alt_is_major = 1;
    double last_freq = 1.0 - ref_freq;
    *maj_freq = (last_freq > 0.0) ? last_freq : 0.0;
  }
}

/* pgenlib_misc.cc:1090-1104: 0 <-> 2, 1 and 3 unchanged. */
void ldo_invert_geno(const uint64_t* raw_geno, uint32_t sample_ct, uint64_t* out) {
  const uint32_t word_ct = ldo_geno_word_ct(sample_ct);
  for (uint32_t w = 0; w != word_ct; ++w) {
    const uint64_t g = raw_geno[w];
    out[w] = g ^ (((~g) << 1) & 0xaaaaaaaaaaaaaaaaULL);
  }
}

/* plink2_ld.cc:2165-2268 with every variant included.  window_max follows :2215-2226 / :2251-2261. */
uint32_t ldo_subcontig_split(const uint32_t* chr_idx, const uint32_t* bps, uint32_t variant_ct, uint32_t prune_window_size, uint32_t* subcontig_info, uint32_t* window_max_out) {
  uint32_t ct = 0;
  uint32_t window_max = 0;
  uint32_t variant_idx = 0;
  if (bps) {
    window_max = 1;
  }
  while (variant_idx < variant_ct) {
    uint32_t chr_end = variant_idx + 1;
    while ((chr_end < variant_ct) && (chr_idx[chr_end] == chr_idx[variant_idx])) {
      ++chr_end;
    }
    const uint32_t chr_variant_ct = chr_end - variant_idx;
    if (chr_variant_ct > 1) {
      if (bps) {
        uint32_t subcontig_idx_first = variant_idx;
        uint32_t window_idx_first = variant_idx;
        uint32_t window_pos_first = bps[variant_idx];
        uint32_t prev_pos = window_pos_first;
        uint32_t vidx = variant_idx + 1;
        do {
          uint32_t variant_bp_thresh = bps[vidx];
          if (variant_bp_thresh < prune_window_size) {
            prev_pos = variant_bp_thresh;
            variant_bp_thresh = 0;
          } else {
            if (variant_bp_thresh - prune_window_size > prev_pos) {
              if (vidx > subcontig_idx_first + 1) {
                subcontig_info[2 * ct] = vidx - subcontig_idx_first;
                subcontig_info[2 * ct + 1] = subcontig_idx_first;
                ++ct;
              }
              subcontig_idx_first = vidx;
            }
            prev_pos = variant_bp_thresh;
            variant_bp_thresh -= prune_window_size;
          }
          if (variant_bp_thresh > window_pos_first) {
            do {
              ++window_idx_first;
              window_pos_first = bps[window_idx_first];
            } while (variant_bp_thresh > window_pos_first);
          } else if (vidx - window_idx_first == window_max) {
            ++window_max;
          }
        } while (++vidx < chr_end);
        if (vidx > subcontig_idx_first + 1) {
          subcontig_info[2 * ct] = vidx - subcontig_idx_first;
          subcontig_info[2 * ct + 1] = subcontig_idx_first;
          ++ct;
        }
      } else {
        subcontig_info[2 * ct] = chr_variant_ct;
        subcontig_info[2 * ct + 1] = variant_idx;
        ++ct;
        if (window_max < prune_window_size) {
          if (chr_variant_ct > window_max) {
            window_max = chr_variant_ct;
          }
        }
      }
    }
    variant_idx = chr_end;
  }
  if ((!bps) && (window_max > prune_window_size)) {
    window_max = prune_window_size;
  }
  if (window_max_out) {
    *window_max_out = window_max;
  }
  return ct;
}

static inline int bit_is_set(const uint64_t* bm, uint32_t i) { return (int)((bm[i / 64] >> (i % 64)) & 1); }
static inline void bit_set(uint64_t* bm, uint32_t i) { bm[i / 64] |= 1ULL << (i % 64); }

typedef struct {
  /* per-variant state for the subcontig being processed (index = variant idx - subcontig first) */
  uint64_t* planes; /* 2*word_ct per variant: hom then ref2het (pairwise) / hap then nm (pairphase) */
  LdoVaggs* vaggs;
  LdoVhaggs* vhaggs;
  int pairphase;
  uint32_t word_ct;
  uint32_t first;
  uint32_t founder_ct; /* pairphase: haplotype count */
  double thresh;
  uint64_t eval_ct;
} SubcontigCtx;

static int pair_exceeds(SubcontigCtx* c, uint32_t first_v, uint32_t second_v) {
  const uint64_t* fp = &c->planes[(uint64_t)(first_v - c->first) * 2 * c->word_ct];
  const uint64_t* sp = &c->planes[(uint64_t)(second_v - c->first) * 2 * c->word_ct];
  ++c->eval_ct;
  if (c->pairphase) {
    LdoHapPairStats hs;
    ldo_hap_pair_stats(fp, fp + c->word_ct, &c->vhaggs[first_v - c->first], sp, sp + c->word_ct, &c->vhaggs[second_v - c->first], c->founder_ct, &hs);
    return ldo_hap_exceeds(&hs, c->thresh);
  }
  LdoPairStats st;
  ldo_pair_stats(fp, fp + c->word_ct, &c->vaggs[first_v - c->first], sp, sp + c->word_ct, &c->vaggs[second_v - c->first], c->founder_ct, &st);
  return ldo_exceeds(&st, c->thresh);
}

/* ---- --indep-pairphase: the same scan over haplotype bit-vectors -------------------------------------------- */

/* include/pgenlib_misc.cc:1887-1931 HapsplitMustPhased.  geno: 2-bit codes (0/1/2 = copies of the counted allele,
 * 3 missing; trailing bits zero); phasepresent/phaseinfo: one bit per sample (phaseinfo only meaningful where
 * phasepresent is set; phasepresent may be NULL = no phase information at all).  hap/nm get 2 bits per sample
 * (hap_ct = 2*sample_ct): hom -> 00 / 11, het -> 01 (unswapped) or 10 (phaseinfo set), missing -> hap 00, nm 00.
 * Returns nonzero when a het call is not covered by phasepresent ("is not fully phased", plink2_ld.cc:2045-2049). */
int ldo_hapsplit_must_phased(const uint64_t* geno, const uint64_t* phasepresent, const uint64_t* phaseinfo, uint32_t sample_ct, uint64_t* hap, uint64_t* nm) {
  const uint32_t word_ct = ldo_geno_word_ct(sample_ct);
  const uint64_t m5 = 0x5555555555555555ULL;
  uint64_t detect_unphased = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    uint64_t g = geno[w];
    if ((w == word_ct - 1) && (sample_ct % 32)) {
      g &= (1ULL << (2 * (sample_ct % 32))) - 1;
    }
    const uint64_t nm_word = 3 * (m5 & (~(g & (g >> 1))));
    const uint64_t g_nm = g & nm_word; /* {00, 01, 10, 00} */
    const uint64_t g_nm_hi = (g_nm >> 1) & m5;
    const uint64_t het = g_nm & m5;
    /* this word covers samples [32w, 32w+32): spread their phase bits to the even positions
     * (UnpackHalfwordToWord, :1913-1914) */
    uint64_t pp = 0, pi = 0;
    if (phasepresent) {
      const uint32_t half = (uint32_t)((phasepresent[w / 2] >> (32 * (w & 1))) & 0xffffffffULL);
      const uint32_t halfi = (uint32_t)((phaseinfo[w / 2] >> (32 * (w & 1))) & 0xffffffffULL);
      for (uint32_t b = 0; b != 32; ++b) {
        pp |= (uint64_t)((half >> b) & 1) << (2 * b);
        pi |= (uint64_t)((halfi >> b) & 1) << (2 * b);
      }
      pi &= pp;
    }
    nm[w] = nm_word;
    /* geno_nm + geno_nm_hi + phaseinfo_word (:1917); without phase information geno_nm | geno_nm_hi (:1899) */
    hap[w] = g_nm + g_nm_hi + pi;
    detect_unphased |= het & ~pp;
  }
  const uint32_t trailing = sample_ct % 32;
  if (trailing) {
    const uint64_t mask = (1ULL << (2 * trailing)) - 1;
    nm[word_ct - 1] &= mask;
    hap[word_ct - 1] &= mask;
  }
  return detect_unphased != 0;
}

/* include/pgenlib_misc.cc:2010-2041 HapsplitHaploid: one haplotype per sample, a het call counts as missing.
 * hap/nm get one bit per sample (hap_ct = sample_ct). */
void ldo_hapsplit_haploid(const uint64_t* geno, uint32_t sample_ct, uint64_t* hap, uint64_t* nm) {
  const uint32_t out_word_ct = ldo_word_ct(sample_ct);
  const uint32_t in_word_ct = ldo_geno_word_ct(sample_ct);
  for (uint32_t w = 0; w != out_word_ct; ++w) {
    const uint64_t g0 = geno[2 * w];
    const uint64_t g1 = (2 * w + 1 < in_word_ct) ? geno[2 * w + 1] : 0;
    /* nm = low bit clear (codes 0, 2); hap = nm and high bit set (code 2) */
    const uint64_t nm0 = ~g0, nm1 = ~g1;
    const uint64_t h0 = nm0 & (g0 >> 1), h1 = nm1 & (g1 >> 1);
    uint64_t nm_word = (uint64_t)pack_even_bits(nm0) | ((uint64_t)pack_even_bits(nm1) << 32);
    uint64_t hap_word = (uint64_t)pack_even_bits(h0) | ((uint64_t)pack_even_bits(h1) << 32);
    if ((w == out_word_ct - 1) && (sample_ct % 64)) {
      const uint64_t mask = (1ULL << (sample_ct % 64)) - 1;
      nm_word &= mask;
      hap_word &= mask;
    }
    nm[w] = nm_word;
    hap[w] = hap_word;
  }
}

/* plink2_ld.cc:1484-1490 FillVhaggs; returns 1 if monomorphic */
int ldo_fill_vhaggs(const uint64_t* hap, const uint64_t* nm, uint32_t word_ct, LdoVhaggs* out) {
  uint32_t nm_ct = 0, sum = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    nm_ct += popcount64(nm[w]);
    sum += popcount64(hap[w]);
  }
  out->nm_ct = nm_ct;
  out->sum = sum;
  return (!sum) || (sum == nm_ct);
}

static uint32_t popcount_intersect(const uint64_t* a, const uint64_t* b, uint32_t word_ct) {
  uint32_t ct = 0;
  for (uint32_t w = 0; w != word_ct; ++w) {
    ct += popcount64(a[w] & b[w]);
  }
  return ct;
}

/* plink2_ld.cc:1456-1481 ComputeIndepPairphaseR2Components, reported as first/second */
void ldo_hap_pair_stats(const uint64_t* first_hap, const uint64_t* first_nm, const LdoVhaggs* first_vh,
                        const uint64_t* second_hap, const uint64_t* second_nm, const LdoVhaggs* second_vh,
                        uint32_t hap_ct, LdoHapPairStats* out) {
  const uint32_t word_ct = ldo_word_ct(hap_ct);
  uint32_t cur_nm_ct = first_vh->nm_ct;
  uint32_t cur_first_sum = first_vh->sum;
  uint32_t second_sum;
  out->dot = popcount_intersect(first_hap, second_hap, word_ct);
  if (cur_nm_ct != hap_ct) {
    second_sum = popcount_intersect(first_nm, second_hap, word_ct);
  } else {
    second_sum = second_vh->sum;
  }
  if (second_vh->nm_ct != hap_ct) {
    cur_first_sum = popcount_intersect(first_hap, second_nm, word_ct);
    if (cur_nm_ct != hap_ct) {
      cur_nm_ct = popcount_intersect(first_nm, second_nm, word_ct);
    } else {
      cur_nm_ct = second_vh->nm_ct;
    }
  }
  out->nm = cur_nm_ct;
  out->sum1 = cur_first_sum;
  out->sum2 = second_sum;
}

/* plink2_ld.cc:1708-1713 (order 1) / :1769-1776 (order 2): the three doubles and the comparison */
int ldo_hap_exceeds(const LdoHapPairStats* s, double prune_ld_thresh) {
  const double cov12 = (double)((int64_t)(s->dot * (uint64_t)s->nm - (uint64_t)s->sum1 * s->sum2));
  const double variance1 = (double)(s->sum1 * (int64_t)(s->nm - s->sum1));
  const double variance2 = (double)(s->sum2 * (int64_t)(s->nm - s->sum2));
  return cov12 * cov12 > prune_ld_thresh * variance1 * variance2;
}

static int prune_scan(int pairphase, const uint64_t* geno, uint64_t stride_words, uint32_t variant_ct, uint32_t founder_ct,
                      const uint32_t* chr_idx, const uint32_t* bps_in, const double* maj_freqs,
                      uint32_t prune_window_size, uint32_t window_incr, int window_is_bp,
                      double r2, int plink1_order, uint64_t* removed, uint64_t* pair_eval_ct);

int ldo_indep_pairwise(const uint64_t* geno, uint64_t stride_words, uint32_t variant_ct, uint32_t founder_ct,
                       const uint32_t* chr_idx, const uint32_t* bps_in, const double* maj_freqs,
                       uint32_t prune_window_size, uint32_t window_incr, int window_is_bp,
                       double r2, int plink1_order, uint64_t* removed, uint64_t* pair_eval_ct) {
  return prune_scan(0, geno, stride_words, variant_ct, founder_ct, chr_idx, bps_in, maj_freqs, prune_window_size, window_incr, window_is_bp, r2, plink1_order, removed, pair_eval_ct);
}

/* plink2_ld.cc:1549-1800 IndepPairphaseThread: the scan of IndepPairwiseThread over haplotype vectors.
 * hap_nm: per variant hap bits then nm bits, ldo_word_ct(hap_ct) words each (the loader_hap_then_nm_vecs rows of
 * :2018-2019), hap bit = haplotype carries the NON-major allele. */
int ldo_indep_pairphase(const uint64_t* hap_nm, uint64_t stride_words, uint32_t variant_ct, uint32_t hap_ct,
                        const uint32_t* chr_idx, const uint32_t* bps_in, const double* maj_freqs,
                        uint32_t prune_window_size, uint32_t window_incr, int window_is_bp,
                        double r2, int plink1_order, uint64_t* removed, uint64_t* pair_eval_ct) {
  return prune_scan(1, hap_nm, stride_words, variant_ct, hap_ct, chr_idx, bps_in, maj_freqs, prune_window_size, window_incr, window_is_bp, r2, plink1_order, removed, pair_eval_ct);
}

static int prune_scan(int pairphase, const uint64_t* geno, uint64_t stride_words, uint32_t variant_ct, uint32_t founder_ct,
                      const uint32_t* chr_idx, const uint32_t* bps_in, const double* maj_freqs,
                      uint32_t prune_window_size, uint32_t window_incr, int window_is_bp,
                      double r2, int plink1_order, uint64_t* removed, uint64_t* pair_eval_ct) {
  const uint32_t* bps = window_is_bp ? bps_in : NULL; /* plink2_ld.cc:2551-2553 */
  memset(removed, 0, ((variant_ct + 63) / 64) * sizeof(uint64_t));
  if (pair_eval_ct) {
    *pair_eval_ct = 0;
  }
  if (!variant_ct) {
    return 0;
  }
  uint32_t* subcontig_info = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)variant_ct);
  uint32_t* win = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)variant_ct);       /* winpos -> variant idx */
  uint8_t* win_removed = (uint8_t*)calloc((size_t)variant_ct + 1, 1);             /* cur_window_removed */
  uint32_t* first_unchecked = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)variant_ct);
  if (!subcontig_info || !win || !win_removed || !first_unchecked) {
    return 1;
  }
  uint32_t window_max;
  const uint32_t subcontig_ct = ldo_subcontig_split(chr_idx, bps, variant_ct, prune_window_size, subcontig_info, &window_max);
  SubcontigCtx ctx;
  ctx.word_ct = ldo_word_ct(founder_ct);
  ctx.founder_ct = founder_ct;
  ctx.pairphase = pairphase;
  ctx.thresh = ldo_prune_thresh(r2);
  ctx.eval_ct = 0;
  for (uint32_t sidx = 0; sidx != subcontig_ct; ++sidx) {
    const uint32_t subcontig_len = subcontig_info[2 * sidx];
    const uint32_t subcontig_first = subcontig_info[2 * sidx + 1];
    const uint32_t subcontig_end = subcontig_first + subcontig_len;
    ctx.first = subcontig_first;
    ctx.planes = (uint64_t*)malloc(sizeof(uint64_t) * 2 * ctx.word_ct * (size_t)subcontig_len);
    ctx.vaggs = (LdoVaggs*)malloc(sizeof(LdoVaggs) * (size_t)subcontig_len);
    ctx.vhaggs = (LdoVhaggs*)malloc(sizeof(LdoVhaggs) * (size_t)subcontig_len);
    if (!ctx.planes || !ctx.vaggs || !ctx.vhaggs) {
      return 1;
    }
    /* LdPruneNextSubcontig, plink2_ld.cc:605-633 */
    uint32_t window_start = subcontig_first;
    uint32_t next_window_end;
    uint32_t winstart_v = subcontig_first;
    uint32_t winend_v = subcontig_first;
    if (bps) {
      const uint32_t variant_bp_thresh = bps[winstart_v] + prune_window_size;
      uint32_t first_window_len = 1;
      do {
        ++winend_v;
      } while ((bps[winend_v] <= variant_bp_thresh) && (++first_window_len < subcontig_len));
      next_window_end = subcontig_first + first_window_len;
    } else {
      next_window_end = subcontig_first + ((subcontig_len < prune_window_size) ? subcontig_len : prune_window_size);
    }
    uint32_t cur_window_size = 0;
    uint32_t winpos_split = 0;
    for (uint32_t cur = subcontig_first; cur < subcontig_end;) {
      /* load one variant, :877-925 */
      uint64_t* hom = &ctx.planes[(uint64_t)(cur - subcontig_first) * 2 * ctx.word_ct];
      int monomorphic;
      if (pairphase) {
        /* :1620-1624 */
        memcpy(hom, &geno[(uint64_t)cur * stride_words], sizeof(uint64_t) * 2 * ctx.word_ct);
        monomorphic = ldo_fill_vhaggs(hom, hom + ctx.word_ct, ctx.word_ct, &ctx.vhaggs[cur - subcontig_first]);
      } else {
        ldo_split_hom_ref2het(&geno[(uint64_t)cur * stride_words], founder_ct, hom, hom + ctx.word_ct);
        LdoVaggs* va = &ctx.vaggs[cur - subcontig_first];
        ldo_fill_vaggs(hom, hom + ctx.word_ct, ctx.word_ct, va);
        monomorphic = ldo_is_monomorphic(va);
      }
      if (monomorphic) {
        win_removed[cur_window_size] = 1;
        bit_set(removed, cur);
      } else {
        first_unchecked[cur] = cur + 1;
      }
      win[cur_window_size++] = cur;
      ++cur;
      if (cur != next_window_end) {
        continue;
      }
      if (plink1_order) {
        /* :931-1037 */
        uint32_t cur_removed_ct = 0;
        for (uint32_t p = 0; p != cur_window_size; ++p) {
          cur_removed_ct += win_removed[p];
        }
        uint32_t prev_removed_ct;
        do {
          prev_removed_ct = cur_removed_ct;
          for (uint32_t first_winpos = 0;; ++first_winpos) {
            while ((first_winpos < cur_window_size) && win_removed[first_winpos]) {
              ++first_winpos;
            }
            if (first_winpos >= cur_window_size) {
              break;
            }
            const uint32_t first_v = win[first_winpos];
            const uint32_t cur_first_unchecked = first_unchecked[first_v];
            if (cur_first_unchecked == cur) {
              continue;
            }
            uint32_t second_winpos = first_winpos;
            int found = 0;
            /* :961-968 */
            while (1) {
              ++second_winpos;
              while ((second_winpos < cur_window_size) && win_removed[second_winpos]) {
                ++second_winpos;
              }
              if (second_winpos >= cur_window_size) {
                first_unchecked[first_v] = cur;
                break;
              }
              if (win[second_winpos] >= cur_first_unchecked) {
                found = 1;
                break;
              }
            }
            if (!found) {
              continue;
            }
            while (1) {
              const uint32_t second_v = win[second_winpos];
              if (pair_exceeds(&ctx, first_v, second_v)) {
                if (maj_freqs[first_v] > maj_freqs[second_v] * (1 + kSmallEpsilon)) {
                  win_removed[first_winpos] = 1;
                  bit_set(removed, first_v);
                } else {
                  win_removed[second_winpos] = 1;
                  bit_set(removed, second_v);
                  uint32_t next_start_winpos = second_winpos + 1;
                  while ((next_start_winpos < cur_window_size) && win_removed[next_start_winpos]) {
                    ++next_start_winpos;
                  }
                  if (next_start_winpos < cur_window_size) {
                    first_unchecked[first_v] = win[next_start_winpos];
                  } else {
                    first_unchecked[first_v] = cur;
                  }
                }
                break;
              }
              ++second_winpos;
              while ((second_winpos < cur_window_size) && win_removed[second_winpos]) {
                ++second_winpos;
              }
              if (second_winpos >= cur_window_size) {
                first_unchecked[first_v] = cur;
                break;
              }
            }
          }
          cur_removed_ct = 0;
          for (uint32_t p = 0; p != cur_window_size; ++p) {
            cur_removed_ct += win_removed[p];
          }
        } while (cur_removed_ct > prev_removed_ct);
      } else {
        /* :1042-1100 */
        const uint32_t second_winpos_stop = winpos_split ? winpos_split : 1;
        for (uint32_t second_winpos = cur_window_size; second_winpos != second_winpos_stop;) {
          --second_winpos;
          const uint32_t second_v = win[second_winpos];
          for (uint32_t first_winpos = second_winpos; first_winpos;) {
            --first_winpos;
            if (win_removed[first_winpos]) {
              continue;
            }
            const uint32_t first_v = win[first_winpos];
            if (pair_exceeds(&ctx, first_v, second_v)) {
              if (maj_freqs[first_v] <= maj_freqs[second_v] * (1 + kSmallEpsilon)) {
                win_removed[second_winpos] = 1;
                bit_set(removed, second_v);
                break;
              }
              win_removed[first_winpos] = 1;
              bit_set(removed, first_v);
            }
          }
        }
      }
      /* LdPruneNextWindow, :635-689 */
      const uint32_t prev_window_size = cur_window_size;
      if (next_window_end == subcontig_end) {
        cur_window_size = 0;
        window_start = subcontig_end;
      } else {
        uint32_t next_window_start = window_start;
        if (bps) {
          const uint32_t window_start_min_bp = bps[winend_v] - prune_window_size;
          uint32_t window_start_bp;
          do {
            ++next_window_start;
            ++winstart_v;
            window_start_bp = bps[winstart_v];
          } while (window_start_bp < window_start_min_bp);
          const uint32_t window_end_thresh = window_start_bp + prune_window_size;
          do {
            if (++next_window_end == subcontig_end) {
              break;
            }
            ++winend_v;
          } while (bps[winend_v] <= window_end_thresh);
        } else {
          next_window_start += window_incr;
          next_window_end = next_window_start + prune_window_size;
          if (next_window_end > subcontig_end) {
            next_window_end = subcontig_end;
          }
        }
        uint32_t winpos_write = 0;
        for (uint32_t winpos_read = 0; winpos_read != cur_window_size; ++winpos_read) {
          const uint32_t v = win[winpos_read];
          if (win_removed[winpos_read] || (v < next_window_start)) {
            continue;
          }
          win[winpos_write++] = v;
        }
        cur_window_size = winpos_write;
        window_start = next_window_start;
      }
      winpos_split = cur_window_size;
      memset(win_removed, 0, prev_window_size);
    }
    free(ctx.planes);
    free(ctx.vaggs);
    free(ctx.vhaggs);
  }
  if (pair_eval_ct) {
    *pair_eval_ct = ctx.eval_ct;
  }
  free(subcontig_info);
  free(win);
  free(win_removed);
  free(first_unchecked);
  return 0;
}
