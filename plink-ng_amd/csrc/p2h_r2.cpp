// p2h_r2.cpp -- plink2-hip: the r^2 outputs: --r2-unphased / --r-unphased matrices and tables (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

// ---- the r^2 outputs: --r2-unphased matrices and tables, --clump ----
// --ld-snp / --ld-snps / --ld-snp-list (VcorTable, plink2_ld.cc:11083-11150): the row variants.  A row variant is
// reported against every variant of its window, on both sides (UpdateVcorWindow :10984 with row_snp_subset), as the
// A of the line; a pair of two row variants appears once, lower index first (:10806-10815).
// Returns one flag per included variant (empty: no row subset).
std::vector<uint8_t> vcor_row_variants(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, uint32_t variant_ct, double thresh) {
  std::vector<uint8_t> is_row;
  if (A.ld_snps.empty() && A.ld_snp_list.empty()) {
    return is_row;
  }
  if (thresh < 0.0) {
    die(63, "Error: a negative --ld-window-r2 with --ld-snp/--ld-snps/--ld-snp-list is not supported by plink2-hip.\n");
  }
  is_row.assign(variant_ct, 0);
  std::unordered_map<std::string, std::vector<uint32_t>> by_id;
  by_id.reserve(static_cast<size_t>(variant_ct) * 2);
  for (uint32_t k = 0; k < variant_ct; ++k) {
    by_id[V.id[inc[k]]].push_back(k);
  }
  if (!A.ld_snp_list.empty()) {  // (TokenExtractExclude, plink2_filter.cc:367: unknown IDs are skipped, every variant carrying a listed ID counts)
    const std::string text = slurp(A.ld_snp_list);
    std::vector<std::string> ids;
    for (size_t p0 = 0; p0 < text.size();) {
      while ((p0 < text.size()) && (static_cast<unsigned char>(text[p0]) <= ' ')) {
        ++p0;
      }
      size_t p1 = p0;
      while ((p1 < text.size()) && (static_cast<unsigned char>(text[p1]) > ' ')) {
        ++p1;
      }
      if (p1 > p0) {
        ids.emplace_back(text, p0, p1 - p0);
      }
      p0 = p1;
    }
    for (const std::string& id : ids) {
      const auto it = by_id.find(id);
      if (it == by_id.end()) {
        continue;
      }
      for (uint32_t k : it->second) {
        is_row[k] = 1;
      }
    }
  }
  for (const auto& pr : A.ld_snps) {  // (InterpretVariantRangeList, plink2_filter.cc:216-271)
    const auto a = by_id.find(pr.first);
    if (a == by_id.end()) {
      die(7, "Error: --ld-snps variant '%s' not found.\n", pr.first.c_str());
    }
    if (pr.second.empty()) {
      for (uint32_t k : a->second) {
        is_row[k] = 1;
      }
      continue;
    }
    if (a->second.size() > 1) {
      die(7, "Error: --ld-snps range-starting variant ID '%s' appears multiple times.\n", pr.first.c_str());
    }
    const auto b = by_id.find(pr.second);
    if (b == by_id.end()) {
      die(7, "Error: --ld-snps variant '%s' not found.\n", pr.second.c_str());
    }
    if (b->second.size() > 1) {
      die(7, "Error: --ld-snps range-ending variant ID '%s' appears multiple times.\n", pr.second.c_str());
    }
    const uint32_t k0 = std::min(a->second[0], b->second[0]), k1 = std::max(a->second[0], b->second[0]);
    for (uint32_t k = k0; k <= k1; ++k) {
      is_row[k] = 1;
    }
  }
  return is_row;
}

// What the two writers of the r^2 outputs share (run_r2 sets it up: engine planned and fed, sex chromosomes prepared)
struct R2Job {
  Session& S;
  ldp_engine* e = nullptr;
  uint32_t shard_first = 0, shard_end = 0;  // --parallel k n: this piece's rows
  std::string piece_suffix, base;
  std::vector<uint8_t> is_x;                // per engine row: a chrX variant whose pairs take the male-weighted sums
  bool any_x = false;
  XWeighted xw;
  std::unordered_map<uint32_t, std::pair<uint32_t, double>> multi_maj;  // multiallelic variant -> (major allele, its frequency)
  std::vector<uint8_t> x_maj_alt;           // chrX-aware major allele (MAJ / NONMAJ columns)
  std::vector<double> x_maj_freq;
  explicit R2Job(Session& s) : S(s) {}
  // the entries of dense rows [r0, r0 + rows) x columns [c0, c0 + cols) (second variant j = row, first variant i = column,
  // i < j) that involve chrX, recomputed in place
  void x_fix_dense(void* buf, bool as_float, uint32_t r0, uint32_t rows, uint32_t c0, uint32_t cols, uint64_t ld) const {
    if (!any_x) {
      return;
    }
    if (g_dbg.x_host) {  // (test hook --debug-x-host: pair lists through ldp_pair_stats and the host arithmetic, as the band writers do)
      std::vector<uint32_t> fi, se;
      std::vector<double> vals;
      for (uint32_t q = 0; q < rows; ++q) {
        const uint32_t j = r0 + q;
        for (uint32_t i = c0; i < std::min(j, c0 + cols); ++i) {
          if (is_x[i] || is_x[j]) {
            fi.push_back(i);
            se.push_back(j);
          }
        }
      }
      xw.pairs(fi, se, &vals);
      for (size_t q = 0; q < fi.size(); ++q) {
        const uint64_t idx = static_cast<uint64_t>(se[q] - r0) * ld + (fi[q] - c0);
        if (as_float) {
          static_cast<float*>(buf)[idx] = static_cast<float>(vals[q]);
        } else {
          static_cast<double*>(buf)[idx] = vals[q];
        }
      }
      return;
    }
    // both engines' tuples of the block's chrX rows / columns from the pair kernels, combined on the device (ldp_r2_unphased_block_x)
    if (ldp_r2_unphased_block_x(xw.all, xw.male, xw.is_x.data(), xw.flip_all.empty() ? nullptr : xw.flip_all.data(), xw.flip_male.empty() ? nullptr : xw.flip_male.data(),
                                r0, rows, c0, cols, as_float ? 1 : 0, xw.unsquared ? 1 : 0, buf, ld)) {
      die(16, "Error: %s\n", ldp_last_error(xw.all));
    }
  }
};

// The column set of the .vcor table (VcorTable :11250-11390, VcorTableWriteThread :10836-10960): what each variant prints
// in front of the r^2, and the header line.
struct VcorColumns {
  const R2Job& J;
  const Args& A;
  const Variants& V;
  const std::vector<uint32_t>& inc;
  const std::vector<uint32_t>& bps;
  uint32_t cols = 0;
  std::vector<uint8_t> prov_bits;
  bool prov_all = false, provref_col = false;
  std::vector<uint8_t> maj_allele;
  std::vector<double> nonmaj_freq;
  explicit VcorColumns(const R2Job& job) : J(job), A(job.S.A), V(job.S.V), inc(job.S.inc), bps(job.S.bps) {
    ldp_engine* const e = J.e;
    ldp_pgen* const pg = J.S.pg;
    const uint32_t variant_ct = J.S.variant_ct, raw_variant_ct = J.S.raw_variant_ct;
    const std::vector<uint8_t>& is_x = J.is_x;
    const auto& multi_maj = J.multi_maj;
    const std::vector<uint8_t>& x_maj_alt = J.x_maj_alt;
    const std::vector<double>& x_maj_freq = J.x_maj_freq;
    cols = A.r2_cols;
    if (cols & kVcorColRef) {  // ProvrefCol (plink2_common.h:1549): 'provref' always, 'maybeprovref' when some included variant is flagged
      prov_bits.assign((static_cast<size_t>(raw_variant_ct) + 7) / 8, 0);
      int storage = ldp_pgen_provisional_ref(pg, prov_bits.data(), prov_bits.size());
      if ((storage == 0) && V.info_pr_header) {  // the .pgen leaves it to the .pvar's INFO/PR
        storage = 3;
        std::copy(V.info_pr.begin(), V.info_pr.begin() + std::min(V.info_pr.size(), prov_bits.size()), prov_bits.begin());
      }
      prov_all = (storage == 2);
      if (cols & kVcorColProvref) {
        provref_col = true;
      } else if (cols & kVcorColMaybeprovref) {
        provref_col = prov_all;
        for (uint32_t k = 0; (storage == 3) && (!provref_col) && (k < variant_ct); ++k) {
          provref_col = (prov_bits[inc[k] >> 3] >> (inc[k] & 7)) & 1;
        }
      }
    }
    // major allele and non-major frequency per variant (the allele-frequency pass: plink2_filter.cc:2137-2147, GetMajIdx)
    if (cols & (kVcorColMaj | kVcorColNonmaj | kVcorColFreq)) {
      std::vector<ldp_variant_rec> recs(variant_ct);
      if (variant_ct && ldp_get_variant_recs(e, 0, variant_ct, recs.data())) {
        die(16, "Error: %s\n", ldp_last_error(e));
      }
      maj_allele.assign(variant_ct, 0);
      nonmaj_freq.assign(variant_ct, 0.0);
      for (uint32_t k = 0; k < variant_ct; ++k) {
        const auto it = multi_maj.find(k);
        double maj_freq;
        if (it != multi_maj.end()) {  // (several ALT alleles, on chrX too: the allele-frequency pass's own major allele)
          maj_allele[k] = static_cast<uint8_t>(it->second.first);
          maj_freq = it->second.second;
        } else if (is_x[k]) {
          maj_allele[k] = x_maj_alt[k];
          maj_freq = x_maj_freq[k];
        } else {
          const uint64_t ref_ct = 2ull * recs[k].n_homref + recs[k].n_het, alt_ct = 2ull * recs[k].n_homalt + recs[k].n_het, tot = ref_ct + alt_ct;
          double ref_freq = 0.5;
          if (tot) {
            ref_freq = static_cast<double>(ref_ct) * (1.0 / static_cast<double>(tot));
          }
          maj_allele[k] = (ref_freq >= 0.5) ? 0 : 1;
          maj_freq = maj_allele[k] ? (1.0 - ref_freq) : ref_freq;  // GetAlleleFreq: the last allele's frequency is 1 - the others
        }
        nonmaj_freq[k] = 1.0 - maj_freq;
      }
    }
    // one variant's columns, each followed by a tab
  }
  void allele_text(uint32_t k, uint32_t allele, std::string* out) const {
    const uint32_t v = inc[k];
    if (!allele) {
      *out += V.ref[v];
      return;
    }
    const std::string& alt = V.alt[v];
    size_t p0 = 0;
    for (uint32_t a = 1; a < allele; ++a) {
      p0 = alt.find(',', p0) + 1;
    }
    out->append(alt, p0, std::min(alt.find(',', p0), alt.size()) - p0);
  }
  // one variant's columns, each followed by a tab
  void put(uint32_t k, const std::string& chr_name, std::string* out) const {
    char num[40];
    if (cols & kVcorColChrom) {
      *out += chr_name;
      *out += '\t';
    }
    if (cols & kVcorColPos) {
      *out += std::to_string(bps[k]);
      *out += '\t';
    }
    if (cols & kVcorColId) {
      *out += V.id[inc[k]];
      *out += '\t';
    }
    if (cols & kVcorColRef) {
      *out += V.ref[inc[k]];
      *out += '\t';
    }
    if (cols & kVcorColAlt1) {
      allele_text(k, 1, out);
      *out += '\t';
    }
    if (cols & kVcorColAlt) {
      *out += V.alt[inc[k]];
      *out += '\t';
    }
    if (provref_col) {
      *out += (prov_all || ((!prov_bits.empty()) && ((prov_bits[inc[k] >> 3] >> (inc[k] & 7)) & 1))) ? 'Y' : 'N';
      *out += '\t';
    }
    if (cols & kVcorColMaj) {
      allele_text(k, maj_allele[k], out);
      *out += '\t';
    }
    if (cols & kVcorColNonmaj) {
      const uint32_t allele_ct = static_cast<uint32_t>(V.alt_ct[inc[k]]) + 1;
      for (uint32_t a = 0; a < allele_ct; ++a) {
        if (a != maj_allele[k]) {
          allele_text(k, a, out);
          *out += ',';
        }
      }
      out->back() = '\t';
    }
    if (cols & kVcorColFreq) {
      out->append(num, format_g6(nonmaj_freq[k], num) - num);
      *out += '\t';
    }
  }
  std::string header() const {
  std::string hdr = "#";
  for (const char side : {'A', 'B'}) {
    const std::pair<uint32_t, const char*> names[] = {{kVcorColChrom, "CHROM_"}, {kVcorColPos, "POS_"}, {kVcorColId, "ID_"}, {kVcorColRef, "REF_"},
                                                      {kVcorColAlt1, "ALT1_"}, {kVcorColAlt, "ALT_"}, {0, "PROVISIONAL_REF_"}, {kVcorColMaj, "MAJ_"},
                                                      {kVcorColNonmaj, "NONMAJ_"}, {kVcorColFreq, "NONMAJ_FREQ_"}};
    for (const auto& nm : names) {
      if (nm.first ? ((cols & nm.first) != 0) : provref_col) {
        hdr += nm.second;
        hdr += side;
        if (!nm.first) {
          hdr += '?';
        }
        hdr += '\t';
      }
    }
  }
  hdr += A.r_unsquared ? "UNPHASED_R\n" : "UNPHASED_R2\n";
    return hdr;
  }
};

// ---- the .vcor table: windowed (VcorTable, plink2_ld.cc:11025) or inter-chr ----
int write_vcor_table(R2Job& J) {
  Session& S = J.S;
  const Args& A = S.A;
  const Variants& V = S.V;
  ldp_engine* const e = J.e;
  ldp_pgen* const pg = S.pg;
  const std::vector<uint32_t>&inc = S.inc, &chr_idx = S.chr_idx, &bps = S.bps;
  const uint32_t variant_ct = S.variant_ct, raw_variant_ct = S.raw_variant_ct;
  const uint32_t shard_first = J.shard_first, shard_end = J.shard_end;
  const std::string &piece_suffix = J.piece_suffix, &base = J.base;
  const std::vector<uint8_t>& is_x = J.is_x;
  const bool any_x = J.any_x;
  const auto& multi_maj = J.multi_maj;
  const std::vector<uint8_t>& x_maj_alt = J.x_maj_alt;
  const std::vector<double>& x_maj_freq = J.x_maj_freq;
  auto x_pairs_r2 = [&](const std::vector<uint32_t>& first, const std::vector<uint32_t>& second, std::vector<double>* out) { J.xw.pairs(first, second, out); };
  auto x_fix_dense = [&](void* buf, bool as_float, uint32_t r0, uint32_t rows, uint32_t c0, uint32_t cols, uint64_t ld) {
    J.x_fix_dense(buf, as_float, r0, rows, c0, cols, ld);
  };
  (void)raw_variant_ct; (void)x_pairs_r2; (void)x_fix_dense; (void)multi_maj; (void)x_maj_alt; (void)x_maj_freq; (void)chr_idx; (void)bps; (void)base; (void)pg;
  // ---- windowed table (VcorTable, plink2_ld.cc:11025): one line per pair A < B inside the window whose r^2 passes
  //      --ld-window-r2, A-major; default column set (plink2_ld.h:101)
  std::vector<uint32_t> lo(std::max<uint32_t>(variant_ct, 1));
  uint64_t cand = 0;
  if (!A.r2_inter) {
    ldp_get_band(e, lo.data(), &cand);
  }
  // hi[i] = last second variant paired with i (lo is nondecreasing inside a chromosome and == j outside windows)
  std::vector<uint32_t> hi(variant_ct);
  if (!A.r2_inter) {
    uint32_t j = 0;
    for (uint32_t i = 0; i < variant_ct; ++i) {
      j = std::max(j, i);
      while ((j + 1 < variant_ct) && (lo[j + 1] <= i) && (chr_idx[j + 1] == chr_idx[i])) {
        ++j;
      }
      hi[i] = j;
    }
  }
  // names as the reference prints them (chrtoa with the default --output-chr: bare numbers, XY/PAR1/PAR2, contig names)
  auto chrom_out = [&](const std::string& raw) {
    std::string name = raw;
    if (name.size() > 3 && (name[0] | 32) == 'c' && (name[1] | 32) == 'h' && (name[2] | 32) == 'r') {
      bool zero = false;
      const std::string rest = name.substr(3);
      bool numeric = !rest.empty();
      for (char c : rest) {
        numeric = numeric && (c >= '0' && c <= '9');
      }
      if (numeric || ieq(rest.c_str(), "XY") || ieq(rest.c_str(), "PAR1") || ieq(rest.c_str(), "PAR2")) {
        name = rest;
      }
      (void)zero;
    }
    bool numeric = !name.empty();
    for (char c : name) {
      numeric = numeric && (c >= '0' && c <= '9');
    }
    if (numeric) {
      const long v = strtol(name.c_str(), nullptr, 10);
      return (v == 25) ? std::string("XY") : std::to_string(v);
    }
    if (ieq(name.c_str(), "XY")) return std::string("XY");
    if (ieq(name.c_str(), "PAR1")) return std::string("PAR1");
    if (ieq(name.c_str(), "PAR2")) return std::string("PAR2");
    return name;
  };
  const std::string tpath = A.out + ".vcor" + piece_suffix + (A.r2_zs ? ".zst" : "");
  OutFile tf;
  tf.open(tpath, A.r2_zs);
  const VcorColumns columns(J);
  auto put_variant = [&](uint32_t k, const std::string& chr_name, std::string* out) { columns.put(k, chr_name, out); };
  if (A.parallel_idx == 0) {
    const std::string hdr = columns.header();
    tf.write(hdr.data(), hdr.size());
  }
  // (--r-unphased filters |r| against the root of --ld-window-r2: VcorTable :11575-11579)
  const double thresh = A.r_unsquared ? ((A.ld_min_r2 < 0.0) ? -1.0 : sqrt(A.ld_min_r2)) : A.ld_min_r2;
  const std::vector<uint8_t> is_row = vcor_row_variants(A, V, inc, variant_ct, thresh);
  const bool row_subset = !is_row.empty();
  if (A.r2_inter || (thresh > 0.0) || row_subset) {
    // ---- inter-chr: every pair A < B of the whole variant set, chromosome 0 included (plink2_ld.cc:11082-11116).
    // The r^2 values come row chunk by row chunk (second variant B) from the all-pairs plan; pairs that pass
    // --ld-window-r2 are kept as (A, B, r^2) and bucketed by A afterwards, which gives the file's A-major order.
    // ---- windowed table with a positive threshold (the default): the same, over the band's pairs.
    struct Hit {
      uint32_t i, j;
      double r2;
    };
    std::vector<Hit> hits;
    std::vector<double> chunk;
    const uint32_t nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    // With a positive threshold the filter runs in the kernel's epilogue (ldp_r2_unphased_hits) and only the
    // passing pairs cross PCIe; a row chunk whose hits overflow the buffer is redone through the dense path below.
    const bool device_filter = (thresh > 0.0) || row_subset;  // (threshold 0: every defined r^2 passes, NaN does not, :10816)
    std::vector<ldp_r2_hit> dev_hits(device_filter ? (1u << 24) : 0);
    uint32_t big_rows = 65536;
    // (a shard owns the pairs whose FIRST variant lies in [shard_first, shard_end): second variants from shard_first + 1 on)
    for (uint32_t r0 = (A.parallel_tot == 1) ? 0 : shard_first; r0 < variant_ct;) {
      uint32_t rows = static_cast<uint32_t>(std::max<uint64_t>(32, (1ull << 28) / (static_cast<uint64_t>(r0 + 4096) * 8)));
      rows = std::min(std::min(rows, variant_ct - r0), 65536u);
      if (device_filter) {
        const uint32_t big = A.r2_inter ? std::min(std::min<uint32_t>(rows * 16, variant_ct - r0), 65536u)  // (no dense buffer to size)
                                        : std::min(big_rows, variant_ct - r0);
        uint64_t found = 0;
        if ((A.r2_inter && (A.parallel_tot != 1))
                ? ldp_r2_unphased_block_hits(e, r0, big, shard_first, shard_end - shard_first, thresh, dev_hits.data(), dev_hits.size(), &found)
                : ldp_r2_unphased_hits(e, r0, big, thresh, dev_hits.data(), dev_hits.size(), &found)) {
          die(16, "Error: %s\n", ldp_last_error(e));
        }
        if (found <= dev_hits.size()) {
          std::vector<Hit> fresh;
          for (uint64_t q = 0; q < found; ++q) {
            if ((dev_hits[q].first >= shard_first) && (dev_hits[q].first < shard_end) && !(is_x[dev_hits[q].first] || is_x[dev_hits[q].second])) {
              fresh.push_back({dev_hits[q].first, dev_hits[q].second, dev_hits[q].r2});
            }
          }
          bool x_done = false;
          if (any_x && A.r2_inter && (thresh >= 0.0) && !g_dbg.x_host) {
            // all-pairs plan: the chunk's pairs with a chrX variant from the pair kernels too, weighted and filtered on the device
            // (a chunk whose passing pairs do not fit the buffer goes through the lists below)
            uint64_t x_found = 0;
            const XWeighted& xw = J.xw;
            if (ldp_r2_unphased_block_x_hits(xw.all, xw.male, xw.is_x.data(), xw.flip_all.empty() ? nullptr : xw.flip_all.data(),
                                             xw.flip_male.empty() ? nullptr : xw.flip_male.data(), r0, big, shard_first, shard_end - shard_first, xw.unsquared ? 1 : 0,
                                             thresh, dev_hits.data(), dev_hits.size(), &x_found)) {
              die(16, "Error: %s\n", ldp_last_error(xw.all));
            }
            if (x_found <= dev_hits.size()) {
              for (uint64_t q = 0; q < x_found; ++q) {
                fresh.push_back({dev_hits[q].first, dev_hits[q].second, dev_hits[q].r2});
              }
              x_done = true;
            }
          }
          if (any_x && (!x_done) && (!A.r2_inter) && (thresh >= 0.0) && J.xw.band_ready()) {  // (a negative --ld-window-r2 keeps NaN pairs: the host list path below)
            // windowed plan: the chrX run's own all-pairs engines give the windows' pairs on the pair kernels, weighted and filtered on the device
            const XWeighted& xw = J.xw;
            const uint32_t j0 = std::max(r0, xw.band_first), j1 = std::min(r0 + big, xw.band_first + xw.band_ct);
            if (j0 < j1) {
              xw.band_hits(lo.data(), j0, j1, thresh, &dev_hits, [&](uint32_t i, uint32_t j, double v) {
                if ((i >= shard_first) && (i < shard_end)) {
                  fresh.push_back({i, j, v});
                }
              });
            }
            x_done = true;
          }
          if (any_x && !x_done) {  // the chunk's pairs with a chrX variant: values and filter on the host
            std::vector<uint32_t> fi, se;
            std::vector<double> vals;
            for (uint32_t j = r0; j < r0 + big; ++j) {
              const uint32_t i0 = A.r2_inter ? shard_first : std::max(lo[j], shard_first);
              const uint32_t i1 = std::min(j, shard_end);
              for (uint32_t i = i0; i < i1; ++i) {
                if (is_x[i] || is_x[j]) {
                  fi.push_back(i);
                  se.push_back(j);
                }
              }
            }
            x_pairs_r2(fi, se, &vals);
            for (size_t q = 0; q < fi.size(); ++q) {
              if ((thresh >= 0.0) && (!(fabs(vals[q]) >= thresh))) {
                continue;
              }
              fresh.push_back({fi[q], se[q], vals[q]});
            }
          }
          std::sort(fresh.begin(), fresh.end(), [](const Hit& a, const Hit& b) { return (a.j != b.j) ? (a.j < b.j) : (a.i < b.i); });
          hits.insert(hits.end(), fresh.begin(), fresh.end());
          r0 += big;
          continue;
        }
        if (!A.r2_inter) {
          if (big == 1) {
            die(2, "Error: one variant has more passing partners than the filter buffer holds.\n");
          }
          big_rows = std::max(1u, big / 2);  // more hits than the buffer holds: fewer second variants per call
          continue;
        }
      }
      const uint64_t ld = static_cast<uint64_t>(r0) + rows;
      chunk.assign(static_cast<size_t>(rows) * ld, 0.0);
      if (ldp_r2_unphased_rows(e, r0, rows, 0, chunk.data(), ld)) {
        die(16, "Error: %s\n", ldp_last_error(e));
      }
      x_fix_dense(chunk.data(), false, r0, rows, 0, static_cast<uint32_t>(ld), ld);
      std::vector<std::vector<Hit>> part(nthreads);
      std::vector<std::thread> pool;
      for (uint32_t t = 0; t < nthreads; ++t) {
        pool.emplace_back([&, t]() {
          const uint32_t q0 = static_cast<uint32_t>(static_cast<uint64_t>(rows) * t / nthreads);
          const uint32_t q1 = static_cast<uint32_t>(static_cast<uint64_t>(rows) * (t + 1) / nthreads);
          for (uint32_t q = q0; q < q1; ++q) {
            const uint32_t j = r0 + q;
            const double* row = chunk.data() + static_cast<uint64_t>(q) * ld;
            for (uint32_t i = shard_first; (i < j) && (i < shard_end); ++i) {
              const double r2 = row[i];
              if ((thresh >= 0.0) && (!(fabs(r2) >= thresh))) {  // VcorTableWriteThread :10816-10821
                continue;
              }
              part[t].push_back({i, j, r2});
            }
          }
        });
      }
      for (std::thread& th : pool) {
        th.join();
      }
      for (const std::vector<Hit>& v : part) {
        hits.insert(hits.end(), v.begin(), v.end());
      }
      r0 += rows;
    }
    if (row_subset) {
      // the row variant becomes the A of each line; pairs without one drop out; lines run by (A, B)
      size_t kept = 0;
      for (const Hit& h : hits) {
        if (is_row[h.i]) {
          hits[kept++] = h;
        } else if (is_row[h.j]) {
          hits[kept++] = {h.j, h.i, h.r2};
        }
      }
      hits.resize(kept);
      std::sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return (a.i != b.i) ? (a.i < b.i) : (a.j < b.j); });
    }
    // stable bucket by first variant (second variants arrive in increasing order)
    std::vector<uint64_t> start(static_cast<size_t>(variant_ct) + 1, 0);
    for (const Hit& h : hits) {
      ++start[h.i + 1];
    }
    for (uint32_t i = 0; i < variant_ct; ++i) {
      start[i + 1] += start[i];
    }
    std::vector<Hit> sorted(hits.size());
    {
      std::vector<uint64_t> cursor(start.begin(), start.end() - 1);
      for (const Hit& h : hits) {
        sorted[cursor[h.i]++] = h;
      }
    }
    std::vector<Hit>().swap(hits);
    std::vector<std::string> chr_name;  // by chromosome order index
    for (uint32_t k = 0; k < variant_ct; ++k) {
      if (chr_idx[k] >= chr_name.size()) {
        chr_name.resize(chr_idx[k] + 1);
        chr_name[chr_idx[k]] = chrom_out(V.chrom[inc[k]]);
      }
    }
    std::string out;
    out.reserve(1 << 22);
    char num[40];
    for (const Hit& h : sorted) {
      put_variant(h.i, chr_name[chr_idx[h.i]], &out);
      put_variant(h.j, chr_name[chr_idx[h.j]], &out);
      out.append(num, format_g6(h.r2, num) - num);
      out += '\n';
      if (out.size() > (1u << 21)) {
        tf.write(out.data(), out.size());
        out.clear();
      }
    }
    tf.write(out.data(), out.size());
    tf.close();
    logprintf("--r2-unphased: %llu variant pair%s written to %s .\n", static_cast<unsigned long long>(sorted.size()), sorted.size() == 1 ? "" : "s", tpath.c_str());
    ldp_destroy(e);
    ldp_pgen_close(pg);
    if (g_log) {
      fclose(g_log);
    }
    return 0;
  }
  std::vector<double> band;
  std::vector<uint64_t> off;
  std::string linebuf;
  linebuf.reserve(1 << 22);
  std::string chr_a_name;
  uint32_t chr_a_idx = 0xffffffffu;
  uint64_t written = 0;
  const uint64_t kMaxPairs = 1ull << 25;  // 256 MiB of doubles per chunk
  for (uint32_t a0 = shard_first; a0 < shard_end;) {
    // first variants [a0, a1): their partners are the second variants (a0, hi[a1-1]]
    uint32_t a1 = a0;
    uint64_t pairs = 0;
    uint32_t row_end = a0 + 1;
    while (a1 < shard_end) {
      const uint32_t new_end = std::max(row_end, hi[a1] + 1);
      uint64_t add = 0;
      for (uint32_t j = row_end; j < new_end; ++j) {
        add += j - lo[j];
      }
      if ((a1 > a0) && (pairs + add > kMaxPairs)) {
        break;
      }
      pairs += add;
      row_end = new_end;
      ++a1;
    }
    const uint32_t row_first = a0;
    const uint32_t row_ct = row_end - row_first;
    off.assign(static_cast<size_t>(row_ct) + 1, 0);
    for (uint32_t q = 0; q < row_ct; ++q) {
      off[q + 1] = off[q] + ((row_first + q) - lo[row_first + q]);
    }
    band.resize(std::max<uint64_t>(off[row_ct], 1));
    if (off[row_ct] && ldp_r2_unphased_band_rows(e, row_first, row_ct, 0, band.data(), off[row_ct])) {
      die(16, "Error: %s\n", ldp_last_error(e));
    }
    if (any_x && J.xw.band_ready()) {  // (a window never leaves its chromosome: the pairs of the chrX rows, from the run's own all-pairs engines)
      const XWeighted& xw = J.xw;
      const uint32_t j0 = std::max(row_first, xw.band_first), j1 = std::min(row_first + row_ct, xw.band_first + xw.band_ct);
      if (j0 < j1) {
        xw.band_dense(lo.data(), j0, j1, [&](uint32_t i, uint32_t j, double v) { band[off[j - row_first] + (i - lo[j])] = v; });
      }
    } else if (any_x) {
      std::vector<uint32_t> fi, se;
      std::vector<double> vals;
      for (uint32_t q = 0; q < row_ct; ++q) {
        const uint32_t j = row_first + q;
        for (uint32_t i = lo[j]; is_x[j] && (i < j); ++i) {
          fi.push_back(i);
          se.push_back(j);
        }
      }
      x_pairs_r2(fi, se, &vals);
      for (size_t q = 0; q < fi.size(); ++q) {
        band[off[se[q] - row_first] + (fi[q] - lo[se[q]])] = vals[q];
      }
    }
    char num[40];
    for (uint32_t i = a0; i < a1; ++i) {
      if (chr_idx[i] != chr_a_idx) {
        chr_a_idx = chr_idx[i];
        chr_a_name = chrom_out(V.chrom[inc[i]]);
      }
      for (uint32_t j = i + 1; j <= hi[i]; ++j) {
        const double r2 = band[off[j - row_first] + (i - lo[j])];
        if ((thresh >= 0.0) && (!(fabs(r2) >= thresh))) {  // VcorTableWriteThread :10816-10821 (NaN never passes)
          continue;
        }
        put_variant(i, chr_a_name, &linebuf);
        put_variant(j, chr_a_name, &linebuf);  // same chromosome: the table never pairs across chromosomes without inter-chr
        linebuf.append(num, format_g6(r2, num) - num);
        linebuf += '\n';
        ++written;
      }
      if (linebuf.size() > (1u << 21)) {
        tf.write(linebuf.data(), linebuf.size());
        linebuf.clear();
      }
    }
    a0 = a1;
  }
  tf.write(linebuf.data(), linebuf.size());
  tf.close();
  logprintf("--r2-unphased: %llu variant pair%s written to %s .\n", static_cast<unsigned long long>(written), written == 1 ? "" : "s", tpath.c_str());
  ldp_destroy(e);
  ldp_pgen_close(pg);
  if (g_log) {
    fclose(g_log);
  }
  return 0;
}

// ---- the matrix shapes (VcorMatrix, plink2_ld.cc:9766): square / square0 / triangle as bin, bin4 or text ----
int write_vcor_matrix(R2Job& J) {
  Session& S = J.S;
  const Args& A = S.A;
  ldp_engine* const e = J.e;
  ldp_pgen* const pg = S.pg;
  const uint32_t variant_ct = S.variant_ct;
  const uint32_t shard_first = J.shard_first, shard_end = J.shard_end;
  const std::string &piece_suffix = J.piece_suffix, &base = J.base;
  auto x_fix_dense = [&](void* buf, bool as_float, uint32_t r0, uint32_t rows, uint32_t c0, uint32_t cols, uint64_t ld) {
    J.x_fix_dense(buf, as_float, r0, rows, c0, cols, ld);
  };
  const size_t esz = A.r2_float ? 4 : 8;
  const std::string mpath = base + piece_suffix + ((A.r2_text && A.r2_zs) ? ".zst" : "");
  OutFile mf;
  mf.open(mpath, A.r2_text && A.r2_zs);
  // square needs the mirrored upper triangle: the shard's rows, full width, in host memory.  The lower part of row j comes
  // from the engine's row j; the upper part (columns i > j) from the column block [shard rows] of the later rows i.
  const uint32_t piece_rows = shard_end - shard_first;
  std::vector<uint8_t> full;
  if (A.r2_shape == 0) {
    full.assign(static_cast<size_t>(piece_rows) * variant_ct * esz, 0);
  }
  std::vector<uint8_t> chunk;
  std::string textbuf;
  for (uint32_t r0 = shard_first; r0 < shard_end;) {
    // rows per chunk: about 1 GiB of output
    uint32_t rows = static_cast<uint32_t>(std::max<uint64_t>(32, (1ull << 30) / (static_cast<uint64_t>(r0 + 4096) * esz)));
    rows = std::min(std::min(rows, shard_end - r0), 65536u);
    const uint64_t ld = static_cast<uint64_t>(r0) + rows;
    chunk.assign(static_cast<size_t>(rows) * ld * esz, 0);
    if (ldp_r2_unphased_rows(e, r0, rows, A.r2_float, chunk.data(), ld)) {
      die(16, "Error: %s\n", ldp_last_error(e));
    }
    x_fix_dense(chunk.data(), A.r2_float != 0, r0, rows, 0, static_cast<uint32_t>(ld), ld);
    for (uint32_t q = 0; q < rows; ++q) {
      const uint32_t j = r0 + q;
      const uint8_t* row = chunk.data() + static_cast<uint64_t>(q) * ld * esz;
      if (A.r2_text && (A.r2_shape != 0)) {
        // VcorMatrixWriteThread :9733-9752: dtoa_g values, tab-separated, square0 padded with "0" entries
        const double* drow = reinterpret_cast<const double*>(row);
        textbuf.clear();
        char num[40];
        for (uint32_t i = 0; i <= j; ++i) {
          textbuf.append(num, format_g6(drow[i], num) - num);
          textbuf += '\t';
        }
        if (A.r2_shape == 1) {
          for (uint32_t i = j + 1; i < variant_ct; ++i) {
            textbuf += "0\t";
          }
        }
        textbuf.back() = '\n';
        mf.write(textbuf.data(), textbuf.size());
      } else if (A.r2_shape == 2) {
        mf.write(row, esz * (static_cast<size_t>(j) + 1));
      } else if (A.r2_shape == 1) {
        mf.write(row, esz * (static_cast<size_t>(j) + 1));
        static const std::vector<uint8_t> zeros(1 << 20, 0);
        for (uint64_t left = (static_cast<uint64_t>(variant_ct) - j - 1) * esz; left;) {
          const size_t w = static_cast<size_t>(std::min<uint64_t>(left, zeros.size()));
          mf.write(zeros.data(), w);
          left -= w;
        }
      } else {
        memcpy(full.data() + static_cast<uint64_t>(j - shard_first) * variant_ct * esz, row, (static_cast<size_t>(j) + 1) * esz);
      }
    }
    r0 += rows;
  }
  if (A.r2_shape == 0 && !full.empty()) {
    if (piece_rows == variant_ct) {
      // the whole matrix is here: mirror it
      for (uint32_t j = 1; j < variant_ct; ++j) {
        for (uint32_t i = 0; i < j; ++i) {
          memcpy(full.data() + (static_cast<uint64_t>(i) * variant_ct + j) * esz, full.data() + (static_cast<uint64_t>(j) * variant_ct + i) * esz, esz);
        }
      }
    }
    // upper parts of a shard: second variants i in (shard_first, M), first variants = the shard's rows
    for (uint32_t r0 = (piece_rows == variant_ct) ? variant_ct : (shard_first + 1); r0 < variant_ct;) {
      uint32_t rows = static_cast<uint32_t>(std::max<uint64_t>(32, (1ull << 29) / (static_cast<uint64_t>(piece_rows) * esz)));
      rows = std::min(std::min(rows, variant_ct - r0), 65536u);
      chunk.assign(static_cast<size_t>(rows) * piece_rows * esz, 0);
      if (ldp_r2_unphased_block(e, r0, rows, shard_first, piece_rows, A.r2_float, chunk.data(), piece_rows)) {
        die(16, "Error: %s\n", ldp_last_error(e));
      }
      x_fix_dense(chunk.data(), A.r2_float != 0, r0, rows, shard_first, piece_rows, piece_rows);
      for (uint32_t q = 0; q < rows; ++q) {
        const uint32_t i = r0 + q;  // second variant
        const uint32_t jmax = std::min(i, shard_end);  // first variants j in [shard_first, jmax)
        const uint8_t* brow = chunk.data() + static_cast<uint64_t>(q) * piece_rows * esz;
        for (uint32_t j = shard_first; j < jmax; ++j) {
          memcpy(full.data() + (static_cast<uint64_t>(j - shard_first) * variant_ct + i) * esz, brow + static_cast<uint64_t>(j - shard_first) * esz, esz);
        }
      }
      r0 += rows;
    }
    if (A.r2_text) {
      const double* dm = reinterpret_cast<const double*>(full.data());
      char num[40];
      for (uint32_t j = 0; j < piece_rows; ++j) {
        textbuf.clear();
        for (uint32_t i = 0; i < variant_ct; ++i) {
          textbuf.append(num, format_g6(dm[static_cast<uint64_t>(j) * variant_ct + i], num) - num);
          textbuf += '\t';
        }
        textbuf.back() = '\n';
        mf.write(textbuf.data(), textbuf.size());
      }
    } else {
      mf.write(full.data(), full.size());
    }
  }
  mf.close();
  logprintf("--r2-unphased: Matrix%s written to %s .\n", (A.parallel_tot == 1) ? "" : " piece", mpath.c_str());
  ldp_destroy(e);
  ldp_pgen_close(pg);
  if (g_log) {
    fclose(g_log);
  }
  return 0;
}

int run_r2(Session& S) {
  const Args& A = S.A;
  const Variants& V = S.V;
  const std::vector<uint8_t>& is_founder = S.is_founder;
  const std::vector<uint8_t>& sex = S.sex;
  const uint32_t raw_sample_ct = S.raw_sample_ct, founder_ct = S.founder_ct, raw_variant_ct = S.raw_variant_ct;
  const std::string& gpath = S.gpath;
  ldp_pgen* const pg = S.pg;
  const int storage_mode = S.storage_mode, encoding = S.encoding;
  const uint64_t rec_bytes = S.rec_bytes;
  const uint8_t* const direct_rows = S.direct_rows;
  const std::vector<uint32_t>&inc = S.inc, &chr_idx = S.chr_idx, &bps = S.bps;
  const std::vector<uint8_t>& vcls = S.vcls;
  const uint32_t variant_ct = S.variant_ct;
  auto join_hip = [&S]() { S.join_hip(); };
  // ---- --r2-unphased {square|square0|triangle} {bin|bin4}: every variant, every pair (Vcor, plink2_ld.cc:12050)
  if ((!A.r2_table) && variant_ct > 400000 && (A.parallel_tot == 1) && !A.yes_really) {  // plink2_ld.cc:9788
    die(7, "Error: Gigantic (over 400k variants) --r2-unphased unfiltered, non-distributed\ncomputation.  Rerun with the 'yes-really' modifier if you are SURE you have enough\nhard drive space and want to do this.\n");
  }
  // host rows of the listed variants (raw file indices, in engine order) -> engine: decode / direct rows, founder columns
  // cols: the sample columns the engine keeps (nullptr: the founders)
  auto feed_rows_cols = [&](ldp_engine* eng, const std::vector<uint32_t>& incl, const std::vector<uint32_t>* cols) {
    const uint32_t n_incl = static_cast<uint32_t>(incl.size());
    const bool all_founders = (!cols) && (founder_ct == raw_sample_ct);
    std::vector<uint32_t> founder_idx;
    if (cols) {
      founder_idx = *cols;
    } else {
      for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
        if (is_founder[sx]) {
          founder_idx.push_back(sx);
        }
      }
    }
    const uint32_t kChunk = std::max<uint32_t>(1, static_cast<uint32_t>((1024ull << 20) / std::max<uint64_t>(rec_bytes, 1)));
    std::vector<uint8_t> decoded;
    if ((!all_founders) && ldp_set_sample_map(eng, raw_sample_ct, founder_idx.data(), nullptr)) {
      die(16, "Error: %s\n", ldp_last_error(eng));
    }
    for (uint32_t k = 0; k < n_incl;) {
      // a run of included variants that are consecutive in the file (chromosome 0 is stripped in table mode)
      const uint32_t raw_first = incl[k];
      uint32_t run = 1;
      while ((run < kChunk) && (k + run < n_incl) && (incl[k + run] == raw_first + run)) {
        ++run;
      }
      const uint8_t* src;
      uint64_t stride = rec_bytes;
      if (direct_rows) {
        src = direct_rows + static_cast<uint64_t>(raw_first) * rec_bytes;
      } else {
        decoded.resize(static_cast<size_t>(run) * rec_bytes);
        if (ldp_pgen_read(pg, raw_first, run, decoded.data(), rec_bytes, 0)) {
          die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
        }
        src = decoded.data();
      }
      // (the founder columns, CopyNyparrNonemptySubset pgenlib_misc.cc:32,185, are picked on the device)
      if (ldp_load_genotypes(eng, k, run, src, stride, LDP_MEM_HOST, encoding | (all_founders ? 0 : LDP_GENO_MAPPED))) {
        die(16, "Error: %s\n", ldp_last_error(eng));
      }
      k += run;
    }
  };
  auto feed_rows = [&](ldp_engine* eng, const std::vector<uint32_t>& incl) { feed_rows_cols(eng, incl, nullptr); };
  // chrY rows of the r^2 outputs and --clump: the female founders' calls count as missing (InterleavedSetMissing, plink2_ld.cc
  // :8833, :10290, :11845).  Reloads engine row `row` from raw variant `raw` that way.
  // --clump on a variant with several ALT alleles (aidx >= 0): the row is PgrGetInv1's for that allele -- copies of the OTHER alleles,
  // 3 = missing (pgenlib_read.cc:5544-5563) -- over all samples of the file, from the per-sample allele pairs of the host reader.
  std::vector<uint8_t> al_lo, al_hi;
  auto allele_row = [&](ldp_engine* eng, uint32_t row_idx, uint32_t raw, int32_t aidx, bool fem_missing, bool mapped) {
    std::vector<uint8_t> row(rec_bytes, 0);
    const uint8_t missing_code = (encoding == LDP_GENO_BED) ? 1 : 3;
    if (aidx >= 0) {
      al_lo.resize(raw_sample_ct);
      al_hi.resize(raw_sample_ct);
      if (ldp_pgen_read_alleles(pg, raw, V.alt_ct[raw], al_lo.data(), al_hi.data())) {
        die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
      }
      for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
        const uint32_t c = (al_lo[sx] == 255) ? 3u : (static_cast<uint32_t>(al_lo[sx] != aidx) + static_cast<uint32_t>(al_hi[sx] != aidx));
        row[sx >> 2] |= static_cast<uint8_t>(c << (2 * (sx & 3)));
      }
    } else if (direct_rows) {
      memcpy(row.data(), direct_rows + static_cast<uint64_t>(raw) * rec_bytes, rec_bytes);
    } else if (ldp_pgen_read(pg, raw, 1, row.data(), rec_bytes, 0)) {
      die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
    }
    for (uint32_t sx = 0; fem_missing && (sx < raw_sample_ct); ++sx) {
      if (is_founder[sx] && (sex[sx] == 2)) {
        uint8_t& b = row[sx >> 2];
        b = static_cast<uint8_t>((b & ~(3u << (2 * (sx & 3)))) | (missing_code << (2 * (sx & 3))));
      }
    }
    if (ldp_load_genotypes(eng, row_idx, 1, row.data(), rec_bytes, LDP_MEM_HOST, encoding | (mapped ? LDP_GENO_MAPPED : 0))) {
      die(16, "Error: %s\n", ldp_last_error(eng));
    }
  };
  if (A.have_clump) {
    join_hip();
    ClumpSex SX;
    SX.vcls = &vcls;
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      if (is_founder[sx]) {
        SX.founder_male_ct += (sex[sx] == 1);
        SX.founder_female_ct += (sex[sx] == 2);
        SX.founder_nosex_ct += (sex[sx] != 1) && (sex[sx] != 2);
        if (sex[sx] == 1) {
          SX.male_cols.push_back(sx);
        }
      }
    }
    SX.prov_bits.assign((static_cast<size_t>(raw_variant_ct) + 7) / 8, 0);
    SX.prov_storage = ldp_pgen_provisional_ref(pg, SX.prov_bits.data(), SX.prov_bits.size());
    if ((SX.prov_storage == 0) && V.info_pr_header) {  // the .pgen leaves it to the .pvar's INFO/PR
      SX.prov_storage = 3;
      std::copy(V.info_pr.begin(), V.info_pr.begin() + std::min(V.info_pr.size(), SX.prov_bits.size()), SX.prov_bits.begin());
    }
    SX.feed_cols = feed_rows_cols;
    SX.allele_row = allele_row;
    SX.raw_sample_ct = raw_sample_ct;
    const int rc = clump_reports(A, V, inc, chr_idx, bps, founder_ct, feed_rows, SX);
    if (g_log) {
      fclose(g_log);
    }
    return rc;
  }
  ldp_params RP;
  memset(&RP, 0, sizeof(RP));
  RP.founder_ct = founder_ct;
  RP.prune_window_size = 2;
  RP.prune_window_incr = 1;
  RP.prune_last_param = 0.5;
  RP.device = 0;
  join_hip();
  if (ldp_device_count() < 1) {
    die(16, "Error: no usable HIP device (plink2-hip has no CPU compute path).\n");
  }
  ldp_engine* e = nullptr;
  if (ldp_create(&RP, &e)) {
    die(16, "Error: engine setup failed.\n");
  }
  if (g_dbg.x_rows) {
    (void)ldp_debug_set_option(e, "x_rows", static_cast<double>(g_dbg.x_rows));
  }
  if (ldp_set_r_signed(e, A.r_unsquared ? (A.r2_ref_based ? 2 : 1) : 0)) {
    die(16, "Error: %s\n", ldp_last_error(e));
  }
  if (A.r2_table && !A.r2_allow_ambiguous) {  // plink2_ld.cc:11042-11074
    bool multiallelic = false;
    for (uint32_t k = 0; (!multiallelic) && (k < variant_ct); ++k) {
      multiallelic = V.alt_ct[inc[k]] > 1;
    }
    if (A.r_unsquared) {
      // the sign of r refers to an allele: some column has to name it
      bool ambiguous = false;
      if (!A.r2_ref_based) {
        ambiguous = !(A.r2_cols & (kVcorColMaj | kVcorColNonmaj));
      } else {
        const uint32_t relevant = A.r2_cols & (kVcorColRef | kVcorColAlt1 | kVcorColAlt);
        if (relevant != kVcorColAlt1) {
          ambiguous = !relevant;
        } else if (multiallelic) {
          die(7, "Error: The meaning of r's sign cannot be consistently inferred from just the\n--r2-unphased 'alt1' column-set at multiallelic variants. Either filter out\nmultiallelic variants, revise the column-set, or use the\n'allow-ambiguous-allele' modifier to override this error.\n");
        }
      }
      if (ambiguous) {
        die(7, "Error: --r2-unphased column-set doesn't include allele columns which clarify\nthe meaning of r's sign. Either switch to --r2-unphased, add a disambiguating\ncolumn-set, or use the 'allow-ambiguous-allele' modifier to override this\nerror.\n");
      }
    } else {
      const bool ambiguous = A.r2_ref_based ? !(A.r2_cols & (kVcorColRef | kVcorColAlt)) : !(A.r2_cols & (kVcorColMaj | kVcorColNonmaj));
      if (ambiguous && multiallelic) {
        die(7, "Error: --r2-unphased column-set doesn't include allele columns which clarify\nwhich calculation is being performed at multiallelic variants. Either filter\nout multiallelic variants, revise the column-set (with e.g. \"cols=+%s\"), or\nuse the 'allow-ambiguous-allele' modifier to override this error.\n", A.r2_ref_based ? "ref" : "maj");
      }
    }
  }
  if (A.r2_inter && (A.ld_min_r2 <= 0.0) && (variant_ct > 400000) && (A.parallel_tot == 1) && !A.yes_really) {  // plink2_ld.cc:11087
    die(7, "Error: Gigantic (over 400k variants) --r2-unphased unfiltered, non-distributed\ncomputation.  Rerun with the 'yes-really' modifier if you are SURE you have enough\nhard drive space and want to do this.\n");
  }
  std::vector<double> cms;  // --ld-window-cm (a file without non-zero CM values has no CM window: Vcor hands UpdateVcorWindow a null array)
  if (A.r2_table && (!A.r2_inter) && (A.ld_cm_radius != -1.0)) {
    if (V.cm_unsorted) {  // plink2.cc:2948-2951
      die(7, "Error: --ld-window-cm requires nondecreasing CM values on each chromosome.\nRetry this command after regenerating your CM coordinates.\n");
    }
    if (V.cm_any_nonzero) {
      cms.resize(variant_ct);
      for (uint32_t k = 0; k < variant_ct; ++k) {
        cms[k] = V.cm[inc[k]];
      }
    }
  }
  if ((A.r2_table && !A.r2_inter) ? ldp_set_variants_vcor_cm(e, variant_ct, chr_idx.data(), bps.data(), cms.empty() ? nullptr : cms.data(), A.ld_bp_radius,
                                                             A.ld_cm_radius, A.ld_var_ct_radius)
                                  : ldp_set_variants_matrix(e, variant_ct)) {
    die(16, "Error: engine setup failed: %s\n", ldp_last_error(e));
  }
  const std::string base = A.out + ".unphased.vcor" + (A.r_unsquared ? "1" : "2") + (A.r2_text ? "" : ".bin");  // (VcorMatrix :9849-9857)
  // --parallel k n: the reference's row shards.  Matrix (VcorMatrix, plink2_ld.cc:9800-9824): `square` takes rows
  // [M k / n, M (k+1) / n); the triangular shapes take ParallelBounds() rows (equal numbers of lower-triangle entries) and
  // a piece that does not reach the last row behaves as if the later variants did not exist (its .vars file, written by piece
  // 1 only, lists just the variants before its last row; square0's zero padding still runs to the full width).  Table
  // (VcorTable, :11157-11168): first variants [M k / n, M (k+1) / n); the header goes to piece 1.  Pieces are named
  // <file>.<k> and concatenate to the undistributed output.
  uint32_t shard_first = 0, shard_end = variant_ct, vars_ct = variant_ct;
  const std::string piece_suffix = (A.parallel_tot == 1) ? std::string() : ("." + std::to_string(A.parallel_idx + 1));
  if (A.parallel_tot != 1) {
    if ((!A.r2_table) && (variant_ct < 2 * A.parallel_tot)) {
      die(7, "Error: Too few variants in --r2-unphased run for --parallel %u %u.\n", A.parallel_idx + 1, A.parallel_tot);
    }
    if ((!A.r2_table) && (A.r2_shape != 0)) {
      // smallest v with v (v + 1) >= x (TriangleDivide, plink2_common.cc:4936, modif = 1)
      auto tri = [](uint64_t x) {
        if (!x) {
          return static_cast<uint64_t>(0);
        }
        uint64_t v = static_cast<uint64_t>(sqrt(static_cast<double>(x)));
        while ((v >= 1) && ((v - 1) * v >= x)) {
          --v;
        }
        while (v * (v + 1) < x) {
          ++v;
        }
        return v;
      };
      const uint64_t tot = static_cast<uint64_t>(variant_ct) * (static_cast<uint64_t>(variant_ct) + 1);
      shard_first = static_cast<uint32_t>(tri(tot * A.parallel_idx / A.parallel_tot));
      shard_end = static_cast<uint32_t>(tri(tot * (A.parallel_idx + 1) / A.parallel_tot));
      vars_ct = shard_end;
    } else {
      shard_first = static_cast<uint32_t>(static_cast<uint64_t>(variant_ct) * A.parallel_idx / A.parallel_tot);
      shard_end = static_cast<uint32_t>(static_cast<uint64_t>(variant_ct) * (A.parallel_idx + 1) / A.parallel_tot);
    }
  }
  if ((!A.r2_table) && (A.parallel_idx == 0)) {
    FILE* vf = fopen((base + ".vars").c_str(), "wb");
    if (!vf) {
      die(3, "Error: Failed to open %s.vars for writing.\n", base.c_str());
    }
    for (uint32_t k = 0; k < vars_ct; ++k) {
      fputs(V.id[inc[k]].c_str(), vf);
      fputc('\n', vf);
    }
    fclose(vf);
    logprintf("--r2-unphased: Variant IDs written to %s.vars .\n", base.c_str());
  }
  // genotype rows -> engine (same feeder as the prune path)
  std::unordered_map<uint32_t, std::pair<uint32_t, double>> multi_maj;  // multiallelic variant -> (major allele, its frequency), for the MAJ / NONMAJ / NONMAJ_FREQ columns
  std::vector<std::pair<uint32_t, uint32_t>> collapsed_on;  // (row, major allele) of the rows loaded as copies of the non-major alleles
  std::unordered_map<uint32_t, double> multi_maj_all;       // ... and the major allele's frequency
  {
    feed_rows(e, inc);
    const uint64_t out_rec = (static_cast<uint64_t>(founder_ct) + 3) / 4;
    std::vector<uint32_t> founder_idx;
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      if (is_founder[sx]) {
        founder_idx.push_back(sx);
      }
    }
    // Multiallelic variants (R2NondosageVariant works on PgrGetInv1(major allele) rows, plink2_ld.cc:6039-6048):
    // collapsed major-vs-rest on the host, as for the prune.  With 'ref-based' the collapse is REF-vs-rest, which
    // is what the main track's codes already are.
    const bool want_maj = A.r2_table && (A.r2_cols & (kVcorColMaj | kVcorColNonmaj | kVcorColFreq));
    if ((!A.r2_ref_based) || want_maj) {
      std::vector<uint8_t> lo(raw_sample_ct), hi(raw_sample_ct), inv_row(out_rec);
      for (uint32_t k = 0; k < variant_ct; ++k) {
        const uint32_t alts = V.alt_ct[inc[k]];
        if (alts < 2) {
          continue;
        }
        if (storage_mode == 0x01) {
          die(6, "Error: multiallelic variant in a .bim/.bed fileset.\n");
        }
        double mf = 0.0;
        uint32_t maj = 0;
        if (vcls[k] >= 3) {
          // chrX / chrY / MT (round 5; refused before): the major allele by the chromosome's own allele-frequency weights -- chrX: a male founder's allele copy counts 1, anybody
          // else's 2; chrY: the non-female founders; MT: every founder (plink2_data.cc:2752-2895) --, the row PgrGetInv1's collapse on it over all founders
          SexPlan sp;
          for (uint32_t sidx : founder_idx) {
            if (vcls[k] == 3) {
              (sex[sidx] == 1 ? sp.part1 : sp.part2).push_back(sidx);
            } else if ((vcls[k] == 5) || (sex[sidx] != 2)) {
              sp.part1.push_back(sidx);
            }
          }
          maj = sex_major_allele(pg, inc[k], alts, sp, &lo, &hi, &mf);
          memset(inv_row.data(), 0, out_rec);
          uint32_t f = 0;
          for (uint32_t sidx : founder_idx) {
            const uint32_t code = (lo[sidx] == 255) ? 3u : (static_cast<uint32_t>(lo[sidx] != maj) + static_cast<uint32_t>(hi[sidx] != maj));
            inv_row[f >> 2] |= static_cast<uint8_t>(code << (2 * (f & 3)));
            ++f;
          }
        } else {
          multiallelic_inverse_row(pg, inc[k], alts, founder_idx, &lo, &hi, inv_row.data(), out_rec, &mf, nullptr, 0, nullptr, &maj);
        }
        if (want_maj) {
          multi_maj[k] = std::make_pair(maj, mf);
        }
        if (A.r2_ref_based) {
          continue;  // (the main track's REF-vs-rest codes are the rows; only the major allele and its frequency were wanted)
        }
        collapsed_on.emplace_back(k, maj);
        multi_maj_all[k] = mf;
        if (ldp_load_genotypes(e, k, 1, inv_row.data(), out_rec, LDP_MEM_HOST, LDP_GENO_INVERSE) || ldp_set_maj_freqs(e, k, 1, &mf)) {
          die(16, "Error: %s\n", ldp_last_error(e));
        }
      }
    }
  }
  // ---- chrX: the kernels' values of pairs with a chrX variant are replaced on the host (XWeighted above) ----
  std::vector<uint8_t> is_x(variant_ct, 0);
  bool any_x = false, any_ymt = false;
  for (uint32_t k = 0; k < variant_ct; ++k) {
    is_x[k] = (vcls[k] == 3);
    any_x = any_x || is_x[k];
    any_ymt = any_ymt || (vcls[k] >= 4);
  }
  // chrY: the female founders' calls count as missing (InterleavedSetMissing, VcorMatrix :10290 / VcorTable :11845), unless
  // every founder is male or none is female (:10025-10043); MT rows are ordinary.  What would need the haploid
  // allele-frequency arithmetic of these chromosomes is the major allele: the sign of a major-oriented r, the MAJ / NONMAJ /
  // NONMAJ_FREQ columns, and the rounding of the chrX-weighted sums when a chrX variant is paired with them.
  uint32_t founder_male_ct = 0, founder_female_ct = 0;
  for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
    founder_male_ct += is_founder[sx] && (sex[sx] == 1);
    founder_female_ct += is_founder[sx] && (sex[sx] == 2);
  }
  if (founder_female_ct && (founder_male_ct != founder_ct)) {
    for (uint32_t k = 0; k < variant_ct; ++k) {
      if (vcls[k] == 4) {
        int32_t on = -1;  // (a variant with several ALT alleles: its collapsed row, not the main track)
        for (const auto& km : collapsed_on) {
          if (km.first == k) {
            on = static_cast<int32_t>(km.second);
          }
        }
        allele_row(e, k, inc[k], on, true, founder_ct != raw_sample_ct);
      }
    }
  }
  // chrX is only special when the founders are of both kinds (:9946-9951, :11470-11480)
  if ((!founder_male_ct) || (founder_male_ct == founder_ct)) {
    any_x = false;
    std::fill(is_x.begin(), is_x.end(), 0);
  }
  if (any_ymt) {
    if (A.r_unsquared && !A.r2_ref_based) {
      die(63, "Error: --r-unphased on chrY/MT variants needs 'ref-based' in plink2-hip.\n");
    }
    if (A.r2_table && (A.r2_cols & (kVcorColMaj | kVcorColNonmaj | kVcorColFreq))) {
      die(63, "Error: the maj/nonmaj/freq columns of chrY/MT variants are not supported by plink2-hip.\n");
    }
    if (any_x && (!A.r2_ref_based) && (A.r2_inter || !A.r2_table)) {
      die(63, "Error: all-pairs --r2-unphased over chrX together with chrY/MT needs 'ref-based' in plink2-hip.\n");
    }
  }
  ldp_engine* e_male = nullptr;
  std::vector<uint8_t> x_flip_all, x_flip_male, x_maj_alt;  // (x_maj_alt: the chrX-aware major allele, for the MAJ / NONMAJ columns)
  std::vector<uint8_t> x_target;                             // the orientation the values are wanted in, per row (1: ALT counted as the major allele)
  struct EngineGuard {
    ldp_engine** p;
    ~EngineGuard() {
      if (*p) {
        ldp_destroy(*p);
      }
    }
  } male_guard{&e_male};
  std::vector<double> x_maj_freq;
  if (any_x) {
    std::vector<uint32_t> male_cols;
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      if (is_founder[sx] && (sex[sx] == 1)) {
        male_cols.push_back(sx);
      }
    }
    std::vector<ldp_variant_rec> recs_all(variant_ct), recs_male(variant_ct);
    memset(recs_male.data(), 0, recs_male.size() * sizeof(ldp_variant_rec));
    if (ldp_get_variant_recs(e, 0, variant_ct, recs_all.data())) {
      die(16, "Error: %s\n", ldp_last_error(e));
    }
    if (!male_cols.empty()) {
      ldp_params MP = RP;
      MP.founder_ct = static_cast<uint32_t>(male_cols.size());
      if (ldp_create(&MP, &e_male) || ldp_set_variants_matrix(e_male, variant_ct)) {
        die(16, "Error: engine setup failed.\n");
      }
      feed_rows_cols(e_male, inc, &male_cols);
      // (a pair of a chrX variant with a multiallelic one takes the male founders' tuple of BOTH, ComputeXR2 plink2_ld.cc:7122-7190: the male engine's row of the
      // multiallelic variant is the same collapse -- copies of the alleles other than the major one the ALL-founder counts chose -- over its columns)
      for (const auto& km : collapsed_on) {
        allele_row(e_male, km.first, inc[km.first], static_cast<int32_t>(km.second), false, true);
      }
      if (ldp_get_variant_recs(e_male, 0, variant_ct, recs_male.data())) {
        die(16, "Error: %s\n", ldp_last_error(e_male));
      }
    }
    x_flip_all.assign(variant_ct, 0);
    x_flip_male.assign(variant_ct, 0);
    x_target.assign(variant_ct, 0);
    x_maj_alt.assign(variant_ct, 0);
    x_maj_freq.assign(variant_ct, 0.0);
    for (uint32_t k = 0; k < variant_ct; ++k) {
      uint32_t target_alt = recs_all[k].flags & 1u;  // the engine's own choice: diploid allele counts over the founders
      const auto mm = multi_maj_all.find(k);
      if (is_x[k] && (mm != multi_maj_all.end())) {
        // several ALT alleles: the row already counts the copies of the alleles other than the chrX rule's major one (loaded as LDP_GENO_INVERSE: flags bit 0 clear)
        target_alt = 0;
        x_maj_freq[k] = mm->second;
      } else if (is_x[k]) {
        // the allele-frequency pass on chrX counts a male once (the arithmetic of build_sex_row above)
        const uint64_t g1 = recs_all[k].n_het, g2 = recs_all[k].n_homalt, n_all = static_cast<uint64_t>(recs_all[k].n_homref) + g1 + g2;
        const uint64_t m1 = recs_male[k].n_het, m2 = recs_male[k].n_homalt, n_male = static_cast<uint64_t>(recs_male[k].n_homref) + m1 + m2;
        const uint64_t alt_ct = 4 * g2 + 2 * g1 - 2 * m2 - m1, tot = 2 * (2 * n_all - n_male), ref_ct = tot - alt_ct;
        double ref_freq = 0.5;
        if (tot) {
          ref_freq = static_cast<double>(ref_ct) * (1.0 / static_cast<double>(tot));
        }
        target_alt = (ref_freq >= 0.5) ? 0 : 1;
        x_maj_freq[k] = target_alt ? (1.0 - ref_freq) : ref_freq;
      }
      x_maj_alt[k] = static_cast<uint8_t>(target_alt);
      if (A.r2_ref_based) {
        target_alt = 0;
      }
      x_target[k] = static_cast<uint8_t>(target_alt);
      x_flip_all[k] = static_cast<uint8_t>((recs_all[k].flags & 1u) ^ target_alt);
      x_flip_male[k] = static_cast<uint8_t>((recs_male[k].flags & 1u) ^ target_alt);
    }
  }
  // windowed table: the chrX run a second time in two all-pairs engines (XWeighted: band_*), so that its windows' pairs come from the pair kernels too
  XWeighted band;
  struct BandGuard {
    XWeighted* p;
    ~BandGuard() { p->band_destroy(); }
  } band_guard{&band};
  if (any_x && A.r2_table && (!A.r2_inter) && !g_dbg.x_host) {
    uint32_t x0 = 0, x1 = 0;
    bool one_run = true;
    for (uint32_t k = 0; k < variant_ct; ++k) {
      if (is_x[k]) {
        if (x1 == 0) {
          x0 = k;
        } else if (x1 != k) {
          one_run = false;
        }
        x1 = k + 1;
      }
    }
    if (one_run && (x1 > x0)) {
      const uint32_t cnt = x1 - x0;
      const std::vector<uint32_t> x_inc(inc.begin() + x0, inc.begin() + x1);
      if (ldp_create(&RP, &band.band_all) || ldp_set_variants_matrix(band.band_all, cnt)) {
        die(16, "Error: engine setup failed.\n");
      }
      if (g_dbg.x_rows) {
        (void)ldp_debug_set_option(band.band_all, "x_rows", static_cast<double>(g_dbg.x_rows));
      }
      feed_rows(band.band_all, x_inc);
      if (e_male) {
        ldp_params MP = RP;
        MP.founder_ct = founder_male_ct;
        if (ldp_create(&MP, &band.band_male) || ldp_set_variants_matrix(band.band_male, cnt)) {
          die(16, "Error: engine setup failed.\n");
        }
        std::vector<uint32_t> male_cols;
        for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
          if (is_founder[sx] && (sex[sx] == 1)) {
            male_cols.push_back(sx);
          }
        }
        feed_rows_cols(band.band_male, x_inc, &male_cols);
      }
      for (const auto& km : collapsed_on) {  // (variants of the run with several ALT alleles: the same collapse in both)
        if ((km.first >= x0) && (km.first < x1)) {
          allele_row(band.band_all, km.first - x0, inc[km.first], static_cast<int32_t>(km.second), false, founder_ct != raw_sample_ct);
          if (band.band_male) {
            allele_row(band.band_male, km.first - x0, inc[km.first], static_cast<int32_t>(km.second), false, true);
          }
        }
      }
      band.band_first = x0;
      band.band_ct = cnt;
      // (the same rows as in `e` / `e_male` up to how a collapsed row was loaded: each engine's own orientation against the same target)
      std::vector<ldp_variant_rec> rb(cnt);
      band.band_flip_all.assign(cnt, 0);
      band.band_flip_male.assign(cnt, 0);
      if (ldp_get_variant_recs(band.band_all, 0, cnt, rb.data())) {
        die(16, "Error: %s\n", ldp_last_error(band.band_all));
      }
      for (uint32_t q = 0; q < cnt; ++q) {
        band.band_flip_all[q] = static_cast<uint8_t>((rb[q].flags & 1u) ^ x_target[x0 + q]);
      }
      if (band.band_male) {
        if (ldp_get_variant_recs(band.band_male, 0, cnt, rb.data())) {
          die(16, "Error: %s\n", ldp_last_error(band.band_male));
        }
        for (uint32_t q = 0; q < cnt; ++q) {
          band.band_flip_male[q] = static_cast<uint8_t>((rb[q].flags & 1u) ^ x_target[x0 + q]);
        }
      }
    }
  }
  R2Job J(S);
  J.e = e;
  J.shard_first = shard_first;
  J.shard_end = shard_end;
  J.piece_suffix = piece_suffix;
  J.base = base;
  J.is_x = is_x;
  J.any_x = any_x;
  J.xw.all = e;
  J.xw.male = e_male;
  J.xw.is_x = is_x;
  J.xw.flip_all = x_flip_all;
  J.xw.flip_male = x_flip_male;
  J.xw.unsquared = A.r_unsquared;
  J.xw.band_all = band.band_all;  // (owned by `band`: destroyed when this function returns)
  J.xw.band_male = band.band_male;
  J.xw.band_first = band.band_first;
  J.xw.band_ct = band.band_ct;
  J.xw.band_flip_all = band.band_flip_all;
  J.xw.band_flip_male = band.band_flip_male;
  J.multi_maj = std::move(multi_maj);
  J.x_maj_alt = x_maj_alt;
  J.x_maj_freq = x_maj_freq;
  return A.r2_table ? write_vcor_table(J) : write_vcor_matrix(J);
}


}  // namespace p2h
