// p2h_clump.cpp -- plink2-hip: --clump (ClumpReports) and the chrX-weighted r^2 arithmetic (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

// ---- --clump (ClumpReports, plink2_ld.cc:7506-9480) --------------------------------------------------------------
// What plink2-hip covers: one association report, biallelic diploid variants, --clump-unphased (the hardcall r^2 of
// ComputeR2, plink2_ld.cc:6654-6682 -- the quantity the matrix-pipe kernels produce), default column set, and the
// --clump-p1/-p2/-r2/-kb/-id-field/-p-field/-test/-test-field settings.  The reference walks the index candidates in
// p-value order and, for each one still unclumped, computes r^2 against the unclumped variants of its window.  Here
// the r^2 > threshold pairs of the WHOLE band (every observed variant against its +-kb neighbours) come from one pass
// of the windowed-table kernels, filtered in the kernel epilogue; the rank-ordered greedy assignment then runs on the
// host over that sparse pair list.  The pair set tested is a superset of the reference's, each r^2 is the same
// double, and the greedy pass only ever looks at (index, window member) pairs, so the clumps are identical.

const double kLn10 = 2.3025850929940457;
const double kRecipLn10 = 0.43429448190325176;

// ln of a nonnegative decimal number, as ScanadvLn (include/plink2_string.cc:1530-1760) derives it: up to ~17
// significant digits accumulate in an integer, the rest only move the decimal exponent, and ln = log(digits) +
// e10 * ln(10) -- so "1e-400" works, and the doubles (hence the candidate order and the bins) match the reference's.
// Returns the end of the number, or nullptr when there is none; zero gives -DBL_MAX.
const char* scan_ln(const char* s, double* ln_out) {
  const bool neg = (*s == '-');
  if (neg || (*s == '+')) {
    ++s;
  }
  int64_t digits = 0;
  long e10 = 0;
  bool any = false;
  bool full = false;  // 10^16 reached: later digits are not read
  for (; (*s >= '0') && (*s <= '9'); ++s) {
    any = true;
    if (!full) {
      digits = digits * 10 + (*s - '0');
      full = (digits >= 10000000000000000LL);
    } else {
      ++e10;
    }
  }
  if (*s == '.') {
    ++s;
    if ((!any) && !((*s >= '0') && (*s <= '9'))) {
      return nullptr;
    }
    for (; (*s >= '0') && (*s <= '9'); ++s) {
      any = true;
      if (!full) {
        digits = digits * 10 + (*s - '0');
        --e10;
        full = (digits >= 10000000000000000LL);
      }
    }
  }
  if (!any || (neg && digits)) {
    return nullptr;
  }
  if ((*s == 'e') || (*s == 'E')) {
    ++s;
    const bool eneg = (*s == '-');
    if (eneg || (*s == '+')) {
      ++s;
    }
    long ex = 0;
    for (; (*s >= '0') && (*s <= '9'); ++s) {
      if (ex >= 107374182) {
        if (!eneg) {
          return nullptr;
        }
        while ((*s >= '0') && (*s <= '9')) {
          ++s;
        }
        *ln_out = -DBL_MAX;
        return s;
      }
      ex = ex * 10 + (*s - '0');
    }
    e10 += eneg ? -ex : ex;
  }
  if (!digits) {
    *ln_out = -DBL_MAX;
    return s;
  }
  double ln = log(static_cast<double>(digits));
  if (e10) {
    ln += static_cast<double>(e10) * kLn10;
  }
  *ln_out = ln;
  return s;
}

// exp(ln_val) with 6 significant digits, as lntoa_g prints p-values (include/plink2_string.cc:2876-2946): plain
// decimals down to 1e-4, d.ddddde-XX below, mantissa and exponent taken from the logarithm so that values under
// DBL_MIN still print.
char* format_ln_g6(double ln_val, char* out) {
  if (ln_val < 13.81551005796414) {
    if (ln_val > -9.210340871976317) {
      if (ln_val > -5.000001349509205e-7) {
        if (ln_val < 4.999987599993995e-6) {
          *out++ = '1';
          return out;
        }
        return format_g6(exp(ln_val), out);
      }
      double x = exp(ln_val);
      *out++ = '0';
      *out++ = '.';
      if (x < 9.9999949999999e-3) {
        x *= 100;
        *out++ = '0';
        *out++ = '0';
      }
      if (x < 9.9999949999999e-2) {
        x *= 10;
        *out++ = '0';
      }
      return put_digits_trimmed(banker_round(x * 1000000), 6, 1, out);
    }
    if (ln_val < 2147483643.0 * (-kLn10)) {
      *out++ = '0';
      return out;
    }
  } else if (ln_val > 2147483643.0 * kLn10) {
    memcpy(out, "inf", 3);
    return out + 3;
  }
  int32_t xp10 = static_cast<int32_t>(fma(ln_val, kRecipLn10, 5.000001349509205e-7 * kRecipLn10));
  double mantissa = exp(fma(static_cast<double>(xp10), -kLn10, ln_val));
  if (mantissa < 0.99999949999999) {
    mantissa *= 10;
    xp10 -= 1;
  } else if (mantissa > 9.9999949999999) {
    mantissa *= 0.1;
    xp10 += 1;
  }
  const uint32_t t = banker_round(mantissa * 100000);
  *out++ = static_cast<char>('0' + t / 100000);
  if (t % 100000) {
    *out++ = '.';
    out = put_digits_trimmed(t % 100000, 5, 1, out);
  }
  *out++ = 'e';
  *out++ = (xp10 < 0) ? '-' : '+';
  const uint32_t ax = static_cast<uint32_t>((xp10 < 0) ? -xp10 : xp10);
  if (ax < 10) {
    *out++ = '0';
  }
  return out + snprintf(out, 12, "%u", ax);
}

// digit runs compare as numbers, everything else bytewise (the order NsortDedupAndWrite gives the .missing_id list)
bool natural_less(const std::string& a, const std::string& b) {
  size_t i = 0, j = 0;
  while ((i < a.size()) && (j < b.size())) {
    const bool da = (a[i] >= '0') && (a[i] <= '9'), db = (b[j] >= '0') && (b[j] <= '9');
    if (da && db) {
      size_t i1 = i, j1 = j;
      while ((i1 < a.size()) && (a[i1] == '0')) {
        ++i1;
      }
      while ((j1 < b.size()) && (b[j1] == '0')) {
        ++j1;
      }
      size_t i2 = i1, j2 = j1;
      while ((i2 < a.size()) && (a[i2] >= '0') && (a[i2] <= '9')) {
        ++i2;
      }
      while ((j2 < b.size()) && (b[j2] >= '0') && (b[j2] <= '9')) {
        ++j2;
      }
      if ((i2 - i1) != (j2 - j1)) {
        return (i2 - i1) < (j2 - j1);
      }
      const int c = a.compare(i1, i2 - i1, b, j1, j2 - j1);
      if (c) {
        return c < 0;
      }
      i = i2;
      j = j2;
      continue;
    }
    if (a[i] != b[j]) {
      return static_cast<unsigned char>(a[i]) < static_cast<unsigned char>(b[j]);
    }
    ++i;
    ++j;
  }
  if ((i == a.size()) != (j == b.size())) {
    return i == a.size();
  }
  return a < b;
}

// bin boundaries of the default 'bins' column set (kClumpDefaultLnBinBounds, plink2_ld.cc:7498): ln of 1e-4, 1e-3, 1e-2, 0.05
const double kClumpLnBins[4] = {-9.210340371976706, -6.907755278982529, -4.605170185988353, -2.995732273554161};

// What the reference keys by "allele index" (plink2_ld.cc:7776-7819): a biallelic variant is ONE entity whatever the report's A1 says
// (--clump-force-a1 only remembers which allele the best line named), a variant with several ALT alleles is one entity PER ALLELE, REF
// included -- each with its own p-values, its own rank among the index candidates and its own genotype row (copies of that allele).  A slot
// here is that entity: slot_base[k] .. slot_base[k + 1] are the slots of included variant k (one, or allele count many), in dataset order.
struct ClumpSlots {
  std::vector<uint32_t> base;  // per included variant (+ 1)
  std::vector<uint32_t> var;   // slot -> included variant
  bool any_multiallelic = false;
  void build(const Variants& V, const std::vector<uint32_t>& inc) {
    base.assign(inc.size() + 1, 0);
    for (size_t k = 0; k < inc.size(); ++k) {
      const uint32_t n = (V.alt_ct[inc[k]] > 1) ? (static_cast<uint32_t>(V.alt_ct[inc[k]]) + 1) : 1u;
      any_multiallelic = any_multiallelic || (n > 1);
      base[k + 1] = base[k] + n;
    }
    var.resize(base.back());
    for (size_t k = 0; k < inc.size(); ++k) {
      for (uint32_t q = base[k]; q < base[k + 1]; ++q) {
        var[q] = static_cast<uint32_t>(k);
      }
    }
  }
  uint32_t count() const { return base.back(); }
  bool multiallelic(uint32_t slot) const { return base[var[slot] + 1] - base[var[slot]] > 1; }
  uint32_t aidx(uint32_t slot) const { return slot - base[var[slot]]; }
};

// allele `idx` of raw variant v as the reference's flat allele table holds it (REF, then the ALT alleles): one past the last allele is the
// next variant's REF -- which is what its SP2 printer reads when a stale forced-A1 bit rides on a multiallelic entry (clump_load_report)
std::string allele_text(const Variants& V, uint32_t v, uint32_t idx) {
  if (idx == 0) {
    return V.ref[v];
  }
  const std::string& alt = V.alt[v];
  size_t pos = 0;
  for (uint32_t a = 1; a <= V.alt_ct[v]; ++a) {
    const size_t comma = std::min(alt.find(',', pos), alt.size());
    if (a == idx) {
      return alt.substr(pos, comma - pos);
    }
    pos = comma + 1;
  }
  return (static_cast<size_t>(v) + 1 < V.ref.size()) ? V.ref[v + 1] : std::string();
}

struct ClumpData {
  // per slot (ClumpSlots)
  std::vector<double> best_ln;                // lowest ln p among the lines at or below the load threshold; 0 without one
  std::vector<uint32_t> nonsig;               // lines above every bin boundary
  std::vector<std::vector<uint32_t>> entries; // one per loaded line, in the order read (last report first): (file << 12) | (bin << 1) | (ln p > ln p2)
  std::vector<double> ln_bins;                // bin boundaries in use (empty: the 'bins' column set is off)
  std::vector<uint8_t> best_a1;               // --clump-force-a1: the best line's A1 is the ALT allele (a biallelic variant's slot)
  std::vector<std::string> missing_pairs;     // top (ID, A1) pairs whose allele the dataset's variant does not have
  std::vector<uint16_t> best_file;            // report (1-based) the best p-value came from; ties go to the first report
  std::vector<uint8_t> observed;
  std::vector<std::string> missing_ids;       // top (p <= p1) IDs absent from the dataset
};

uint32_t clump_bin(const std::vector<double>& ln_bins, double ln_pval) {  // LowerBoundNonemptyD: boundaries strictly below
  uint32_t b = 0;
  while ((b < ln_bins.size()) && (ln_pval > ln_bins[b])) {
    ++b;
  }
  return b;
}

// The report -> per-variant p-value lists (plink2_ld.cc:7667-7858).
void clump_load_report(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, const ClumpSlots& SL, ClumpData* D) {
  const uint32_t variant_ct = static_cast<uint32_t>(inc.size());
  const uint32_t slot_ct = SL.count();
  D->best_ln.assign(slot_ct, 0.0);
  D->nonsig.assign(slot_ct, 0);
  D->entries.assign(slot_ct, std::vector<uint32_t>());
  D->best_file.assign(slot_ct, 1);
  D->best_a1.assign(slot_ct, 0);
  D->observed.assign(slot_ct, 0);
  // ID -> included-variant index; kDup marks IDs the dataset holds more than once (an error only when the report names one)
  bool pvar_multiallelic = false;
  for (size_t v = 0; (!pvar_multiallelic) && (v < V.alt_ct.size()); ++v) {
    pvar_multiallelic = V.alt_ct[v] > 1;
  }
  // the forced-A1 bit of the latest line of a BIALLELIC variant (biallelic_forced_a1_alt, :7646): the reference declares it outside the loops
  // over reports and lines and writes it on biallelic lines only, so an entry of a multiallelic variant's allele carries whatever the
  // previous biallelic line left there.  Reproduced (the bit shows in SP2 and in the bounds scan).
  uint32_t a1_alt = 0;
  const uint32_t kDup = 0xffffffffu;
  std::unordered_map<std::string, uint32_t> by_id;
  by_id.reserve(static_cast<size_t>(variant_ct) * 2);
  for (uint32_t k = 0; k < variant_ct; ++k) {
    auto it = by_id.emplace(V.id[inc[k]], k);
    if (!it.second) {
      it.first->second = kDup;
    }
  }
  // what is kept of a line depends on the column set (:7577-7612): bins and their boundaries, p2 only when SP2 or the bounds
  // want it, entries at all only for total / bins / SP2 / bounds, the above-every-boundary counts only for total / bins
  D->ln_bins.clear();
  if (A.clump_cols & kClumpColBins) {
    D->ln_bins = A.clump_ln_bins.empty() ? std::vector<double>(kClumpLnBins, kClumpLnBins + 4) : A.clump_ln_bins;
  }
  const bool ranges_col = !A.clump_range.empty();
  const bool bounds_col = (A.clump_cols & kClumpColBounds) || ((A.clump_cols & kClumpColMaybeBounds) && ranges_col);
  const bool sp2_col = (A.clump_cols & kClumpColSp2) != 0;
  const double ln_p1 = A.clump_ln_p1, ln_p2 = (sp2_col || bounds_col || ranges_col) ? A.clump_ln_p2 : -1.7976931348623157e308;
  double load_thresh = std::max(ln_p1, ln_p2);
  if ((!D->ln_bins.empty()) && (load_thresh < D->ln_bins.back())) {
    load_thresh = D->ln_bins.back();
  }
  const bool keep_entries = (A.clump_cols & (kClumpColTotal | kClumpColBins | kClumpColSp2)) || bounds_col || ranges_col;
  const bool nonsig_needed = (A.clump_cols & (kClumpColTotal | kClumpColBins)) && (load_thresh < 0.0);
  if (A.clump_files.size() > 4000) {
    die(63, "Error: too many --clump reports.\n");
  }
  for (size_t file_idx1 = A.clump_files.size(); file_idx1; --file_idx1) {  // last report first (plink2_ld.cc:7644-7654)
    const std::string& fname = A.clump_files[file_idx1 - 1];
    const std::string text = slurp(fname);
    const char* p = text.c_str();
    const char* const end = p + text.size();
    size_t line_idx = 0;
    auto next_line = [&](const char** ls, const char** le) {
      if (p >= end) {
        return false;
      }
      ++line_idx;
      *ls = p;
      const char* nl = static_cast<const char*>(memchr(p, '\n', end - p));
      *le = nl ? nl : end;
      p = nl ? nl + 1 : end;
      while ((*ls < *le) && ((**ls == ' ') || (**ls == '\t'))) {
        ++*ls;
      }
      return true;
    };
    auto tokens_of = [](const char* ls, const char* le, std::vector<std::pair<const char*, uint32_t>>* out) {
      out->clear();
      while (ls < le) {
        while ((ls < le) && ((*ls == ' ') || (*ls == '\t') || (*ls == '\r'))) {
          ++ls;
        }
        const char* t0 = ls;
        while ((ls < le) && (*ls != ' ') && (*ls != '\t') && (*ls != '\r')) {
          ++ls;
        }
        if (ls > t0) {
          out->emplace_back(t0, static_cast<uint32_t>(ls - t0));
        }
      }
    };
    const char* ls;
    const char* le;
    std::vector<std::pair<const char*, uint32_t>> toks;
    // The first nonblank line is the header.  (The reference means to skip '##' lines first, but its test compares three
    // bytes -- "##" and a terminator, plink2_ld.cc:7680 -- which no line of a text file matches; a '##' line is
    // therefore read as the header there, and here.)
    do {
      if (!next_line(&ls, &le)) {
        die(6, "Error: %s is empty.\n", fname.c_str());
      }
    } while (ls == le);  // (the reference's text reader skips blank lines)
    if (*ls == '#') {
      ++ls;
    }
    tokens_of(ls, le, &toks);
    // column search (SearchHeaderLine, plink2_cmdline.cc:4270): per field a priority list of names
    std::vector<std::string> want[3];
    want[0] = A.clump_id_field.empty() ? std::vector<std::string>{"ID", "SNP"} : A.clump_id_field;
    if (!A.clump_no_test) {
      want[1] = A.clump_test_field.empty() ? std::vector<std::string>{"TEST"} : A.clump_test_field;
    }
    want[2] = A.clump_p_field.empty() ? (A.clump_in_log10 ? std::vector<std::string>{"LOG10_P", "NEG_LOG10_P", "P"} : std::vector<std::string>{"P"})
                                      : A.clump_p_field;  // (:7631)
    // (search_a1, :7624: asked for, or the dataset has a variant with several ALT alleles and --clump-a1-field was not given empty)
    const bool search_a1 = A.clump_force_a1 || (pvar_multiallelic && !A.clump_no_a1);
    const std::vector<std::string> want_a1 = search_a1 ? (A.clump_a1_field.empty() ? std::vector<std::string>{"A1"} : A.clump_a1_field) : std::vector<std::string>();
    int col_a1 = -1;
    size_t prio_a1 = ~size_t(0);
    int col[3] = {-1, -1, -1};
    size_t prio[3] = {~size_t(0), ~size_t(0), ~size_t(0)};
    for (size_t c = 0; c < toks.size(); ++c) {
      const std::string name(toks[c].first, toks[c].second);
      for (size_t q = 0; q < want_a1.size(); ++q) {
        if ((want_a1[q] == name) && (prio_a1 >= q)) {
          if (prio_a1 == q) {
            die(6, "Error: Duplicate column header '%s' in --clump file.\n", name.c_str());
          }
          prio_a1 = q;
          col_a1 = static_cast<int>(c);
        }
      }
      for (int t = 0; t < 3; ++t) {
        for (size_t q = 0; q < want[t].size(); ++q) {
          if (want[t][q] == name && prio[t] >= q) {
            if (prio[t] == q) {
              die(6, "Error: Duplicate column header '%s' in --clump file.\n", name.c_str());
            }
            prio[t] = q;
            col[t] = static_cast<int>(c);
          }
        }
      }
    }
    if ((col[0] < 0) || (col[2] < 0)) {
      die(7, "Error: --clump requires ID and P columns.\n");
    }
    const int last_col = std::max(std::max(col[0], col_a1), std::max(col[1], col[2]));
    const std::vector<std::string> test_names = A.clump_test.empty() ? std::vector<std::string>{"ADD"} : A.clump_test;
    while (next_line(&ls, &le)) {
      if (ls == le) {
        continue;
      }
      tokens_of(ls, le, &toks);
      if (toks.empty()) {
        continue;
      }
      if (static_cast<int>(toks.size()) <= last_col) {
        die(7, "Error: Line %zu of %s has fewer tokens than expected.\n", line_idx, fname.c_str());
      }
      if (col[1] >= 0) {
        const std::string t(toks[col[1]].first, toks[col[1]].second);
        if (std::find(test_names.begin(), test_names.end(), t) == test_names.end()) {
          continue;
        }
      }
      const std::string ptok(toks[col[2]].first, toks[col[2]].second);
      double ln_pval = 0.0;
      bool scanned;
      if (!A.clump_in_log10) {
        const char* pe = scan_ln(ptok.c_str(), &ln_pval);
        scanned = pe && !*pe;
      } else {  // -log10(p) (:7744-7752)
        double neglog10;
        const char* pe;
        scanned = scan_double_plink(ptok.c_str(), &neglog10, &pe) && !*pe;
        if (scanned) {
          ln_pval = neglog10 * -2.3025850929940457;
          if (ln_pval > 0.0) {
            die(7, "Error: Invalid p-value on line %zu of %s.\n", line_idx, fname.c_str());
          }
        }
      }
      if (!scanned) {
        std::string low = ptok;
        for (char& ch : low) {
          ch = static_cast<char>(tolower(static_cast<unsigned char>(ch)));
        }
        if ((low == "na") || (low == "nan")) {
          continue;
        }
        if ((ptok == "INF") || (A.clump_in_log10 && (ptok == "inf"))) {  // PLINK 1.x underflow
          ln_pval = -708.3964185322641;
        } else {
          die(7, "Error: Invalid p-value on line %zu of %s.\n", line_idx, fname.c_str());
        }
      }
      const std::string id(toks[col[0]].first, toks[col[0]].second);
      const auto it = by_id.find(id);
      if (it == by_id.end()) {
        if (ln_pval <= ln_p1) {
          D->missing_ids.push_back(id);
        }
        continue;
      }
      if (it->second == kDup) {
        die(7, "Error: --clump variant ID '%s' appears multiple times in main dataset.\n", id.c_str());
      }
      const uint32_t k = it->second;
      const uint32_t iv = inc[k];
      const bool multi_k = V.alt_ct[iv] > 1;
      uint32_t aidx = 0;
      if (multi_k || A.clump_force_a1) {  // (:7782-7818)
        if (col_a1 < 0) {
          die(7, "Error: Variant ID on line %zu of %s is multiallelic, but there is no A1 column.\n", line_idx, fname.c_str());
        }
        const std::string a1(toks[col_a1].first, toks[col_a1].second);
        const uint32_t allele_ct = static_cast<uint32_t>(V.alt_ct[iv]) + 1;
        for (; aidx < allele_ct; ++aidx) {
          if (a1 == allele_text(V, iv, aidx)) {
            break;
          }
        }
        if (aidx == allele_ct) {
          if (ln_pval <= ln_p1) {
            D->missing_pairs.push_back(id + "\t" + a1);
          }
          continue;
        }
        if (!multi_k) {
          a1_alt = aidx;
          aidx = 0;
        }
      }
      const uint32_t slot = SL.base[k] + aidx;
      if (ln_pval > load_thresh) {
        if (ln_pval > 0.0) {
          die(6, "Error: p-value > 1 on line %zu of %s.\n", line_idx, fname.c_str());
        }
        if (nonsig_needed && (D->ln_bins.empty() || (ln_pval > D->ln_bins.back()))) {
          D->nonsig[slot] += 1;
          D->observed[slot] = 1;
        }
        continue;
      }
      if (D->best_ln[slot] >= ln_pval) {  // (>=: the reports are read last to first, so ties end up with the first one, :7833)
        D->best_ln[slot] = ln_pval;
        D->best_file[slot] = static_cast<uint16_t>(file_idx1);
        D->best_a1[slot] = static_cast<uint8_t>(a1_alt);
      }
      D->observed[slot] = 1;
      if (keep_entries) {
        D->entries[slot].push_back(static_cast<uint32_t>((a1_alt << 30) | (file_idx1 << 12) | (clump_bin(D->ln_bins, ln_pval) << 1) | (ln_pval > ln_p2)));
      }
    }
  }
}


// --clump-range[0] (LoadAndSortIntervalBed / LoadIntervalBed, plink2_set.cc:39-330, :495-638): lines `chrom first last name`;
// per chromosome the names in natural order, each with its intervals -- stretched by the border, half-open, sorted, merged.
struct ClumpRanges {
  // chromosome key (the numeric code of a standard name, else the name itself) -> (name, flattened [start, end) pairs)
  std::map<std::string, std::vector<std::pair<std::string, std::vector<uint32_t>>>> by_chr;
  static std::string key_of(const std::string& chrom) {
    const int code = chrom_code(chrom);
    return (code >= 0) ? std::to_string(code) : chrom;
  }
  void load(const Args& A, const Variants& V, const std::vector<uint32_t>& inc) {
    std::unordered_set<std::string> known;  // chromosomes the dataset names (an unknown non-standard name is an error there)
    for (uint32_t v : inc) {
      known.insert(key_of(V.chrom[v]));
    }
    const std::string text = slurp(A.clump_range);
    std::map<std::string, std::map<std::string, std::vector<std::pair<uint32_t, uint32_t>>, bool (*)(const std::string&, const std::string&)>> raw;
    size_t line_idx = 0;
    for (size_t p0 = 0; p0 < text.size();) {
      size_t p1 = text.find('\n', p0);
      if (p1 == std::string::npos) {
        p1 = text.size();
      }
      ++line_idx;
      std::vector<std::string> tok;
      for (size_t q = p0; q < p1;) {
        while ((q < p1) && (static_cast<unsigned char>(text[q]) <= ' ')) {
          ++q;
        }
        size_t q1 = q;
        while ((q1 < p1) && (static_cast<unsigned char>(text[q1]) > ' ')) {
          ++q1;
        }
        if (q1 > q) {
          tok.emplace_back(text, q, q1 - q);
        }
        q = q1;
      }
      p0 = p1 + 1;
      if (tok.empty()) {
        continue;
      }
      if (tok.size() < 4) {
        die(6, "Error: Line %zu of %s has fewer tokens than expected.\n", line_idx, A.clump_range.c_str());
      }
      const std::string key = key_of(tok[0]);
      if ((chrom_code(tok[0]) < 0) && !known.count(key)) {
        die(6, "Error: Invalid chromosome code on line %zu of %s.\n", line_idx, A.clump_range.c_str());
      }
      uint64_t first = 0, last = 0;
      for (int w = 1; w <= 2; ++w) {
        uint64_t val = 0;
        bool ok = !tok[w].empty();
        for (char ch : tok[w]) {
          ok = ok && (ch >= '0') && (ch <= '9') && (val < 0x7fffffffull);
          val = val * 10 + static_cast<uint64_t>(ch - '0');
        }
        if ((!ok) || (val > 0x7ffffffeull)) {
          die(6, "Error: Invalid range %s position on line %zu of %s.\n", (w == 1) ? "start" : "end", line_idx, A.clump_range.c_str());
        }
        ((w == 1) ? first : last) = val;
      }
      first += A.clump_range0 ? 1 : 0;
      if (last < first) {
        die(6, "Error: Range end position smaller than range start on line %zu of %s.\n", line_idx, A.clump_range.c_str());
      }
      first = (A.clump_range_border > first) ? 0 : (first - A.clump_range_border);
      last += A.clump_range_border;
      auto it = raw.find(key);
      if (it == raw.end()) {
        it = raw.emplace(key, std::map<std::string, std::vector<std::pair<uint32_t, uint32_t>>, bool (*)(const std::string&, const std::string&)>(natural_less)).first;
      }
      it->second[tok[3]].emplace_back(static_cast<uint32_t>(first), static_cast<uint32_t>(last + 1));
    }
    for (auto& chr : raw) {
      auto& out = by_chr[chr.first];
      for (auto& g : chr.second) {
        std::sort(g.second.begin(), g.second.end());
        std::vector<uint32_t> flat;
        for (const auto& iv : g.second) {
          if ((!flat.empty()) && (iv.first <= flat.back())) {
            flat.back() = std::max(flat.back(), iv.second);
          } else {
            flat.push_back(iv.first);
            flat.push_back(iv.second);
          }
        }
        out.emplace_back(g.first, std::move(flat));
      }
    }
  }
  // names of `chrom` with an interval meeting [first_bp, end_bp), comma-separated (empty: none)
  std::string overlaps(const std::string& chrom, uint32_t first_bp, uint32_t end_bp) const {
    std::string names;
    const auto it = by_chr.find(key_of(chrom));
    if (it == by_chr.end()) {
      return names;
    }
    for (const auto& g : it->second) {
      bool hit = false;
      for (size_t k = 0; (k < g.second.size()) && !hit; k += 2) {
        hit = (g.second[k] < end_bp) && (g.second[k + 1] > first_bp);
      }
      if (hit) {
        names += g.first;
        names += ',';
      }
    }
    if (!names.empty()) {
      names.pop_back();
    }
    return names;
  }
};


int clump_reports(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, const std::vector<uint32_t>& chr_idx,
                  const std::vector<uint32_t>& bps, uint32_t founder_ct,
                  const std::function<void(ldp_engine*, const std::vector<uint32_t>&)>& feed, const ClumpSex& SX) {
  if (founder_ct < 2) {
    die(7, "Error: --clump requires at least two founders.  (--make-founders may come in handy\nhere.)\n");
  }
  const double t_start = now_s();
  ClumpSlots SL;
  SL.build(V, inc);
  ClumpData D;
  clump_load_report(A, V, inc, SL, &D);
  if (!D.missing_ids.empty()) {  // natural-sorted, deduplicated (plink2_ld.cc:7909-7931)
    std::sort(D.missing_ids.begin(), D.missing_ids.end(), natural_less);
    D.missing_ids.erase(std::unique(D.missing_ids.begin(), D.missing_ids.end()), D.missing_ids.end());
    const std::string path = A.out + ".clumps.missing_id" + (A.clump_zs ? ".zst" : "");
    OutFile mf;
    mf.open(path, A.clump_zs);
    for (const std::string& s : D.missing_ids) {
      mf.write(s.data(), s.size());
      mf.write("\n", 1);
    }
    mf.close();
    const size_t n = D.missing_ids.size();
    logprintf("Warning: %zu top variant ID%s in --clump file%s missing from main dataset.  ID%s written to %s .\n", n, (n == 1) ? "" : "s",
              (A.clump_files.size() == 1) ? "" : "s", (n == 1) ? "" : "s", path.c_str());
  }
  if (!D.missing_pairs.empty()) {  // (:7933-7952)
    std::sort(D.missing_pairs.begin(), D.missing_pairs.end(), natural_less);
    D.missing_pairs.erase(std::unique(D.missing_pairs.begin(), D.missing_pairs.end()), D.missing_pairs.end());
    const std::string path = A.out + ".clumps.missing_allele" + (A.clump_zs ? ".zst" : "");
    OutFile mf;
    mf.open(path, A.clump_zs);
    for (const std::string& s : D.missing_pairs) {
      mf.write(s.data(), s.size());
      mf.write("\n", 1);
    }
    mf.close();
    const size_t n = D.missing_pairs.size();
    logprintf("Warning: %zu top (variant ID, A1 allele) pair%s in --clump file%s missing from main dataset due to allele rather than variant ID.  (Variant ID, A1 allele) pair%s written to %s .\n",
              n, (n == 1) ? "" : "s", (A.clump_files.size() == 1) ? "" : "s", (n == 1) ? "" : "s", path.c_str());
  }
  // observed variants (named by a usable report line) in dataset order, and the index candidates among them:
  // best p <= p1, ranked by (ln p, position in the dataset) (ClumpPvalCmp; plink2_ld.cc:7996-8040)
  std::vector<uint32_t> obs;  // -> slot (ClumpSlots: a biallelic variant, or one allele of a multiallelic one)
  for (uint32_t slot = 0; slot < D.observed.size(); ++slot) {
    if (D.observed[slot]) {
      obs.push_back(slot);
    }
  }
  auto var_of = [&](uint32_t o) { return SL.var[obs[o]]; };  // observed index -> index into inc[]
  const uint32_t n_obs = static_cast<uint32_t>(obs.size());
  std::vector<uint32_t> cand;  // -> observed index, rank order
  for (uint32_t o = 0; o < n_obs; ++o) {
    if (D.best_ln[obs[o]] <= A.clump_ln_p1) {
      cand.push_back(o);
    }
  }
  if (cand.empty()) {
    logprintf("Warning: No significant --clump results.  Skipping.\n");
    return 0;
  }
  std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) {
    const double la = D.best_ln[obs[a]], lb = D.best_ln[obs[b]];
    return (la != lb) ? (la < lb) : (a < b);
  });
  const uint32_t cand_ct = static_cast<uint32_t>(cand.size());
  std::vector<uint8_t> is_cand(n_obs, 0);
  for (uint32_t o : cand) {
    is_cand[o] = 1;
  }

  // r^2 > threshold pairs within +-bp_radius, one endpoint an index candidate: the windowed-table kernels with the
  // filter in their epilogue.  The kernel keeps r^2 >= the number the user gave; the reference's test is r^2 >
  // that number * (1 + 2^-44) (plink2.cc:5059, ClumpHighmemR2 plink2_ld.cc:7352), applied here on the same doubles.
  std::vector<std::pair<uint32_t, uint32_t>> links;  // (candidate, partner), observed indices
  double t_rows = t_start, t_pairs = t_start;
  // Only observed variants within the radius of some index candidate can ever be tested (the reference's islands,
  // GetNextIslandIdxs plink2_ld.cc:5642, make the same cut): the engine holds those, in dataset order.
  std::vector<uint32_t> o_chr(n_obs), o_bp(n_obs);
  for (uint32_t o = 0; o < n_obs; ++o) {
    o_chr[o] = chr_idx[var_of(o)];
    o_bp[o] = bps[var_of(o)];
  }
  std::vector<int32_t> cover(static_cast<size_t>(n_obs) + 1, 0);
  bool any_pair = false;
  for (uint32_t o : cand) {
    uint32_t lo = o, hi = o;
    for (uint32_t step = 1; step;) {  // galloping search on both sides
      step = 0;
      uint32_t jump = 1;
      while ((lo >= jump) && (o_chr[lo - jump] == o_chr[o]) && (o_bp[o] - o_bp[lo - jump] <= A.clump_bp_radius)) {
        lo -= jump;
        jump *= 2;
        step = 1;
      }
      jump = 1;
      while ((hi + jump < n_obs) && (o_chr[hi + jump] == o_chr[o]) && (o_bp[hi + jump] - o_bp[o] <= A.clump_bp_radius)) {
        hi += jump;
        jump *= 2;
        step = 1;
      }
    }
    any_pair |= (hi > lo);
    ++cover[lo];
    --cover[hi + 1];
  }
  std::vector<uint32_t> sub;  // engine row -> observed index
  {
    int32_t depth = 0;
    for (uint32_t o = 0; o < n_obs; ++o) {
      depth += cover[o];
      if (depth > 0) {
        sub.push_back(o);
      }
    }
  }
  const uint32_t n_sub = static_cast<uint32_t>(sub.size());
  if (any_pair) {
    ldp_params RP;
    memset(&RP, 0, sizeof(RP));
    RP.founder_ct = founder_ct;
    RP.prune_window_size = 2;
    RP.prune_window_incr = 1;
    RP.prune_last_param = 0.5;
    RP.device = 0;
    if (ldp_device_count() < 1) {
      die(16, "Error: no usable HIP device (plink2-hip has no CPU compute path).\n");
    }
    ldp_engine* e = nullptr;
    if (ldp_create(&RP, &e)) {
      die(16, "Error: engine setup failed.\n");
    }
    std::vector<uint32_t> s_chr(n_sub), s_bp(n_sub), s_raw(n_sub);
    for (uint32_t q = 0; q < n_sub; ++q) {
      s_chr[q] = o_chr[sub[q]];
      s_bp[q] = o_bp[sub[q]];
      s_raw[q] = inc[var_of(sub[q])];
    }
    if (ldp_set_variants_vcor(e, n_sub, s_chr.data(), s_bp.data(), A.clump_bp_radius, 0xffffffffu)) {
      die(16, "Error: engine setup failed: %s\n", ldp_last_error(e));
    }
    feed(e, s_raw);
    // an allele of a multiallelic variant: the row PgrGetInv1 returns for it (copies of the OTHER alleles, pgenlib_read.cc:5544-5563; r^2 does not
    // see the inversion), built on the host over the main-track row feed() has just loaded (:8813-8835)
    std::vector<int32_t> s_aidx(n_sub, -1);
    for (uint32_t q = 0; q < n_sub; ++q) {
      if (SL.multiallelic(obs[sub[q]])) {
        s_aidx[q] = static_cast<int32_t>(SL.aidx(obs[sub[q]]));
        SX.allele_row(e, q, s_raw[q], s_aidx[q], false, founder_ct != SX.raw_sample_ct);
      }
    }
    // sex chromosomes: chrY rows with the female founders' calls missing; chrX pairs through the male-weighted sums when the
    // founders are of both kinds (is_x, :8472-8481), their own engine for the male founders' tuples
    std::vector<uint8_t> s_is_x(n_sub, 0);
    bool any_x = false, any_y = false;
    for (uint32_t q = 0; q < n_sub; ++q) {
      const uint8_t cls = (*SX.vcls)[var_of(sub[q])];
      any_y = any_y || (cls == 4);
      if ((cls == 3) && SX.founder_male_ct && (SX.founder_male_ct != founder_ct)) {
        s_is_x[q] = 1;
        any_x = true;
      }
    }
    if (any_y) {
      if (!(SX.founder_male_ct + SX.founder_nosex_ct)) {  // :8162-8166 (there: an index variant on chrY; here: any chrY row that can be tested)
        die(7, "Error: --clump: chrY index variant(s) are present, but all founders in the main\ndataset are females.\n");
      }
      if (SX.founder_male_ct + SX.founder_nosex_ct != founder_ct) {
        for (uint32_t q = 0; q < n_sub; ++q) {
          if ((*SX.vcls)[var_of(sub[q])] == 4) {
            SX.allele_row(e, q, s_raw[q], s_aidx[q], true, founder_ct != SX.raw_sample_ct);
          }
        }
      }
    }
    XWeighted xw;
    std::vector<uint32_t> band_lo;
    if (any_x) {
      band_lo.resize(n_sub);
      uint64_t cand_pairs = 0;
      ldp_get_band(e, band_lo.data(), &cand_pairs);
      // the chrX rows are one run of the engine's rows (a chromosome is): they go into two all-pairs engines of their own -- every founder, the male
      // founders -- whose dense blocks give the windows' pairs on the pair kernels, weighted on the device (XWeighted::band_hits).  --debug-x-host: the
      // round-4 path instead, pair lists through ldp_pair_stats and the weighting on the host.
      uint32_t x0 = 0, x1 = 0;
      bool one_run = true;
      for (uint32_t q = 0; q < n_sub; ++q) {
        if (s_is_x[q]) {
          if (x1 == 0) {
            x0 = q;
          } else if (x1 != q) {
            one_run = false;
          }
          x1 = q + 1;
        }
      }
      const bool on_device = one_run && !g_dbg.x_host;
      const uint32_t first = on_device ? x0 : 0, cnt = on_device ? (x1 - x0) : n_sub;
      const std::vector<uint32_t> x_raw(s_raw.begin() + first, s_raw.begin() + first + cnt);
      ldp_engine* xa = e;
      ldp_engine* xm = nullptr;
      ldp_params MP = RP;
      MP.founder_ct = SX.founder_male_ct;
      if (ldp_create(&MP, &xm) || ldp_set_variants_matrix(xm, cnt)) {
        die(16, "Error: engine setup failed.\n");
      }
      SX.feed_cols(xm, x_raw, &SX.male_cols);
      if (on_device) {
        xa = nullptr;
        if (ldp_create(&RP, &xa) || ldp_set_variants_matrix(xa, cnt)) {
          die(16, "Error: engine setup failed.\n");
        }
        if (g_dbg.x_rows) {
          (void)ldp_debug_set_option(xa, "x_rows", static_cast<double>(g_dbg.x_rows));
        }
        feed(xa, x_raw);
      }
      for (uint32_t q = 0; q < cnt; ++q) {
        if (s_aidx[first + q] >= 0) {
          SX.allele_row(xm, q, x_raw[q], s_aidx[first + q], false, true);
          if (on_device) {
            SX.allele_row(xa, q, x_raw[q], s_aidx[first + q], false, founder_ct != SX.raw_sample_ct);
          }
        }
      }
      // one orientation for both tuples of a pair: the all-founders engine's (the male engine chose its major alleles from the male
      // founders alone).  Which one it is does not matter inside chrX: the weight is dyadic and every sum exact.
      std::vector<ldp_variant_rec> ra(cnt), rm(cnt);
      if (ldp_get_variant_recs(xa, 0, cnt, ra.data()) || ldp_get_variant_recs(xm, 0, cnt, rm.data())) {
        die(16, "Error: %s\n", ldp_last_error(xa));
      }
      std::vector<uint8_t> flip(cnt);
      for (uint32_t q = 0; q < cnt; ++q) {
        flip[q] = static_cast<uint8_t>((ra[q].flags ^ rm[q].flags) & 1u);
      }
      if (on_device) {
        xw.band_all = xa;
        xw.band_male = xm;
        xw.band_first = x0;
        xw.band_ct = cnt;
        xw.band_flip_male = flip;
      } else {
        xw.all = e;
        xw.male = xm;
        xw.is_x = s_is_x;
        xw.flip_male = flip;
      }
    }
    t_rows = now_s();
    std::vector<ldp_r2_hit> hits(1u << 24);
    const double min_r2 = std::max(A.clump_r2_raw, 0.0);
    uint32_t rows_per_call = 65536;
    for (uint32_t r0 = 0; r0 < n_sub;) {
      const uint32_t rows = std::min(rows_per_call, n_sub - r0);
      uint64_t found = 0;
      if (ldp_r2_unphased_hits(e, r0, rows, min_r2, hits.data(), hits.size(), &found)) {
        die(16, "Error: %s\n", ldp_last_error(e));
      }
      if (found > hits.size()) {
        if (rows == 1) {
          die(2, "Error: one variant has more --clump-r2 partners than the filter buffer holds.\n");
        }
        rows_per_call = std::max(1u, rows / 2);
        continue;
      }
      auto link = [&](uint32_t first, uint32_t second) {
        const uint32_t a = sub[first], b = sub[second];
        if (is_cand[a]) {
          links.emplace_back(a, b);
        }
        if (is_cand[b]) {
          links.emplace_back(b, a);
        }
      };
      for (uint64_t q = 0; q < found; ++q) {
        const ldp_r2_hit& h = hits[q];
        if ((!(h.r2 > A.clump_r2)) || s_is_x[h.second]) {  // (a window never leaves its chromosome: chrX rows pair with chrX rows)
          continue;
        }
        link(h.first, h.second);
      }
      if (any_x && xw.band_ready()) {
        const uint32_t j0 = std::max(r0, xw.band_first), j1 = std::min(r0 + rows, xw.band_first + xw.band_ct);
        if (j0 < j1) {
          xw.band_hits(band_lo.data(), j0, j1, min_r2, &hits, [&](uint32_t i, uint32_t j, double v) {
            if ((v > A.clump_r2) && (is_cand[sub[i]] || is_cand[sub[j]])) {
              link(i, j);
            }
          });
        }
      } else if (any_x) {
        std::vector<uint32_t> fi, se;
        std::vector<double> vals;
        for (uint32_t j = r0; j < r0 + rows; ++j) {
          for (uint32_t i = band_lo[j]; s_is_x[j] && (i < j); ++i) {
            if (is_cand[sub[i]] || is_cand[sub[j]]) {
              fi.push_back(i);
              se.push_back(j);
            }
          }
        }
        xw.pairs(fi, se, &vals);
        for (size_t q = 0; q < fi.size(); ++q) {
          if (vals[q] > A.clump_r2) {
            link(fi[q], se[q]);
          }
        }
      }
      r0 += rows;
    }
    if (xw.male) {
      ldp_destroy(xw.male);
    }
    xw.band_destroy();
    ldp_destroy(e);
    t_pairs = now_s();
  }
  std::sort(links.begin(), links.end());
  std::vector<uint64_t> link_off(static_cast<size_t>(n_obs) + 1, 0);
  for (const auto& l : links) {
    ++link_off[l.first + 1];
  }
  for (uint32_t o = 0; o < n_obs; ++o) {
    link_off[o + 1] += link_off[o];
  }

  // the greedy pass (plink2_ld.cc:8610-8700): candidates in rank order; one already inside a clump is skipped, the
  // others take every still-unclumped window member above the threshold.  --clump-allow-overlap (:8135-8141,7471-7490): a
  // member stays available to later clumps -- only index variants leave the pool -- but joining a clump still takes a
  // candidate off the list of future index variants.
  std::vector<uint64_t> mem_off(static_cast<size_t>(cand_ct) + 1, 0);
  std::vector<uint32_t> members;
  uint32_t clump_ct = 0;
  {
    std::vector<uint8_t> in_pool(n_obs, 1), may_lead(n_obs, 1);
    std::vector<uint32_t> cur;
    for (uint32_t r = 0; r < cand_ct; ++r) {
      const uint32_t o = cand[r];
      mem_off[r] = members.size();
      if (!(A.clump_allow_overlap ? may_lead[o] : in_pool[o])) {
        continue;
      }
      ++clump_ct;
      in_pool[o] = 0;
      cur.assign(1, o);
      for (uint64_t q = link_off[o]; q < link_off[o + 1]; ++q) {
        const uint32_t m = links[q].second;
        if (in_pool[m]) {
          cur.push_back(m);
          may_lead[m] = 0;
          if (!A.clump_allow_overlap) {
            in_pool[m] = 0;
          }
        }
      }
      std::sort(cur.begin(), cur.end());  // members in dataset order (ordered_members, :8936-8970)
      members.insert(members.end(), cur.begin(), cur.end());
    }
    mem_off[cand_ct] = members.size();
  }
  logprintf("--clump: %u clump%s formed from %u index candidate%s.\n", clump_ct, (clump_ct == 1) ? "" : "s", cand_ct, (cand_ct == 1) ? "" : "s");

  // <out>.clumps (plink2_ld.cc:9003-9405): [chrom pos] ID [ref alt1 alt provref a1 f] P [total] [bounds] [bins] [sp2]
  const std::string path = A.out + ".clumps" + (A.clump_zs ? ".zst" : "");
  OutFile f;
  f.open(path, A.clump_zs);
  // (several reports: an F column names the report of the index variant's best p-value, and SP2 entries carry theirs)
  const bool multi = (A.clump_files.size() > 1);
  const uint32_t cols = A.clump_cols;
  const bool f_col = (cols & kClumpColF) || ((cols & kClumpColMaybeF) && multi);
  const bool sp2_col = (cols & kClumpColSp2) != 0;
  const bool f_in_sp2 = sp2_col && ((cols & kClumpColF) || multi);
  const bool ranges_col = !A.clump_range.empty();
  ClumpRanges ranges;
  if (ranges_col) {
    ranges.load(A, V, inc);
  }
  const bool bounds_col = (cols & kClumpColBounds) || ((cols & kClumpColMaybeBounds) && ranges_col);
  const bool save_all_fidxs = (multi || A.clump_force_a1) && sp2_col;  // (:7633)
  const bool a1_col = (cols & kClumpColA1) || ((cols & kClumpColMaybeA1) && SL.any_multiallelic);  // (:9017)
  const size_t bin_bound_ct = D.ln_bins.size();
  bool provref_col = false;
  if (cols & kClumpColRef) {  // ProvrefCol (plink2_common.h:1549)

    if (cols & kClumpColProvref) {
      provref_col = true;
    } else if (cols & kClumpColMaybeprovref) {
      provref_col = (SX.prov_storage == 2);
      for (size_t k = 0; (SX.prov_storage == 3) && (!provref_col) && (k < inc.size()); ++k) {
        provref_col = (SX.prov_bits[inc[k] >> 3] >> (inc[k] & 7)) & 1;
      }
    }
  }
  char num[64];
  std::string buf = "#";
  if (cols & kClumpColChrom) buf += "CHROM\t";
  if (cols & kClumpColPos) buf += "POS\t";
  buf += "ID\t";
  if (cols & kClumpColRef) buf += "REF\t";
  if (cols & kClumpColAlt1) buf += "ALT1\t";
  if (cols & kClumpColAlt) buf += "ALT\t";
  if (provref_col) buf += "PROVISIONAL_REF?\t";
  if (a1_col) buf += "A1\t";
  if (f_col) buf += "F\t";
  buf += A.clump_out_log10 ? "NEG_LOG10_P" : "P";
  if (cols & kClumpColTotal) buf += "\tTOTAL";
  if (bounds_col) buf += "\tCLUMP_FIRST_POS\tCLUMP_LAST_POS";
  if (bin_bound_ct) {
    buf += "\tNONSIG";
    for (size_t b = bin_bound_ct; b; --b) {
      buf += "\tS";
      buf.append(num, format_ln_g6(D.ln_bins[b - 1], num) - num);
    }
  }
  if (sp2_col) buf += "\tSP2";
  if (ranges_col) buf += "\tRANGES";
  buf += '\n';
  std::vector<uint64_t> bins(bin_bound_ct + 1);
  for (uint32_t r = 0; r < cand_ct; ++r) {
    if (mem_off[r] == mem_off[r + 1]) {
      continue;
    }
    const uint32_t io = cand[r];
    const uint32_t iv = inc[var_of(io)];
    const double index_ln = D.best_ln[obs[io]];
    if (cols & kClumpColChrom) {
      buf += V.chrom[iv];
      buf += '\t';
    }
    if (cols & kClumpColPos) {
      buf += std::to_string(V.bp[iv]);
      buf += '\t';
    }
    buf += V.id[iv];
    buf += '\t';
    if (cols & kClumpColRef) {
      buf += V.ref[iv];
      buf += '\t';
    }
    if (cols & kClumpColAlt1) {
      buf.append(V.alt[iv], 0, std::min(V.alt[iv].find(','), V.alt[iv].size()));
      buf += '\t';
    }
    if (cols & kClumpColAlt) {
      buf += V.alt[iv];
      buf += '\t';
    }
    if (provref_col) {
      buf += ((SX.prov_storage == 2) || ((SX.prov_storage == 3) && ((SX.prov_bits[iv >> 3] >> (iv & 7)) & 1))) ? 'Y' : 'N';
      buf += '\t';
    }
    if (a1_col) {  // (a biallelic variant: the best line's A1 with --clump-force-a1, else '.'; an allele of a multiallelic one: itself, :9186-9201)
      if (SL.multiallelic(obs[io])) {
        buf += allele_text(V, iv, SL.aidx(obs[io]));
        buf += '\t';
      } else if (A.clump_force_a1) {
        buf += D.best_a1[obs[io]] ? V.alt[iv] : V.ref[iv];
        buf += '\t';
      } else {
        buf += ".\t";
      }
    }
    const uint32_t index_file = D.best_file[obs[io]];
    if (f_col) {
      buf += std::to_string(index_file);
      buf += '\t';
    }
    if (A.clump_out_log10) {
      buf.append(num, format_g6(-0.43429448190325176 * index_ln, num) - num);  // (:9214-9216)
    } else {
      buf.append(num, format_ln_g6(index_ln, num) - num);
    }
    if ((cols & kClumpColTotal) || bin_bound_ct) {
      uint64_t total = 0;
      std::fill(bins.begin(), bins.end(), 0);
      for (uint64_t q = mem_off[r]; q < mem_off[r + 1]; ++q) {
        const uint32_t k = obs[members[q]];  // (slot)
        bins[bin_bound_ct] += D.nonsig[k];
        for (uint32_t en : D.entries[k]) {
          ++bins[bin_bound_ct ? ((en >> 1) & 2047) : 0];
        }
      }
      // (the index variant's own line is the clump, not one of its members: with bins it leaves its bin, without them the
      // plain count, :9240-9262)
      --bins[bin_bound_ct ? clump_bin(D.ln_bins, index_ln) : 0];
      for (uint64_t b : bins) {
        total += b;
      }
      if (cols & kClumpColTotal) {
        buf += '\t';
        buf += std::to_string(total);
      }
    }
    uint32_t first_bp = 0xffffffffu, last_bp = 0;
    if (bounds_col || ranges_col) {
      // bp range of the members with a line at or below p2 (:9270-9311)
      for (uint64_t q = mem_off[r]; q < mem_off[r + 1]; ++q) {
        const uint32_t k = obs[members[q]];
        // (with several reports -- or --clump-force-a1 -- and SP2 the reference keeps a word behind every entry, report number
        // times two plus the forced-A1 bit, and this scan, :9271-9279, does not step over it: it tests that word's low bit like
        // an entry's.  Reproduced: entries are walked the way its list is, latest read first.)
        bool hit = false;
        const std::vector<uint32_t>& ent_k = D.entries[k];
        for (size_t x = ent_k.size(); x && !hit; --x) {
          const uint32_t en = ent_k[x - 1];
          hit = (!(en & 1)) || (save_all_fidxs && !((en >> 30) & 1));
        }
        if (hit) {
          if (first_bp == 0xffffffffu) {
            first_bp = V.bp[inc[SL.var[k]]];
          }
          last_bp = V.bp[inc[SL.var[k]]];
        }
      }
    }
    if (bounds_col) {
      buf += '\t';
      if (first_bp != 0xffffffffu) {
        buf += std::to_string(first_bp);
        buf += '\t';
        buf += std::to_string(last_bp);
      } else {
        buf += ".\t.";
      }
    }
    for (size_t b = bin_bound_ct + 1; bin_bound_ct && b; --b) {
      buf += '\t';
      buf += std::to_string(bins[b - 1]);
    }
    bool nonempty = false;
    if (sp2_col) {
      buf += '\t';
    }
    for (uint64_t q = mem_off[r]; sp2_col && (q < mem_off[r + 1]); ++q) {
      const uint32_t m = members[q];
      // a member's lines, latest read first (the reference walks its linked list from the head, :7851,9330): report 1's
      // lines bottom-up, then report 2's, ...; the index variant's own line in its own report is the clump itself
      const std::vector<uint32_t>& ent = D.entries[obs[m]];
      for (size_t x = ent.size(); x; --x) {
        const uint32_t en = ent[x - 1];
        const uint32_t file = (en >> 12) & 0x3ffff;
        if ((en & 1) || ((m == io) && (file == index_file))) {
          continue;
        }
        buf += V.id[inc[var_of(m)]];
        if (A.clump_force_a1 || SL.multiallelic(obs[m])) {  // (:9355-9358: the entity's allele, plus the entry's forced-A1 bit)
          buf += '(';
          buf += allele_text(V, inc[var_of(m)], SL.aidx(obs[m]) + ((A.clump_force_a1 && sp2_col) ? ((en >> 30) & 1) : 0));
          buf += ')';
        }
        if (f_in_sp2) {
          buf += '(';
          buf += std::to_string(file);
          buf += ')';
        }
        buf += ',';
        nonempty = true;
      }
    }
    if (sp2_col) {
      if (nonempty) {
        buf.pop_back();
      } else {
        buf += '.';
      }
    }
    if (ranges_col) {  // (:9377-9400)
      const std::string names = (first_bp != 0xffffffffu) ? ranges.overlaps(V.chrom[iv], first_bp, last_bp + 1) : std::string();
      buf += '\t';
      buf += names.empty() ? std::string(".") : names;
    }
    buf += '\n';
    if (buf.size() > (1u << 20)) {
      f.write(buf.data(), buf.size());
      buf.clear();
    }
  }
  f.write(buf.data(), buf.size());
  f.close();
  logprintf("Results written to %s .\n", path.c_str());
  if (A.timing) {
    fprintf(stderr, "[timing] clump: %u observed variants (%u near an index candidate), %u index candidates, %zu links; report+rows %.3f s, pair kernels %.3f s, greedy+write %.3f s\n",
            n_obs, n_sub, cand_ct, links.size(), t_rows - t_start, t_pairs - t_rows, now_s() - t_pairs);
  }
  return 0;
}


}  // namespace p2h
