// p2h_inputs.cpp -- plink2-hip: chromosome / sample filters and load_inputs(): everything the commands share (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

// --chr / --not-chr / --autosome: a chromosome's numeric code (1..22, X 23, Y 24, XY 25, MT 26, 0; -1 for other names)
int chrom_code(const std::string& name_in) {
  std::string name = name_in;
  if (name.size() > 3 && (name[0] | 32) == 'c' && (name[1] | 32) == 'h' && (name[2] | 32) == 'r') {
    name = name.substr(3);
  }
  bool numeric = !name.empty();
  for (char c : name) {
    numeric = numeric && (c >= '0' && c <= '9');
  }
  if (numeric) {
    const long v = strtol(name.c_str(), nullptr, 10);
    return (v <= 26) ? static_cast<int>(v) : -1;
  }
  if (ieq(name.c_str(), "X")) return 23;
  if (ieq(name.c_str(), "Y")) return 24;
  if (ieq(name.c_str(), "XY")) return 25;
  if (ieq(name.c_str(), "MT") || ieq(name.c_str(), "M")) return 26;
  return -1;
}

// is chromosome `name` named by one of the --chr style terms ("7", "chr7", "3-9", "X", "contig_12")?
bool chrom_listed(const std::vector<std::string>& terms, const std::string& name) {
  const int code = chrom_code(name);
  std::string bare = name;
  if (bare.size() > 3 && (bare[0] | 32) == 'c' && (bare[1] | 32) == 'h' && (bare[2] | 32) == 'r') {
    bare = bare.substr(3);
  }
  for (const std::string& t : terms) {
    const size_t dash = t.find('-');
    if ((dash != std::string::npos) && (dash > 0) && (dash + 1 < t.size())) {
      const int lo = chrom_code(t.substr(0, dash)), hi = chrom_code(t.substr(dash + 1));
      if ((lo >= 0) && (hi >= lo)) {
        if ((code >= lo) && (code <= hi)) {
          return true;
        }
        continue;
      }
    }
    const int tc = chrom_code(t);
    if (tc >= 0) {
      if (tc == code) {
        return true;
      }
      continue;
    }
    std::string tb = t;
    if (tb.size() > 3 && (tb[0] | 32) == 'c' && (tb[1] | 32) == 'h' && (tb[2] | 32) == 'r') {
      tb = tb.substr(3);
    }
    if (tb == bare) {
      return true;
    }
  }
  return false;
}

std::vector<std::string> tokens_of_file(const std::string& path) {
  const std::string text = slurp(path);
  std::vector<std::string> out;
  for (size_t p0 = 0; p0 < text.size();) {
    while ((p0 < text.size()) && (static_cast<unsigned char>(text[p0]) <= ' ')) {
      ++p0;
    }
    size_t p1 = p0;
    while ((p1 < text.size()) && (static_cast<unsigned char>(text[p1]) > ' ')) {
      ++p1;
    }
    if (p1 > p0) {
      out.emplace_back(text, p0, p1 - p0);
    }
    p0 = p1;
  }
  return out;
}

// --keep / --remove files (LoadXidHeader + LoadSampleIds, plink2_common.cc:1313,1707): "FID<tab>IID" keys.  A header line
// "#FID IID ..." or "#IID ..." names the columns; without one, a line of two or more tokens is FID IID and a line of one is
// an IID with FID "0".
void load_sample_id_list(const std::string& path, const char* flag, std::vector<std::string>* keys) {
  std::ifstream in(path);
  if (!in) {
    die(3, "Error: Failed to open %s.\n", path.c_str());
  }
  std::string line;
  int mode = 0;  // 0: no header (FID IID or IID), 1: #FID IID, 2: #IID
  bool first = true;
  size_t line_idx = 0;
  while (std::getline(in, line)) {
    ++line_idx;
    std::vector<std::string> t = split_ws(line);
    if (t.empty()) {
      continue;
    }
    if (t[0][0] == '#') {
      if (first && ((t[0] == "#FID") || (t[0] == "#IID"))) {
        first = false;
        if (t[0] == "#FID") {
          if ((t.size() < 2) || (t[1] != "IID")) {
            die(6, "Error: No IID column on line %zu of --%s file.\n", line_idx, flag);
          }
          mode = 1;
        } else {
          mode = 2;
        }
        if ((t.size() > static_cast<size_t>(3 - mode)) && (t[3 - mode] == "SID")) {
          die(63, "Error: SID columns in --%s files are not supported by plink2-hip.\n", flag);
        }
      }
      continue;  // (other '#' lines before the data are comments)
    }
    first = false;
    if (mode == 2) {
      keys->push_back("0\t" + t[0]);
    } else if ((mode == 1) || (t.size() >= 2)) {
      if (t.size() < 2) {
        die(6, "Error: Line %zu of --%s file has fewer tokens than expected.\n", line_idx, flag);
      }
      keys->push_back(t[0] + "\t" + t[1]);
    } else {
      keys->push_back("0\t" + t[0]);
    }
  }
}

// Genotype counts of rows for --maf / --max-maf / --geno: per row the hom-REF / het / hom-ALT calls among the founders and the
// missing calls among all kept samples.  m_f / m_s: one bit pair (01) per founder / kept sample, 32 samples per word.
struct RowCounts {
  uint32_t ref2, het, alt2;  // founders
  uint32_t missing;          // kept samples
};
void count_rows(const uint8_t* rows, uint64_t stride, uint32_t n_rows, bool bed, uint32_t raw_sample_ct, const std::vector<uint64_t>& m_f,
                const std::vector<uint64_t>& m_s, RowCounts* out) {
  const uint64_t kLo = 0x5555555555555555ull;
  const uint64_t row_bytes = (static_cast<uint64_t>(raw_sample_ct) + 3) / 4;
  const size_t words = m_f.size();
  for (uint32_t r = 0; r < n_rows; ++r) {
    const uint8_t* row = rows + static_cast<uint64_t>(r) * stride;
    uint32_t c00 = 0, c01 = 0, c10 = 0, miss = 0;
    for (size_t w = 0; w < words; ++w) {
      uint64_t x = 0;
      const uint64_t left = row_bytes - 8 * w;
      memcpy(&x, row + 8 * w, (left < 8) ? left : 8);
      const uint64_t lo = x & kLo, hi = (x >> 1) & kLo;
      const uint64_t b00 = ~(lo | hi) & kLo, b01 = lo & ~hi, b10 = hi & ~lo, b11 = lo & hi;
      c00 += static_cast<uint32_t>(__builtin_popcountll(b00 & m_f[w]));
      c01 += static_cast<uint32_t>(__builtin_popcountll(b01 & m_f[w]));
      c10 += static_cast<uint32_t>(__builtin_popcountll(b10 & m_f[w]));
      miss += static_cast<uint32_t>(__builtin_popcountll((bed ? b01 : b11) & m_s[w]));
    }
    if (bed) {  // 00 hom-ALT, 01 missing, 10 het, 11 hom-REF (pgenlib_read.cc:2157)
      uint32_t c11 = 0;
      for (size_t w = 0; w < words; ++w) {
        uint64_t x = 0;
        const uint64_t left = row_bytes - 8 * w;
        memcpy(&x, row + 8 * w, (left < 8) ? left : 8);
        c11 += static_cast<uint32_t>(__builtin_popcountll(x & (x >> 1) & kLo & m_f[w]));
      }
      out[r] = {c11, c10, c00, miss};
    } else {    // 00 hom-REF, 01 het, 10 hom-ALT, 11 missing
      out[r] = {c00, c01, c10, miss};
    }
  }
}


void load_inputs(Session& S, int argc, char** argv) {
  S.t_begin = now_s();
  S.A = parse_args(argc, argv);
  const Args& A = S.A;
  const double t_begin = S.t_begin;
  g_log = fopen((A.out + ".log").c_str(), "w");
  logprintf("plink2-hip: MI355X-native --indep-pairwise (drop-in for that path of PLINK v2.0)\n");
  logprintf("Options in effect:\n ");
  for (int i = 1; i < argc; ++i) {
    logprintf(" %s", argv[i]);
  }
  logprintf("\n\n");

  // the variant table parses on its own thread and the HIP runtime initialises on another while the
  // sample file is read
  Variants& V = S.V;
  std::thread t_variants([&]() { load_variants(A, &V); });
  // (the HIP runtime start-up AND the context of device 0 -- its queues, the first pinned allocation -- beside the table parsing)
  S.t_hip = std::thread([&S]() { const double t0 = now_s(); if (ldp_device_count() > 0) { (void)ldp_prewarm(0); } S.t_hip_init = now_s() - t0; });
  std::vector<uint8_t>& is_founder = S.is_founder;
  std::vector<std::string> sample_keys;
  const bool sample_filter = (!A.keep_files.empty()) || (!A.remove_files.empty());
  std::vector<std::pair<std::string, std::string>> parent_keys;
  load_samples(A, &S.is_founder, &S.sex, (sample_filter || A.make_founders) ? &sample_keys : nullptr, A.make_founders ? &parent_keys : nullptr);
  // --make-founders (MakeFounders, plink2_filter.cc:4372-4443): a non-founder with a parent (both, with 'require-2-missing') that
  // is not among the samples in play becomes a founder; 'first' applies it before --keep / --remove, else after them
  auto make_founders = [&](const std::vector<uint8_t>* included) {
    std::unordered_set<std::string> present;
    bool any_nonfounder = false;
    for (size_t sx = 0; sx < sample_keys.size(); ++sx) {
      if ((!included) || (*included)[sx]) {
        present.insert(sample_keys[sx]);
        any_nonfounder = any_nonfounder || !S.is_founder[sx];
      }
    }
    if (!any_nonfounder) {
      logprintf("Note: Skipping --make-founders since there are no nonfounders.\n");
      return;
    }
    uint32_t affected = 0;
    for (size_t sx = 0; sx < sample_keys.size(); ++sx) {
      if (S.is_founder[sx] || (included && !(*included)[sx])) {
        continue;
      }
      const uint32_t missing = (present.count(parent_keys[sx].first) ? 0u : 1u) + (present.count(parent_keys[sx].second) ? 0u : 1u);
      if (missing > (A.make_founders_require2 ? 1u : 0u)) {
        S.is_founder[sx] = 1;
        ++affected;
      }
    }
    logprintf("--make-founders: %u sample%s affected.\n", affected, (affected == 1) ? "" : "s");
  };
  if (A.make_founders && A.make_founders_first) {
    make_founders(nullptr);
  }
  if (sample_filter) {  // KeepOrRemove, plink2_filter.cc:1227-1261 (--keep first, then --remove, plink2.cc)
    std::vector<uint8_t> in(S.is_founder.size(), 1);
    for (int pass = 0; pass < 2; ++pass) {
      const std::vector<std::string>& files = pass ? A.remove_files : A.keep_files;
      if (files.empty()) {
        continue;
      }
      const char* flag = pass ? "remove" : "keep";
      std::vector<std::string> keys;
      for (const std::string& fn : files) {
        load_sample_id_list(fn, flag, &keys);
      }
      std::unordered_set<std::string> listed;
      size_t dups = 0;
      for (std::string& k : keys) {
        dups += listed.insert(std::move(k)).second ? 0 : 1;
      }
      uint32_t remaining = 0;
      for (size_t sx = 0; sx < in.size(); ++sx) {
        const bool hit = listed.count(sample_keys[sx]) != 0;
        in[sx] = static_cast<uint8_t>(in[sx] && (pass ? !hit : hit));
        remaining += in[sx];
      }
      logprintf("--%s: %u sample%s remaining.\n", flag, remaining, (remaining == 1) ? "" : "s");
      if (dups) {
        logprintf("Warning: At least %zu duplicate ID%s in --%s file(s).\n", dups, (dups == 1) ? "" : "s", flag);
      }
    }
    for (size_t sx = 0; sx < in.size(); ++sx) {
      S.is_founder[sx] = static_cast<uint8_t>(S.is_founder[sx] && in[sx]);
    }
    S.sample_kept = in;
    if (std::find(in.begin(), in.end(), 1) == in.end()) {  // plink2.cc:1836-1838
      die(13, "Error: No samples remaining after main filters.\n");
    }
  }
  if (A.make_founders && !A.make_founders_first) {
    make_founders(S.sample_kept.empty() ? nullptr : &S.sample_kept);
  }
  t_variants.join();
  S.t_parse = now_s() - t_begin;
  S.raw_sample_ct = static_cast<uint32_t>(is_founder.size());
  const uint32_t raw_sample_ct = S.raw_sample_ct;
  uint32_t& founder_ct = S.founder_ct;
  for (uint8_t f : is_founder) {
    founder_ct += f;
  }
  logprintf("%u sample%s loaded from %s (%u founder%s).\n", raw_sample_ct, raw_sample_ct == 1 ? "" : "s",
            (A.psam.empty() ? A.fam : A.psam).c_str(), founder_ct, founder_ct == 1 ? "" : "s");
  S.raw_variant_ct = static_cast<uint32_t>(V.id.size());
  const uint32_t raw_variant_ct = S.raw_variant_ct;
  logprintf("%u variant%s loaded from %s.\n", raw_variant_ct, raw_variant_ct == 1 ? "" : "s", (A.pvar.empty() ? A.bim : A.pvar).c_str());

  if (A.have_prune && founder_ct < 50 && !A.bad_ld) {  // plink2.cc:2063-2071
    if (raw_sample_ct < 50) {
      die(13, "Error: This run estimates linkage disequilibrium between variants, but there\nare less than 50 samples to estimate from.  You should perform this operation\non a larger dataset.\n(Strictly speaking, you can also override this error with --bad-ld, but this is\nalmost always a bad idea.)\n");
    }
    die(13, "Error: This run estimates linkage disequilibrium between variants, but there\nare less than 50 founders to estimate from.  --make-founders may help.\n(Strictly speaking, you can also override this error with --bad-ld, but this is\nalmost always a bad idea.)\n");
  }
  if (founder_ct < 2) {
    die(7, "Error: %s requires at least two founders. (--make-founders may come in handy here.)\n", A.have_prune ? (A.pairphase ? "--indep-pairphase" : "--indep-pairwise") : "--r2-unphased");
  }

  // ---- genotype file (.bed / fixed-width .pgen / standard variable-width .pgen)
  S.is_bed = !A.bed.empty();
  S.gpath = S.is_bed ? A.bed : A.pgen;
  const std::string& gpath = S.gpath;
  ldp_pgen*& pg = S.pg;
  if (ldp_pgen_open_indexed(gpath.c_str(), A.pgi.empty() ? nullptr : A.pgi.c_str(), raw_sample_ct, raw_variant_ct, &pg)) {
    die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
  }
  ldp_pgen_info(pg, nullptr, nullptr, &S.storage_mode, &S.encoding, &S.has_multiallelic);
  S.has_dosage = ldp_pgen_has_dosage(pg) != 0;
  if (S.has_dosage) {
    // The reference takes allele frequencies from the dosages when a file has them (plink2_data.cc:2421-2443).  For
    // --indep-pairwise that is the major allele's frequency in the tie-break -- r^2 itself is computed from the hardcalls
    // (plink2_ld.cc:699-723) --, which run_prune() reproduces (ldp_pgen_dosage_sums); --maf / --max-maf compare the same
    // frequencies.  Everything else that would read dosages (the r^2 of --r2-unphased / --clump, phased dosages) is refused
    // rather than computed from hardcalls.
    const char* what = A.have_r2 ? "--r2-unphased / --clump" : (A.pairphase ? "--indep-pairphase" : nullptr);
    if (what) {
      ldp_pgen_close(pg);
      die(63, "Error: %s holds dosage data, which plink2-hip reads for --indep-pairwise only (%s would be\ncomputed from hardcalls, unlike plink2).  Use plink2 --make-pgen erase-dosage first.\n", gpath.c_str(), what);
    }
  }
  S.rec_bytes = (static_cast<uint64_t>(raw_sample_ct) + 3) / 4;
  S.direct_rows = static_cast<const uint8_t*>(ldp_pgen_direct_rows(pg, &S.rec_bytes));  // NULL for variable-width

  // ---- variant table: strip chromosome 0, chromosome order index, sortedness, unique IDs
  std::vector<uint32_t>& inc = S.inc;
  std::vector<uint32_t>& chr_idx = S.chr_idx;
  std::vector<uint32_t>& bps = S.bps;
  std::vector<uint8_t>& vcls = S.vcls;
  uint32_t skipped = 0;
  // variant filters: --chr / --not-chr / --autosome by chromosome, then --extract, then --exclude by ID
  // (TokenExtractExclude, plink2_filter.cc:367: every variant carrying a listed ID, unknown IDs ignored)
  std::unordered_set<std::string> extract_ids, exclude_ids;
  for (const std::string& fn : A.extract_files) {
    for (std::string& t : tokens_of_file(fn)) {
      extract_ids.insert(std::move(t));
    }
  }
  for (const std::string& fn : A.exclude_files) {
    for (std::string& t : tokens_of_file(fn)) {
      exclude_ids.insert(std::move(t));
    }
  }
  const bool chr_filter = (!A.chr_keep.empty()) || (!A.chr_drop.empty()) || A.autosome;
  uint32_t after_extract = 0, after_exclude = 0;
  // --geno / --maf / --max-maf need genotype counts before the variant list is final: one multi-threaded pass over the rows of
  // the variants the table filters leave (host popcounts; the rows are read again when they go to the device)
  std::vector<uint8_t> drop_by_counts;
  const bool mac_filter = (A.min_allele_ddosage != 0) || (A.max_allele_ddosage != ~0ull);
  const bool freq_filter = (A.min_maf != 0.0) || (A.max_maf != 1.0) || mac_filter;
  if (mac_filter && (!A.ac_founders)) {
    // (plink2.cc:2102-2105; plink2-hip counts alleles over the founders: the --nonfounders alternative is not offered)
    uint32_t kept = 0;
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      kept += (S.sample_kept.empty() || S.sample_kept[sx]) ? 1u : 0u;
    }
    if (kept != founder_ct) {
      die(7, "Error: --mac/--max-mac/\"--freq counts\" specified, but with neither\n--ac-founders nor --nonfounders; and nonfounders are present.\n");
    }
  }
  if (freq_filter || (A.geno != 1.0)) {
    std::unordered_map<std::string, uint8_t> chr_state;  // 1 = filtered out by chromosome
    std::vector<uint32_t> todo;
    for (uint32_t v = 0; v < raw_variant_ct; ++v) {
      auto it = chr_state.find(V.chrom[v]);
      if (it == chr_state.end()) {
        const std::string& cur = V.chrom[v];
        const int code = chrom_code(cur);
        const bool out = chr_filter && (((!A.chr_keep.empty()) && !chrom_listed(A.chr_keep, cur)) || ((!A.chr_drop.empty()) && chrom_listed(A.chr_drop, cur)) ||
                                         (A.autosome && !((code >= 1) && (code <= 22))));
        bool zero = false;
        const int cls = chrom_class(cur, A.allow_extra_chr, &zero);
        if ((!out) && (cls >= 3)) {
          die(63, "Error: --maf / --max-maf / --mac / --max-mac / --geno on chrX, chrY or MT ('%s') are not supported by plink2-hip: filter them out (--autosome, --chr) or pre-filter with plink2.\n", cur.c_str());
        }
        it = chr_state.emplace(cur, static_cast<uint8_t>(out)).first;
      }
      if (it->second || ((!A.extract_files.empty()) && !extract_ids.count(V.id[v])) || ((!A.exclude_files.empty()) && exclude_ids.count(V.id[v]))) {
        continue;
      }
      if ((allele_ct_for_filter(V, v) > A.max_alleles) || (A.snps_only && V.not_snp[v])) {
        continue;
      }
      if (V.alt_ct[v] > 1) {
        die(63, "Error: --maf / --max-maf / --mac / --max-mac / --geno with multiallelic variants ('%s') are not supported by plink2-hip.\n", V.id[v].c_str());
      }
      todo.push_back(v);
    }
    const size_t words = (static_cast<size_t>(raw_sample_ct) + 31) / 32;
    std::vector<uint64_t> m_f(words, 0), m_s(words, 0);
    uint32_t kept_samples = 0;
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      if (S.sample_kept.empty() || S.sample_kept[sx]) {
        m_s[sx >> 5] |= 1ull << (2 * (sx & 31));
        ++kept_samples;
      }
      if (is_founder[sx]) {
        m_f[sx >> 5] |= 1ull << (2 * (sx & 31));
      }
    }
    std::vector<RowCounts> counts(todo.size());
    const bool bed = (S.storage_mode == 0x01);
    const uint32_t nthreads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    if (S.direct_rows) {
      std::atomic<size_t> next(0);
      const size_t kTask = 2048;
      std::vector<std::thread> pool;
      for (uint32_t t = 0; t < nthreads; ++t) {
        pool.emplace_back([&]() {
          for (size_t q0 = next.fetch_add(kTask); q0 < todo.size(); q0 = next.fetch_add(kTask)) {
            const size_t q1 = std::min(todo.size(), q0 + kTask);
            for (size_t q = q0; q < q1; ++q) {
              count_rows(S.direct_rows + static_cast<uint64_t>(todo[q]) * S.rec_bytes, S.rec_bytes, 1, bed, raw_sample_ct, m_f, m_s, &counts[q]);
            }
          }
        });
      }
      for (std::thread& th : pool) {
        th.join();
      }
    } else {
      // variable-width records: decode runs of file-consecutive variants (all host threads), then count them
      const uint32_t max_run = std::max<uint32_t>(1, static_cast<uint32_t>((256ull << 20) / std::max<uint64_t>(S.rec_bytes, 1)));
      std::vector<uint8_t> decoded;
      for (size_t q0 = 0; q0 < todo.size();) {
        uint32_t run = 1;
        while ((q0 + run < todo.size()) && (todo[q0 + run] == todo[q0] + run) && (run < max_run)) {
          ++run;
        }
        decoded.resize(static_cast<size_t>(run) * S.rec_bytes);
        if (ldp_pgen_read(pg, todo[q0], run, decoded.data(), S.rec_bytes, 0)) {
          die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
        }
        std::atomic<uint32_t> next(0);
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < nthreads; ++t) {
          pool.emplace_back([&]() {
            for (uint32_t r0 = next.fetch_add(256); r0 < run; r0 = next.fetch_add(256)) {
              const uint32_t n = std::min(256u, run - r0);
              count_rows(decoded.data() + static_cast<uint64_t>(r0) * S.rec_bytes, S.rec_bytes, n, bed, raw_sample_ct, m_f, m_s, &counts[q0 + r0]);
            }
          });
        }
        for (std::thread& th : pool) {
          th.join();
        }
        q0 += run;
      }
    }
    if (S.has_dosage && freq_filter) {
      std::vector<uint32_t> with_track;
      for (uint32_t v : todo) {
        if (ldp_pgen_variant_has_dosage(pg, v)) {
          with_track.push_back(v);
        }
      }
      S.need_dosage_sums(with_track);
    }
    drop_by_counts.assign(raw_variant_ct, 0);
    uint32_t geno_removed = 0, freq_removed = 0;
    const uint32_t missing_max = static_cast<uint32_t>(static_cast<int32_t>(A.geno * (1 + kSmallEpsilon) * static_cast<double>(kept_samples)));
    const double min_maf = A.min_maf * (1.0 - kSmallEpsilon), max_maf = A.max_maf * (1.0 + kSmallEpsilon);
    for (size_t q = 0; q < todo.size(); ++q) {
      const RowCounts& c = counts[q];
      if ((A.geno != 1.0) && (c.missing > missing_max)) {
        drop_by_counts[todo[q]] = 1;
        ++geno_removed;
        continue;
      }
      if (freq_filter) {
        // allele counts in 16384ths of a copy: the hardcalls', or -- a record with dosages -- the founders' dosage sums
        uint64_t ref_ct = (2ull * c.ref2 + c.het) * 16384ull, alt_ct = (2ull * c.alt2 + c.het) * 16384ull;
        const auto dd = S.dosage_sums.find(todo[q]);
        if (dd != S.dosage_sums.end()) {
          ref_ct = dd->second.first;
          alt_ct = dd->second.second;
        }
        const uint64_t tot = ref_ct + alt_ct;
        if (mac_filter) {
          // GetTypedDdosage, nonmajor mode, two alleles (plink2_filter.cc:3765-3767) on allele_ddosages = 2 x these sums
          // (plink2_data.cc:2441-2442)
          const uint64_t typed_dd = 2 * std::min(ref_ct, alt_ct);
          if ((typed_dd < A.min_allele_ddosage) || (typed_dd > A.max_allele_ddosage)) {
            drop_by_counts[todo[q]] = 1;
            ++freq_removed;
            continue;
          }
        }
        const double ref_freq = tot ? (static_cast<double>(ref_ct) * (1.0 / static_cast<double>(tot))) : 0.5;  // plink2_filter.cc:2137-2147
        const double nonref_freq = 1.0 - ref_freq;
        const double typed = (nonref_freq < ref_freq) ? nonref_freq : ref_freq;  // GetTypedFreq, nonmajor mode, two alleles (:3715-3723)
        if ((((A.min_maf != 0.0) || (A.max_maf != 1.0))) && (((A.min_maf != 0.0) && (typed < min_maf)) || ((A.max_maf < 1.0) && (typed > max_maf)))) {
          drop_by_counts[todo[q]] = 1;
          ++freq_removed;
        }
      }
    }
    if (A.geno != 1.0) {
      logprintf("--geno: %u variant%s removed due to missing genotype data.\n", geno_removed, (geno_removed == 1) ? "" : "s");
    }
    if (freq_filter) {
      logprintf("%u variant%s removed due to allele frequency threshold(s)\n(--maf/--max-maf/--mac/--max-mac).\n", freq_removed, (freq_removed == 1) ? "" : "s");
    }
  }
  {
    std::unordered_set<std::string> seen_chr;
    std::string cur;
    uint32_t fo = 0;
    bool first = true;
    bool zero = false;
    bool chr_out = false;
    int cls = 0;
    inc.reserve(raw_variant_ct);
    chr_idx.reserve(raw_variant_ct);
    bps.reserve(raw_variant_ct);
    for (uint32_t v = 0; v < raw_variant_ct; ++v) {
      if (first || V.chrom[v] != cur) {
        if (!seen_chr.insert(V.chrom[v]).second) {
          die(6, "Error: %s has a split chromosome. Use --make-pgen + --sort-vars to remedy this.\n", (A.pvar.empty() ? A.bim : A.pvar).c_str());
        }
        cur = V.chrom[v];
        if (!first) {
          ++fo;
        }
        first = false;
        cls = chrom_class(cur, A.allow_extra_chr, &zero);
        chr_out = false;
        if (chr_filter) {
          const int code = chrom_code(cur);
          chr_out = ((!A.chr_keep.empty()) && !chrom_listed(A.chr_keep, cur)) || ((!A.chr_drop.empty()) && chrom_listed(A.chr_drop, cur)) ||
                    (A.autosome && !((code >= 1) && (code <= 22)));
        }
      }
      if (chr_out || ((allele_ct_for_filter(V, v) > A.max_alleles) || (A.snps_only && V.not_snp[v]))) {
        continue;
      }
      if ((!A.extract_files.empty()) && !extract_ids.count(V.id[v])) {
        continue;
      }
      ++after_extract;
      if ((!A.exclude_files.empty()) && exclude_ids.count(V.id[v])) {
        continue;
      }
      ++after_exclude;
      if ((!drop_by_counts.empty()) && drop_by_counts[v]) {
        continue;
      }
      if (zero && (A.have_prune || (A.r2_table && !A.r2_inter))) {  // (the all-pairs modes keep chromosome 0)
        ++skipped;
        continue;
      }
      if (cls == 2) {
        die(6, "Error: Invalid chromosome code '%s'. (Use --allow-extra-chr to force it to be accepted.)\n", cur.c_str());
      }
      vcls.push_back(static_cast<uint8_t>(cls));
      inc.push_back(v);
      chr_idx.push_back(fo);
      bps.push_back(V.bp[v]);
    }
  }
  if (!A.extract_files.empty()) {
    logprintf("--extract: %u variant%s remaining.\n", after_extract, (after_extract == 1) ? "" : "s");
  }
  if (!A.exclude_files.empty()) {
    logprintf("--exclude: %u variant%s remaining.\n", after_exclude, (after_exclude == 1) ? "" : "s");
  }
  // filters applied while the variant table loads (--autosome / --chr / --not-chr / --max-alleles) that leave nothing:
  // plink2.cc:1025-1050, kPglRetInconsistentInput, flag names in kLoadFilterLogFlagnames order
  if ((chr_filter || (A.max_alleles != 0xffffffffu) || A.snps_only) && raw_variant_ct) {
    bool any_loaded = false;
    std::unordered_map<std::string, uint8_t> chr_state;
    for (uint32_t v = 0; (v < raw_variant_ct) && !any_loaded; ++v) {
      auto it = chr_state.find(V.chrom[v]);
      if (it == chr_state.end()) {
        const std::string& cur = V.chrom[v];
        const int code = chrom_code(cur);
        const bool out = chr_filter && (((!A.chr_keep.empty()) && !chrom_listed(A.chr_keep, cur)) || ((!A.chr_drop.empty()) && chrom_listed(A.chr_drop, cur)) ||
                                         (A.autosome && !((code >= 1) && (code <= 22))));
        it = chr_state.emplace(cur, static_cast<uint8_t>(out)).first;
      }
      any_loaded = (!it->second) && (allele_ct_for_filter(V, v) <= A.max_alleles) && !(A.snps_only && V.not_snp[v]);
    }
    if (!any_loaded) {
      std::string flags;
      for (const char* nm : {A.autosome ? "autosome" : "", A.chr_keep.empty() ? "" : "chr", A.chr_drop.empty() ? "" : "not-chr",
                             (A.max_alleles != 0xffffffffu) ? "max-alleles" : "", A.snps_only ? "snps-only" : ""}) {
        if (*nm) {
          flags += (flags.empty() ? "--" : " + --");
          flags += nm;
        }
      }
      die(7, "Error: All %u variant%s in %s excluded by %s.\n", raw_variant_ct, (raw_variant_ct == 1) ? "" : "s", (A.pvar.empty() ? A.bim : A.pvar).c_str(), flags.c_str());
    }
  }
  const bool any_main_filter = chr_filter || (!A.extract_files.empty()) || (!A.exclude_files.empty()) || (!drop_by_counts.empty()) || (A.max_alleles != 0xffffffffu) || A.snps_only;
  if (any_main_filter && inc.empty() && (!skipped)) {  // plink2.cc:2484-2487 (kPglRetDegenerateData)
    die(13, "Error: No variants remaining after main filters.\n");
  }
  if (skipped) {
    logprintf("--%s: Ignoring %u chromosome 0 variant%s.\n", A.have_prune ? (A.pairphase ? "indep-pairphase" : "indep-pairwise") : (A.have_clump ? "clump" : "r2-unphased"), skipped, skipped == 1 ? "" : "s");
  }
  S.variant_ct = static_cast<uint32_t>(inc.size());
  const uint32_t variant_ct = S.variant_ct;
  if (A.window_is_bp || A.r2_table) {
    for (uint32_t k = 1; k < variant_ct; ++k) {
      if (chr_idx[k] == chr_idx[k - 1] && bps[k] < bps[k - 1]) {
        if (A.have_prune) {  // plink2.cc:2926-2929
          die(6, "Error: When the window size is in kb units, LD-based pruning requires a sorted\n.pvar/.bim.  Retry this command after using --make-pgen/--make-bed +\n--sort-vars to sort your data.\n");
        }
        if (A.have_clump) {  // plink2.cc:2998-3001
          die(7, "Error: --clump requires a sorted .pvar/.bim.  Retry this command after using\n--make-pgen/--make-bed + --sort-vars to sort your data.\n");
        }
        die(6, "Error: --r[2]-[un]phased runs require a sorted .pvar/.bim.  Retry this command\nafter using --make-pgen/--make-bed + --sort-vars to sort your data.\n");  // plink2.cc:2944-2947
      }
    }
  }

  // chrX and chrY variants run on engines of their own (different sample sets); MT stays with the autosomes --
  // except under --indep-pairphase, where the autosomes carry two haplotypes per founder and MT one
  // (IndepPairphaseUpdateSubcontig, plink2_ld.cc:1491-1511)
  std::vector<uint32_t>&mk = S.mk, &xk = S.xk, &yk = S.yk, &tk = S.tk;
  for (uint32_t k = 0; k < variant_ct; ++k) {
    (vcls[k] == 3 ? xk : (vcls[k] == 4 ? yk : ((vcls[k] == 5 && A.pairphase) ? tk : mk))).push_back(k);
  }
  S.m_ct = static_cast<uint32_t>(mk.size());
  const uint32_t m_ct = S.m_ct;
  std::vector<uint32_t>&m_chr = S.m_chr, &m_bps = S.m_bps;
  m_chr.resize(m_ct);
  m_bps.resize(m_ct);
  for (uint32_t q = 0; q < m_ct; ++q) {
    m_chr[q] = chr_idx[mk[q]];
    m_bps[q] = bps[mk[q]];
  }
}


}  // namespace p2h
