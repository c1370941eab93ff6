// p2h_tables.cpp -- plink2-hip: sample / variant tables, chromosome classes, host-built rows of the sex chromosomes and multiallelic variants (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

// founder <=> PAT and MAT are both exactly "0" (plink2_psam.cc:804-806); absent columns => founder
// sex: 1 = male, 2 = female, anything else = unknown (plink2_psam.cc:808-813)
void load_samples(const Args& A, std::vector<uint8_t>* is_founder, std::vector<uint8_t>* sex, std::vector<std::string>* fid_iid,
                  std::vector<std::pair<std::string, std::string>>* parents) {
  const bool psam = !A.psam.empty();
  const std::string& path = psam ? A.psam : A.fam;
  std::ifstream in(path);
  if (!in) {
    die(3, "Error: Failed to open %s.\n", path.c_str());
  }
  std::string line;
  int pat_col = -1, mat_col = -1, sex_col = -1, iid_col = 0;
  bool header_seen = false, has_fid = false;
  while (std::getline(in, line)) {
    if (line.empty()) {
      continue;
    }
    if (psam && line[0] == '#') {
      if (line.rfind("#FID", 0) == 0 || line.rfind("#IID", 0) == 0) {
        std::vector<std::string> cols = split_ws(line);
        for (size_t c = 0; c < cols.size(); ++c) {
          if (cols[c] == "PAT") pat_col = static_cast<int>(c);
          if (cols[c] == "MAT") mat_col = static_cast<int>(c);
          if (cols[c] == "SEX") sex_col = static_cast<int>(c);
        }
        has_fid = (cols[0] == "#FID");
        iid_col = has_fid ? 1 : 0;
        header_seen = true;
      }
      continue;
    }
    std::vector<std::string> t = split_ws(line);
    if (t.empty()) {
      continue;
    }
    if (!psam || !header_seen) {
      // .fam layout: FID IID PAT MAT SEX PHENO
      if (t.size() < 5) {
        die(6, "Error: Fewer tokens than expected in %s.\n", path.c_str());
      }
      is_founder->push_back((t[2] == "0") && (t[3] == "0"));
      if (fid_iid) {
        fid_iid->push_back(t[0] + "\t" + t[1]);
      }
      if (parents) {
        parents->emplace_back(t[0] + "\t" + t[2], t[0] + "\t" + t[3]);
      }
      const std::string& v = t[4];  // CharToSex on a one-character token (plink2_psam.cc:505-509), for .fam as for .psam
      sex->push_back((v == "1" || v == "M" || v == "m") ? 1 : ((v == "2" || v == "F" || v == "f") ? 2 : 0));
    } else {
      bool founder = true;
      if (pat_col >= 0 && mat_col >= 0) {
        if (static_cast<size_t>(std::max(pat_col, mat_col)) >= t.size()) {
          die(6, "Error: Fewer tokens than expected in %s.\n", path.c_str());
        }
        founder = (t[pat_col] == "0") && (t[mat_col] == "0");
      }
      is_founder->push_back(founder);
      if (parents) {
        const std::string fid = has_fid ? t[0] : std::string("0");
        parents->emplace_back(fid + "\t" + ((pat_col >= 0) ? t[pat_col] : std::string("0")), fid + "\t" + ((mat_col >= 0) ? t[mat_col] : std::string("0")));
      }
      if (fid_iid) {  // (no FID column: FID "0", as the reference keys its samples)
        if (static_cast<size_t>(iid_col) >= t.size()) {
          die(6, "Error: Fewer tokens than expected in %s.\n", path.c_str());
        }
        fid_iid->push_back((has_fid ? t[0] : std::string("0")) + "\t" + t[iid_col]);
      }
      uint8_t sx = 0;
      if (sex_col >= 0 && static_cast<size_t>(sex_col) < t.size()) {
        const std::string& v = t[sex_col];
        sx = (v == "1" || v == "M" || v == "m") ? 1 : ((v == "2" || v == "F" || v == "f") ? 2 : 0);
      }
      sex->push_back(sx);
    }
  }
}


// allele count as --max-alleles sees it (LoadPvar, plink2_pvar.cc:1937-1953): a lone ALT that is a missing code counts as one allele

// whole file -> memory; the variant/sample tables are a few tens of MB even at 10M variants
std::string slurp(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    die(3, "Error: Failed to open %s.\n", path.c_str());
  }
  std::string buf;
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(sz > 0 ? static_cast<size_t>(sz) : 0);
  if (sz > 0 && fread(&buf[0], 1, buf.size(), f) != buf.size()) {
    die(4, "Error: Failed to read %s.\n", path.c_str());
  }
  fclose(f);
  return buf;
}

// zstd-compressed text (.pvar.zst / .bim.zst): the image ships libzstd.so.1 without headers, so the few streaming
// entry points are bound by hand (stable C ABI since zstd 1.0: zstd.h "Streaming decompression").
std::string slurp_zst(const std::string& path) {
  struct InBuf {
    const void* src;
    size_t size, pos;
  };
  struct OutBuf {
    void* dst;
    size_t size, pos;
  };
  void* lib = dlopen("libzstd.so.1", RTLD_NOW);
  if (!lib) {
    die(63, "Error: %s is zstd-compressed and libzstd.so.1 could not be loaded (%s).\n", path.c_str(), dlerror());
  }
  auto create = reinterpret_cast<void* (*)()>(dlsym(lib, "ZSTD_createDStream"));
  auto destroy = reinterpret_cast<size_t (*)(void*)>(dlsym(lib, "ZSTD_freeDStream"));
  auto init = reinterpret_cast<size_t (*)(void*)>(dlsym(lib, "ZSTD_initDStream"));
  auto step = reinterpret_cast<size_t (*)(void*, OutBuf*, InBuf*)>(dlsym(lib, "ZSTD_decompressStream"));
  auto is_error = reinterpret_cast<unsigned (*)(size_t)>(dlsym(lib, "ZSTD_isError"));
  if (!create || !destroy || !init || !step || !is_error) {
    die(63, "Error: libzstd.so.1 lacks the streaming decompression API.\n");
  }
  const std::string in = slurp(path);
  void* ds = create();
  if (!ds || is_error(init(ds))) {
    die(63, "Error: zstd decompressor setup failed.\n");
  }
  std::string out;
  std::vector<char> chunk(4u << 20);
  InBuf ib = {in.data(), in.size(), 0};
  size_t last = 0;  // 0 = at a frame boundary with everything flushed
  while (ib.pos < ib.size) {
    OutBuf ob = {chunk.data(), chunk.size(), 0};
    last = step(ds, &ob, &ib);
    if (is_error(last)) {
      die(6, "Error: %s is not a valid zstd stream.\n", path.c_str());
    }
    out.append(chunk.data(), ob.pos);
  }
  while (last != 0) {  // input exhausted inside a frame: the decoder may still hold output
    OutBuf ob = {chunk.data(), chunk.size(), 0};
    InBuf none = {in.data(), ib.size, ib.size};
    last = step(ds, &ob, &none);
    if (is_error(last)) {
      die(6, "Error: %s is not a valid zstd stream.\n", path.c_str());
    }
    out.append(chunk.data(), ob.pos);
    if (!ob.pos && last) {
      die(6, "Error: %s ends inside a zstd frame.\n", path.c_str());
    }
  }
  destroy(ds);
  return out;
}

struct Tok {
  const char* p;
  size_t n;
  bool eq(const char* s) const { return strlen(s) == n && !memcmp(p, s, n); }
};

// split [p, e) on spaces/tabs into at most `cap` tokens; returns the token count (capped)
inline int tokenize(const char* p, const char* e, Tok* out, int cap) {
  int n = 0;
  while (p < e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) {
      ++p;
    }
    if (p >= e) {
      break;
    }
    const char* q = p;
    while (q < e && *q != ' ' && *q != '\t' && *q != '\r') {
      ++q;
    }
    if (n < cap) {
      out[n].p = p;
      out[n].n = static_cast<size_t>(q - p);
    }
    ++n;
    p = q;
  }
  return n;
}

void load_variants(const Args& A, Variants* V) {
  const bool pvar = !A.pvar.empty();
  const std::string& path = pvar ? A.pvar : A.bim;
  const bool zst = (path.size() > 4) && (path.compare(path.size() - 4, 4, ".zst") == 0);
  const std::string buf = zst ? slurp_zst(path) : slurp(path);
  bool header = false;
  int c_chrom = 0, c_pos = 3, c_id = 1, c_alt = -1, c_ref = -1, c_cm = -1, c_info = -1;
  const bool keep_pr = (A.have_r2 && (A.r2_cols & kVcorColRef)) || (A.have_clump && (A.clump_cols & kClumpColRef));
  const bool keep_cm = A.have_r2 && (A.ld_cm_radius != -1.0);
  double last_cm = -1.7976931348623157e308;
  std::string last_cm_chrom;
  const bool keep_alleles = (A.have_r2 && (A.r2_cols & (kVcorColRef | kVcorColAlt1 | kVcorColAlt | kVcorColMaj | kVcorColNonmaj))) ||
                            A.have_clump;  // (--clump: an A1 column names alleles of multiallelic variants, and of all variants under --clump-force-a1)
  constexpr int kCap = 64;
  Tok t[kCap];
  const char* p = buf.data();
  const char* end = p + buf.size();
  size_t guess = std::count(buf.begin(), buf.end(), '\n') + 1;
  V->chrom.reserve(guess);
  V->id.reserve(guess);
  V->bp.reserve(guess);
  while (p < end) {
    const char* eol = static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(end - p)));
    if (!eol) {
      eol = end;
    }
    const char* line = p;
    p = (eol < end) ? eol + 1 : end;
    if (line == eol) {
      continue;
    }
    if (*line == '#') {
      if ((eol - line) >= 6 && !memcmp(line, "#CHROM", 6)) {
        const int nc = std::min(tokenize(line, eol, t, kCap), kCap);
        c_chrom = 0;
        c_pos = c_id = -1;
        for (int c = 0; c < nc; ++c) {
          if (t[c].eq("POS")) c_pos = c;
          if (t[c].eq("ID")) c_id = c;
          if (t[c].eq("ALT")) c_alt = c;
          if (t[c].eq("REF")) c_ref = c;
          if (t[c].eq("CM")) c_cm = c;
          if (t[c].eq("INFO")) c_info = c;
        }
        if (c_pos < 0 || c_id < 0) {
          die(6, "Error: %s header lacks POS/ID.\n", path.c_str());
        }
        header = true;
      } else if ((eol - line) >= 14 && !memcmp(line, "##INFO=<ID=PR,", 14)) {
        // (only a Flag definition counts, plink2_pvar.cc:1254-1259)
        const std::string hl(line, static_cast<size_t>(eol - line));
        const size_t tp = hl.find("Type=");
        V->info_pr_header = (tp != std::string::npos) && (hl.compare(tp + 5, 4, "Flag") == 0) && ((tp + 9 >= hl.size()) || (hl[tp + 9] == ',') || (hl[tp + 9] == '>'));
      }
      continue;
    }
    const int nt = tokenize(line, eol, t, kCap);
    if (!nt) {
      continue;
    }
    if (!header) {
      // .bim layout: chrom id cM bp A1 A2 (5-column variant without cM also accepted by plink2)
      if (nt == 5) {
        c_pos = 2;
      } else if (nt < 6) {
        die(6, "Error: Fewer tokens than expected in %s.\n", path.c_str());
      }
    }
    if (std::max(std::max(c_chrom, c_pos), c_id) >= std::min(nt, kCap)) {
      die(6, "Error: Fewer tokens than expected in %s.\n", path.c_str());
    }
    uint32_t alts = 1;
    if (c_alt >= 0 && c_alt < std::min(nt, kCap)) {
      alts += static_cast<uint32_t>(std::count(t[c_alt].p, t[c_alt].p + t[c_alt].n, ','));
    }
    if (alts > 254) {
      die(63, "Error: variant '%.*s' has more than 254 ALT alleles, which plink2-hip does not support.\n", static_cast<int>(t[c_id].n), t[c_id].p);
    }
    V->alt_ct.push_back(static_cast<uint8_t>(alts));
    if (A.max_alleles != 0xffffffffu) {
      // ('0' is the reference's default --input-missing-genotype character, plink2.cc:4033)
      const int k_alt1 = header ? c_alt : ((nt == 5) ? 3 : 4);
      const bool miss = (alts == 1) && (k_alt1 >= 0) && (k_alt1 < std::min(nt, kCap)) && (t[k_alt1].n == 1) && ((t[k_alt1].p[0] == '.') || (t[k_alt1].p[0] == '0'));
      V->alt_missing.push_back(miss ? 1 : 0);
    }
    if (A.snps_only) {  // LoadPvar, plink2_pvar.cc:1917-1932
      const int k_ref = header ? c_ref : ((nt == 5) ? 4 : 5), k_alt = header ? c_alt : ((nt == 5) ? 3 : 4);
      bool snp = (k_ref >= 0) && (k_alt >= 0) && (std::max(k_ref, k_alt) < std::min(nt, kCap)) && (t[k_ref].n == 1) && (t[k_alt].n == 2 * (alts - 1) + 1);
      if (snp && A.snps_only_acgt) {
        auto acgtm = [](char ch) { return (ch == 'A') || (ch == 'C') || (ch == 'G') || (ch == 'T') || (ch == 'a') || (ch == 'c') || (ch == 'g') || (ch == 't') || (ch == '.') || (ch == '0'); };  // (acgtm_table incl. the default missing-genotype character '0', plink2_pvar.cc:1631)
        snp = acgtm(t[k_ref].p[0]);
        for (uint32_t a = 0; snp && (a < alts); ++a) {
          snp = acgtm(t[k_alt].p[2 * a]);
        }
      }
      V->not_snp.push_back(snp ? 0 : 1);
    }
    if (keep_alleles) {
      // .bim: ... A1 A2 with A1 -> ALT, A2 -> REF (LoadPvar, plink2_pvar.cc:1434-1450)
      const int k_ref = header ? c_ref : ((nt == 5) ? 4 : 5), k_alt = header ? c_alt : ((nt == 5) ? 3 : 4);
      if ((k_ref < 0) || (k_alt < 0) || (std::max(k_ref, k_alt) >= std::min(nt, kCap))) {
        die(6, "Error: %s has no REF/ALT columns.\n", path.c_str());
      }
      V->ref.emplace_back(t[k_ref].p, t[k_ref].n);
      V->alt.emplace_back(t[k_alt].p, t[k_alt].n);
    }
    if (keep_cm) {
      const int k_cm = header ? c_cm : ((nt == 5) ? -1 : 2);
      double cur_cm = 0.0;
      if ((k_cm >= 0) && (k_cm < std::min(nt, kCap))) {
        if ((t[c_chrom].n != last_cm_chrom.size()) || memcmp(t[c_chrom].p, last_cm_chrom.data(), t[c_chrom].n)) {
          last_cm_chrom.assign(t[c_chrom].p, t[c_chrom].n);
          last_cm = -1.7976931348623157e308;
        }
        if (!((t[k_cm].n == 1) && (t[k_cm].p[0] == '0'))) {  // (a bare "0" is taken as is, without the order check)
          const std::string tok(t[k_cm].p, t[k_cm].n);
          const char* endp;
          if (!scan_double_plink(tok.c_str(), &cur_cm, &endp) || *endp) {
            die(6, "Error: Invalid centimorgan position in %s.\n", path.c_str());
          }
          if (cur_cm < last_cm) {
            V->cm_unsorted = true;
          } else {
            last_cm = cur_cm;
          }
          V->cm_any_nonzero = V->cm_any_nonzero || (cur_cm != 0.0);
        }
      }
      V->cm.push_back(cur_cm);
    }
    if (keep_pr && V->info_pr_header && header && (c_info >= 0) && (c_info < std::min(nt, kCap))) {
      const std::string info(t[c_info].p, t[c_info].n);
      const bool pr = (info == "PR") || (info.compare(0, 3, "PR;") == 0) || ((info.size() >= 3) && (info.compare(info.size() - 3, 3, ";PR") == 0)) ||
                      (info.find(";PR;") != std::string::npos);
      const size_t vi = V->chrom.size();
      if (pr) {
        if (V->info_pr.size() <= (vi >> 3)) {
          V->info_pr.resize((vi >> 3) + 1024, 0);
        }
        V->info_pr[vi >> 3] |= static_cast<uint8_t>(1u << (vi & 7));
      }
    }
    V->chrom.emplace_back(t[c_chrom].p, t[c_chrom].n);
    V->id.emplace_back(t[c_id].p, t[c_id].n);
    uint64_t pos = 0;
    const Tok& tp = t[c_pos];
    if (!tp.n || tp.n > 10) {
      die(6, "Error: Invalid bp coordinate in %s.\n", path.c_str());
    }
    for (size_t k = 0; k < tp.n; ++k) {
      if (tp.p[k] < '0' || tp.p[k] > '9') {
        die(6, "Error: Invalid bp coordinate in %s.\n", path.c_str());
      }
      pos = pos * 10 + static_cast<uint64_t>(tp.p[k] - '0');
    }
    if (pos > 0x7ffffffe) {
      die(6, "Error: Invalid bp coordinate in %s.\n", path.c_str());
    }
    V->bp.push_back(static_cast<uint32_t>(pos));
  }
}

// chromosome class: 0 = diploid autosome / PAR (1..22, XY, extra contigs with --allow-extra-chr; *is_zero for
// chromosome 0), 2 = invalid code, 3 = chrX, 4 = chrY, 5 = MT (haploid)
int chrom_class(const std::string& name_in, bool allow_extra, bool* is_zero) {
  std::string name = name_in;
  if (name.size() > 3 && (name[0] | 32) == 'c' && (name[1] | 32) == 'h' && (name[2] | 32) == 'r') {
    name = name.substr(3);
  }
  *is_zero = false;
  bool numeric = !name.empty();
  for (char c : name) {
    numeric = numeric && (c >= '0' && c <= '9');
  }
  if (numeric) {
    const long v = strtol(name.c_str(), nullptr, 10);
    if (v == 0) {
      *is_zero = true;
      return 0;
    }
    if (v <= 22) {
      return 0;
    }
    if (v == 25) {
      return 0;  // XY (pseudo-autosomal) is diploid
    }
    if (v == 23) return 3;
    if (v == 24) return 4;
    if (v == 26) return 5;
    return 2;
  }
  if (ieq(name.c_str(), "X")) return 3;
  if (ieq(name.c_str(), "Y")) return 4;
  if (ieq(name.c_str(), "MT") || ieq(name.c_str(), "M")) return 5;
  if (ieq(name.c_str(), "XY") || ieq(name.c_str(), "PAR1") || ieq(name.c_str(), "PAR2")) {
    return 0;
  }
  return allow_extra ? 0 : 2;
}

// allele counts -> allele frequencies (ComputeAlleleFreqs, plink2_filter.cc:2113-2153: count * (1 / total), 1 / allele_ct each when nothing
// is observed) -> the major allele (GetMajIdxMulti, plink2_common.cc:1042-1070) and its frequency (GetAlleleFreq, plink2_common.h:584-593)
uint32_t pick_major_allele(const std::vector<uint64_t>& cnt, double* maj_freq) {
  const uint32_t allele_ct = static_cast<uint32_t>(cnt.size());
  uint64_t tot = 0;
  for (uint64_t c : cnt) {
    tot += c;
  }
  std::vector<double> freq(allele_ct - 1);
  if (!tot) {
    const double recip = 1.0 / static_cast<double>(allele_ct);
    for (double& f : freq) {
      f = recip;
    }
  } else {
    const double tot_recip = 1.0 / static_cast<double>(tot);
    for (uint32_t a = 0; a + 1 < allele_ct; ++a) {
      freq[a] = static_cast<double>(cnt[a]) * tot_recip;
    }
  }
  uint32_t maj;
  if (freq[0] >= 0.5) {
    maj = 0;
  } else if (allele_ct == 2) {
    maj = 1;
  } else {
    const double alt1_freq = freq[1];
    if (alt1_freq >= 0.5) {
      maj = 1;
    } else {
      const double ref_freq = freq[0];
      maj = 1;
      double max_freq = alt1_freq;
      if (ref_freq >= alt1_freq) {
        maj = 0;
        max_freq = ref_freq;
      }
      double tot_nonlast = ref_freq + alt1_freq;
      for (uint32_t a = 2; a + 1 < allele_ct; ++a) {
        if (freq[a] > max_freq) {
          maj = a;
          max_freq = freq[a];
        }
        tot_nonlast += freq[a];
      }
      if (max_freq + tot_nonlast < 1.0 - kSmallEpsilon) {
        maj = allele_ct - 1;
      }
    }
  }
  if (maj + 1 < allele_ct) {
    *maj_freq = freq[maj];
  } else {
    double last = 1.0 - freq[0];
    for (uint32_t a = 1; a + 1 < allele_ct; ++a) {
      last -= freq[a];
    }
    *maj_freq = (last > 0.0) ? last : 0.0;
  }
  return maj;
}

// Multiallelic variant on the host (rare: a few percent of sites): founder allele counts -> allele frequencies in
// the reference's arithmetic (ComputeAlleleFreqs, plink2_filter.cc:2113-2153: freq[a] = count[a] * (1/total), 1/k
// when nothing is observed) -> major allele (GetMajIdx / GetMajIdxMulti, plink2_common.h:559-567,
// plink2_common.cc:1042-1070) -> its frequency (GetAlleleFreq, plink2_common.h:584-593) -> the 2-bit row
// PgrGetInv1 would return for that allele (pgenlib_read.cc:5544-5563): copies of non-major alleles, 3 = missing.
//
// phase != nullptr (--indep-pairphase; two byte buffers of ceil(raw samples / 8), phasepresent then phaseinfo): the row
// holds two haplotypes per founder instead (haplotype = genotype code 2h, h = carries a non-major allele; index 2f =
// the second haplotype of the file, 2f+1 the first, as the conversion kernel lays out LDP_GENO_PHASED rows), following
// PgrGetInv1P -> Get1MP (pgenlib_read.cc:7016,6962) -> HapsplitMustPhased.  Get1MP hands the file's phaseinfo through
// unchanged, which means "the HIGHER allele of the het is on the first haplotype"; read as "the counted allele is"
// it is off by a swap whenever the major allele is the LOWER allele of a multiallelic het (1|2 with major = 1).  The
// reference prunes with that assignment (reproduced here; the physically right one gives different lists on
// VCF-imported data, tests/test_pairphase.py).  *unphased: a collapsed het (one major allele) without phase.
void multiallelic_inverse_row(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const std::vector<uint32_t>& founder_idx,
                              std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* out_row, uint64_t out_rec, double* maj_freq,
                              uint8_t* phase, uint64_t phase_bytes, bool* unphased, uint32_t* maj_idx) {
  if (phase ? ldp_pgen_read_alleles_phased(pg, raw_variant, alt_ct, lo->data(), hi->data(), phase, phase + phase_bytes)
            : ldp_pgen_read_alleles(pg, raw_variant, alt_ct, lo->data(), hi->data())) {
    die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
  }
  const uint32_t allele_ct = alt_ct + 1;
  std::vector<uint64_t> cnt(allele_ct, 0);
  for (uint32_t s : founder_idx) {
    if ((*lo)[s] != 255) {
      if ((*lo)[s] >= allele_ct || (*hi)[s] >= allele_ct) {
        die(6, "\nError: allele index out of range in multiallelic record.\n");
      }
      ++cnt[(*lo)[s]];
      ++cnt[(*hi)[s]];
    }
  }
  const uint32_t maj = pick_major_allele(cnt, maj_freq);
  if (maj_idx) {
    *maj_idx = maj;
  }
  memset(out_row, 0, out_rec);
  uint32_t f = 0;
  if (phase) {
    const uint8_t* present = phase;
    const uint8_t* info = phase + phase_bytes;
    for (uint32_t s : founder_idx) {
      uint32_t hap_second = 3, hap_first = 3;
      const uint32_t a = (*lo)[s], b = (*hi)[s];
      if (a != 255) {
        const bool swapped = (info[s >> 3] >> (s & 7)) & 1;
        uint32_t first_allele = swapped ? b : a;
        uint32_t second_allele = swapped ? a : b;
        if ((maj >= 1) && (a == maj) && (b != maj)) {
          std::swap(first_allele, second_allele);  // (the reference's reading of phaseinfo, see above)
        }
        hap_first = (first_allele != maj) ? 2 : 0;
        hap_second = (second_allele != maj) ? 2 : 0;
        if (((a == maj) != (b == maj)) && !((present[s >> 3] >> (s & 7)) & 1)) {
          *unphased = true;
        }
      }
      out_row[f >> 2] |= static_cast<uint8_t>(hap_second << (2 * (f & 3)));
      ++f;
      out_row[f >> 2] |= static_cast<uint8_t>(hap_first << (2 * (f & 3)));
      ++f;
    }
    return;
  }
  for (uint32_t s : founder_idx) {
    const uint32_t code = ((*lo)[s] == 255) ? 3u : (static_cast<uint32_t>((*lo)[s] != maj) + static_cast<uint32_t>((*hi)[s] != maj));
    out_row[f >> 2] |= static_cast<uint8_t>(code << (2 * (f & 3)));
    ++f;
  }
}



// raw_row: REF-based pgen codes of all samples.  Writes the PgrGetInv1-style row (+ het->missing) and maj_freq.
// phase != nullptr (--indep-pairphase on chrX, plink2_ld.cc:2060-2097): the non-male founders contribute their two
// haplotypes, split by the phaseinfo bits of all samples (HapsplitMustPhased), instead of their genotype twice; a
// haplotype h is carried as the genotype code 2h (include/ldprune_hip.h, LDP_GENO_PHASED).
void build_sex_row(const SexPlan& sp, const uint8_t* raw_row, uint8_t* out_row, uint64_t out_rec, double* maj_freq, const uint8_t* phase) {
  uint64_t g[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
  for (uint32_t s : sp.part1) {
    ++m[code_at(raw_row, s)];
  }
  for (uint32_t s : sp.part2) {
    ++g[code_at(raw_row, s)];
  }
  uint64_t ref_ct, alt_ct;
  if (sp.x_freq) {
    for (int q = 0; q < 4; ++q) {
      g[q] += m[q];  // all founders
    }
    const uint64_t n_all = g[0] + g[1] + g[2];
    const uint64_t n_male = m[0] + m[1] + m[2];
    alt_ct = 4 * g[2] + 2 * g[1] - 2 * m[2] - m[1];
    const uint64_t tot = 2 * (2 * n_all - n_male);
    ref_ct = tot - alt_ct;
  } else {
    ref_ct = 2 * m[0] + m[1];
    alt_ct = 2 * m[2] + m[1];
  }
  const uint64_t tot = ref_ct + alt_ct;
  double ref_freq = 0.5;
  if (tot) {
    const double tot_recip = 1.0 / static_cast<double>(tot);
    ref_freq = static_cast<double>(ref_ct) * tot_recip;
  }
  const bool alt_major = !(ref_freq >= 0.5);
  double mf = ref_freq;
  if (alt_major) {
    mf = 1.0 - ref_freq;
    if (mf < 0.0) {
      mf = 0.0;
    }
  }
  *maj_freq = mf;
  memset(out_row, 0, out_rec);
  static const uint8_t inv[4] = {2, 1, 0, 3};
  uint32_t f = 0;
  for (uint32_t s : sp.part1) {
    uint32_t c = code_at(raw_row, s);
    c = (c == 1) ? 3u : (alt_major ? inv[c] : c);  // SetHetMissing
    out_row[f >> 2] |= static_cast<uint8_t>(c << (2 * (f & 3)));
    ++f;
  }
  if (phase) {
    for (uint32_t s : sp.part2) {
      const uint32_t c = code_at(raw_row, s);
      const bool ph = (phase[s >> 3] >> (s & 7)) & 1;
      uint32_t hap[2] = {3, 3};
      if (c != 3) {
        const bool alt_first = (c == 2) || ((c == 1) && ph);
        const bool alt_second = (c == 2) || ((c == 1) && !ph);
        hap[0] = (alt_first != alt_major) ? 2 : 0;
        hap[1] = (alt_second != alt_major) ? 2 : 0;
      }
      for (int k = 0; k < 2; ++k) {
        out_row[f >> 2] |= static_cast<uint8_t>(hap[k] << (2 * (f & 3)));
        ++f;
      }
    }
    return;
  }
  for (int rep = 0; rep < 2; ++rep) {
    for (uint32_t s : sp.part2) {
      uint32_t c = code_at(raw_row, s);
      c = alt_major ? inv[c] : c;
      out_row[f >> 2] |= static_cast<uint8_t>(c << (2 * (f & 3)));
      ++f;
    }
  }
}

// A variant with several ALT alleles on chrX / chrY / MT (--indep-pairwise): the allele counts that choose its major allele weigh the
// founders as the reference's allele-frequency pass does (LoadAlleleAndGenoCountsThread, plink2_data.cc:2840-2895: chrX "double all counts,
// then subtract male counts" -- a non-male's allele copy counts 2, a male's 1, a male het half / half --; chrY / MT diploid-style counts over the
// plan's founders, :2752-2835), the row is PgrGetInv1's collapse on that allele (copies of non-major alleles, pgenlib_read.cc:5544-5563) in the
// plan's layout: part 1 once with "one major + one other allele" made missing (plink2_ld.cc:1362-1388), part 2 twice.
void multiallelic_sex_row(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* out_row,
                          uint64_t out_rec, double* maj_freq) {
  if (ldp_pgen_read_alleles(pg, raw_variant, alt_ct, lo->data(), hi->data())) {
    die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
  }
  const uint32_t allele_ct = alt_ct + 1;
  std::vector<uint64_t> cnt(allele_ct, 0);
  auto add = [&](uint32_t s, uint64_t w) {
    if ((*lo)[s] != 255) {
      if ((*lo)[s] >= allele_ct || (*hi)[s] >= allele_ct) {
        die(6, "\nError: allele index out of range in multiallelic record.\n");
      }
      cnt[(*lo)[s]] += w;
      cnt[(*hi)[s]] += w;
    }
  };
  for (uint32_t s : sp.part1) {
    add(s, 1);
  }
  for (uint32_t s : sp.part2) {
    add(s, 2);
  }
  const uint32_t maj = pick_major_allele(cnt, maj_freq);
  memset(out_row, 0, out_rec);
  uint32_t f = 0;
  auto code_of = [&](uint32_t s) { return ((*lo)[s] == 255) ? 3u : (static_cast<uint32_t>((*lo)[s] != maj) + static_cast<uint32_t>((*hi)[s] != maj)); };
  for (uint32_t s : sp.part1) {
    uint32_t c = code_of(s);
    c = (c == 1) ? 3u : c;
    out_row[f >> 2] |= static_cast<uint8_t>(c << (2 * (f & 3)));
    ++f;
  }
  for (int rep = 0; rep < 2; ++rep) {
    for (uint32_t s : sp.part2) {
      out_row[f >> 2] |= static_cast<uint8_t>(code_of(s) << (2 * (f & 3)));
      ++f;
    }
  }
}

// The major allele (and its frequency) of a variant with several ALT alleles under a chromosome's own allele-frequency weights, as multiallelic_sex_row counts them: the
// plan's part-1 founders with weight 1 per allele copy, its part-2 founders with weight 2 (chrX: males | non-males; chrY: the non-female founders; MT: every founder).
// For the r^2 outputs, whose rows stay over all founders (the collapse is PgrGetInv1's on this allele; the sample layouts are the prune's business).
uint32_t sex_major_allele(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, double* maj_freq) {
  if (ldp_pgen_read_alleles(pg, raw_variant, alt_ct, lo->data(), hi->data())) {
    die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
  }
  const uint32_t allele_ct = alt_ct + 1;
  std::vector<uint64_t> cnt(allele_ct, 0);
  auto add = [&](uint32_t s, uint64_t w) {
    if ((*lo)[s] != 255) {
      if ((*lo)[s] >= allele_ct || (*hi)[s] >= allele_ct) {
        die(6, "\nError: allele index out of range in multiallelic record.\n");
      }
      cnt[(*lo)[s]] += w;
      cnt[(*hi)[s]] += w;
    }
  };
  for (uint32_t s : sp.part1) {
    add(s, 1);
  }
  for (uint32_t s : sp.part2) {
    add(s, 2);
  }
  return pick_major_allele(cnt, maj_freq);
}

// ... under --indep-pairphase on chrX with non-male founders (plink2_ld.cc:2060-2097 after PgrGetInv1P on the chromosome's major allele): part 1 -- the males -- one
// haplotype each as above, part 2 two haplotypes per founder split by the file's phase bits in Get1MP's reading (multiallelic_inverse_row above: the bit is passed
// through as "the counted allele is on the first haplotype", which for a major allele other than REF is the complement of what the file says); a haplotype h is the code
// 2h, the first haplotype of the file at the even position (build_sex_row's layout).  *unphased: a het of the collapsed row of a part-2 founder without phase.
// phase: scratch of 2 * ceil(raw samples / 8) bytes.
void multiallelic_sex_row_phased(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* phase,
                                 uint64_t phase_bytes, uint8_t* out_row, uint64_t out_rec, double* maj_freq, bool* unphased) {
  if (ldp_pgen_read_alleles_phased(pg, raw_variant, alt_ct, lo->data(), hi->data(), phase, phase + phase_bytes)) {
    die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
  }
  const uint8_t* present = phase;
  const uint8_t* info = phase + phase_bytes;
  const uint32_t allele_ct = alt_ct + 1;
  std::vector<uint64_t> cnt(allele_ct, 0);
  auto add = [&](uint32_t s, uint64_t w) {
    if ((*lo)[s] != 255) {
      if ((*lo)[s] >= allele_ct || (*hi)[s] >= allele_ct) {
        die(6, "\nError: allele index out of range in multiallelic record.\n");
      }
      cnt[(*lo)[s]] += w;
      cnt[(*hi)[s]] += w;
    }
  };
  for (uint32_t s : sp.part1) {
    add(s, 1);
  }
  for (uint32_t s : sp.part2) {
    add(s, 2);
  }
  const uint32_t maj = pick_major_allele(cnt, maj_freq);
  memset(out_row, 0, out_rec);
  uint32_t f = 0;
  for (uint32_t s : sp.part1) {
    uint32_t c = ((*lo)[s] == 255) ? 3u : (static_cast<uint32_t>((*lo)[s] != maj) + static_cast<uint32_t>((*hi)[s] != maj));
    c = (c == 1) ? 3u : c;
    out_row[f >> 2] |= static_cast<uint8_t>(c << (2 * (f & 3)));
    ++f;
  }
  for (uint32_t s : sp.part2) {
    uint32_t hap[2] = {3, 3};
    const uint32_t a = (*lo)[s], b = (*hi)[s];
    if (a != 255) {
      const bool swapped = (info[s >> 3] >> (s & 7)) & 1;
      uint32_t first_allele = swapped ? b : a;
      uint32_t second_allele = swapped ? a : b;
      if ((maj >= 1) && (a == maj) && (b != maj)) {
        std::swap(first_allele, second_allele);
      }
      hap[0] = (first_allele != maj) ? 2 : 0;
      hap[1] = (second_allele != maj) ? 2 : 0;
      if (((a == maj) != (b == maj)) && !((present[s >> 3] >> (s & 7)) & 1)) {
        *unphased = true;
      }
    }
    for (int k = 0; k < 2; ++k) {
      out_row[f >> 2] |= static_cast<uint8_t>(hap[k] << (2 * (f & 3)));
      ++f;
    }
  }
}

// raw REF-coded row of one variant (decoding / .bed recoding as needed) into `buf`
void fetch_raw_row(ldp_pgen* pg, int storage_mode, uint32_t raw_variant, uint32_t raw_sample_ct, uint64_t rec_bytes, uint8_t* buf) {
  if (ldp_pgen_read(pg, raw_variant, 1, buf, rec_bytes, 1)) {
    die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
  }
  if (storage_mode == 0x01) {
    static const uint8_t conv[4] = {2, 3, 1, 0};  // .bed -> pgen codes (pgenlib_read.cc:2157)
    for (uint32_t sidx = 0; sidx < raw_sample_ct; ++sidx) {
      const uint32_t c = conv[code_at(buf, sidx)];
      uint8_t& b = buf[sidx >> 2];
      const uint32_t sh = 2 * (sidx & 3);
      b = static_cast<uint8_t>((b & ~(3u << sh)) | (c << sh));
    }
  }
}


}  // namespace p2h
