// p2h_args.cpp -- plink2-hip: the command line (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

std::vector<std::string> split_ws(const std::string& line) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) {
      ++i;
    }
    size_t j = i;
    while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') {
      ++j;
    }
    if (j > i) {
      out.emplace_back(line.substr(i, j - i));
    }
    i = j;
  }
  return out;
}

bool ieq(const char* a, const char* b) {
  for (; *a && *b; ++a, ++b) {
    if ((*a | 32) != (*b | 32)) {
      return false;
    }
  }
  return !*a && !*b;
}



// A column-set descriptor: either a plain list (exactly these columns) or +name / -name edits of the default set, never
// both; "-x" also removes "maybex" when x itself is not set.
uint32_t parse_col_descriptor(const std::string& desc, const std::vector<std::string>& names, uint32_t default_cols, const char* flag) {
  auto find = [&](const std::string& id) {
    for (size_t k = 0; k < names.size(); ++k) {
      if (names[k] == id) {
        return static_cast<int>(k);
      }
    }
    return -1;
  };
  uint32_t result = 0;
  if (desc.empty()) {
    return result;
  }
  const bool edits = (desc[0] == '+') || (desc[0] == '-');
  if (edits) {
    result = default_cols;
  }
  for (size_t p0 = 0; p0 <= desc.size();) {
    const size_t p1 = std::min(desc.find(',', p0), desc.size());
    std::string tok = desc.substr(p0, p1 - p0);
    const bool signed_tok = (!tok.empty()) && ((tok[0] == '+') || (tok[0] == '-'));
    if (signed_tok != edits) {
      die(8, "Error: Invalid --%s column set descriptor (either all column set IDs must be\npreceded by +/-, or none of them can be).\n", flag);
    }
    const char sign = edits ? tok[0] : '+';
    if (edits) {
      tok.erase(0, 1);
    }
    const int k = find(tok);
    if (k < 0) {
      die(8, "Error: Unrecognized ID '%s' in --%s column set descriptor.\n", tok.c_str(), flag);
    }
    if (sign == '+') {
      result |= 1u << k;
    } else if (result & (1u << k)) {
      result -= 1u << k;
    } else {
      const int mk = find("maybe" + tok);
      if (mk >= 0) {
        result &= ~(1u << mk);
      }
    }
    p0 = p1 + 1;
  }
  return result;
}

// ---- command line ----
// parse_args(): one pass over argv, each flag offered to the families below in turn (a family returns false for a flag that is not its
// own), then the checks between flags (check_flag_combinations).  Inside a family `i` is the cursor into argv, as in the loop.
struct ArgCursor {
  int argc;
  char** argv;
  int i;
};
#define LDP_ARG_FAMILY_PROLOGUE                                   \
  int& i = c.i;                                                   \
  const int argc = c.argc;                                        \
  char** const argv = c.argv;                                     \
  auto need = [&](int at, int n, const char* flag) {             \
    if (at + n >= argc) {                                         \
      die(8, "Error: Missing argument for %s.\n", flag);         \
    }                                                             \
  };                                                              \
  (void)need;                                                     \
  (void)argv

// the fileset and output names
bool parse_input_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--bfile" || f == "--pfile" || f == "--bpfile") {
    need(i, 1, f.c_str());
    std::string pre = argv[++i];
    // optional 'vzs' modifier: the variant table is zstd-compressed (<prefix>.pvar.zst / .bim.zst)
    std::string vz;
    if (i + 1 < argc && std::string(argv[i + 1]) == "vzs") {
      vz = ".zst";
      ++i;
    }
    if (f == "--bfile") {
      A.bed = pre + ".bed";
      A.bim = pre + ".bim" + vz;
      A.fam = pre + ".fam";
    } else if (f == "--pfile") {
      A.pgen = pre + ".pgen";
      A.pvar = pre + ".pvar" + vz;
      A.psam = pre + ".psam";
    } else {
      A.pgen = pre + ".pgen";
      A.bim = pre + ".bim" + vz;
      A.fam = pre + ".fam";
    }
  } else if (f == "--bed" || f == "--bim" || f == "--fam" || f == "--pgen" || f == "--pgi" || f == "--pvar" || f == "--psam" || f == "--out" || f == "--indep-preferred") {
    need(i, 1, f.c_str());
    std::string v = argv[++i];
    if (f == "--bed") A.bed = v;
    else if (f == "--bim") A.bim = v;
    else if (f == "--fam") A.fam = v;
    else if (f == "--pgen") A.pgen = v;
    else if (f == "--pgi") A.pgi = v;  // (external-index .pgen: plink2.cc:10572-10590)
    else if (f == "--pvar") A.pvar = v;
    else if (f == "--psam") A.psam = v;
    else if (f == "--out") A.out = v;
    else A.preferred = v;
  } else {
    return false;
  }
  return true;
}

// --indep-pairwise / --indep-pairphase / --r2-unphased / --r-unphased and their modifiers
bool parse_command_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--indep-pairwise" || f == "--indep-pairphase") {
    if (A.have_prune) {
      die(8, "Error: --indep-pairwise and --indep-pairphase cannot be used together.\n");
    }
    A.pairphase = (f == "--indep-pairphase");
    const char* fl = f.c_str();
    // <window size>['kb'] [step size (variant ct)] <unphased-hardcall-r^2 threshold>   (plink2.cc:7238-7313)
    std::vector<std::string> par;
    while (i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] == '-')) {
      par.emplace_back(argv[++i]);
    }
    if (par.size() < 2 || par.size() > 4) {
      die(8, "Error: %s accepts 2-4 arguments.\n", fl);
    }
    double first;
    const char* endp;
    if (!scan_double_plink(par[0].c_str(), &first, &endp) || first < 0.0) {
      die(8, "Error: Invalid %s window size '%s'.\n", fl, par[0].c_str());
    }
    size_t next = 1;
    bool is_kb = false;
    if (ieq(endp, "kb")) {
      is_kb = true;
    } else if (*endp) {
      die(8, "Error: Invalid %s window size '%s'.\n", fl, par[0].c_str());
    } else if (ieq(par[1].c_str(), "kb")) {
      is_kb = true;
      next = 2;
    }
    if (is_kb) {
      A.window_is_bp = true;
      if (first > 2147483.646) {
        A.window = 2147483646;
      } else {
        const int32_t w = static_cast<int32_t>(first * 1000 * (1 + kSmallEpsilon));
        if (w < 2) {
          die(8, "Error: %s window size cannot be smaller than 2.\n", fl);
        }
        A.window = w;
      }
    } else {
      A.window = (first > 2147483647) ? 2147483647u : static_cast<uint32_t>(static_cast<int32_t>(first));
    }
    if (next + 2 == par.size()) {
      // explicit step size
      char* e2;
      const long st = strtol(par[next].c_str(), &e2, 10);
      if (*e2 || st < 1 || st > 2147483646) {
        die(8, "Error: Invalid %s window-increment '%s'.\n", fl, par[next].c_str());
      }
      A.step = static_cast<uint32_t>(st);
      if (!is_kb) {
        if (A.step > A.window) {
          die(8, "Error: %s window-increment cannot be larger than window size.\n", fl);
        }
      } else if (A.step != 1) {
        die(8, "Error: %s window-increment must be 1 when window size is in\nkilobase units.\n", fl);
      }
      ++next;
    } else if (next + 1 != par.size()) {
      die(8, "Error: Invalid %s argument sequence.\n", fl);
    }
    const char* e3;
    if (!scan_double_plink(par[next].c_str(), &A.r2, &e3) || *e3 || A.r2 < 0.0 || A.r2 >= 1.0) {
      die(8, "Error: Invalid %s r^2 threshold '%s'.\n", fl, par[next].c_str());
    }
    A.have_prune = true;
  } else if ((f == "--r2-unphased") || (f == "--r-unphased")) {
    if (A.have_r2) {
      die(8, "Error: --r-phased, --r-unphased, --r2-phased, and --r2-unphased are mutually\nexclusive.\n");
    }
    A.r_unsquared = (f == "--r-unphased");
    g_r_unsquared = A.r_unsquared;
    // [{square | square0 | triangle | inter-chr}] ['yes-really'] [{zs | bin | bin4}] ... (plink2.cc:11090-11210)
    while (i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] == '-')) {
      std::string m = argv[++i];
      const bool is_shape = (m == "square") || (m == "square0") || (m == "triangle");
      const bool is_encoding = (m == "bin") || (m == "bin4") || (m == "zs");
      if (is_shape && (A.r2_shape >= 0)) {
        die(8, "Error: Multiple --r2-unphased shape modifiers.\n");  // plink2.cc:11068-11090
      }
      if (is_encoding && ((A.r2_float >= 0) || A.r2_zs)) {
        die(8, "Error: Multiple --r2-unphased encoding modifiers.\n");  // plink2.cc:11106-11118
      }
      if (m == "square") A.r2_shape = 0;
      else if (m == "square0") A.r2_shape = 1;
      else if (m == "triangle") A.r2_shape = 2;
      else if (m == "inter-chr") A.r2_inter = true;
      else if (m == "bin") A.r2_float = 0;
      else if (m == "bin4") A.r2_float = 1;
      else if (m == "zs") A.r2_zs = true;
      else if (m == "yes-really") A.yes_really = true;
      else if (m == "ref-based") A.r2_ref_based = true;          // multiallelic variants: REF vs the rest instead of major vs the rest
      else if (m == "allow-ambiguous-allele") A.r2_allow_ambiguous = true;
      else if (m.compare(0, 5, "cols=") == 0) {
        if (A.r2_cols_given) {
          die(8, "Error: Multiple --r2-unphased cols= modifiers.\n");
        }
        A.r2_cols_given = true;
        A.r2_cols_desc = m.substr(5);
      }
      else if ((m == "d") || (m == "dprime") || (m == "dprime-signed")) {
        die(8, "Error: --r2-unphased does not support computation of D or D'. Use --r2-phased\nwith 'cols=+%s' instead.\n", (m == "d") ? "d" : ((m == "dprime") ? "dprimeabs" : "dprime"));
      }
      else die(63, "Error: --r2-unphased modifier '%s' is not supported by plink2-hip (matrix shapes with bin/bin4, or the default-column table).\n", m.c_str());
    }
    if ((A.r2_shape < 0) && (A.r2_float >= 0)) {
      A.r2_shape = 0;  // an encoding without a shape: square (plink2_help.cc:1015-1017)
    }
    // (r's sign needs an allele to refer to: its default set adds MAJ, or REF with 'ref-based'; plink2.cc:11158-11162, :11196-11203)
    const uint32_t default_cols = kVcorColDefault | (A.r_unsquared ? (A.r2_ref_based ? kVcorColRef : kVcorColMaj) : 0u);
    A.r2_cols = default_cols;
    if (A.r2_cols_given) {  // plink2.cc:11158-11172
      A.r2_cols = parse_col_descriptor(A.r2_cols_desc, {"chrom", "pos", "id", "ref", "alt1", "alt", "maybeprovref", "provref", "maj", "nonmaj", "freq", "d", "dprime", "dprimeabs"},
                                       default_cols, A.r_unsquared ? "r-unphased" : "r2-unphased");
      if (A.r2_cols & (kVcorColD | kVcorColDprime | kVcorColDprimeAbs)) {
        die(8, "Error: --r2-unphased does not support computation of D or D'. Use --r%s-phased\ninstead.\n", A.r_unsquared ? "" : "2");
      }
    }
    if ((A.r2_inter || A.r2_cols_given) && (A.r2_shape >= 0)) {
      die(8, "Error: Matrix-only and table-only --r2-unphased settings cannot be used together.\n");  // plink2.cc:11187-11191
    }
    A.r2_table = (A.r2_shape < 0);
    A.r2_text = (A.r2_shape >= 0) && (A.r2_float < 0);  // shape without bin/bin4: tab-delimited text matrix
    if (A.r2_text) {
      A.r2_float = 0;  // computed as doubles, printed with 6 significant digits
    }
    A.have_r2 = true;
  } else {
    return false;
  }
  return true;
}

// --clump and its companions (plink2.cc:4861-5232)
bool parse_clump_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--clump") {  // plink2.cc:4861-4958
    need(i, 1, "--clump");
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string arg = argv[++i];
      if (arg == "zs") {  // (.clumps and the missing-ID lists through the zstd writer, OutnameZstSet :7920, :7944, :9004)
        if (!A.clump_files.empty()) {
          die(8, "Error: Invalid --clump argument sequence ('zs' must come before\nfilename(s)).\n");
        }
        A.clump_zs = true;
        continue;
      }
      if (arg.compare(0, 5, "cols=") == 0) {  // plink2.cc:4900-4925
        if (!A.clump_files.empty()) {
          die(8, "Error: Invalid --clump argument sequence ('cols=' must come before\nfilename(s)).\n");
        }
        if (A.clump_cols_given) {
          die(8, "Error: Multiple --clump cols= modifiers.\n");
        }
        A.clump_cols_given = true;
        A.clump_cols_desc = arg.substr(5);
        continue;
      }
      size_t p0 = 0;
      while (p0 <= arg.size()) {
        const size_t p1 = std::min(arg.find(',', p0), arg.size());
        if (p1 > p0) {
          A.clump_files.push_back(arg.substr(p0, p1 - p0));
        }
        p0 = p1 + 1;
      }
    }
    A.have_clump = true;
    A.clump_cols = kClumpColDefault;
    if (A.clump_cols_given) {
      A.clump_cols = parse_col_descriptor(A.clump_cols_desc, {"chrom", "pos", "ref", "alt1", "alt", "maybeprovref", "provref", "maybea1", "a1", "maybef", "f", "total",
                                                              "maybebounds", "bounds", "bins", "sp2"}, kClumpColDefault, "clump");
    }
  } else if (f == "--clump-bins") {  // plink2.cc:5139-5192
    need(i, 1, "--clump-bins");
    double prev_ln = -1.7976931348623157e308;
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string arg = argv[++i];
      const char* it = arg.c_str();
      while (true) {
        double cur_ln;
        it = scan_ln(it, &cur_ln);
        if ((!it) || ((*it != '\0') && (*it != ','))) {
          die(8, "Error: Invalid --clump-bins argument '%s'.\n", arg.c_str());
        }
        if (cur_ln <= prev_ln) {  // (the reference means to refuse these too, plink2.cc:5178, but never advances its prev_ln)
          die(8, "Error: --clump-bins values are not in increasing order.\n");
        }
        if (cur_ln >= 0.0) {
          die(8, "Error: --clump-bins values >= 1 do not make sense.\n");
        }
        prev_ln = cur_ln;
        A.clump_ln_bins.push_back(cur_ln * (1.0 + kSmallEpsilon));
        if (*it == '\0') {
          break;
        }
        ++it;
      }
    }
    if (A.clump_ln_bins.size() > 2000) {
      die(63, "Error: more than 2000 --clump-bins boundaries are not supported by plink2-hip.\n");
    }
  } else if (f == "--clump-unphased") {
    A.clump_unphased = true;
  } else if (f == "--clump-allow-overlap") {
    A.clump_allow_overlap = true;
  } else if (f == "--clump-force-a1") {  // plink2.cc:5200-5210
    A.clump_force_a1 = true;
  } else if (f == "--clump-a1-field") {  // plink2.cc:5059-5071
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      A.clump_a1_field.push_back(argv[++i]);
    }
    A.clump_no_a1 = A.clump_a1_field.empty();
  } else if ((f == "--clump-range") || (f == "--clump-range0")) {  // plink2.cc:5092-5120
    need(i, 1, f.c_str());
    if (!A.clump_range.empty()) {
      die(8, "Error: --clump-range and --clump-range0 cannot be used together.\n");
    }
    A.clump_range = argv[++i];
    A.clump_range0 = (f == "--clump-range0");
  } else if (f == "--clump-range-border") {  // plink2.cc:5121-5138
    need(i, 1, "--clump-range-border");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || (d < 0.0)) {
      die(8, "Error: Invalid --clump-range-border argument '%s'.\n", v.c_str());
    }
    A.clump_range_border = (d > 2147483.646) ? 0x7ffffffeu : static_cast<uint32_t>(static_cast<int32_t>(d * 1000 * (1 + kSmallEpsilon)));
    A.clump_range_border_given = true;
  } else if (f == "--clump-log10") {  // plink2.cc:5211-5232
    A.clump_in_log10 = A.clump_out_log10 = true;
    if ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string v = argv[++i];
      if (v == "input-only") {
        A.clump_out_log10 = false;
      } else if (v == "output-only") {
        A.clump_in_log10 = false;
      } else {
        die(8, "Error: Invalid --clump-log10 argument '%s'.\n", v.c_str());
      }
    }
  } else if ((f == "--clump-log10-p1") || (f == "--clump-log10-p2")) {  // plink2.cc:4979-5008
    need(i, 1, f.c_str());
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || (d < 0.0)) {
      die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), v.c_str());
    }
    ((f == "--clump-log10-p1") ? A.clump_ln_p1 : A.clump_ln_p2) = d * (-2.3025850929940457 * (1.0 - kSmallEpsilon));
    ((f == "--clump-log10-p1") ? A.clump_log10_p1 : A.clump_log10_p2) = true;
  } else if ((f == "--clump-p1") || (f == "--clump-p2")) {  // plink2.cc:5015-5046
    ((f == "--clump-p1") ? A.clump_plain_p1 : A.clump_plain_p2) = true;
    need(i, 1, f.c_str());
    const std::string v = argv[++i];
    double ln;
    const char* endp = scan_ln(v.c_str(), &ln);
    if (!endp || *endp || (ln > 0.0)) {
      die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), v.c_str());
    }
    ((f == "--clump-p1") ? A.clump_ln_p1 : A.clump_ln_p2) = ln * (1.0 - kSmallEpsilon);
  } else if (f == "--clump-r2") {  // plink2.cc:5047-5059
    need(i, 1, "--clump-r2");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || (d >= 1.0 - kSmallEpsilon)) {
      die(8, "Error: Invalid --clump-r2 argument '%s'.\n", v.c_str());
    }
    A.clump_r2_raw = d;
    A.clump_r2 = d * (1.0 + kSmallEpsilon);
  } else if (f == "--clump-kb") {  // plink2.cc:4960-4978
    need(i, 1, "--clump-kb");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || (d < 0.001)) {
      die(8, "Error: Invalid --clump-kb argument '%s'.\n", v.c_str());
    }
    d *= 1000;
    A.clump_bp_radius = (d > 2147483647.0) ? 0x7ffffffeu : static_cast<uint32_t>(static_cast<int32_t>(d * (1.0 + kSmallEpsilon) - 1));
  } else if ((f == "--clump-id-field") || (f == "--clump-snp-field") || (f == "--clump-p-field") || (f == "--clump-field") ||
             (f == "--clump-test-field") || (f == "--clump-test")) {
    // one or more names, highest priority first; --clump-test[-field] without arguments turns the TEST filter off
    std::vector<std::string> names;
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      names.push_back(argv[++i]);
    }
    if ((f == "--clump-test") || (f == "--clump-test-field")) {
      if (names.empty()) {
        A.clump_no_test = true;
      }
      ((f == "--clump-test") ? A.clump_test : A.clump_test_field) = names;
    } else {
      if (names.empty()) {
        die(8, "Error: %s needs at least one column name.\n", f.c_str());
      }
      (((f == "--clump-p-field") || (f == "--clump-field")) ? A.clump_p_field : A.clump_id_field) = names;
    }
  } else if (f.compare(0, 7, "--clump") == 0) {
    die(63, "Error: %s is not supported by plink2-hip's --clump yet.\n", f.c_str());
  } else {
    return false;
  }
  return true;
}

// variant and sample filters
bool parse_filter_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--snps-only") {  // plink2.cc:11437-11453
    A.snps_only = true;
    if ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string v = argv[++i];
      if (v != "just-acgt") {
        die(8, "Error: Invalid --snps-only argument '%s'.\n", v.c_str());
      }
      A.snps_only_acgt = true;
    }
  } else if (f == "--make-founders") {  // plink2.cc:9555-9575
    A.make_founders = true;
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string v = argv[++i];
      if (v == "require-2-missing") {
        A.make_founders_require2 = true;
      } else if (v == "first") {
        A.make_founders_first = true;
      } else {
        die(8, "Error: Invalid --make-founders argument '%s'.\n", v.c_str());
      }
    }
  } else if ((f == "--chr") || (f == "--not-chr")) {  // ParseChrRanges, plink2_cmdline.cc: "1-4, 22, X" in one or several arguments
    std::vector<std::string>& dst = (f == "--chr") ? A.chr_keep : A.chr_drop;
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string arg = argv[++i];
      size_t p0 = 0;
      while (p0 < arg.size()) {
        const size_t p1 = std::min(arg.find(',', p0), arg.size());
        if (p1 > p0) {
          dst.push_back(arg.substr(p0, p1 - p0));
        }
        p0 = p1 + 1;
      }
    }
    if (dst.empty()) {
      die(8, "Error: %s requires at least one value.\n", f.c_str());
    }
  } else if (f == "--max-alleles") {  // plink2.cc:9340-9360
    need(i, 1, "--max-alleles");
    const std::string v = argv[++i];
    char* endp;
    const unsigned long n = strtoul(v.c_str(), &endp, 10);
    if (v.empty() || *endp || (n < 1) || (n > 0x7fffffffUL)) {  // (ScanPosintDefcapx: any positive integer)
      die(8, "Error: Invalid --max-alleles argument '%s'.\n", v.c_str());
    }
    A.max_alleles = static_cast<uint32_t>(n);
  } else if (f == "--autosome") {
    A.autosome = true;
  } else if ((f == "--maf") || (f == "--max-maf") || (f == "--geno")) {  // plink2.cc:8690-8742, 8745-8790, 6487-6516
    double d = (f == "--maf") ? 0.01 : 0.1;
    if ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string v = argv[++i];
      const char* endp;
      if (!scan_double_plink(v.c_str(), &d, &endp) || *endp) {
        if (*endp == ':' || !((v[0] >= '0' && v[0] <= '9') || v[0] == '.')) {
          die(63, "Error: %s modifiers ('%s') are not supported by plink2-hip.\n", f.c_str(), v.c_str());
        }
        die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), v.c_str());
      }
      if ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
        die(63, "Error: %s modifiers ('%s') are not supported by plink2-hip.\n", f.c_str(), argv[i + 1]);
      }
      if (d < 0.0) {
        die(8, "Error: %s argument '%s' too small (must be >= 0).\n", f.c_str(), v.c_str());
      }
      if ((f == "--max-maf") ? (d >= 1.0) : (d > 1.0)) {
        die(8, "Error: %s argument '%s' too large (must be %s 1).\n", f.c_str(), v.c_str(), (f == "--max-maf") ? "<" : "<=");
      }
    } else if (f == "--max-maf") {
      die(8, "Error: --max-maf requires a value.\n");
    }
    ((f == "--maf") ? A.min_maf : ((f == "--max-maf") ? A.max_maf : A.geno)) = d;
  } else if ((f == "--mac") || (f == "--max-mac")) {  // plink2.cc:8785-8867 (default mode: the non-major allele's dosage sum over the founders)
    if ((i + 1 >= argc) || (argv[i + 1][0] == '-')) {
      die(8, "Error: %s requires a value.\n", f.c_str());
    }
    const std::string v = argv[++i];
    double d = 0.0;
    const char* endp = v.c_str();  // (scan_double_plink leaves it alone when there is no number at all)
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp) {
      if (*endp == ':') {
        die(63, "Error: %s modifiers ('%s') are not supported by plink2-hip.\n", f.c_str(), v.c_str());
      }
      die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), v.c_str());
    }
    if ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      die(63, "Error: %s modifiers ('%s') are not supported by plink2-hip.\n", f.c_str(), argv[i + 1]);
    }
    if ((d < 0.0) || (d > 2147483646.0)) {
      die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), v.c_str());
    }
    if (f == "--mac") {
      if (d > 0.0) {  // round up, but keep as much precision as possible
        const int32_t int_part = static_cast<int32_t>(d);
        d -= int_part;
        A.min_allele_ddosage = static_cast<uint64_t>(int_part) * 32768ull;
        if (d > 0.0) {
          A.min_allele_ddosage += 1 + static_cast<uint64_t>(d * (32768 * (1 - kSmallEpsilon)));
        }
      }
    } else {
      A.max_allele_ddosage = static_cast<uint64_t>(static_cast<int64_t>(d * 32768));  // round down
    }
  } else if (f == "--ac-founders") {
    A.ac_founders = true;
  } else if ((f == "--extract") || (f == "--exclude") || (f == "--keep") || (f == "--remove")) {
    std::vector<std::string>& dst = (f == "--extract") ? A.extract_files : ((f == "--exclude") ? A.exclude_files : ((f == "--keep") ? A.keep_files : A.remove_files));
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      dst.push_back(argv[++i]);
    }
    if (dst.empty()) {
      die(8, "Error: %s requires at least one filename.\n", f.c_str());
    }
    if (((f == "--extract") || (f == "--exclude")) && ((dst[0] == "range") || (dst[0] == "bed0") || (dst[0] == "bed1") || (dst[0] == "intersect"))) {
      die(63, "Error: the '%s' mode of %s is not supported by plink2-hip.\n", dst[0].c_str(), f.c_str());
    }
  } else {
    return false;
  }
  return true;
}

// --ld-window* / --ld-snp* of the r^2 table
bool parse_ldwindow_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--ld-snp") {
    need(i, 1, "--ld-snp");
    if (!A.ld_snps.empty() || !A.ld_snp_list.empty()) {
      die(8, "Error: --ld-snp cannot be used with --ld-snps or --ld-snp-list.\n");
    }
    A.ld_snps.emplace_back(argv[++i], "");
  } else if (f == "--ld-snps") {  // ParseNameRanges, plink2_cmdline.cc:2247: comma-separated IDs and first-last ranges
    if (!A.ld_snps.empty() || !A.ld_snp_list.empty()) {
      die(8, "Error: --ld-snps cannot be used with --ld-snp or --ld-snp-list.\n");
    }
    while ((i + 1 < argc) && (argv[i + 1][0] != '-')) {
      const std::string arg = argv[++i];
      size_t p0 = 0;
      while (p0 <= arg.size()) {
        const size_t p1 = std::min(arg.find(',', p0), arg.size());
        const std::string piece = arg.substr(p0, p1 - p0);
        const size_t dash = piece.find('-');
        if (piece.empty() || (dash == 0) || (dash + 1 == piece.size())) {
          die(8, "Error: Invalid --ld-snps argument '%s'.\n", arg.c_str());
        }
        if (dash == std::string::npos) {
          A.ld_snps.emplace_back(piece, "");
        } else {
          A.ld_snps.emplace_back(piece.substr(0, dash), piece.substr(dash + 1));
        }
        p0 = p1 + 1;
      }
    }
    if (A.ld_snps.empty()) {
      die(8, "Error: --ld-snps requires at least one value.\n");
    }
  } else if (f == "--ld-snp-list") {
    need(i, 1, "--ld-snp-list");
    if (!A.ld_snps.empty()) {
      die(8, "Error: --ld-snp-list cannot be used with --ld-snp.\n");
    }
    A.ld_snp_list = argv[++i];
  } else if (f == "--ld-window") {  // plink2.cc:7908-7920
    need(i, 1, "--ld-window");
    const std::string v = argv[++i];
    char* endp;
    const unsigned long n = strtoul(v.c_str(), &endp, 10);
    if (v.empty() || *endp || n < 2 || n > 0x7ffffffeul) {
      die(8, "Error: Invalid --ld-window argument '%s'.\n", v.c_str());
    }
    A.ld_var_ct_radius = static_cast<uint32_t>(n) - 1;
  } else if (f == "--ld-window-kb") {  // plink2.cc:7921-7937
    need(i, 1, "--ld-window-kb");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || d < 0) {
      die(8, "Error: Invalid --ld-window-kb argument '%s'.\n", v.c_str());
    }
    d *= 1000 * (1 + kSmallEpsilon);
    A.ld_bp_radius = (d > 2147483646) ? 2147483646u : static_cast<uint32_t>(static_cast<int32_t>(d));
  } else if (f == "--ld-window-cm") {  // plink2.cc:7938-7949
    need(i, 1, "--ld-window-cm");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || d < 0) {
      die(8, "Error: Invalid --ld-window-cm argument '%s'.\n", v.c_str());
    }
    A.ld_cm_radius = d * (1 + kSmallEpsilon);
  } else if (f == "--ld-window-r2") {  // plink2.cc:7950-7964
    need(i, 1, "--ld-window-r2");
    const std::string v = argv[++i];
    double d;
    const char* endp;
    if (!scan_double_plink(v.c_str(), &d, &endp) || *endp || d > 1.0) {
      die(8, "Error: Invalid --ld-window-r2 argument '%s'.\n", v.c_str());
    }
    if (d > 0.0) {
      d *= 1 - kSmallEpsilon;
    }
    A.ld_min_r2 = d;
  } else if (f == "--ld-window-cm" || f == "--ld-snp" || f == "--ld-snps" || f == "--ld-snp-list") {
    die(63, "Error: %s is not supported by plink2-hip.\n", f.c_str());
  } else {
    return false;
  }
  return true;
}

DebugHooks g_dbg;

// everything else (order, threads, debugging aids)
bool parse_misc_flags(Args& A, ArgCursor& c, const std::string& f) {
  LDP_ARG_FAMILY_PROLOGUE;
  if (f == "--silent") {
    g_silent = true;
  } else if (f == "--indep-order") {
    need(i, 1, "--indep-order");
    std::string v = argv[++i];
    if (v == "1") A.order = 1;
    else if (v == "2") A.order = 2;
    else die(8, "Error: Invalid --indep-order mode '%s' ('1' or '2' expected).\n", v.c_str());
  } else if (f == "--bad-ld") {
    A.bad_ld = true;
  } else if (f == "--allow-extra-chr") {
    A.allow_extra_chr = true;
  } else if (f == "--timing") {
    A.timing = true;
  } else if (f == "--dry-run") {
    A.dry_run = true;
  } else if (f == "--debug-alias-devices") {
    g_dbg.alias_devices = true;   // (test hook: the N engines of --gpus N dealt onto the devices there are, host transport for the exchange)
  } else if (f == "--debug-x-host") {
    g_dbg.x_host = true;          // (test hook: chrX pairs as lists through ldp_pair_stats and the host arithmetic)
  } else if (f == "--debug-host-decode") {
    g_dbg.host_decode = true;     // (measurement / test hook: variable-width records decoded by the host reader)
  } else if (f == "--debug-no-bind") {
    g_dbg.no_bind = true;         // (measurement: stay on whatever CPUs the scheduler picks instead of the device's NUMA node)
  } else if (f == "--debug-serial-feed") {
    g_dbg.serial_feed = true;     // (measurement: --gpus N feeds its engines one after the other from one thread, as rounds 2-5 did)
  } else if (f == "--debug-load-map") {
    g_dbg.load_map = true;        // (measurement: fixed-width rows copied out of the mapping instead of pread())
  } else if ((f == "--debug-x-rows") || (f == "--debug-decode-threads")) {
    need(i, 1, f.c_str());
    const int v = atoi(argv[++i]);
    if (v < 1) {
      die(8, "Error: Invalid %s argument '%s'.\n", f.c_str(), argv[i]);
    }
    ((f == "--debug-x-rows") ? g_dbg.x_rows : g_dbg.decode_threads) = static_cast<uint32_t>(v);
  } else if (f == "--debug-format-g6") {
    // test hook (no GPU needed): one hex bit pattern of a double per line in, the .vcor number formatting out
    need(i, 1, "--debug-format-g6");
    FILE* df = fopen(argv[++i], "r");
    if (!df) {
      die(3, "Error: Failed to open %s.\n", argv[i]);
    }
    char line[64], num[40];
    while (fgets(line, sizeof(line), df)) {
      const unsigned long long bits = strtoull(line, nullptr, 16);
      double d;
      memcpy(&d, &bits, sizeof(d));
      *format_g6(d, num) = '\0';
      puts(num);
    }
    fclose(df);
    exit(0);
  } else if (f == "--debug-zstd") {
    // test hook (no GPU needed): <in> <out.zst> through the 'zs' output writer, in odd-sized pieces
    need(i, 2, "--debug-zstd");
    std::ifstream in(argv[i + 1], std::ios::binary);
    if (!in) {
      die(3, "Error: Failed to open %s.\n", argv[i + 1]);
    }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string data = ss.str();
    OutFile of;
    of.open(argv[i + 2], true);
    for (size_t pos = 0, piece = 1; pos < data.size(); pos += piece, piece = piece * 3 + 1) {
      piece = std::min(piece, data.size() - pos);
      of.write(data.data() + pos, piece);
    }
    of.close();
    exit(0);
  } else if (f == "--parallel") {
    need(i, 2, "--parallel");
    char* end = nullptr;
    const long k = strtol(argv[i + 1], &end, 10);
    if ((*end) || (k < 1) || (k > 32768)) {
      die(8, "Error: Invalid --parallel job index '%s'.\n", argv[i + 1]);
    }
    const long n = strtol(argv[i + 2], &end, 10);
    if ((*end) || (n < 2) || (n > 32768) || (n < k)) {
      die(8, "Error: Invalid --parallel total job count '%s'.\n", argv[i + 2]);
    }
    A.parallel_idx = static_cast<uint32_t>(k - 1);
    A.parallel_tot = static_cast<uint32_t>(n);
    i += 2;
  } else if (f == "--gpus") {
    need(i, 1, "--gpus");
    A.gpus = atoi(argv[++i]);
  } else if (f == "--threads" || f == "--memory" || f == "--seed") {
    need(i, 1, f.c_str());
    ++i;  // accepted for command-line compatibility; the work runs on the GPU(s)
  } else {
    return false;
  }
  return true;
}

#undef LDP_ARG_FAMILY_PROLOGUE

// what the reference checks between flags once all of them are read
void check_flag_combinations(Args& A) {
  if (A.have_clump) {
    if (A.have_prune || A.have_r2) {
      die(8, "Error: run --clump on its own.\n");
    }
    if (!A.clump_unphased) {
      // (without it the reference uses phased-hardcall / EM haplotype-frequency r^2, ComputeR2 :6490-6650: not this path)
      die(63, "Error: plink2-hip's --clump computes unphased hardcall r^2 only: add --clump-unphased.\n");
    }
    if (A.parallel_tot != 1) {
      die(8, "Error: --parallel has no effect on --clump.\n");
    }
    // the rest of the program sees a windowed r^2 run: chromosome 0 stripped, sorted positions required
    A.have_r2 = true;
    A.r2_table = true;
  } else if (A.clump_unphased) {
    die(8, "Error: --clump-unphased must be used with --clump.\n");
  }
  if (!A.have_prune && !A.have_r2) {
    die(8, "Error: no command given (plink2-hip implements --indep-pairwise and --r2-unphased matrices).\n");
  }
  if (A.have_prune && (A.parallel_tot != 1)) {
    die(63, "Error: --parallel only distributes the --r2-unphased outputs in plink2-hip (the prune shards by --gpus).\n");
  }
  if (A.have_prune && A.have_r2) {
    die(8, "Error: run --indep-pairwise and --r2-unphased separately.\n");
  }
  if ((A.clump_plain_p1 && A.clump_log10_p1) || (A.clump_plain_p2 && A.clump_log10_p2)) {  // plink2.cc:5014-5016, :5032-5034
    die(8, "Error: --clump-p%d cannot be used with --clump-log10-p%d.\n", (A.clump_plain_p1 && A.clump_log10_p1) ? 1 : 2, (A.clump_plain_p1 && A.clump_log10_p1) ? 1 : 2);
  }
  if ((A.clump_in_log10 || A.clump_out_log10 || A.clump_log10_p1 || A.clump_log10_p2) && !A.have_clump) {
    die(8, "Error: --clump-log10 must be used with --clump.\n");
  }
  // (the reference reads its flags in sorted order, and "clump-range-border" sorts before "clump-range0": with --clump-range0 the
  // border flag finds no range file yet, plink2.cc:5122-5125)
  if (A.clump_force_a1 && A.clump_no_a1) {
    die(8, "Error: --clump-force-a1 does not make sense with empty --clump-a1-field\nargument.\n");
  }
  if ((A.clump_force_a1 || A.clump_no_a1 || !A.clump_a1_field.empty()) && !A.have_clump) {
    die(8, "Error: --clump-force-a1 must be used with --clump.\n");
  }
  if (A.clump_range_border_given && (A.clump_range.empty() || A.clump_range0)) {
    die(8, "Error: --clump-range-border must be used with --clump-range[0].\n");
  }
  if ((!A.clump_range.empty()) && !A.have_clump) {
    die(8, "Error: --clump-range must be used with --clump.\n");
  }
  if (!A.clump_ln_bins.empty()) {  // plink2.cc:5139-5147
    if (!A.have_clump) {
      die(8, "Error: --clump-bins must be used with --clump.\n");
    }
    if (!(A.clump_cols & kClumpColBins)) {
      die(8, "Error: --clump-bins does not make sense when --clump 'bins' column set has been\nexcluded.\n");
    }
  }
  const bool ld_window_given = (A.ld_var_ct_radius != 0x7fffffff) || (A.ld_bp_radius != 0xffffffffu) || (A.ld_cm_radius != -1.0);
  const bool ld_snp_given = (!A.ld_snps.empty()) || (!A.ld_snp_list.empty());
  if (ld_snp_given && (!A.have_r2 || A.have_clump)) {
    die(8, "Error: --ld-window.../--ld-snp... must be used with --r[2]-[un]phased.\n");  // plink2.cc:12960
  }
  if (ld_snp_given && A.have_r2 && (!A.r2_table)) {  // plink2.cc:11186-11191
    die(8, "Error: Matrix-only and table-only --r2-unphased settings cannot be used together.\n");
  }
  if (ld_snp_given && (A.ld_var_ct_radius != 0x7fffffff)) {
    // With a variant-count window the reference's row windows are irregular: its second pass restarts each chromosome's window
    // search at a position its first pass has already cleared, and FindNth1BitFrom (UpdateVcorWindow, plink2_ld.cc:10997-11001)
    // then keeps one variant more on the leading side for the rows that follow (snp101,snp103 with --ld-window 3 pairs snp103
    // with snp100).  Not reproduced.
    die(63, "Error: --ld-window together with --ld-snp/--ld-snps/--ld-snp-list is not supported by plink2-hip (--ld-window-kb is).\n");
  }
  if (ld_snp_given && (A.ld_cm_radius != -1.0)) {
    // (the leading side of a row variant's window keeps cm >= center - radius, the trailing side cm < center + radius,
    // UpdateVcorWindow :10991-10994 / :11010-11013: not the same band seen from the two ends)
    die(63, "Error: --ld-window-cm together with --ld-snp/--ld-snps/--ld-snp-list is not supported by plink2-hip.\n");
  }
  if (ld_snp_given && (A.parallel_tot != 1)) {
    die(63, "Error: --parallel with --ld-snp/--ld-snps/--ld-snp-list is not supported by plink2-hip yet.\n");
  }
  if ((ld_window_given || A.ld_min_r2 != 2.0) && !A.have_r2) {
    die(8, "Error: --ld-window.../--ld-snp... must be used with --r[2]-[un]phased.\n");  // plink2.cc:12960
  }
  if (A.have_r2 && ((!A.r2_table) || A.r2_inter)) {
    if (ld_window_given) {  // plink2.cc:11175-11179
      die(8, "Error: All-pairs --r2-unphased settings cannot be used with --ld-window/--ld-window-kb/--ld-window-cm.\n");
    }
  }
  if (A.have_r2 && !A.r2_table) {
    if (A.ld_min_r2 != 2.0) {  // plink2.cc:11186-11191
      die(8, "Error: Matrix-only and table-only --r2-unphased settings cannot be used together.\n");
    }
  }
  if (A.r2_table) {  // table defaults, plink2.cc:11181-11205
    if (A.ld_bp_radius == 0xffffffffu) {
      A.ld_bp_radius = 1000000;
    }
    if (A.ld_min_r2 == 2.0) {
      A.ld_min_r2 = 0.2 * (1 - kSmallEpsilon);
    }
  }
  if (A.gpus < 1) {
    die(8, "Error: --gpus must be positive.\n");
  }
}

Args parse_args(int argc, char** argv) {
  Args A;
  ArgCursor c{argc, argv, 1};
  for (; c.i < argc; ++c.i) {
    const std::string f = argv[c.i];
    if (!(parse_input_flags(A, c, f) || parse_command_flags(A, c, f) || parse_clump_flags(A, c, f) || parse_filter_flags(A, c, f) || parse_ldwindow_flags(A, c, f) ||
          parse_misc_flags(A, c, f))) {
      die(8, "Error: Unrecognized flag ('%s').  plink2-hip implements the --indep-pairwise path only.\n", f.c_str());
    }
  }
  check_flag_combinations(A);
  return A;
}


}  // namespace p2h
