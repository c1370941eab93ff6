// p2h_cli.h -- what the translation units of `plink2-hip` share (p2h_*.cpp, plink2_hip_cli.cpp): the parsed command line, the variant
// and sample tables, the session, the output stream and the helpers one unit calls in another.  Not a public interface.
#ifndef P2H_CLI_H
#define P2H_CLI_H
#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <fstream>
#include <functional>
#include <sstream>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/ldprune_hip.h"
#include "../../include/ldprune_hip_debug.h"  // (only for the --debug-* test hooks)

namespace p2h {

constexpr double kSmallEpsilon = 0.00000000000005684341886080801486968994140625;  // 2^-44
extern FILE* g_log;
extern bool g_silent;       // --silent
extern bool g_r_unsquared;  // --r-unphased

double now_s();
void logprintf(const char* fmt, ...);
[[noreturn]] void die(int code, const char* fmt, ...);
bool scan_double_plink(const char* s, double* out, const char** endp);
uint32_t banker_round(double v);  // v >= 0, < 2^31
char* put_digits_trimmed(uint32_t u, int digits, int keep, char* out);
char* format_g6(double x, char* out);

// Output file, optionally Zstandard-compressed ('zs': <name>.zst, as the reference's compress stream writes it;
// plink2_compress_stream.cc).  libzstd.so.1 is bound by hand (no zstd headers in the image); the default compression
// level (3) is used, so the bytes may differ from the reference's file while the decompressed text is identical.
class OutFile {
 public:
  void open(const std::string& path, bool zs) {
    path_ = path;
    f_ = fopen(path.c_str(), "wb");
    if (!f_) {
      die(3, "Error: Failed to open %s for writing.\n", path.c_str());
    }
    if (!zs) {
      return;
    }
    void* lib = dlopen("libzstd.so.1", RTLD_NOW);
    if (!lib) {
      die(63, "Error: 'zs' output needs libzstd.so.1, which could not be loaded.\n");
    }
    create_ = reinterpret_cast<void* (*)()>(dlsym(lib, "ZSTD_createCCtx"));
    destroy_ = reinterpret_cast<size_t (*)(void*)>(dlsym(lib, "ZSTD_freeCCtx"));
    step_ = reinterpret_cast<size_t (*)(void*, Buf*, CBuf*, int)>(dlsym(lib, "ZSTD_compressStream2"));
    is_error_ = reinterpret_cast<unsigned (*)(size_t)>(dlsym(lib, "ZSTD_isError"));
    if (!create_ || !destroy_ || !step_ || !is_error_) {
      die(63, "Error: libzstd.so.1 lacks the streaming compression API.\n");
    }
    ctx_ = create_();
    if (!ctx_) {
      die(2, "Error: Out of memory.\n");
    }
    obuf_.resize(1 << 20);
  }
  void write(const void* p, size_t n) {
    if (!ctx_) {
      if (n && (fwrite(p, 1, n, f_) != n)) {
        die(5, "Error: File write failure: %s.\n", path_.c_str());
      }
      return;
    }
    pump(p, n, 0);
  }
  void close() {
    if (ctx_) {
      pump(nullptr, 0, 2);  // ZSTD_e_end
      destroy_(ctx_);
      ctx_ = nullptr;
    }
    if (fclose(f_)) {
      die(5, "Error: File write failure: %s.\n", path_.c_str());
    }
    f_ = nullptr;
  }

 private:
  struct Buf {
    void* dst;
    size_t size, pos;
  };
  struct CBuf {
    const void* src;
    size_t size, pos;
  };
  void pump(const void* p, size_t n, int end_op) {
    CBuf in{p, n, 0};
    while (true) {
      Buf out{obuf_.data(), obuf_.size(), 0};
      const size_t left = step_(ctx_, &out, &in, end_op);
      if (is_error_(left)) {
        die(5, "Error: zstd compression failure: %s.\n", path_.c_str());
      }
      if (out.pos && (fwrite(obuf_.data(), 1, out.pos, f_) != out.pos)) {
        die(5, "Error: File write failure: %s.\n", path_.c_str());
      }
      if (end_op ? (left == 0) : (in.pos == in.size)) {
        break;
      }
    }
  }
  std::string path_;
  FILE* f_ = nullptr;
  void* ctx_ = nullptr;
  void* (*create_)() = nullptr;
  size_t (*destroy_)(void*) = nullptr;
  size_t (*step_)(void*, Buf*, CBuf*, int) = nullptr;
  unsigned (*is_error_)(size_t) = nullptr;
  std::vector<uint8_t> obuf_;
};

struct Args {
  std::string bed, bim, fam, pgen, pgi, pvar, psam, out = "plink2";
  bool have_prune = false;
  bool pairphase = false;  // --indep-pairphase instead of --indep-pairwise
  uint32_t window = 0, step = 1;
  bool window_is_bp = false;
  double r2 = 0.0;
  int order = 2;
  bool bad_ld = false;
  bool allow_extra_chr = false;
  std::string preferred;
  int gpus = 1;
  bool have_r2 = false;
  uint32_t parallel_idx = 0, parallel_tot = 1;  // --parallel k n (0-based index inside, plink2.cc:10109-10117)
  int r2_shape = -1;      // 0 square, 1 square0, 2 triangle
  int r2_float = -1;      // 1 bin4, 0 bin
  bool yes_really = false;
  bool r2_table = false;   // --r2-unphased without a matrix shape: windowed .vcor table
  bool r2_ref_based = false;
  bool r2_allow_ambiguous = false;
  bool r_unsquared = false;        // --r-unphased: r = +-sqrt(r^2) with the sign of the covariance
  uint32_t r2_cols = 0;            // kVcorCol* (set after the modifiers are read: plink2.cc:11158-11207)
  std::string r2_cols_desc;        // the text behind cols=
  bool r2_cols_given = false;
  bool r2_zs = false;      // 'zs': Zstandard-compressed table / text matrix
  bool r2_inter = false;   // 'inter-chr': the table over ALL pairs, chromosome 0 included (plink2_ld.cc:11082-11116)
  bool r2_text = false;    // matrix shape without bin/bin4: text matrix
  uint32_t ld_var_ct_radius = 0x7fffffff;  // --ld-window N: N - 1
  uint32_t ld_bp_radius = 0xffffffffu;     // --ld-window-kb; UINT32_MAX = not given (table default 1000 kb)
  double ld_cm_radius = -1.0;              // --ld-window-cm; -1 = not given
  double ld_min_r2 = 2.0;                  // --ld-window-r2 (after the reference's epsilon); 2.0 = not given
  // variant / sample filters applied before the command (the reference's variant_include / sample_include):
  // --chr / --not-chr (codes and code ranges, or names), --autosome, --extract / --exclude (variant ID lists),
  // --keep / --remove (sample ID lists: "FID IID", "IID", or a #FID / #IID header line)
  std::vector<std::string> chr_keep, chr_drop;
  bool autosome = false;
  // --maf / --max-maf (nonmajor-allele frequency over the founders) and --geno (missing-call rate over the samples), as the
  // reference enforces them (EnforceFreqConstraints plink2_filter.cc:3791, EnforceGenoThresh :3498); 0 / 1 / 1 = not given
  double min_maf = 0.0, max_maf = 1.0, geno = 1.0;
  uint64_t min_allele_ddosage = 0, max_allele_ddosage = ~0ull;  // --mac / --max-mac in 32768ths of an allele copy (plink2.cc:8785-8867)
  bool ac_founders = false;
  uint32_t max_alleles = 0xffffffffu;  // --max-alleles N (applied while the variant table loads, LoadPvar)
  bool snps_only = false, snps_only_acgt = false;  // --snps-only ['just-acgt'] (another load-time filter)
  std::vector<std::string> extract_files, exclude_files, keep_files, remove_files;
  // --ld-snp / --ld-snps / --ld-snp-list (plink2.cc:7966-8003): the table's row variants.  ld_snps: (first, second) ID pairs,
  // second empty for a single ID, otherwise the range first..second in file order
  std::vector<std::pair<std::string, std::string>> ld_snps;
  std::string ld_snp_list;
  // --clump (InitClump, plink2_ld.cc:62-78; parsing plink2.cc:4960-5120)
  bool have_clump = false;
  std::vector<std::string> clump_files;  // one or more reports (plink2.cc:4861-4958: comma- or space-separated)
  bool clump_unphased = false;
  bool clump_allow_overlap = false;
  bool clump_no_test = false;
  std::vector<std::string> clump_id_field, clump_p_field, clump_test_field, clump_test;
  bool make_founders = false, make_founders_require2 = false, make_founders_first = false;  // --make-founders ['require-2-missing'] ['first']
  bool clump_zs = false;
  bool clump_force_a1 = false, clump_no_a1 = false;  // --clump-force-a1; --clump-a1-field without names
  std::vector<std::string> clump_a1_field;
  std::string clump_range;                 // --clump-range / --clump-range0 <file>: regions to report overlaps with
  bool clump_range0 = false;
  uint32_t clump_range_border = 0;         // --clump-range-border <kb>, in bp
  bool clump_range_border_given = false;
  bool clump_in_log10 = false, clump_out_log10 = false;  // --clump-log10 ['input-only' | 'output-only']
  bool clump_log10_p1 = false, clump_log10_p2 = false, clump_plain_p1 = false, clump_plain_p2 = false;
  uint32_t clump_cols = 0;                 // kClumpCol* (plink2_ld.h:51-67), set after the modifiers are read
  std::string clump_cols_desc;
  bool clump_cols_given = false;
  std::vector<double> clump_ln_bins;       // --clump-bins: ln of the boundaries, each times (1 + 2^-44); empty: the default four
  double clump_ln_p1 = 2.3025850929940457 * -4.0 * (1.0 - kSmallEpsilon);
  double clump_ln_p2 = 2.3025850929940457 * -2.0 * (1.0 - kSmallEpsilon);
  double clump_r2_raw = 0.5;
  double clump_r2 = 0.5 * (1.0 + kSmallEpsilon);
  uint32_t clump_bp_radius = 249999;
  bool timing = false;    // --timing: print per-phase wall times
  bool dry_run = false;  // parse + plan only, print the parameters exactly (%a) and exit: used by the CPU tests
};

std::vector<std::string> split_ws(const std::string& line);
bool ieq(const char* a, const char* b);
const char* scan_ln(const char* s, double* ln_out);  // (--clump section below)

// ---- --r2-unphased cols= (plink2_ld.h:87-101, ParseColDescriptor plink2_cmdline.cc:4375) ----
enum : uint32_t {
  kVcorColChrom = 1u << 0, kVcorColPos = 1u << 1, kVcorColId = 1u << 2, kVcorColRef = 1u << 3, kVcorColAlt1 = 1u << 4, kVcorColAlt = 1u << 5,
  kVcorColMaybeprovref = 1u << 6, kVcorColProvref = 1u << 7, kVcorColMaj = 1u << 8, kVcorColNonmaj = 1u << 9, kVcorColFreq = 1u << 10,
  kVcorColD = 1u << 11, kVcorColDprime = 1u << 12, kVcorColDprimeAbs = 1u << 13,
  kVcorColDefault = kVcorColChrom | kVcorColPos | kVcorColId | kVcorColMaybeprovref
};

enum : uint32_t {
  kClumpColChrom = 1u << 0, kClumpColPos = 1u << 1, kClumpColRef = 1u << 2, kClumpColAlt1 = 1u << 3, kClumpColAlt = 1u << 4,
  kClumpColMaybeprovref = 1u << 5, kClumpColProvref = 1u << 6, kClumpColMaybeA1 = 1u << 7, kClumpColA1 = 1u << 8, kClumpColMaybeF = 1u << 9,
  kClumpColF = 1u << 10, kClumpColTotal = 1u << 11, kClumpColMaybeBounds = 1u << 12, kClumpColBounds = 1u << 13, kClumpColBins = 1u << 14,
  kClumpColSp2 = 1u << 15,
  kClumpColDefault = kClumpColChrom | kClumpColPos | kClumpColMaybeprovref | kClumpColMaybeA1 | kClumpColMaybeF | kClumpColTotal | kClumpColMaybeBounds |
                     kClumpColBins | kClumpColSp2
};

// test / measurement hooks of the front-end (hidden --debug-* flags; the library itself reads no environment, csrc/ldp_env.h)
struct DebugHooks {
  bool alias_devices = false, x_host = false, host_decode = false, load_map = false, no_bind = false, serial_feed = false;
  uint32_t x_rows = 0, decode_threads = 0;
};
extern DebugHooks g_dbg;

Args parse_args(int argc, char** argv);
void load_samples(const Args& A, std::vector<uint8_t>* is_founder, std::vector<uint8_t>* sex, std::vector<std::string>* fid_iid = nullptr,
                  std::vector<std::pair<std::string, std::string>>* parents = nullptr);

struct Variants {
  std::vector<std::string> chrom, id;
  std::vector<uint32_t> bp;
  std::vector<uint8_t> alt_ct;  // number of ALT alleles (1 for biallelic / .bim), capped at 255
  std::vector<uint8_t> alt_missing;  // --max-alleles: the single ALT is a missing code ('.' or '0'), which counts as ONE allele (plink2_pvar.cc:1940-1948); empty unless the filter is on
  std::vector<uint8_t> not_snp; // --snps-only: an allele longer than one character (or, with 'just-acgt', outside ACGT / missing)
  std::vector<std::string> ref, alt;  // allele text (ALT comma-separated as in the file); only kept for --r2-unphased allele columns
  bool info_pr_header = false;        // the .pvar declares INFO/PR as a flag (provisional REF alleles are marked per variant there)
  std::vector<uint8_t> info_pr;       // bit v: variant v's INFO carries PR (PrInInfo, plink2_pvar.cc:561); kept when a REF column is printed
  std::vector<double> cm;             // centimorgan positions; only kept for --ld-window-cm (empty when the file has no CM column)
  bool cm_unsorted = false;           // some chromosome's CM values decrease (LoadPvar, plink2_pvar.cc:2121-2134)
  bool cm_any_nonzero = false;
};

inline uint32_t allele_ct_for_filter(const Variants& V, size_t v) {
  return ((v < V.alt_missing.size()) && V.alt_missing[v]) ? 1u : (static_cast<uint32_t>(V.alt_ct[v]) + 1);
}

std::string slurp(const std::string& path);
// This thread (and every thread it starts from here on) onto the CPUs of the NUMA node the device is attached to; a no-op where the host does not say
// or the node's CPUs are not ours to use.  Returns the node, or -1.
struct AffinityMask {
  bool valid = false;
  unsigned long bits[16] = {0};  // a cpu_set_t (1,024 CPUs)
};
int bind_near_device(int device, AffinityMask* saved = nullptr);
void restore_affinity(const AffinityMask& saved);
void load_variants(const Args& A, Variants* V);
int chrom_class(const std::string& name_in, bool allow_extra, bool* is_zero);
void multiallelic_inverse_row(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const std::vector<uint32_t>& founder_idx,
                              std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* out_row, uint64_t out_rec, double* maj_freq,
                              uint8_t* phase = nullptr, uint64_t phase_bytes = 0, bool* unphased = nullptr, uint32_t* maj_idx = nullptr);

// ---- chrX / chrY / MT ---------------------------------------------------------------------------------------
// The reference feeds IndepPairwiseThread differently shaped genotype vectors on these chromosomes
// (plink2_ld.cc:1356-1388): MT/haploid = founders with hets set to missing; chrY = non-female founders, hets to
// missing; chrX = male founders (hets to missing) followed by non-male founders, whose statistics count twice
// (:890-901, :1066-1082).  All of the prune's statistics are sums over samples, so "count twice" is reproduced
// exactly by emitting the non-male block twice; the engine then runs unchanged on founder_ct' samples.
// Allele frequencies follow LoadAlleleAndGenoCountsThread (plink2_data.cc:2421-2700): diploid-style counts over the
// relevant founders (hets count half/half) for MT and chrY, and for chrX non-males weigh twice as much as males
// with a male het counting half (alt = 4*G2 + 2*G1 - 2*M2 - M1 over all / male founders, :2641).
struct SexPlan {
  std::vector<uint32_t> part1;  // samples whose hets become missing
  std::vector<uint32_t> part2;  // chrX non-males (emitted twice); empty otherwise
  bool x_freq = false;
  uint32_t out_ct() const { return static_cast<uint32_t>(part1.size() + 2 * part2.size()); }
};

inline uint32_t code_at(const uint8_t* row, uint32_t s) { return (row[s >> 2] >> (2 * (s & 3))) & 3; }
void build_sex_row(const SexPlan& sp, const uint8_t* raw_row, uint8_t* out_row, uint64_t out_rec, double* maj_freq, const uint8_t* phase = nullptr);
void multiallelic_sex_row(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* out_row,
                          uint64_t out_rec, double* maj_freq);
uint32_t sex_major_allele(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, double* maj_freq);
void multiallelic_sex_row_phased(ldp_pgen* pg, uint32_t raw_variant, uint32_t alt_ct, const SexPlan& sp, std::vector<uint8_t>* lo, std::vector<uint8_t>* hi, uint8_t* phase,
                                 uint64_t phase_bytes, uint8_t* out_row, uint64_t out_rec, double* maj_freq, bool* unphased);
void fetch_raw_row(ldp_pgen* pg, int storage_mode, uint32_t raw_variant, uint32_t raw_sample_ct, uint64_t rec_bytes, uint8_t* buf);

// feed(engine, raw file indices in engine order): the caller's genotype-row feeder
// ---- chrX pairs of the r^2 outputs and --clump (ComputeXR2, plink2_ld.cc:7122-7190) ----
// A pair with a chrX variant weighs the male founders down in all six sums -- by 1/2 when both variants are on chrX, by
// 1 - sqrt(2)/2 when one is -- before the same quotient.  Two integer 6-tuples per pair (all founders from `all`, male
// founders from `male`, an engine fed the same rows through a sample map; ldp_pair_stats) are turned from the engines' +-1
// coding and orientation into the reference's counts of the non-major (non-REF) allele -- exactly, in integers -- and
// then the reference's doubles follow, fma for fma (the documented AVX2 build defines FP_FAST_FMA).  Inside chrX the weight
// is dyadic and every sum exact, so WHICH orientation is the target only matters for pairs with an autosome -- but the two
// tuples of a pair must agree on one (each engine picks its major alleles from its own samples).
struct XWeighted {
  ldp_engine* all = nullptr;
  ldp_engine* male = nullptr;             // nullptr: no male founders
  std::vector<uint8_t> is_x;               // per engine row
  std::vector<uint8_t> flip_all, flip_male;  // per engine row: the engine's orientation differs from the target's
  bool unsquared = false;
  struct G {
    int64_t n, g1, q1, g2, q2, d;
  };
  static G counts(const ldp_pair_stats_t& t, bool flip1, bool flip2) {
    G c;
    c.n = t.nm;
    c.g1 = c.n - t.sum1;
    c.q1 = c.n - 2 * static_cast<int64_t>(t.sum1) + t.ssq1;
    c.g2 = c.n - t.sum2;
    c.q2 = c.n - 2 * static_cast<int64_t>(t.sum2) + t.ssq2;
    c.d = c.n - t.sum1 - t.sum2 + t.dot;
    if (flip1) {  // g -> 2 - g
      c.q1 = 4 * c.n - 4 * c.g1 + c.q1;
      c.g1 = 2 * c.n - c.g1;
      c.d = 2 * c.g2 - c.d;
    }
    if (flip2) {
      c.q2 = 4 * c.n - 4 * c.g2 + c.q2;
      c.g2 = 2 * c.n - c.g2;
      c.d = 2 * c.g1 - c.d;
    }
    return c;
  }
  // the reference's doubles from the two count tuples (ComputeXR2 :7160-7185)
  static double weighted(const G& a, const G& m, bool both_x, bool unsquared, double nan_ref) {
    if (!a.n) {
      return nan_ref;
    }
    const double male_downwt = both_x ? 0.5 : (1.0 - 0.5 * 1.4142135623730951);
    const double w_obs = fma(-male_downwt, static_cast<double>(m.n), static_cast<double>(a.n));
    const double w_g1 = fma(-male_downwt, static_cast<double>(m.g1), static_cast<double>(a.g1));
    const double w_g2 = fma(-male_downwt, static_cast<double>(m.g2), static_cast<double>(a.g2));
    const double w_q1 = fma(-male_downwt, static_cast<double>(m.q1), static_cast<double>(a.q1));
    const double w_q2 = fma(-male_downwt, static_cast<double>(m.q2), static_cast<double>(a.q2));
    const double w_d = fma(-male_downwt, static_cast<double>(m.d), static_cast<double>(a.d));
    const double var1 = fma(w_q1, w_obs, -w_g1 * w_g1);
    const double var2 = fma(w_q2, w_obs, -w_g2 * w_g2);
    if (!((var1 > 0.0) && (var2 > 0.0))) {
      return nan_ref;
    }
    const double var_prod = var1 * var2;
    const double cov = fma(w_d, w_obs, -w_g1 * w_g2);
    const double quot = cov * cov / var_prod;
    double r = (1.0 < quot) ? 1.0 : quot;
    if (unsquared) {
      r = sqrt(r);
      if (cov < 0.0) {
        r = -r;
      }
    }
    return r;
  }
  // ---- windowed plans (ldp_set_variants_vcor; --clump): the chrX run [band_first, band_first + band_ct) of the caller's rows, loaded a second time into two
  // ALL-PAIRS engines that hold nothing else (all founders / male founders).  The pairs of a window on chrX then come from the pair kernels and the
  // device-side weighting (ldp_r2_unphased_block_x*: ComputeXR2 as ClumpHighmemR2 and the windowed table call it, plink2_ld.cc:7122-7190, :7343),
  // a chunk of rows at a time, instead of a list of pairs with one wave each.  (A windowed engine has no dense-block plan of its own.)
  ldp_engine* band_all = nullptr;
  ldp_engine* band_male = nullptr;  // nullptr: no male founders
  uint32_t band_first = 0, band_ct = 0;
  std::vector<uint8_t> band_flip_all, band_flip_male;  // per row of the run (empty: none)
  bool band_ready() const { return band_all != nullptr; }
  void band_destroy() {
    if (band_male) {
      ldp_destroy(band_male);
    }
    if (band_all) {
      ldp_destroy(band_all);
    }
    band_all = band_male = nullptr;
  }
  // rows [j, j + R) of the run and the columns their windows reach: R about the window's width, so that the rectangle is about twice the band
  void band_chunk(const uint32_t* lo, uint32_t j, uint32_t j1, uint32_t max_rows, uint32_t* rows, uint32_t* c0) const {
    *c0 = std::max(lo[j], band_first);
    const uint32_t width = j - *c0 + 1;
    *rows = std::min(std::min(j1 - j, max_rows), std::max(512u, std::min(width, 8192u)));
  }
  // f(i, j, value) for every pair lo[j] <= i < j of the rows j in [j0, j1) (inside the run); NaN where the reference's value is undefined
  template <class F>
  void band_dense(const uint32_t* lo, uint32_t j0, uint32_t j1, F f) const {
    const std::vector<uint8_t> ones(band_ct, 1);
    std::vector<double> blk;
    for (uint32_t j = j0; j < j1;) {
      uint32_t rows, c0;
      band_chunk(lo, j, j1, 0xffffffffu, &rows, &c0);
      const uint32_t cols = j + rows - 1 - c0;
      if (cols) {
        blk.assign(static_cast<size_t>(rows) * cols, 0.0);
        if (ldp_r2_unphased_block_x(band_all, band_male, ones.data(), band_flip_all.empty() ? nullptr : band_flip_all.data(), band_flip_male.empty() ? nullptr : band_flip_male.data(),
                                    j - band_first, rows, c0 - band_first, cols, 0, unsquared ? 1 : 0, blk.data(), cols)) {
          die(16, "Error: %s\n", ldp_last_error(band_all));
        }
        for (uint32_t q = 0; q < rows; ++q) {
          for (uint32_t i = std::max(lo[j + q], c0); i < j + q; ++i) {
            f(i, j + q, blk[static_cast<size_t>(q) * cols + (i - c0)]);
          }
        }
      }
      j += rows;
    }
  }
  // ... those with |value| >= min_r2 (NaN never passes), in no particular order
  template <class F>
  void band_hits(const uint32_t* lo, uint32_t j0, uint32_t j1, double min_r2, std::vector<ldp_r2_hit>* buf, F f) const {
    const std::vector<uint8_t> ones(band_ct, 1);
    if (buf->size() < (1u << 20)) {
      buf->resize(1u << 22);
    }
    uint32_t max_rows = 0xffffffffu;
    for (uint32_t j = j0; j < j1;) {
      uint32_t rows, c0;
      band_chunk(lo, j, j1, max_rows, &rows, &c0);
      const uint32_t cols = j + rows - 1 - c0;
      uint64_t found = 0;
      if (cols && ldp_r2_unphased_block_x_hits(band_all, band_male, ones.data(), band_flip_all.empty() ? nullptr : band_flip_all.data(),
                                               band_flip_male.empty() ? nullptr : band_flip_male.data(), j - band_first, rows, c0 - band_first, cols, unsquared ? 1 : 0, min_r2,
                                               buf->data(), buf->size(), &found)) {
        die(16, "Error: %s\n", ldp_last_error(band_all));
      }
      if (found > buf->size()) {  // more passing pairs than the buffer holds: fewer rows per call
        if (rows == 1) {
          die(2, "Error: one chrX variant has more passing partners than the filter buffer holds.\n");
        }
        max_rows = std::max(1u, rows / 2);
        continue;
      }
      for (uint64_t q = 0; q < found; ++q) {
        const uint32_t i = (*buf)[q].first + band_first, jj = (*buf)[q].second + band_first;
        if ((i >= lo[jj]) && (i < jj)) {
          f(i, jj, (*buf)[q].r2);
        }
      }
      j += rows;
    }
  }
  // r^2 (or r) of the listed pairs, each with at least one chrX variant; NaN where the reference's is undefined
  void pairs(const std::vector<uint32_t>& first, const std::vector<uint32_t>& second, std::vector<double>* out) const {
    const size_t n = first.size();
    out->resize(n);
    std::vector<ldp_pair_stats_t> ta, tm;
    double nan_ref;  // (the bits the reference's `0.0 / 0.0` has on x86: sign set)
    {
      const uint64_t bits = 0xfff8000000000000ull;
      memcpy(&nan_ref, &bits, 8);
    }
    for (size_t p0 = 0; p0 < n; p0 += (1u << 21)) {
      const uint32_t cnt = static_cast<uint32_t>(std::min<size_t>(n - p0, 1u << 21));
      ta.resize(cnt);
      tm.assign(cnt, ldp_pair_stats_t{0, 0, 0, 0, 0, 0});
      if (ldp_pair_stats(all, cnt, first.data() + p0, second.data() + p0, ta.data())) {
        die(16, "Error: %s\n", ldp_last_error(all));
      }
      if (male && ldp_pair_stats(male, cnt, first.data() + p0, second.data() + p0, tm.data())) {
        die(16, "Error: %s\n", ldp_last_error(male));
      }
      for (uint32_t q = 0; q < cnt; ++q) {
        const uint32_t i = first[p0 + q], j = second[p0 + q];
        const G a = counts(ta[q], (!flip_all.empty()) && flip_all[i], (!flip_all.empty()) && flip_all[j]);
        const G m = male ? counts(tm[q], (!flip_male.empty()) && flip_male[i], (!flip_male.empty()) && flip_male[j]) : G{0, 0, 0, 0, 0, 0};
        const double r = weighted(a, m, is_x[i] && is_x[j], unsquared, nan_ref);
        (*out)[p0 + q] = r;
      }
    }
  }
};

int chrom_code(const std::string& name_in);
// what --clump needs to know about sex chromosomes (ClumpReports :8150-8215, :8460-8482)
struct ClumpSex {
  const std::vector<uint8_t>* vcls = nullptr;  // per included variant: 3 chrX, 4 chrY
  std::vector<uint32_t> male_cols;             // raw sample indices of the male founders
  uint32_t founder_male_ct = 0, founder_female_ct = 0, founder_nosex_ct = 0;
  int prov_storage = 1;                        // ldp_pgen_provisional_ref
  std::vector<uint8_t> prov_bits;
  std::function<void(ldp_engine*, const std::vector<uint32_t>&, const std::vector<uint32_t>*)> feed_cols;
  uint32_t raw_sample_ct = 0;
  // reloads engine row `row` from raw variant `raw`: the main track (aidx < 0) or the copies-of-the-other-alleles row of allele `aidx` of a
  // multiallelic variant; females_missing: the female founders' calls become missing (chrY); mapped: the engine picks its sample columns itself
  std::function<void(ldp_engine*, uint32_t row, uint32_t raw, int32_t aidx, bool females_missing, bool mapped)> allele_row;
};

int clump_reports(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, const std::vector<uint32_t>& chr_idx,
                  const std::vector<uint32_t>& bps, uint32_t founder_ct,
                  const std::function<void(ldp_engine*, const std::vector<uint32_t>&)>& feed, const ClumpSex& SX);

// Everything the commands share: the parsed command line, the variant and sample tables, the open genotype file and the
// included-variant index (chromosome 0 stripped where the reference strips it).  load_inputs() fills it; run_r2() (the
// --r2-unphased outputs and --clump) and run_prune() (--indep-pairwise / --indep-pairphase) consume it.
struct Session {
  double t_begin = 0.0, t_hip_init = 0.0, t_parse = 0.0, t_joined = 0.0;
  Args A;
  Variants V;
  std::thread t_hip;  // HIP runtime start-up, beside the file parsing; joined where the first engine is created, or on the way out
  std::vector<uint8_t> is_founder, sex;  // (a sample --keep / --remove drops is no founder from here on)
  std::vector<uint8_t> sample_kept;      // empty: no sample filter
  uint32_t raw_sample_ct = 0, founder_ct = 0, raw_variant_ct = 0;
  bool is_bed = false;
  std::string gpath;
  ldp_pgen* pg = nullptr;
  int storage_mode = 0, encoding = LDP_GENO_REF, has_multiallelic = 0;
  bool has_dosage = false;  // some record carries a dosage track: --indep-pairwise takes the allele frequencies from them
  // founders' (ref, alt) dosage sums of the variants that have a dosage track (ldp_pgen_dosage_sums), computed once: the
  // frequency filters and the prune's tie-break both want them
  std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> dosage_sums;
  void need_dosage_sums(const std::vector<uint32_t>& raw_variants);
  uint64_t rec_bytes = 0;
  const uint8_t* direct_rows = nullptr;  // NULL for variable-width files
  std::vector<uint32_t> inc;             // raw index of every included variant
  std::vector<uint32_t> chr_idx, bps;
  std::vector<uint8_t> vcls;             // per included variant: 0 diploid, 3 chrX, 4 chrY, 5 MT
  uint32_t variant_ct = 0;
  std::vector<uint32_t> mk, xk, yk, tk;  // indices into inc[]: main engine, chrX, chrY, MT under --indep-pairphase
  uint32_t m_ct = 0;
  std::vector<uint32_t> m_chr, m_bps;
  // (... and, once the runtime is up, this thread -- the one that creates the engines, their copy threads and their pinned staging -- moves next to the device)
  void join_hip() {
    if (t_hip.joinable()) {
      t_hip.join();
      t_joined = now_s();
      if ((A.gpus <= 1) && !g_dbg.no_bind) {
        bind_near_device(0, &load_affinity);  // (several GPUs: every engine's feeding thread binds itself next to ITS device, p2h_prune.cpp)
      }
    }
  }
  AffinityMask load_affinity;  // what this thread ran on before join_hip() moved it next to device 0: file_to_hbm_done() puts it back
  void file_to_hbm_done() {
    restore_affinity(load_affinity);
    load_affinity.valid = false;
  }
  ~Session() {
    if (t_hip.joinable()) {
      t_hip.join();
    }
  }
};

void load_inputs(Session& S, int argc, char** argv);
int run_r2(Session& S);
int run_prune(Session& S);

}  // namespace p2h
#endif
