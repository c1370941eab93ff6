// plink2_hip_cli.cpp -- `plink2-hip`: process-level drop-in for the --indep-pairwise path of plink2.
//
// Same flag spellings (2.0/plink2.cc:7238-7337, 2.0/plink2_help.cc:948-969), same inputs
// (.bed/.bim/.fam, fixed- and variable-width .pgen/.pvar/.psam) and byte-identical <out>.prune.in / <out>.prune.out
// (LdPruneWrite, 2.0/plink2_ld.cc:2464-2528).  Everything between "files are open" and "bitmap of removed
// variants" goes through the C ABI in include/ldprune_hip.h, i.e. through the HIP kernels; there is no CPU
// compute path here, so without a usable GPU the program exits with an error.
//
// What this front-end does itself (host C++): argument parsing incl. the reference's decimal scanner
// (ScanadvDouble, 2.0/include/plink2_string.cc:1264-1528), .fam/.psam founder detection
// (2.0/plink2_psam.cc:804-813), .bim/.pvar parsing, chromosome-0 stripping (StripUnplacedK,
// plink2_ld.cc:113-164), the sorted-positions and unique-ID checks (plink2.cc:2926, plink2_ld.cc:2573-2592),
// the <50-founders guard (plink2.cc:2063-2071) and the output writer.
// Multiallelic variants are collapsed major-vs-rest on the host (Get1Multiallelic semantics); chrX / chrY / MT get
// their sample-mapped rows (males het->missing, non-males x2, ...) built on the host as well.
// --r2-unphased / --r-unphased: the matrix shapes (square/square0/triangle as bin, bin4 or text, zs), the windowed and the
// inter-chr .vcor table with cols=, --ld-window, --ld-window-kb, --ld-window-cm, --ld-window-r2, --ld-snp / --ld-snps / --ld-snp-list,
// --parallel; number formatting restated from dtoa_g.  --clump (several reports, --clump-allow-overlap, cols=, bins, -log10,
// ranges, sex chromosomes).
// Not yet supported (reported as such with exit 63, never silently mis-handled): dosage data outside --indep-pairwise on the autosomes,
// more than 254 ALT alleles, multiallelic sites on chrX/Y/MT and in --clump, major-allele-oriented r^2 outputs on chrY/MT.
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

namespace {

constexpr double kSmallEpsilon = 0.00000000000005684341886080801486968994140625;  // 2^-44
FILE* g_log = nullptr;
bool g_silent = false;       // --silent: the log file still gets every line, the terminal only errors
bool g_r_unsquared = false;  // --r-unphased: the messages below name that flag where they say --r2-unphased

// (--r-unphased shares every code path with --r2-unphased; the reference prints the flag actually given)
void name_the_flag(char* buf) {
  if (!g_r_unsquared) {
    return;
  }
  static const char kFrom[] = "--r2-unphased";
  for (char* p = strstr(buf, kFrom); p; p = strstr(p, kFrom)) {
    memmove(p + 3, p + 4, strlen(p + 4) + 1);  // "--r2-..." -> "--r-..."
  }
}

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

void logprintf(const char* fmt, ...) {
  char buf[4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  name_the_flag(buf);
  if (!g_silent) {
    fputs(buf, stdout);
  }
  if (g_log) {
    fputs(buf, g_log);
  }
}

[[noreturn]] void die(int code, const char* fmt, ...) {
  char buf[4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  name_the_flag(buf);
  fflush(stdout);
  fputs(buf, stderr);
  if (g_log) {
    fputs(buf, g_log);
    fclose(g_log);
  }
  // (not exit(): loader threads may still be running, and static destructors under their feet end in a crash instead of
  // the exit code)
  fflush(nullptr);
  _exit(code);
}

// The reference's decimal scanner: up to 16-17 significant digits accumulated in an int64, then ONE
// multiplication by a table power of ten -- so "0.3" parses as 3 * 0.1 = 0.30000000000000004, not as strtod
// would.  Returns false on malformed input.  (plink2_string.cc:1264-1528; exponents beyond the tables unsupported)
bool scan_double_plink(const char* s, double* out, const char** endp) {
  static const double kNegPow10[16] = {1.0, 1.0e-1, 1.0e-2, 1.0e-3, 1.0e-4, 1.0e-5, 1.0e-6, 1.0e-7, 1.0e-8, 1.0e-9, 1.0e-10, 1.0e-11, 1.0e-12, 1.0e-13, 1.0e-14, 1.0e-15};
  static const double kPosPow10[16] = {1.0, 1.0e1, 1.0e2, 1.0e3, 1.0e4, 1.0e5, 1.0e6, 1.0e7, 1.0e8, 1.0e9, 1.0e10, 1.0e11, 1.0e12, 1.0e13, 1.0e14, 1.0e15};
  const char* p = s;
  bool neg = false;
  if (*p == '-' || *p == '+') {
    neg = (*p == '-');
    ++p;
  }
  int64_t digits = 0;
  long e10 = 0;
  bool any = false;
  bool seen_dot = false;
  for (;; ++p) {
    if (*p >= '0' && *p <= '9') {
      any = true;
      if (digits < 10000000000000000LL) {
        digits = digits * 10 + (*p - '0');
        if (seen_dot) {
          --e10;
        }
      } else if (!seen_dot) {
        ++e10;
      }
    } else if (*p == '.' && !seen_dot) {
      seen_dot = true;
    } else {
      break;
    }
  }
  if (!any) {
    return false;
  }
  if (*p == 'e' || *p == 'E') {
    const char* q = p + 1;
    bool eneg = false;
    if (*q == '-' || *q == '+') {
      eneg = (*q == '-');
      ++q;
    }
    if (*q >= '0' && *q <= '9') {
      long ev = 0;
      while (*q >= '0' && *q <= '9' && ev < 100000) {
        ev = ev * 10 + (*q - '0');
        ++q;
      }
      e10 += eneg ? -ev : ev;
      p = q;
    }
  }
  double d = static_cast<double>(neg ? -digits : digits);
  if (digits && e10) {
    if (e10 < 0) {
      if (-e10 > 15) {
        long pe = -e10;
        d *= kNegPow10[pe & 15];
        for (pe /= 16; pe > 0; --pe) {
          d *= 1.0e-16;
        }
      } else {
        d *= kNegPow10[-e10];
      }
    } else {
      if (e10 > 15) {
        long pe = e10;
        d *= kPosPow10[pe & 15];
        for (pe /= 16; pe > 0; --pe) {
          d *= 1.0e16;
        }
      } else {
        d *= kPosPow10[e10];
      }
    }
  }
  *out = d;
  if (endp) {
    *endp = p;
  }
  return true;
}

// The reference's 6-significant-digit double formatter (dtoa_g, include/plink2_string.cc:2507-2639, with its
// BankerRoundD* helpers :2231-2295), restated: the value is scaled by the same sequence of powers of ten (each
// product rounds, so the sequence matters), rounded to six digits with ties-to-even inside a 5e-9 tolerance band,
// and printed without trailing zeros; exponent form below 1e-4 and from 1e6.
uint32_t banker_round(double v) {  // v >= 0, < 2^31
  static const double kTie[2] = {0.499999995, 0.500000005};
  const uint32_t t = static_cast<uint32_t>(static_cast<int32_t>(v));
  return t + static_cast<uint32_t>(static_cast<int32_t>((v - static_cast<double>(t)) + kTie[t & 1]));
}

// `digits` decimal digits of u, zero-padded, trailing zeros dropped (at least `keep` stay)
char* put_digits_trimmed(uint32_t u, int digits, int keep, char* out) {
  char buf[16];
  for (int k = digits - 1; k >= 0; --k) {
    buf[k] = static_cast<char>('0' + u % 10);
    u /= 10;
  }
  int n = digits;
  while (n > keep && buf[n - 1] == '0') {
    --n;
  }
  memcpy(out, buf, n);
  return out + n;
}

char* format_g6(double x, char* out) {
  if (x != x) {
    memcpy(out, "nan", 3);
    return out + 3;
  }
  if (x < 0) {
    *out++ = '-';
    x = -x;
  }
  if (x == 0.0) {
    *out++ = '0';
    return out;
  }
  static const int kExp[9] = {256, 128, 64, 32, 16, 8, 4, 2, 1};
  static const double kUp[9] = {1.0e256, 1.0e128, 1.0e64, 1.0e32, 1.0e16, 100000000, 10000, 100, 10};
  static const double kDown[9] = {1.0e-256, 1.0e-128, 1.0e-64, 1.0e-32, 1.0e-16, 1.0e-8, 1.0e-4, 1.0e-2, 1.0e-1};
  static const double kSmallBound[9] = {9.9999949999999e-256, 9.9999949999999e-128, 9.9999949999999e-64, 9.9999949999999e-32, 9.9999949999999e-16,
                                        9.9999949999999e-8,   9.9999949999999e-4,   9.9999949999999e-2,  9.9999949999999e-1};
  static const double kLargeBound[9] = {9.9999949999999e255, 9.9999949999999e127, 9.9999949999999e63, 9.9999949999999e31, 9.9999949999999e15,
                                        9.9999949999999e7,   9.9999949999999e3,   9.9999949999999e1,  9.9999949999999e0};
  const bool small = (x < 9.9999949999999e-5);
  if (small || (x >= 999999.49999999)) {
    if ((!small) && (x > 1.7976931348623157e308)) {
      memcpy(out, "inf", 3);
      return out + 3;
    }
    int xp10 = 0;
    for (int k = 0; k < 9; ++k) {
      if (small ? (x < kSmallBound[k]) : (x >= kLargeBound[k])) {
        x *= small ? kUp[k] : kDown[k];
        xp10 += kExp[k];
        if (k == 0) {
          ++k;  // (the reference takes either the 256 or the 128 step, never both)
        }
      }
    }
    const uint32_t t = banker_round(x * 100000);
    *out++ = static_cast<char>('0' + t / 100000);
    if (t % 100000) {
      *out++ = '.';
      out = put_digits_trimmed(t % 100000, 5, 1, out);
    }
    *out++ = 'e';
    *out++ = small ? '-' : '+';
    if (xp10 >= 100) {
      *out++ = static_cast<char>('0' + xp10 / 100);
      xp10 %= 100;
    }
    *out++ = static_cast<char>('0' + xp10 / 10);
    *out++ = static_cast<char>('0' + xp10 % 10);
    return out;
  }
  if (x >= 0.99999949999999) {
    // six significant digits of a number in [1, 1e6): the digits before the point, then what is left of the six
    int int_digits = 1;
    double bound = 9.9999949999999;
    while ((int_digits < 6) && (x >= bound)) {
      ++int_digits;
      bound = (int_digits == 2) ? 99.999949999999 : ((int_digits == 3) ? 999.99949999999 : ((int_digits == 4) ? 9999.9949999999 : 99999.949999999));
    }
    static const double kScale[7] = {0, 100000, 10000, 1000, 100, 10, 1};
    static const uint32_t kDiv[7] = {0, 100000, 10000, 1000, 100, 10, 1};
    const uint32_t t = banker_round(x * kScale[int_digits]);
    const uint32_t q = t / kDiv[int_digits], r = t % kDiv[int_digits];
    char tmp[16];
    const int n = snprintf(tmp, sizeof(tmp), "%u", q);
    memcpy(out, tmp, n);
    out += n;
    if (r) {
      *out++ = '.';
      out = put_digits_trimmed(r, 6 - int_digits, 1, out);
    }
    return out;
  }
  // [~1e-4, 1): "0." + leading zeros + six significant digits
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

// test / measurement hooks of the front-end (hidden --debug-* flags; the library itself reads no environment, csrc/ldp_env.h)
struct DebugHooks {
  bool alias_devices = false, x_host = false, host_decode = false, load_map = false;
  uint32_t x_rows = 0, decode_threads = 0;
} g_dbg;

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

// founder <=> PAT and MAT are both exactly "0" (plink2_psam.cc:804-806); absent columns => founder
// sex: 1 = male, 2 = female, anything else = unknown (plink2_psam.cc:808-813)
void load_samples(const Args& A, std::vector<uint8_t>* is_founder, std::vector<uint8_t>* sex, std::vector<std::string>* fid_iid = nullptr,
                  std::vector<std::pair<std::string, std::string>>* parents = nullptr) {
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

// allele count as --max-alleles sees it (LoadPvar, plink2_pvar.cc:1937-1953): a lone ALT that is a missing code counts as one allele
inline uint32_t allele_ct_for_filter(const Variants& V, size_t v) {
  return ((v < V.alt_missing.size()) && V.alt_missing[v]) ? 1u : (static_cast<uint32_t>(V.alt_ct[v]) + 1);
}

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
                            (A.have_clump && ((A.clump_cols & (kClumpColRef | kClumpColAlt1 | kClumpColAlt)) || A.clump_force_a1));
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
                              uint8_t* phase = nullptr, uint64_t phase_bytes = 0, bool* unphased = nullptr, uint32_t* maj_idx = nullptr) {
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
  if (maj_idx) {
    *maj_idx = maj;
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

// raw_row: REF-based pgen codes of all samples.  Writes the PgrGetInv1-style row (+ het->missing) and maj_freq.
// phase != nullptr (--indep-pairphase on chrX, plink2_ld.cc:2060-2097): the non-male founders contribute their two
// haplotypes, split by the phaseinfo bits of all samples (HapsplitMustPhased), instead of their genotype twice; a
// haplotype h is carried as the genotype code 2h (include/ldprune_hip.h, LDP_GENO_PHASED).
void build_sex_row(const SexPlan& sp, const uint8_t* raw_row, uint8_t* out_row, uint64_t out_rec, double* maj_freq, const uint8_t* phase = nullptr) {
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

struct ClumpData {
  // per dataset variant (index into the caller's included-variant list)
  std::vector<double> best_ln;                // lowest ln p among the lines at or below the load threshold; 0 without one
  std::vector<uint32_t> nonsig;               // lines above every bin boundary
  std::vector<std::vector<uint32_t>> entries; // one per loaded line, in the order read (last report first): (file << 12) | (bin << 1) | (ln p > ln p2)
  std::vector<double> ln_bins;                // bin boundaries in use (empty: the 'bins' column set is off)
  std::vector<uint8_t> best_a1;               // --clump-force-a1: the best line's A1 is the ALT allele
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
void clump_load_report(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, ClumpData* D) {
  const uint32_t variant_ct = static_cast<uint32_t>(inc.size());
  D->best_ln.assign(variant_ct, 0.0);
  D->nonsig.assign(variant_ct, 0);
  D->entries.assign(variant_ct, std::vector<uint32_t>());
  D->best_file.assign(variant_ct, 1);
  D->best_a1.assign(variant_ct, 0);
  D->observed.assign(variant_ct, 0);
  // ID -> included-variant index; kDup marks IDs the dataset holds more than once (an error only when the report names one)
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
    const std::vector<std::string> want_a1 = A.clump_force_a1 ? (A.clump_a1_field.empty() ? std::vector<std::string>{"A1"} : A.clump_a1_field) : std::vector<std::string>();
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
      uint32_t a1_alt = 0;
      if (A.clump_force_a1) {  // (:7783-7818)
        if (col_a1 < 0) {
          die(7, "Error: Variant ID on line %zu of %s is multiallelic, but there is no A1 column.\n", line_idx, fname.c_str());
        }
        const std::string a1(toks[col_a1].first, toks[col_a1].second);
        if (a1 == V.ref[inc[k]]) {
          a1_alt = 0;
        } else if (a1 == V.alt[inc[k]]) {
          a1_alt = 1;
        } else {
          if (ln_pval <= ln_p1) {
            D->missing_pairs.push_back(id + "\t" + a1);
          }
          continue;
        }
      }
      if (ln_pval > load_thresh) {
        if (ln_pval > 0.0) {
          die(6, "Error: p-value > 1 on line %zu of %s.\n", line_idx, fname.c_str());
        }
        if (nonsig_needed && (D->ln_bins.empty() || (ln_pval > D->ln_bins.back()))) {
          D->nonsig[k] += 1;
          D->observed[k] = 1;
        }
        continue;
      }
      if (D->best_ln[k] >= ln_pval) {  // (>=: the reports are read last to first, so ties end up with the first one, :7833)
        D->best_ln[k] = ln_pval;
        D->best_file[k] = static_cast<uint16_t>(file_idx1);
        D->best_a1[k] = static_cast<uint8_t>(a1_alt);
      }
      D->observed[k] = 1;
      if (keep_entries) {
        D->entries[k].push_back(static_cast<uint32_t>((a1_alt << 30) | (file_idx1 << 12) | (clump_bin(D->ln_bins, ln_pval) << 1) | (ln_pval > ln_p2)));
      }
    }
  }
}

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

// --clump-range[0] (LoadAndSortIntervalBed / LoadIntervalBed, plink2_set.cc:39-330, :495-638): lines `chrom first last name`;
// per chromosome the names in natural order, each with its intervals -- stretched by the border, half-open, sorted, merged.
int chrom_code(const std::string& name_in);
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

// what --clump needs to know about sex chromosomes (ClumpReports :8150-8215, :8460-8482)
struct ClumpSex {
  const std::vector<uint8_t>* vcls = nullptr;  // per included variant: 3 chrX, 4 chrY
  std::vector<uint32_t> male_cols;             // raw sample indices of the male founders
  uint32_t founder_male_ct = 0, founder_female_ct = 0, founder_nosex_ct = 0;
  int prov_storage = 1;                        // ldp_pgen_provisional_ref
  std::vector<uint8_t> prov_bits;
  std::function<void(ldp_engine*, const std::vector<uint32_t>&, const std::vector<uint32_t>*)> feed_cols;
  std::function<void(ldp_engine*, uint32_t, uint32_t)> females_missing;
};

int clump_reports(const Args& A, const Variants& V, const std::vector<uint32_t>& inc, const std::vector<uint32_t>& chr_idx,
                  const std::vector<uint32_t>& bps, uint32_t founder_ct,
                  const std::function<void(ldp_engine*, const std::vector<uint32_t>&)>& feed, const ClumpSex& SX) {
  if (founder_ct < 2) {
    die(7, "Error: --clump requires at least two founders.  (--make-founders may come in handy\nhere.)\n");
  }
  const double t_start = now_s();
  ClumpData D;
  clump_load_report(A, V, inc, &D);
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
  std::vector<uint32_t> obs;  // -> index into inc[]
  for (uint32_t k = 0; k < D.observed.size(); ++k) {
    if (D.observed[k]) {
      obs.push_back(k);
    }
  }
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
    o_chr[o] = chr_idx[obs[o]];
    o_bp[o] = bps[obs[o]];
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
      s_raw[q] = inc[obs[sub[q]]];
    }
    if (ldp_set_variants_vcor(e, n_sub, s_chr.data(), s_bp.data(), A.clump_bp_radius, 0xffffffffu)) {
      die(16, "Error: engine setup failed: %s\n", ldp_last_error(e));
    }
    feed(e, s_raw);
    // sex chromosomes: chrY rows with the female founders' calls missing; chrX pairs through the male-weighted sums when the
    // founders are of both kinds (is_x, :8472-8481), their own engine for the male founders' tuples
    std::vector<uint8_t> s_is_x(n_sub, 0);
    bool any_x = false, any_y = false;
    for (uint32_t q = 0; q < n_sub; ++q) {
      const uint8_t cls = (*SX.vcls)[obs[sub[q]]];
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
          if ((*SX.vcls)[obs[sub[q]]] == 4) {
            SX.females_missing(e, q, s_raw[q]);
          }
        }
      }
    }
    XWeighted xw;
    std::vector<uint32_t> band_lo;
    if (any_x) {
      ldp_params MP = RP;
      MP.founder_ct = SX.founder_male_ct;
      if (ldp_create(&MP, &xw.male) || ldp_set_variants_matrix(xw.male, n_sub)) {
        die(16, "Error: engine setup failed.\n");
      }
      SX.feed_cols(xw.male, s_raw, &SX.male_cols);
      xw.all = e;
      xw.is_x = s_is_x;
      // one orientation for both tuples of a pair: the main engine's (the male engine chose its major alleles from the male
      // founders alone).  Which one it is does not matter inside chrX: the weight is dyadic and every sum exact.
      std::vector<ldp_variant_rec> ra(n_sub), rm(n_sub);
      if (ldp_get_variant_recs(e, 0, n_sub, ra.data()) || ldp_get_variant_recs(xw.male, 0, n_sub, rm.data())) {
        die(16, "Error: %s\n", ldp_last_error(e));
      }
      xw.flip_male.resize(n_sub);
      for (uint32_t q = 0; q < n_sub; ++q) {
        xw.flip_male[q] = static_cast<uint8_t>((ra[q].flags ^ rm[q].flags) & 1u);
      }
      band_lo.resize(n_sub);
      uint64_t cand_pairs = 0;
      ldp_get_band(e, band_lo.data(), &cand_pairs);
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
      if (any_x) {
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
  const bool a1_col = (cols & kClumpColA1) != 0;  // ('maybea1' wants a multiallelic variant in the dataset: those are refused above)
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
    const uint32_t iv = inc[obs[io]];
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
    if (a1_col) {  // (a biallelic variant: the best line's A1 with --clump-force-a1, else '.', :9186-9196)
      if (A.clump_force_a1) {
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
        const uint32_t k = obs[members[q]];
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
            first_bp = V.bp[inc[k]];
          }
          last_bp = V.bp[inc[k]];
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
        buf += V.id[inc[obs[m]]];
        if (A.clump_force_a1) {  // (:9355-9358)
          buf += '(';
          buf += ((en >> 30) & 1) ? V.alt[inc[obs[m]]] : V.ref[inc[obs[m]]];
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
  void join_hip() {
    if (t_hip.joinable()) {
      t_hip.join();
      t_joined = now_s();
    }
  }
  ~Session() {
    if (t_hip.joinable()) {
      t_hip.join();
    }
  }
};

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
      die(7, "Error: This run estimates linkage disequilibrium between variants, but there\nare less than 50 samples to estimate from.  You should perform this operation\non a larger dataset.\n(Strictly speaking, you can also override this error with --bad-ld, but this is\nalmost always a bad idea.)\n");
    }
    die(7, "Error: This run estimates linkage disequilibrium between variants, but there\nare less than 50 founders to estimate from.  --make-founders may help.\n(Strictly speaking, you can also override this error with --bad-ld, but this is\nalmost always a bad idea.)\n");
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
      if (cls >= 3 && V.alt_ct[v] > 1) {
        die(63, "Error: multiallelic variant '%s' on chrX/chrY/MT is not supported yet by plink2-hip.\n", V.id[v].c_str());
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
        if (is_x[k]) {
          maj_allele[k] = x_maj_alt[k];
          maj_freq = x_maj_freq[k];
        } else if (it != multi_maj.end()) {
          maj_allele[k] = static_cast<uint8_t>(it->second.first);
          maj_freq = it->second.second;
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
    if (any_x) {  // (a window never leaves its chromosome: the pairs of the chrX rows)
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
  const double t_begin = S.t_begin;
  const std::vector<uint8_t>& is_founder = S.is_founder;
  const std::vector<uint8_t>& sex = S.sex;
  const uint32_t raw_sample_ct = S.raw_sample_ct, founder_ct = S.founder_ct, raw_variant_ct = S.raw_variant_ct;
  const std::string& gpath = S.gpath;
  ldp_pgen* const pg = S.pg;
  const int storage_mode = S.storage_mode, encoding = S.encoding, has_multiallelic = S.has_multiallelic;
  const uint64_t rec_bytes = S.rec_bytes;
  const uint8_t* const direct_rows = S.direct_rows;
  const std::vector<uint32_t>&inc = S.inc, &chr_idx = S.chr_idx, &bps = S.bps;
  const std::vector<uint8_t>& vcls = S.vcls;
  const uint32_t variant_ct = S.variant_ct, m_ct = S.m_ct;
  const std::vector<uint32_t>&mk = S.mk, &xk = S.xk, &yk = S.yk, &tk = S.tk, &m_chr = S.m_chr, &m_bps = S.m_bps;
  auto join_hip = [&S]() { S.join_hip(); };
  const double &t_hip_init = S.t_hip_init, &t_parse = S.t_parse, &t_joined = S.t_joined;
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
  auto females_missing = [&](ldp_engine* eng, uint32_t row_idx, uint32_t raw) {
    std::vector<uint8_t> row(rec_bytes);
    const uint8_t missing_code = (encoding == LDP_GENO_BED) ? 1 : 3;
    if (direct_rows) {
      memcpy(row.data(), direct_rows + static_cast<uint64_t>(raw) * rec_bytes, rec_bytes);
    } else if (ldp_pgen_read(pg, raw, 1, row.data(), rec_bytes, 0)) {
      die(6, "Error: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
    }
    for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
      if (is_founder[sx] && (sex[sx] == 2)) {
        uint8_t& b = row[sx >> 2];
        b = static_cast<uint8_t>((b & ~(3u << (2 * (sx & 3)))) | (missing_code << (2 * (sx & 3))));
      }
    }
    if (ldp_load_genotypes(eng, row_idx, 1, row.data(), rec_bytes, LDP_MEM_HOST, encoding | ((founder_ct == raw_sample_ct) ? 0 : LDP_GENO_MAPPED))) {
      die(16, "Error: %s\n", ldp_last_error(eng));
    }
  };
  if (A.have_clump) {
    for (uint32_t k = 0; k < variant_ct; ++k) {
      if (V.alt_ct[inc[k]] > 1) {  // (the reference clumps (variant, A1 allele) pairs there, plink2_ld.cc:7776-7817)
        die(63, "Error: multiallelic variant '%s': plink2-hip's --clump handles biallelic variants only.\n", V.id[inc[k]].c_str());
      }
    }
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
    SX.females_missing = females_missing;
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
        multiallelic_inverse_row(pg, inc[k], alts, founder_idx, &lo, &hi, inv_row.data(), out_rec, &mf, nullptr, 0, nullptr, &maj);
        if (want_maj) {
          multi_maj[k] = std::make_pair(maj, mf);
        }
        if (A.r2_ref_based) {
          continue;  // (the main track's REF-vs-rest codes are the rows; only the major allele and its frequency were wanted)
        }
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
        females_missing(e, k, inc[k]);
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
      if (ldp_get_variant_recs(e_male, 0, variant_ct, recs_male.data())) {
        die(16, "Error: %s\n", ldp_last_error(e_male));
      }
    }
    x_flip_all.assign(variant_ct, 0);
    x_flip_male.assign(variant_ct, 0);
    x_maj_alt.assign(variant_ct, 0);
    x_maj_freq.assign(variant_ct, 0.0);
    for (uint32_t k = 0; k < variant_ct; ++k) {
      uint32_t target_alt = recs_all[k].flags & 1u;  // the engine's own choice: diploid allele counts over the founders
      if (is_x[k]) {
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
      x_flip_all[k] = static_cast<uint8_t>((recs_all[k].flags & 1u) ^ target_alt);
      x_flip_male[k] = static_cast<uint8_t>((recs_male[k].flags & 1u) ^ target_alt);
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
  J.multi_maj = std::move(multi_maj);
  J.x_maj_alt = x_maj_alt;
  J.x_maj_freq = x_maj_freq;
  return A.r2_table ? write_vcor_table(J) : write_vcor_matrix(J);
}

// ---- --indep-pairwise / --indep-pairphase ----
void Session::need_dosage_sums(const std::vector<uint32_t>& raw_variants) {
  std::vector<uint32_t> todo;
  for (uint32_t v : raw_variants) {
    if (!dosage_sums.count(v)) {
      todo.push_back(v);
    }
  }
  if (todo.empty()) {
    return;
  }
  std::vector<uint8_t> founder_mask((static_cast<size_t>(raw_sample_ct) + 7) / 8, 0);
  for (uint32_t sx = 0; sx < raw_sample_ct; ++sx) {
    if (is_founder[sx]) {
      founder_mask[sx >> 3] |= static_cast<uint8_t>(1u << (sx & 7));
    }
  }
  const uint8_t* mask = (founder_ct == raw_sample_ct) ? nullptr : founder_mask.data();
  std::vector<std::pair<uint64_t, uint64_t>> out(todo.size());
  std::atomic<uint32_t> next(0);
  std::atomic<int> bad(0);
  auto worker = [&]() {
    for (uint32_t t = next.fetch_add(64); (t < todo.size()) && !bad.load(); t = next.fetch_add(64)) {
      for (uint32_t q = t; q < std::min<size_t>(todo.size(), t + 64); ++q) {
        if (ldp_pgen_dosage_sums(pg, todo[q], mask, &out[q].first, &out[q].second)) {
          bad.store(1);
          return;
        }
      }
    }
  };
  const uint32_t nthreads = std::max<uint32_t>(1, std::min<uint32_t>({64u, std::thread::hardware_concurrency(), static_cast<uint32_t>((todo.size() + 63) / 64)}));
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < nthreads; ++t) {
    pool.emplace_back(worker);
  }
  worker();
  for (std::thread& t : pool) {
    t.join();
  }
  if (bad.load()) {
    die(6, "\nError: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
  }
  for (size_t q = 0; q < todo.size(); ++q) {
    dosage_sums[todo[q]] = out[q];
  }
}

// --indep-pairwise / --indep-pairphase: one run, phase by phase in the order run() calls them (LdPrune, plink2_ld.cc:2530-2720;
// IndepPairwise / IndepPairphase :1284-1450, :2020-2330; LdPruneWrite :2464-2528).  The members are what the phases share.
struct PruneJob {
  Session& S;
  const Args& A = S.A;
  const Variants& V = S.V;
  const double t_begin;
  const std::vector<uint8_t>& is_founder = S.is_founder;
  const std::vector<uint8_t>& sex = S.sex;
  const uint32_t raw_sample_ct = S.raw_sample_ct, founder_ct = S.founder_ct, raw_variant_ct = S.raw_variant_ct;
  const std::string& gpath = S.gpath;
  ldp_pgen* const pg = S.pg;
  const int storage_mode = S.storage_mode, encoding = S.encoding, has_multiallelic = S.has_multiallelic;
  const uint64_t rec_bytes = S.rec_bytes;
  const uint8_t* const direct_rows = S.direct_rows;
  const std::vector<uint32_t>&inc = S.inc, &chr_idx = S.chr_idx, &bps = S.bps;
  const std::vector<uint8_t>& vcls = S.vcls;
  const uint32_t variant_ct = S.variant_ct, m_ct = S.m_ct;
  const std::vector<uint32_t>&mk = S.mk, &xk = S.xk, &yk = S.yk, &tk = S.tk, &m_chr = S.m_chr, &m_bps = S.m_bps;
  const double &t_hip_init = S.t_hip_init, &t_parse = S.t_parse, &t_joined = S.t_joined;

  ldp_params P;
  bool duplicate_ids = false;
  double t_tables_done = 0, t_planned = 0, t_load0 = 0, t_load1 = 0, t_run1 = 0;
  int world = 1, n_devices = 1;
  bool alias_devices = false;
  std::vector<ldp_engine*> eng;
  uint32_t subcontig_ct = 0;
  std::vector<uint64_t> removed;    // bit k: variant k (include order) is pruned
  std::vector<uint64_t> preferred;  // --indep-preferred, same indexing; empty: none
  std::vector<uint8_t> founder_mask;
  std::vector<uint32_t> founder_idx;
  // geometry of the rows on their way to the engines (set_row_geometry)
  bool all_founders = false;
  uint64_t in_rec = 0, in_phase_off = 0, out_rec = 0, direct_off = 0;
  int load_encoding = 0, direct_fd = -1;
  const uint8_t* direct = nullptr;
  // what the bulk load leaves for the host-built rows
  bool device_multi = false;
  uint32_t pending_unphased = UINT32_MAX;

  explicit PruneJob(Session& s) : S(s), t_begin(s.t_begin) {}

  [[noreturn]] void die_unphased(uint32_t raw_v) const {
    die(7, "\nError: --indep-pairphase: 0-based variant #%u is not fully phased.\n", raw_v);  // plink2_ld.cc:2047
  }
  // --indep-preferred bits of a subset of the variants (ks: include-order indices, in the subset's engine order)
  std::vector<uint64_t> sub_preferred(const std::vector<uint32_t>& ks) const {
      std::vector<uint64_t> out;
      if (!preferred.empty()) {
        out.assign((ks.size() + 63) / 64 + 1, 0);
        for (size_t q = 0; q < ks.size(); ++q) {
          if ((preferred[ks[q] >> 6] >> (ks[q] & 63)) & 1) {
            out[q >> 6] |= 1ull << (q & 63);
          }
        }
      }
      return out;
  }
  // an engine's removed bits (its own variant order) into the run's bitmap
  void scatter(const std::vector<uint64_t>& bm, const std::vector<uint32_t>& ks) {
      for (size_t q = 0; q < ks.size(); ++q) {
        if ((bm[q >> 6] >> (q & 63)) & 1) {
          removed[ks[q] >> 6] |= 1ull << (ks[q] & 63);
        }
      }
  }

  void set_params() {
    memset(&P, 0, sizeof(P));
    P.founder_ct = A.pairphase ? 2 * founder_ct : founder_ct;  // --indep-pairphase: haplotypes (plink2_ld.cc:1506)
    P.prune_window_size = A.window;
    P.prune_window_incr = A.step;
    P.window_is_bp = A.window_is_bp;
    P.plink1_order = (A.order == 1);
    P.prune_last_param = A.r2;
  }

  // --dry-run: the plan only (no device)
  int dry_run() {
    ldp_engine* e = nullptr;
    P.device = -1;
    S.join_hip();
    const double t_plan0 = now_s();
    if (ldp_create(&P, &e) || ldp_set_variants(e, m_ct, m_chr.data(), A.window_is_bp ? m_bps.data() : nullptr)) {
      die(16, "Error: planning failed.\n");
    }
    if (A.timing) {
      logprintf("[timing] table parse %.3f s, joined at %.3f s, variant table passes %.3f s, engine plan %.3f s\n", t_parse, t_joined - t_begin,
                t_plan0 - t_joined, now_s() - t_plan0);
    }
    uint32_t sct = 0;
    uint64_t cand = 0;
    ldp_get_subcontigs(e, &sct, nullptr, 0);
    ldp_get_band(e, nullptr, &cand);
    logprintf("dry-run: founders=%u variants=%u window=%u step=%u window_is_bp=%d r2=%a order=%d subcontigs=%u candidate_pairs=%llu\n",
              founder_ct, m_ct, A.window, A.step, A.window_is_bp ? 1 : 0, A.r2, A.order, sct, static_cast<unsigned long long>(cand));
    if (!xk.empty() || !yk.empty() || !tk.empty()) {
      logprintf("dry-run: chrX variants=%zu chrY variants=%zu%s (separate engines)\n", xk.size(), yk.size(), tk.empty() ? "" : " + MT");
    }
    ldp_destroy(e);
    return 0;
  }

  // unique IDs (plink2_ld.cc:2573-2592): checked here, beside the HIP start-up, reported where the reference does
  void check_unique_ids() {
    {
      // open-addressing table of variant indices keyed by a 64-bit FNV-1a hash of the ID
      uint32_t bits = 4;
      while ((1ull << bits) < 2ull * variant_ct) {
        ++bits;
      }
      const uint64_t mask = (1ull << bits) - 1;
      std::vector<uint32_t> table(static_cast<size_t>(1) << bits, 0xffffffffu);
      for (uint32_t k = 0; (k < variant_ct) && !duplicate_ids; ++k) {
        const std::string& id = V.id[inc[k]];
        uint64_t h = 0xcbf29ce484222325ull;
        for (unsigned char ch : id) {
          h = (h ^ ch) * 0x100000001b3ull;
        }
        uint64_t slot = (h ^ (h >> 29)) & mask;
        while (table[slot] != 0xffffffffu) {
          if (V.id[inc[table[slot]]] == id) {
            duplicate_ids = true;
            break;
          }
          slot = (slot + 1) & mask;
        }
        table[slot] = k;
      }
    }
  }

  // One GPU: the engine is created and planned (host work: ldp_create binds the device lazily) while the HIP runtime
  // is still starting; several GPUs: the device count decides how many engines there are, so wait for it first.
  void plan_engines() {
    if (A.gpus > 1) {
      S.join_hip();
      const int ndev = ldp_device_count();
      if (ndev < 1) {
        die(16, "Error: no usable HIP device (plink2-hip has no CPU compute path).\n");
      }
      // --debug-alias-devices (tests, one-GPU boxes): as many engines as --gpus asks for, dealt round-robin onto the devices
      // there are -- every host-side step of the N-device run (shard plan, per-engine loads, one thread per engine, segment pack /
      // exchange / stitch) then runs on a single device; RCCL refuses a device twice, so the exchange is the host transport.
      alias_devices = g_dbg.alias_devices;
      n_devices = ndev;
      world = alias_devices ? A.gpus : std::min(A.gpus, ndev);
    }
    eng.assign(world, nullptr);
    for (int r = 0; r < world; ++r) {
      P.device = r % n_devices;
      int rc = ldp_create(&P, &eng[r]);
      if (rc) {
        die(16, "Error: ldp_create failed (%d).\n", rc);
      }
      rc = ldp_set_variants(eng[r], m_ct, m_chr.data(), A.window_is_bp ? m_bps.data() : nullptr);
      if (rc) {
        die(16, "Error: %s\n", ldp_last_error(eng[r]));
      }
      ldp_get_subcontigs(eng[r], &subcontig_ct, nullptr, 0);
      if (world > 1) {
        rc = ldp_set_shard(eng[r], r, world, nullptr);
        if (rc) {
          die(16, "Error: %s\n", ldp_last_error(eng[r]));
        }
      }
    }
  }

  void check_before_loading() {
    if (duplicate_ids) {  // plink2_ld.cc:2590-2592
      die(7, "Error: --indep-pair%s requires unique variant IDs. (--set-all-var-ids and/or --rm-dup may help.)\n", A.pairphase ? "phase" : "wise");
    }
    if (S.has_dosage) {
      // chrX / chrY / MT: the reference's dosage-aware counts weigh males and females differently there (plink2_data.cc:2467-2620)
      for (const std::vector<uint32_t>* ks : {&xk, &yk, &tk}) {
        for (uint32_t k : *ks) {
          if (ldp_pgen_variant_has_dosage(pg, inc[k])) {
            die(63, "Error: variant '%s' on a sex chromosome or chrM has dosages, which plink2-hip reads on the autosomes only.\n", V.id[inc[k]].c_str());
          }
        }
      }
    }
  }

  // --indep-preferred (plink2_ld.cc:2594-2640)
  void read_preferred() {
    if (!A.preferred.empty()) {
      std::unordered_set<std::string> want;
      std::ifstream pin(A.preferred);
      if (!pin) {
        die(3, "Error: Failed to open %s.\n", A.preferred.c_str());
      }
      std::string tok;
      while (pin >> tok) {
        want.insert(tok);
      }
      preferred.assign((static_cast<size_t>(variant_ct) + 63) / 64, 0);
      uint32_t ct = 0;
      for (uint32_t k = 0; k < variant_ct; ++k) {
        if (want.count(V.id[inc[k]])) {
          preferred[k >> 6] |= 1ull << (k & 63);
          ++ct;
        }
      }
      logprintf("--indep-preferred: %u variant%s loaded.\n", ct, ct == 1 ? "" : "s");
    }
  }

  // ---- genotype rows of the diploid (+MT) variants -> engines, straight from the mapping (or the decoder's buffers);
  // the founder columns (CopyNyparrNonemptySubset, pgenlib_misc.cc:32,185) are picked on the device.
  void set_row_geometry() {
    all_founders = (founder_ct == raw_sample_ct);
    // --indep-pairphase rows: 2-bit codes, padding to a dword, phaseinfo bits (LDP_GENO_PHASED, ldprune_hip.h)
    in_rec = A.pairphase ? ldp_phased_row_bytes(2 * raw_sample_ct) : rec_bytes;
    in_phase_off = ldp_phased_phase_offset(2 * raw_sample_ct);
    out_rec = A.pairphase ? ldp_phased_row_bytes(2 * founder_ct) : ((static_cast<uint64_t>(founder_ct) + 3) / 4);
    load_encoding = A.pairphase ? (LDP_GENO_REF | LDP_GENO_PHASED) : encoding;
    direct = A.pairphase ? nullptr : direct_rows;  // phased rows always come through the decoder
    direct_off = 0;
    // Fixed-width rows go from the file to the engine's pinned ring with pread() (ldp_load_genotypes_fd), not by memcpy out of the
    // mapping: a 12 GB mapping is faulted in page run by page run, and what that costs swung between 0.35 and 0.97 s from one run to
    // the next on the same host, while 32 readers take 0.36-0.38 s every time (--debug-load-map: the mapping)
    direct_fd = (direct && !g_dbg.load_map) ? ldp_pgen_direct_fd(pg, &direct_off, nullptr) : -1;
    founder_mask.assign((static_cast<size_t>(raw_sample_ct) + 7) / 8, 0);
    for (uint32_t sidx = 0; sidx < raw_sample_ct; ++sidx) {
      if (is_founder[sidx]) {
        founder_mask[sidx >> 3] |= static_cast<uint8_t>(1u << (sidx & 7));
      }
    }
    founder_idx.clear();
    for (uint32_t s = 0; s < raw_sample_ct; ++s) {
      if (is_founder[s]) {
        founder_idx.push_back(s);
      }
    }
  }

  // the diploid (+MT) variants' rows, file -> engines (the loop of IndepPairwise, plink2_ld.cc:1345-1390)
  void load_diploid_rows() {
    // Chunks of ~256 MiB of decoded rows.  Variable-width .pgen: the next chunk is decoded (all host threads, see
    // ldp_pgen_read) while the engine takes the current one, two buffers alternating; small enough that the
    // buffers' first-touch page faults are paid once, large enough for ~60 decode tasks per chunk.
    const uint32_t kChunk = std::max<uint32_t>(1, static_cast<uint32_t>((direct ? (1024ull << 20) : (256ull << 20)) / std::max<uint64_t>(in_rec, 1)));
    // two buffers of one chunk each, malloc'ed (a vector would zero-fill them on this thread: 2 x 256 MiB of page faults and
    // memset before the first record is decoded; this way the decoder's threads touch the pages first, in parallel) and
    // never freed: returning ~0.5 GiB of touched pages to the kernel costs tens of ms and the process exits soon
    uint8_t* decoded[2] = {nullptr, nullptr};
    std::vector<uint8_t> gather;
    // the runs (maximal stretches of included variants that are contiguous in the file, capped at kChunk)
    struct Run {
      uint32_t q, raw0, n;
    };
    std::vector<Run> runs;
    for (uint32_t q = 0; q < m_ct;) {
      const uint32_t raw0 = inc[mk[q]];
      uint32_t run = 1;
      while (q + run < m_ct && inc[mk[q + run]] == raw0 + run && run < kChunk) {
        ++run;
      }
      runs.push_back({q, raw0, run});
      q += run;
    }
    // Non-founders in the file: the engines pick the founder columns themselves while converting (ldp_set_sample_map), so the
    // rows go up as the file has them.  (--indep-pairphase rows carry phase bits the gather does not move: host subset.)
    const bool device_subset = (!all_founders) && !A.pairphase;
    if (device_subset) {
      for (int r = 0; r < world; ++r) {
        if (ldp_set_sample_map(eng[r], raw_sample_ct, founder_idx.data(), nullptr)) {
          die(16, "\nError: %s\n", ldp_last_error(eng[r]));
        }
      }
    }
    // Variable-width records are decoded ON THE DEVICE from the file's own bytes (ldp_load_pgen_records: main track of every
    // record type, LD-compressed chains, and -- when the engine's samples are the file's -- the collapse of variants with more
    // than one ALT allele); --indep-pairphase rows (phase track) and --debug-host-decode take the host decoder below.
    // --indep-pairphase: main AND phase track on the device (ldp_load_pgen_records_phased) when every sample is a founder and no
    // variant has more than one ALT allele (whose phase refers to allele pairs: host rows, below); otherwise the host decoder.
    const bool device_phase = A.pairphase && all_founders && (!has_multiallelic) && (storage_mode != 0x01) && (storage_mode != 0x02) &&
                              (!g_dbg.host_decode);
    const bool device_decode = (!direct) && ((!A.pairphase) || device_phase) && (storage_mode != 0x01) && (storage_mode != 0x02) && (!g_dbg.host_decode);
    // (records with several ALT alleles are collapsed on the device as well: over the file's samples, or over the founders when the
    // engines pick those through a subset sample map)
    device_multi = device_decode && (all_founders || device_subset) && !A.pairphase;
    uint64_t file_size = 0;
    const void* file_bytes = device_decode ? ldp_pgen_file_bytes(pg, &file_size) : nullptr;
    std::vector<ldp_pgen_rec> rec_index;
    std::thread decoder;
    // The decoder runs beside the engine's copy threads (ldp_load_genotypes: 16 of them feeding the pinned ring); a record
    // takes microseconds, so a few dozen threads keep ahead of PCIe and more only get in the copies' way.
    const uint32_t decode_threads = g_dbg.decode_threads ? g_dbg.decode_threads : 32;
    double t_wait_decode = 0.0, t_load_calls = 0.0;
    int decode_rc = 0;
    uint32_t unphased_at = 0;
    pending_unphased = UINT32_MAX;
    auto start_decode = [&](size_t k) {
      if (direct || device_decode || k >= runs.size()) {
        return;
      }
      if (!decoded[k & 1]) {
        uint32_t longest = 0;
        for (const Run& rn : runs) {
          longest = std::max(longest, rn.n);
        }
        decoded[k & 1] = static_cast<uint8_t*>(malloc(static_cast<size_t>(longest) * in_rec + 64));
        if (!decoded[k & 1]) {
          die(2, "\nError: Out of memory.\n");
        }
      }
      decoder = std::thread([&, k]() {
        decode_rc = A.pairphase ? ldp_pgen_read_phased(pg, runs[k].raw0, runs[k].n, decoded[k & 1], in_rec, founder_mask.data(), decode_threads, &unphased_at)
                                : ldp_pgen_read(pg, runs[k].raw0, runs[k].n, decoded[k & 1], rec_bytes, decode_threads);
      });
    };
    start_decode(0);
    for (size_t k = 0; k < runs.size(); ++k) {
      const uint32_t q = runs[k].q;
      const uint32_t raw0 = runs[k].raw0;
      const uint32_t run = runs[k].n;
      const uint8_t* src;
      uint64_t stride = in_rec;
      if (device_decode) {
        rec_index.resize(run);
        uint32_t base_v = UINT32_MAX;
        ldp_pgen_rec base_rec;
        if (ldp_pgen_record_index(pg, raw0, run, rec_index.data(), &base_v) || ((base_v != UINT32_MAX) && ldp_pgen_record_index(pg, base_v, 1, &base_rec, nullptr))) {
          die(6, "\nError: %s: malformed variant record index.\n", gpath.c_str());
        }
        if (device_multi) {
          for (uint32_t t = 0; t < run; ++t) {
            const uint32_t alts = V.alt_ct[raw0 + t];
            if ((alts > 1) && (vcls[mk[q + t]] != 5)) {
              if (alts > 254) {
                die(63, "\nError: variant '%s' has more than 254 ALT alleles: not supported by plink2-hip.\n", V.id[raw0 + t].c_str());
              }
              rec_index[t].allele_ct = static_cast<uint8_t>(alts + 1);
            }
          }
        }
        const double tl0 = now_s();
        for (int r = 0; r < world; ++r) {
          uint32_t bad_q = UINT32_MAX;
          const int rc = device_phase ? ldp_load_pgen_records_phased(eng[r], q, run, file_bytes, file_size, LDP_MEM_HOST, rec_index.data(),
                                                                     (base_v != UINT32_MAX) ? &base_rec : nullptr, raw_sample_ct, &bad_q)
                                      : ldp_load_pgen_records(eng[r], q, run, file_bytes, file_size, LDP_MEM_HOST, rec_index.data(),
                                                              (base_v != UINT32_MAX) ? &base_rec : nullptr, raw_sample_ct, nullptr);
          if ((rc == LDP_ERR_UNPHASED) && (bad_q != UINT32_MAX)) {
            die_unphased(inc[mk[bad_q]]);  // (chunks and launches run in variant order: the first one to fail holds the lowest variant)
          }
          if (rc) {
            die((rc == LDP_ERR_INVALID) ? 6 : 16, "\nError: %s: %s\n", gpath.c_str(), ldp_last_error(eng[r]));
          }
        }
        t_load_calls += now_s() - tl0;
        continue;
      }
      if (direct) {
        src = direct + static_cast<uint64_t>(raw0) * rec_bytes;
      } else {
        const double tw0 = now_s();
        decoder.join();
        t_wait_decode += now_s() - tw0;
        if (decode_rc == LDP_ERR_UNPHASED) {
          pending_unphased = unphased_at;  // reported below, unless a multiallelic variant before it is unphased too
          break;
        }
        if (decode_rc) {
          die(6, "\nError: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
        }
        src = decoded[k & 1];
        start_decode(k + 1);
      }
      if ((!all_founders) && !device_subset) {
        // gather the founder columns (CopyNyparrNonemptySubset, pgenlib_misc.cc:32,185)
        // (+ CopyBitarrSubset of phaseinfo under --indep-pairphase, plink2_ld.cc:2075), all host threads
        gather.resize(static_cast<size_t>(run) * out_rec);
        if (ldp_subset_samples(src, direct ? rec_bytes : in_rec, run, raw_sample_ct, founder_mask.data(), gather.data(), out_rec, A.pairphase ? 1 : 0, 0)) {
          die(16, "\nError: founder subsetting failed.\n");
        }
        src = gather.data();
        stride = out_rec;
      }
      const double tl0 = now_s();
      for (int r = 0; r < world; ++r) {
        // fixed-width rows as the file has them: with pread() straight into the engine's
        // pinned ring (ldp_load_genotypes_fd)
        const bool from_fd = direct && (src == direct + static_cast<uint64_t>(raw0) * rec_bytes) && (direct_fd >= 0);
        const int rc = from_fd ? ldp_load_genotypes_fd(eng[r], q, run, direct_fd, direct_off + static_cast<uint64_t>(raw0) * rec_bytes, stride,
                                                       load_encoding | (device_subset ? LDP_GENO_MAPPED : 0))
                               : ldp_load_genotypes(eng[r], q, run, src, stride, LDP_MEM_HOST, load_encoding | (device_subset ? LDP_GENO_MAPPED : 0));
        if (rc) {
          die(16, "Error: %s\n", ldp_last_error(eng[r]));
        }
      }
      t_load_calls += now_s() - tl0;
    }
    if (A.timing && !direct) {
      logprintf("\n[timing] variable-width records: %zu chunks, waited %.3f s for the decoder, %.3f s inside ldp_load_genotypes\n", runs.size(), t_wait_decode,
                t_load_calls);
    }
  }

  // rows that need host treatment overwrite their bulk-loaded versions: variants with more than one ALT
  // allele (collapsed major-vs-rest) and MT variants (hets -> missing, plink2_ld.cc:1362-1364)
  void patch_host_built_rows() {
    uint32_t multi_ct = 0, mt_ct = 0, multi_device = 0;
    // (--indep-pairphase: a multiallelic row is 2 haplotypes per founder as plain 2-bit codes on the 2N-haplotype engine)
    const uint64_t host_rec = A.pairphase ? ((2ull * founder_ct + 3) / 4) : out_rec;
    const uint64_t raw_phase_bytes = (static_cast<uint64_t>(raw_sample_ct) + 7) / 8;
    std::vector<uint8_t> lo(raw_sample_ct), hi(raw_sample_ct), inv_row(host_rec), raw_row(rec_bytes + 8), phase_buf(2 * raw_phase_bytes);
    uint32_t multi_unphased = UINT32_MAX;
    SexPlan mt_plan;
    mt_plan.part1 = founder_idx;
    // A multiallelic variant whose REF allele is the major one needs nothing: the main track already counts REF
    // copies (0/1/2 non-REF alleles = 0/1/2 non-major ones), and GetMajIdxMulti's first test (plink2_common.cc:1042,
    // freq[REF] >= 0.5 with freq = count * (1 / total), plink2_filter.cc:2137-2147) is the biallelic rule the
    // conversion kernel applied to the bulk-loaded row.  Its genotype counts say which variants those are.
    std::vector<uint8_t> ref_is_major;
    uint32_t multi_skipped = 0;
    if ((!A.pairphase) && !device_multi) {
      bool any_multi = false;
      for (uint32_t qq = 0; (qq < m_ct) && !any_multi; ++qq) {
        any_multi = (V.alt_ct[inc[mk[qq]]] > 1) && (vcls[mk[qq]] != 5);
      }
      if (any_multi) {
        ref_is_major.assign(m_ct, 0);
        std::vector<ldp_variant_rec> recs(m_ct);
        for (int r = 0; r < world; ++r) {  // (a variant's counts are zero on the engines that do not own it)
          if (ldp_get_variant_recs(eng[r], 0, m_ct, recs.data())) {
            die(16, "\nError: %s\n", ldp_last_error(eng[r]));
          }
          for (uint32_t qq = 0; qq < m_ct; ++qq) {
            const uint64_t ref_ct = 2ull * recs[qq].n_homref + recs[qq].n_het;
            const uint64_t tot = 2ull * (static_cast<uint64_t>(recs[qq].n_homref) + recs[qq].n_het + recs[qq].n_homalt);
            if (tot && (static_cast<double>(ref_ct) * (1.0 / static_cast<double>(tot)) >= 0.5)) {
              ref_is_major[qq] = 1;
            }
          }
        }
      }
    }
    for (uint32_t qq = 0; qq < m_ct; ++qq) {
      const uint32_t raw_v = inc[mk[qq]];
      const uint32_t alts = V.alt_ct[raw_v];
      const bool is_mt = (vcls[mk[qq]] == 5);
      if (alts < 2 && !is_mt) {
        continue;
      }
      if (device_multi && !is_mt) {
        ++multi_device;  // (collapsed by ldp_load_pgen_records)
        continue;
      }
      if ((!is_mt) && (!ref_is_major.empty()) && ref_is_major[qq]) {
        ++multi_skipped;
        continue;
      }
      double mf = 0.0;
      if (is_mt) {
        fetch_raw_row(pg, storage_mode, raw_v, raw_sample_ct, rec_bytes, raw_row.data());
        build_sex_row(mt_plan, raw_row.data(), inv_row.data(), out_rec, &mf);
        ++mt_ct;
      } else {
        if (storage_mode == 0x01) {
          die(6, "\nError: multiallelic variant in a .bim/.bed fileset.\n");
        }
        if (A.pairphase) {
          bool unphased = false;
          multiallelic_inverse_row(pg, raw_v, alts, founder_idx, &lo, &hi, inv_row.data(), host_rec, &mf, phase_buf.data(), raw_phase_bytes, &unphased);
          if (unphased) {
            multi_unphased = std::min(multi_unphased, raw_v);
          }
        } else {
          multiallelic_inverse_row(pg, raw_v, alts, founder_idx, &lo, &hi, inv_row.data(), out_rec, &mf);
        }
        ++multi_ct;
      }
      for (int r = 0; r < world; ++r) {
        if (ldp_load_genotypes(eng[r], qq, 1, inv_row.data(), host_rec, LDP_MEM_HOST, LDP_GENO_INVERSE) ||
            ldp_set_maj_freqs(eng[r], qq, 1, &mf)) {
          die(16, "\nError: %s\n", ldp_last_error(eng[r]));
        }
      }
    }
    if (std::min(multi_unphased, pending_unphased) != UINT32_MAX) {
      die_unphased(std::min(multi_unphased, pending_unphased));
    }
    if ((multi_ct || mt_ct || multi_skipped || multi_device) && A.timing) {
      logprintf("\n[timing] host-built rows: %u multiallelic (%u more have REF as the major allele: main track as loaded; %u collapsed on the device), %u MT\n",
                multi_ct, multi_skipped, multi_device, mt_ct);
    }
  }

  // Variants whose records carry dosages: the major allele's frequency comes from the founders' dosage sums (a sample's
  // dosage where it has one, its hardcall otherwise: ldp_pgen_dosage_sums), in ComputeAlleleFreqs' arithmetic
  // (plink2_filter.cc:2137-2147: ref * (1 / (ref + alt)); the factor 2 of the diploid case cancels exactly) with
  // GetMajIdx's rule (REF unless its frequency is below 0.5).  The rows themselves stay the hardcalls.
  void set_dosage_frequencies() {
    if (S.has_dosage) {
      std::vector<uint32_t> todo;
      for (uint32_t qq = 0; qq < m_ct; ++qq) {
        const uint32_t raw_v = inc[mk[qq]];
        if (!ldp_pgen_variant_has_dosage(pg, raw_v)) {
          continue;
        }
        if ((V.alt_ct[raw_v] > 1) || (vcls[mk[qq]] == 5)) {
          die(63, "\nError: variant '%s' has dosages and %s, which plink2-hip does not read yet.\n", V.id[raw_v].c_str(),
              (vcls[mk[qq]] == 5) ? "lies on chrM" : "several ALT alleles");
        }
        todo.push_back(qq);
      }
      std::vector<double> mfs(todo.size(), 0.0);
      {
        std::vector<uint32_t> raw_todo(todo.size());
        for (size_t q = 0; q < todo.size(); ++q) {
          raw_todo[q] = inc[mk[todo[q]]];
        }
        S.need_dosage_sums(raw_todo);
        for (size_t q = 0; q < todo.size(); ++q) {
          const std::pair<uint64_t, uint64_t>& dd = S.dosage_sums[raw_todo[q]];
          const uint64_t tot = dd.first + dd.second;
          const double ref_freq = tot ? (static_cast<double>(static_cast<int64_t>(dd.first)) * (1.0 / static_cast<double>(static_cast<int64_t>(tot)))) : 0.5;
          mfs[q] = (ref_freq < 0.5) ? (1.0 - ref_freq) : ref_freq;
        }
      }
      for (size_t q = 0; q < todo.size(); ++q) {
        for (int r = 0; r < world; ++r) {
          if (ldp_set_maj_freqs(eng[r], todo[q], 1, &mfs[q])) {
            die(16, "\nError: %s\n", ldp_last_error(eng[r]));
          }
        }
      }
      if (A.timing) {
        logprintf("\n[timing] allele frequencies of %zu variants from their dosages\n", todo.size());
      }
    }
  }

  // every engine prunes its shard; several engines: their removed-bit segments meet (stitch, plink2_ld.cc:1418-1426)
  void run_diploid_engines() {
    t_load1 = now_s();
    const std::vector<uint64_t> pref_m = sub_preferred(mk);
    const size_t m_words = (static_cast<size_t>(m_ct) + 63) / 64 + 1;
    std::vector<std::vector<uint64_t>> part(world, std::vector<uint64_t>(m_words, 0));
    std::vector<int> rcs(world, 0);
    // several devices: every engine prunes its shard on a host thread of its own; the shards' results then meet in ONE RCCL
    // all-gather of their removed-bit segments (ldp_allgather_removed: the cross-device form of the stitch at
    // plink2_ld.cc:1418-1426).  Without RCCL -- or with engines that share a device -- the same segments are packed, copied
    // between the engines by the host and stitched by every rank (ldp_pack_removed_segment / ldp_stitch_removed_segments).
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) {
      th.emplace_back([&, r]() {
        if (!pref_m.empty()) {
          ldp_set_preferred(eng[r], pref_m.data());
        }
        rcs[r] = ldp_run(eng[r], part[r].data());
      });
    }
    for (std::thread& t : th) {
      t.join();
    }
    th.clear();
    // (a rank whose run failed must not leave the others waiting in a collective: nobody enters it then)
    for (int r = 0; r < world; ++r) {
      if (rcs[r]) {
        die(16, "\nError: %s\n", ldp_last_error(eng[r]));
      }
    }
    if (world == 1) {
      scatter(part[0], mk);
    } else {
      std::vector<void*> comms(world, nullptr);
      std::vector<std::vector<uint64_t>> full(world, std::vector<uint64_t>(m_words, 0));
      bool use_rccl = false;
      if (!alias_devices) {
        std::vector<int> devs(world);
        for (int r = 0; r < world; ++r) {
          devs[r] = r;
        }
        use_rccl = (ldp_comm_init_all(world, devs.data(), comms.data()) == 0);
      }
      if (use_rccl) {
        for (int r = 0; r < world; ++r) {
          th.emplace_back([&, r]() { rcs[r] = ldp_allgather_removed(eng[r], comms[r], part[r].data(), full[r].data()); });
        }
        for (std::thread& t : th) {
          t.join();
        }
        for (int r = 0; r < world; ++r) {
          if (rcs[r]) {  // (the failing rank aborted its communicator; the process ends here, nothing is destroyed twice)
            die(16, "\nError: %s\n", ldp_last_error(eng[r]));
          }
        }
        for (int r = 0; r < world; ++r) {
          ldp_comm_destroy(comms[r]);
        }
      } else {
        uint64_t seg_words = 0;
        if (ldp_shard_segment_words(eng[0], &seg_words)) {
          die(16, "\nError: %s\n", ldp_last_error(eng[0]));
        }
        std::vector<uint64_t> segs(static_cast<size_t>(seg_words) * world, 0);
        for (int r = 0; r < world; ++r) {
          if (ldp_pack_removed_segment(eng[r], part[r].data(), segs.data() + static_cast<size_t>(r) * seg_words)) {
            die(16, "\nError: packing the removed bits of shard %d failed.\n", r);
          }
        }
        for (int r = 0; r < world; ++r) {
          if (ldp_stitch_removed_segments(eng[r], segs.data(), full[r].data())) {
            die(16, "\nError: stitching the removed bits on shard %d failed.\n", r);
          }
        }
      }
      for (int r = 1; r < world; ++r) {  // every rank holds the same global bitmap
        if (memcmp(full[r].data(), full[0].data(), ((static_cast<size_t>(m_ct) + 63) / 64) * sizeof(uint64_t)) != 0) {
          die(16, "\nError: the shards disagree about the stitched prune bitmap (rank %d).\n", r);
        }
      }
      scatter(full[0], mk);
      if (A.timing) {
        logprintf("\n[timing] %d engines on %d device%s, exchange: %s\n", world, std::min(world, n_devices), (std::min(world, n_devices) == 1) ? "" : "s",
                  use_rccl ? "RCCL all-gather" : "host transport");
      }
    }
    t_run1 = now_s();
  }

  void report_load_and_run() {
    ldp_counters c;
    ldp_get_counters(eng[0], &c);
    logprintf("\n[timing] setup+parse %.3f s | genotype load (file -> HBM bit-planes) %.3f s | run %.3f s (pair kernel %.1f ms, replay %.1f ms; %llu candidate pairs) | buffer release %.3f s\n",
              t_load0 - t_begin, t_load1 - t_load0, (t_run1 ? t_run1 : now_s()) - t_load1, c.ms_pair_kernel, c.ms_replay, static_cast<unsigned long long>(c.candidate_pairs), t_run1 ? now_s() - t_run1 : 0.0);
  }

  // ---- chrX, chrY: their own sample sets, rows built on the host, one engine each on device 0
  // (--indep-pairphase: MT too, one haplotype per founder with hets missing -- HapsplitHaploid, plink2_ld.cc:2051)
  void run_sex_chromosomes() {
    for (int which = 0; which < 3; ++which) {
      const std::vector<uint32_t>& ks = (which == 0) ? xk : ((which == 1) ? yk : tk);
      if (ks.empty()) {
        continue;
      }
      static const char* const kSexName[3] = {"X", "Y", "MT"};
      SexPlan sp;
      for (uint32_t sidx : founder_idx) {
        if (which == 0) {
          (sex[sidx] == 1 ? sp.part1 : sp.part2).push_back(sidx);  // males | non-males (female + unknown)
        } else if ((which == 2) || (sex[sidx] != 2)) {
          sp.part1.push_back(sidx);                                  // non-females (chrY) / every founder (MT)
        }
      }
      sp.x_freq = (which == 0);
      const bool x_phased = A.pairphase && (which == 0) && !sp.part2.empty();
      std::vector<uint8_t> nonmale_mask;
      if (x_phased) {
        nonmale_mask.assign((static_cast<size_t>(raw_sample_ct) + 7) / 8, 0);
        for (uint32_t sidx : sp.part2) {
          nonmale_mask[sidx >> 3] |= static_cast<uint8_t>(1u << (sidx & 7));
        }
      }
      const uint32_t fct = sp.out_ct();
      if (fct < 2) {
        die(63, "\nError: fewer than two usable founders on chr%s; not supported by plink2-hip.\n", kSexName[which]);
      }
      ldp_params SP = P;
      SP.founder_ct = fct;
      SP.device = 0;
      ldp_engine* se = nullptr;
      std::vector<uint32_t> s_chr(ks.size()), s_bps(ks.size());
      for (size_t w = 0; w < ks.size(); ++w) {
        s_chr[w] = chr_idx[ks[w]];
        s_bps[w] = bps[ks[w]];
      }
      if (ldp_create(&SP, &se) || ldp_set_variants(se, static_cast<uint32_t>(ks.size()), s_chr.data(), A.window_is_bp ? s_bps.data() : nullptr)) {
        die(16, "\nError: chr%s engine setup failed.\n", kSexName[which]);
      }
      const uint64_t s_rec = (static_cast<uint64_t>(fct) + 3) / 4;
      const uint32_t chunk = std::max<uint32_t>(1, static_cast<uint32_t>((256ull << 20) / std::max<uint64_t>(s_rec, 1)));
      std::vector<uint8_t> rows;
      std::vector<double> mfs;
      std::atomic<uint32_t> x_unphased(UINT32_MAX);
      if (!x_phased) {
        // The rows are built on the device (ldp_set_sample_map): the file's rows go up as they are, in runs of variants
        // that are consecutive in the file, and a conversion-time gather picks the founders -- the haploid ones first, with
        // their het calls made missing, then (chrX) the others twice.  The host used to do this per sample and variant.
        std::vector<uint32_t> src_sample;
        std::vector<uint8_t> het_missing;
        src_sample.reserve(fct);
        het_missing.reserve(fct);
        for (uint32_t sidx : sp.part1) {
          src_sample.push_back(sidx);
          het_missing.push_back(1);
        }
        for (int rep = 0; rep < 2; ++rep) {
          for (uint32_t sidx : sp.part2) {
            src_sample.push_back(sidx);
            het_missing.push_back(0);
          }
        }
        if (ldp_set_sample_map(se, raw_sample_ct, src_sample.data(), het_missing.data())) {
          die(16, "\nError: %s\n", ldp_last_error(se));
        }
        const uint32_t max_run = std::max<uint32_t>(1, static_cast<uint32_t>((256ull << 20) / std::max<uint64_t>(rec_bytes, 1)));
        std::vector<uint8_t> decoded;
        for (uint32_t w0 = 0; w0 < ks.size();) {
          const uint32_t raw0 = inc[ks[w0]];
          uint32_t run = 1;
          while ((w0 + run < ks.size()) && (inc[ks[w0 + run]] == raw0 + run) && (run < max_run)) {
            ++run;
          }
          const uint8_t* rows_at = nullptr;
          if (direct_rows) {
            rows_at = direct_rows + static_cast<uint64_t>(raw0) * rec_bytes;
          } else {
            decoded.resize(static_cast<size_t>(run) * rec_bytes);
            if (ldp_pgen_read(pg, raw0, run, decoded.data(), rec_bytes, 0)) {
              die(6, "\nError: %s: %s\n", gpath.c_str(), ldp_pgen_last_error(pg));
            }
            rows_at = decoded.data();
          }
          if (ldp_load_genotypes(se, w0, run, rows_at, rec_bytes, LDP_MEM_HOST, encoding | LDP_GENO_MAPPED)) {
            die(16, "\nError: %s\n", ldp_last_error(se));
          }
          w0 += run;
        }
      }
      for (uint32_t w0 = 0; x_phased && (w0 < ks.size()); w0 += chunk) {
        const uint32_t cnt = std::min<uint32_t>(chunk, static_cast<uint32_t>(ks.size()) - w0);
        rows.assign(static_cast<size_t>(cnt) * s_rec, 0);
        mfs.assign(cnt, 0.0);
        const uint32_t nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < nthreads; ++t) {
          pool.emplace_back([&, t]() {
            std::vector<uint8_t> raw_row(in_rec + 8);
            for (uint32_t w = t; w < cnt; w += nthreads) {
              const uint32_t raw_v = inc[ks[w0 + w]];
              if (x_phased) {
                uint32_t at = 0;
                const int prc = ldp_pgen_read_phased(pg, raw_v, 1, raw_row.data(), in_rec, nonmale_mask.data(), 1, &at);
                if (prc == LDP_ERR_UNPHASED) {
                  uint32_t cur = x_unphased.load();
                  while ((raw_v < cur) && !x_unphased.compare_exchange_weak(cur, raw_v)) {
                  }
                  continue;
                }
                if (prc) {
                  die(6, "\nError: %s\n", ldp_pgen_last_error(pg));
                }
                build_sex_row(sp, raw_row.data(), rows.data() + static_cast<uint64_t>(w) * s_rec, s_rec, &mfs[w], raw_row.data() + in_phase_off);
                continue;
              }
              fetch_raw_row(pg, storage_mode, raw_v, raw_sample_ct, rec_bytes, raw_row.data());
              build_sex_row(sp, raw_row.data(), rows.data() + static_cast<uint64_t>(w) * s_rec, s_rec, &mfs[w]);
            }
          });
        }
        for (std::thread& t : pool) {
          t.join();
        }
        if (x_unphased.load() != UINT32_MAX) {
          die_unphased(x_unphased.load());
        }
        if (ldp_load_genotypes(se, w0, cnt, rows.data(), s_rec, LDP_MEM_HOST, LDP_GENO_INVERSE) || ldp_set_maj_freqs(se, w0, cnt, mfs.data())) {
          die(16, "\nError: %s\n", ldp_last_error(se));
        }
      }
      const std::vector<uint64_t> pref_s = sub_preferred(ks);
      if (!pref_s.empty()) {
        ldp_set_preferred(se, pref_s.data());
      }
      std::vector<uint64_t> bm((ks.size() + 63) / 64 + 1, 0);
      if (ldp_run(se, bm.data())) {
        die(16, "\nError: %s\n", ldp_last_error(se));
      }
      scatter(bm, ks);
      ldp_destroy(se);
    }
  }

  void write_lists() {
    uint32_t removed_ct = 0;
    for (uint64_t w : removed) {
      removed_ct += static_cast<uint32_t>(__builtin_popcountll(w));
    }
    logprintf("%u/%u variants removed.\n", removed_ct, variant_ct);  // plink2_ld.cc:2707
    // LdPruneWrite, plink2_ld.cc:2464-2528
    for (int pass = 0; pass < 2; ++pass) {
      const std::string path = A.out + (pass ? ".prune.out" : ".prune.in");
      FILE* f = fopen(path.c_str(), "wb");
      if (!f) {
        die(3, "Error: Failed to open %s for writing.\n", path.c_str());
      }
      for (uint32_t k = 0; k < variant_ct; ++k) {
        const bool rem = (removed[k >> 6] >> (k & 63)) & 1;
        if (rem == static_cast<bool>(pass)) {
          fputs(V.id[inc[k]].c_str(), f);
          fputc('\n', f);
        }
      }
      if (fclose(f)) {
        die(5, "Error: File write failure: %s.\n", path.c_str());
      }
    }
    logprintf("Variant lists written to %s.prune.in and %s.prune.out .\n", A.out.c_str(), A.out.c_str());
  }

  [[noreturn]] void finish() {
    if (A.timing) {
      // (wall-clock stamps: what a caller's stopwatch sees beyond `total` is process start-up before main() and teardown after _exit)
      const double unix_now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
      logprintf("[timing] total %.3f s (main() entered at unix time %.3f, leaving at %.3f)\n", now_s() - t_begin, unix_now - (now_s() - t_begin), unix_now);
    }
    if (g_log) {
      fclose(g_log);
    }
    fflush(nullptr);
    // everything is on disk; releasing tens of GB of device memory and unmapping the input only costs time
    _exit(0);
  }

  int run() {
    set_params();
    if (A.dry_run) {
      return dry_run();
    }
    check_unique_ids();
    t_tables_done = now_s();
    plan_engines();
    t_planned = now_s();
    S.join_hip();
    if (ldp_device_count() < 1) {
      die(16, "Error: no usable HIP device (plink2-hip has no CPU compute path).\n");
    }
    removed.assign((static_cast<size_t>(variant_ct) + 63) / 64 + 1, 0);
    if (subcontig_ct || !xk.empty() || !yk.empty() || !tk.empty()) {
      check_before_loading();
      read_preferred();
      logprintf("--indep-pair%s (%d GPU%s): ", A.pairphase ? "phase" : "wise", world, world == 1 ? "" : "s");
      fflush(stdout);
      t_load0 = now_s();
      if (A.timing) {
        logprintf("\n[timing] table parse %.3f s, variant-table passes + ID check done at %.3f s, engine planned at %.3f s, HIP init %.3f s (concurrent; joined at %.3f s)\n",
                  t_parse, t_tables_done - t_begin, t_planned - t_begin, t_hip_init, t_joined - t_begin);
      }
      set_row_geometry();
      t_load1 = now_s();
      t_run1 = 0;
      if (subcontig_ct) {
        load_diploid_rows();
        patch_host_built_rows();
        set_dosage_frequencies();
        run_diploid_engines();
      }
      if (A.timing) {
        report_load_and_run();
      }
      run_sex_chromosomes();
    }
    write_lists();
    finish();
  }
};

int run_prune(Session& S) {
  PruneJob job(S);
  return job.run();
}


}  // namespace

// test hook (no GPU needed): lines `both_x unsquared flip1 flip2 mflip1 mflip2  nm sum1 ssq1 sum2 ssq2 dot  (the same six for the male
// founders)` in, the chrX-weighted r^2 (or r) out as the hex bits of the double -- XWeighted's arithmetic against values the
// reference computed (tests/test_r2_flags.py)
int debug_xweighted(const char* path) {
  FILE* df = fopen(path, "r");
  if (!df) {
    die(3, "Error: Failed to open %s.\n", path);
  }
  double nan_ref;
  {
    const uint64_t bits = 0xfff8000000000000ull;
    memcpy(&nan_ref, &bits, 8);
  }
  int bx, us, f1, f2, g1, g2;
  long long v[12];
  while (fscanf(df, "%d %d %d %d %d %d %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld %lld", &bx, &us, &f1, &f2, &g1, &g2, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5],
                &v[6], &v[7], &v[8], &v[9], &v[10], &v[11]) == 18) {
    ldp_pair_stats_t ta, tm;
    ta.nm = static_cast<uint32_t>(v[0]); ta.sum1 = static_cast<int32_t>(v[1]); ta.ssq1 = static_cast<uint32_t>(v[2]);
    ta.sum2 = static_cast<int32_t>(v[3]); ta.ssq2 = static_cast<uint32_t>(v[4]); ta.dot = static_cast<int32_t>(v[5]);
    tm.nm = static_cast<uint32_t>(v[6]); tm.sum1 = static_cast<int32_t>(v[7]); tm.ssq1 = static_cast<uint32_t>(v[8]);
    tm.sum2 = static_cast<int32_t>(v[9]); tm.ssq2 = static_cast<uint32_t>(v[10]); tm.dot = static_cast<int32_t>(v[11]);
    const double r = XWeighted::weighted(XWeighted::counts(ta, f1 != 0, f2 != 0), XWeighted::counts(tm, g1 != 0, g2 != 0), bx != 0, us != 0, nan_ref);
    uint64_t bits;
    memcpy(&bits, &r, 8);
    printf("%016llx\n", static_cast<unsigned long long>(bits));
  }
  fclose(df);
  return 0;
}

int main(int argc, char** argv) {
  if ((argc == 3) && (strcmp(argv[1], "--debug-xweighted") == 0)) {
    return debug_xweighted(argv[2]);
  }
  Session S;
  load_inputs(S, argc, argv);
  return S.A.have_r2 ? run_r2(S) : run_prune(S);
}
