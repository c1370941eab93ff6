// p2h_util.cpp -- plink2-hip: logging, the reference's number scanner and formatter (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include <sched.h>

#include "p2h_cli.h"

namespace p2h {


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




// The device's NUMA node (ldp_device_numa_node) -> its CPUs (sysfs cpulist: "0-63,128-191") -> this THREAD's affinity, intersected with what the process may use.
// Threads started afterwards by this thread inherit it, and memory they touch first lands on that node: the copy pool and the pinned staging ring of a load
// (ldp_engine_load.cpp) then feed the device's DMA engines without crossing the inter-socket fabric.
// Only when the node leaves this thread enough to work with -- at least 8 of its CPUs, or at least half of them (a container whose cpuset holds one or two CPUs
// of that node would otherwise put 32 copy threads on them) --, and `saved` (optional) receives the mask it had, for restore_affinity() once the file -> HBM leg is over.
int bind_near_device(int device, AffinityMask* saved) {
  if (saved) {
    saved->valid = false;
  }
  const int node = ldp_device_numa_node(device);
  if (node < 0) {
    return -1;
  }
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) {
    return -1;
  }
  char buf[4096];
  const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[got] = 0;
  cpu_set_t want, cur;
  CPU_ZERO(&want);
  for (const char* p = buf; *p;) {
    while (*p && !isdigit(static_cast<unsigned char>(*p))) {
      ++p;
    }
    if (!*p) {
      break;
    }
    char* e = nullptr;
    long a = strtol(p, &e, 10), b = a;
    if (*e == '-') {
      b = strtol(e + 1, &e, 10);
    }
    for (long c = a; (c <= b) && (c < CPU_SETSIZE); ++c) {
      CPU_SET(static_cast<int>(c), &want);
    }
    p = e;
  }
  if (sched_getaffinity(0, sizeof(cur), &cur)) {
    return -1;
  }
  CPU_AND(&want, &want, &cur);
  const int keep = CPU_COUNT(&want), have = CPU_COUNT(&cur);
  if ((keep == 0) || ((keep < 8) && (2 * keep < have))) {
    return -1;
  }
  if (sched_setaffinity(0, sizeof(want), &want)) {
    return -1;
  }
  if (saved) {
    static_assert(sizeof(saved->bits) >= sizeof(cpu_set_t), "AffinityMask holds a cpu_set_t");
    memcpy(saved->bits, &cur, sizeof(cur));
    saved->valid = true;
  }
  return node;
}

void restore_affinity(const AffinityMask& saved) {
  if (saved.valid) {
    cpu_set_t old;
    memcpy(&old, saved.bits, sizeof(old));
    (void)sched_setaffinity(0, sizeof(old), &old);
  }
}

}  // namespace p2h
