// ldp_pair_device.h -- device-side pieces shared by the two pair kernels (ldp_kernels.hip: popcount tiles,
// ldp_pair_mfma.hip: matrix-pipe tiles): the FP64 prune predicate, the r^2 of --r2-unphased, the per-pair output
// step and the early-termination bound.  Device code only (included by .hip translation units).
#ifndef LDP_PAIR_DEVICE_H
#define LDP_PAIR_DEVICE_H

#include "ldp_device.h"

namespace ldp {

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += __shfl_down(v, off, 64);
  }
  return v;
}

// Wait until at most `allowed` of this wave's memory operations are still in flight (they complete in order, so
// everything older has landed), then the workgroup barrier.  Hand-written because __syncthreads() always drains
// to zero, which would serialise the ring.  The "memory" clobber keeps LDS reads and DMA issues on their side.
__device__ __forceinline__ void wait_dma_then_barrier(uint32_t allowed) {
#define LDP_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); break;
  switch (allowed) {
    LDP_WAIT_CASE(1) LDP_WAIT_CASE(2) LDP_WAIT_CASE(3) LDP_WAIT_CASE(4) LDP_WAIT_CASE(5) LDP_WAIT_CASE(6) LDP_WAIT_CASE(7)
    LDP_WAIT_CASE(8) LDP_WAIT_CASE(9) LDP_WAIT_CASE(10) LDP_WAIT_CASE(11) LDP_WAIT_CASE(12) LDP_WAIT_CASE(13) LDP_WAIT_CASE(14)
    LDP_WAIT_CASE(15) LDP_WAIT_CASE(16) LDP_WAIT_CASE(17) LDP_WAIT_CASE(18) LDP_WAIT_CASE(19) LDP_WAIT_CASE(20) LDP_WAIT_CASE(21)
    LDP_WAIT_CASE(22) LDP_WAIT_CASE(23) LDP_WAIT_CASE(24) LDP_WAIT_CASE(25) LDP_WAIT_CASE(26) LDP_WAIT_CASE(27) LDP_WAIT_CASE(28)
    default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
  }
#undef LDP_WAIT_CASE
}

// plink2_ld.cc:1085-1090, no FMA contraction possible (multiplies only); var1 belongs to the FIRST variant.
__device__ __forceinline__ bool exceeds(const ldp_pair_stats_t& s, double thresh) {
  const double cov12 = static_cast<double>(static_cast<int64_t>(s.dot) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum2));
  const double var1 = static_cast<double>(static_cast<int64_t>(s.ssq1) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum1));
  const double var2 = static_cast<double>(static_cast<int64_t>(s.ssq2) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum2) * static_cast<int64_t>(s.sum2));
  return __dmul_rn(cov12, cov12) > __dmul_rn(__dmul_rn(thresh, var1), var2);
}

// the same with negative variances read as zero (hypothetical sums of squares below what the sums allow: the low end of an
// interval; true variances are never negative, and the right-hand side is monotone in each of them from zero upwards)
__device__ __forceinline__ bool exceeds_clamped(const ldp_pair_stats_t& s, double thresh) {
  const double cov12 = static_cast<double>(static_cast<int64_t>(s.dot) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum2));
  const int64_t v1 = static_cast<int64_t>(s.ssq1) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum1);
  const int64_t v2 = static_cast<int64_t>(s.ssq2) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum2) * static_cast<int64_t>(s.sum2);
  const double var1 = static_cast<double>((v1 > 0) ? v1 : 0), var2 = static_cast<double>((v2 > 0) ? v2 : 0);
  return __dmul_rn(cov12, cov12) > __dmul_rn(__dmul_rn(thresh, var1), var2);
}

// r^2 of --r2-unphased exactly as ComputeR2 writes it (plink2_ld.cc:6654-6682): NaN when there is no joint
// observation or a zero variance product, else cov01*cov01 / (double(var0)*double(var1)).  The NaN bit
// patterns are the ones the reference's `0.0 / 0.0` produces on x86 (sign bit set).
__device__ __forceinline__ double r2_unphased(const ldp_pair_stats_t& s) {
  const double nan_ref = __longlong_as_double(static_cast<long long>(0xfff8000000000000ull));
  if (!s.nm) {
    return nan_ref;
  }
  const int64_t var0 = static_cast<int64_t>(s.ssq1) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum1);
  const int64_t var1 = static_cast<int64_t>(s.ssq2) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum2) * static_cast<int64_t>(s.sum2);
  const double variance_prod = __dmul_rn(static_cast<double>(var0), static_cast<double>(var1));
  if (variance_prod == 0.0) {
    return nan_ref;
  }
  const double cov01 = static_cast<double>(static_cast<int64_t>(s.dot) * static_cast<int64_t>(s.nm) - static_cast<int64_t>(s.sum1) * static_cast<int64_t>(s.sum2));
  return __ddiv_rn(__dmul_rn(cov01, cov01), variance_prod);
}

// returns true when the prune predicate holds (the caller counts)
__device__ __forceinline__ bool emit_pair(const PairKernelArgs& A, uint32_t i, uint32_t j, uint32_t lo_j, const ldp_pair_stats_t& st) {
  if (A.stats) {
    A.stats[A.pair_off[j] + (i - lo_j)] = st;
  }
  if (A.r2_out || A.r2_hits) {
    if ((j < A.r2_row_first) || (j >= A.r2_row_end) || (i < A.r2_col_first) || (i >= A.r2_col_end)) {
      return false;  // a tile can straddle the edge of the requested rows / columns
    }
    if (A.r2_float == 2) {
      // the pair's six integers themselves, dense layout (r2_tuples_impl: the chrX-weighted r^2 combines two engines' tuples)
      static_cast<ldp_pair_stats_t*>(A.r2_out)[static_cast<uint64_t>(j - A.r2_row_first) * A.r2_ld + (i - A.r2_col_first)] = st;
      return false;
    }
    double r2 = r2_unphased(st);
    if (A.r_signed && (r2 == r2)) {
      // --r-unphased (plink2_ld.cc:9633-9641, :10640-10647): sqrt of the same quotient, negative when the covariance is
      r2 = __dsqrt_rn(r2);
      const int64_t cov = static_cast<int64_t>(st.dot) * static_cast<int64_t>(st.nm) - static_cast<int64_t>(st.sum1) * static_cast<int64_t>(st.sum2);
      bool neg = cov < 0;
      if ((A.r_signed == 2) && cov) {  // (a zero covariance stays +0 in either orientation)
        neg ^= ((A.recs[i].flags ^ A.recs[j].flags) & 1u) != 0;
      }
      if (neg) {
        r2 = -r2;
      }
    }
    if (A.r2_hits) {
      if (fabs(r2) >= A.r2_min) {  // (false for NaN)
        const unsigned long long slot = atomicAdd(&A.counters[3], 1ull);
        if (slot < A.r2_hit_capacity) {
          ldp_r2_hit h;
          h.first = i;
          h.second = j;
          h.r2 = r2;
          A.r2_hits[slot] = h;
        }
      }
      return false;
    }
    // dense rows of the lower triangle (matrix shapes), or the band itself (windowed table)
    const uint64_t idx = A.r2_ld ? (static_cast<uint64_t>(j - A.r2_row_first) * A.r2_ld + (i - A.r2_col_first)) : (A.pair_off[j] - A.r2_band_base + (i - lo_j));
    if (A.r2_float) {
      const float f = (r2 != r2) ? __uint_as_float(0xffc00000u) : static_cast<float>(r2);
      static_cast<float*>(A.r2_out)[idx] = f;
    } else {
      static_cast<double*>(A.r2_out)[idx] = r2;
    }
    return false;
  }
  if (exceeds(st, A.thresh)) {
    atomicOr(&A.pred[A.row_off[j] + ((i >> 5) - (lo_j >> 5))], 1u << (i & 31));
    return true;
  }
  return false;
}

// Early termination test, complete data (see ldp_device.h).  After the chunks before checkpoint `cp` the partial dot
// product of pair (i,j) is dot_p = acc[0] - 2*acc[1].  |N*dot - S_i*S_j| <= |c0| + B with
//   c0 = N*dot_p + a_i*a_j - S_i*S_j,  B = b_i*b_j   (a, b = the checkpoint slot of each variant)
// and the pair cannot reach the threshold when |c0| + B + 1 < t_i*t_j (t = the scaled sqrt(variance numerator)).
// FP64 error budget, for every founder count the matrix pipe accepts (N <= kMfMaxFounders = 4,000,000; the same test serves the
// tile kernels' checkpoints, ldp_pair_wide.hip):
//   * every integer here is below N^2 = 1.6e13 < 2^53: N*dot_p and S_i*S_j are exact, the two fma of c0 round once each at a
//     magnitude <= 4.8e13 (half an ulp there: 2^-8), and the stored slots a = s_R*sqrt(N/n_R), b = sqrt(N*(q_R - s_R^2/n_R)) carry
//     a few ulp of relative error at a magnitude <= N, i.e. <= 2e-9 each, <= 0.02 in a_i*a_j or b_i*b_j.  |c0| + B as computed is
//     therefore within 0.05 of its real value: the "+ 1" on the left is twenty times that.
//   * the right side is scaled by (1 - 1e-6) (cp_tv_scale), ten orders of magnitude more than the ~1e-15 relative rounding of t_i*t_j
//     AND of the reference's own predicate cov^2 > thr*var1*var2 (three multiplications, plink2_ld.cc:1085-1090): a pair dropped
//     here has |cov| < (1 - 1e-6) sqrt(thr var1 var2) - 0.95 in exact arithmetic, which no rounding of that predicate can turn true.
//   * the partial sums a checkpoint needs (s_R back from a * sqrt(n_R / N)) are integers recovered by rint() from a value that is off
//     by <= 4e6 * 5e-16: exact.
// The budget would hold up to N ~ 3e7 (N^2 < 2^53 / 8); the accumulators' own limit (4 N < 2^24) is what sets kMfMaxFounders.
__device__ __forceinline__ bool pair_hopeless(const PairKernelArgs& A, const uint32_t (&c)[2], const cp_slot& ci, const cp_slot& gi, const cp_slot& cj,
                                              const cp_slot& gj) {
  const double dot_p = static_cast<double>(static_cast<int32_t>(c[0] - 2 * c[1]));
  const double c0 = fma(static_cast<double>(A.founder_ct), dot_p, fma(ci.a, cj.a, -(gi.a * gj.a)));
  const double bound = fabs(c0) + fma(ci.b, cj.b, 1.0);
  return bound < gi.b * gj.b;
}


// ---- rows with a few missing calls: what per-variant counts say about a pair's pairwise-complete statistics ----
// x = 0 where a call is missing, so the product the matrix pipe accumulates in the +-2 coding is already the exact `dot` of
// ComputeIndepPairwiseR2Components (plink2_ld.cc:699-723); open are nm, the two sums and the two sums of squares.  With a row's calls
// split by value -- plus (x = +1), minus (x = -1), het (x = 0); nm0 = their total, M = N - nm0 its missing calls -- and (u, w, a) of
// row i's plus / minus / het calls sitting on samples where row j is missing:
//     nm = nm0_i - (u + w + a),   sum1 = S_i - u + w,   ssq1 = Q_i - u - w          (S = plus - minus, Q = plus + minus)
// with 0 <= u <= min(M_j, plus_i), 0 <= w <= min(M_j, minus_i), 0 <= a <= min(M_j, het_i).  var1 = ssq1 nm - sum1^2 is multilinear
// in (u, w, a) and falls in each of them over that whole box (d/du = -((het - a) + 4 (minus - w)), d/dw = -((het - a) + 4 (plus - u)),
// d/da = -ssq1), so its minimum over the box is the far corner and its maximum the near one -- tighter than independent intervals
// for the three sums exactly where those are widest (a rare allele: removing hom-major calls hardly moves the variance, and there
// are few others to remove).  cov = dot nm - sum1 sum2 takes plain interval arithmetic over nm in [N - M_i - M_j, N - max(M_i, M_j)]
// and the two sums' ranges.  Every quantity is an integer below 2^53 (N <= kMfMaxFounders), so the bounds are exact in FP64 until
// the squares and the triple product of the comparison: 1e-9 relative slack + 1 covers those roundings and the reference's own.
// All inputs in ONE orientation per row (dot's): the records' (major allele) in the epilogues, the image's at a checkpoint.
struct SparseRow {
  double nm0, plus, minus, het;
};
__device__ __forceinline__ SparseRow sparse_row_of(const ldp_variant_rec& r) {
  SparseRow s;
  s.nm0 = static_cast<double>(r.nm_ct);
  s.plus = 0.5 * (static_cast<double>(r.ssq) + static_cast<double>(r.sum));
  s.minus = 0.5 * (static_cast<double>(r.ssq) - static_cast<double>(r.sum));
  s.het = static_cast<double>(r.nm_ct) - static_cast<double>(r.ssq);
  return s;
}
// the whole-row cp_gen_slot of the count pass (calls, sum z, sum z^2 with z = 0 / 1 / 2 for codes 00 / 01 / 10): the image's orientation
__device__ __forceinline__ SparseRow sparse_row_of(const cp_gen_slot& g) {
  SparseRow s;
  const double n2 = 0.5 * (static_cast<double>(g.zq_r) - static_cast<double>(g.zs_r));
  s.nm0 = static_cast<double>(g.nm_r);
  s.het = static_cast<double>(g.zs_r) - 2.0 * n2;
  s.minus = n2;
  s.plus = s.nm0 - s.het - n2;
  return s;
}
// dot is known to lie in [d_lo, d_hi] (integers; equal in the epilogues).  Returns 0: the reference's predicate is false whatever the
// open statistics are, 1: it is true, 2: open (the caller recounts the pair, or keeps it alive at a checkpoint).
__device__ __forceinline__ int sparse_decide(double thresh, double N, double d_lo, double d_hi, const SparseRow& I, const SparseRow& J) {
  const double Mi = N - I.nm0, Mj = N - J.nm0;
  const double n_lo = fmax(N - Mi - Mj, 0.0), n_hi = fmin(I.nm0, J.nm0);
  const double u1 = fmin(Mj, I.plus), w1 = fmin(Mj, I.minus), a1 = fmin(Mj, I.het);
  const double u2 = fmin(Mi, J.plus), w2 = fmin(Mi, J.minus), a2 = fmin(Mi, J.het);
  const double Si = I.plus - I.minus, Sj = J.plus - J.minus, Qi = I.plus + I.minus, Qj = J.plus + J.minus;
  const double s1_lo = Si - u1, s1_hi = Si + w1, s2_lo = Sj - u2, s2_hi = Sj + w2;
  // cov = dot nm - sum1 sum2
  const double a_lo = fmin(d_lo * n_lo, d_lo * n_hi), a_hi = fmax(d_hi * n_lo, d_hi * n_hi);  // (nm >= 0)
  const double p0 = s1_lo * s2_lo, p1 = s1_lo * s2_hi, p2 = s1_hi * s2_lo, p3 = s1_hi * s2_hi;
  const double b_lo = fmin(fmin(p0, p1), fmin(p2, p3)), b_hi = fmax(fmax(p0, p1), fmax(p2, p3));
  const double c_lo = a_lo - b_hi, c_hi = a_hi - b_lo;
  const double c2_hi = fmax(c_lo * c_lo, c_hi * c_hi);
  const double c2_lo = ((c_lo <= 0.0) && (c_hi >= 0.0)) ? 0.0 : fmin(c_lo * c_lo, c_hi * c_hi);
  // var = ssq nm - sum^2: the far and the near corner of the box
  const double f1 = Si - u1 + w1, f2 = Sj - u2 + w2;
  const double v1_lo = fmax((Qi - u1 - w1) * (I.nm0 - u1 - w1 - a1) - f1 * f1, 0.0), v1_hi = Qi * I.nm0 - Si * Si;
  const double v2_lo = fmax((Qj - u2 - w2) * (J.nm0 - u2 - w2 - a2) - f2 * f2, 0.0), v2_hi = Qj * J.nm0 - Sj * Sj;
  const double rhs_lo = thresh * v1_lo * v2_lo, rhs_hi = thresh * v1_hi * v2_hi;
  if (c2_lo > rhs_hi * (1.0 + 1e-9) + 1.0) {
    return 1;
  }
  if (c2_hi * (1.0 + 1e-9) + 1.0 < rhs_lo) {
    return 0;
  }
  return 2;
}

// the five pairwise-complete counts of (i, j) by the whole wave, from the two rows of the code image; every lane returns the
// same tuple (dot is the caller's).  The integers are in major-allele orientation, like the records: si / sj = the rows'
// ALT-major flags.
__device__ __forceinline__ ldp_pair_stats_t wave_pair_counts(const PairKernelArgs& A, uint32_t i, uint32_t j, int32_t dot, uint32_t lane, bool alt_i, bool alt_j) {
  // 16-byte loads, two per row in flight (a lone wave is latency-bound here)
  const uint4* __restrict__ r1 = reinterpret_cast<const uint4*>(A.codes + static_cast<uint64_t>(i) * A.code_row_bytes);
  const uint4* __restrict__ r2 = reinterpret_cast<const uint4*>(A.codes + static_cast<uint64_t>(j) * A.code_row_bytes);
  const uint32_t n_quads = static_cast<uint32_t>(A.code_row_bytes / 16);
  uint32_t c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
  const uint4 pad = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);  // "missing": counts nothing
  for (uint32_t q = lane; q < n_quads; q += 128) {
    const uint32_t qb = q + 64;
    const bool two = qb < n_quads;
    const uint4 w1a = r1[q], w2a = r2[q];
    uint4 w1b = r1[two ? qb : q], w2b = r2[two ? qb : q];
    if (!two) {
      w1b = pad;
      w2b = pad;
    }
    // sixteen samples per dword at the even bit positions: h = homozygous, p = hom-REF (x = +1), n = call present
#define LDP_SPARSE_COUNT(W1, W2)                                                                  \
  {                                                                                               \
    const uint32_t h1 = ~(W1) & 0x55555555u, h2 = ~(W2) & 0x55555555u;                             \
    const uint32_t p1 = h1 & ~((W1) >> 1), p2 = h2 & ~((W2) >> 1);                                 \
    const uint32_t n1 = ~((W1) & ((W1) >> 1)) & 0x55555555u, n2 = ~((W2) & ((W2) >> 1)) & 0x55555555u; \
    c2 += __popc(n1 & n2);                                                                        \
    c3 += __popc(n1 & h2);                                                                        \
    c4 += __popc(n1 & p2);                                                                        \
    c5 += __popc(n2 & h1);                                                                        \
    c6 += __popc(n2 & p1);                                                                        \
  }
    LDP_SPARSE_COUNT(w1a.x, w2a.x)
    LDP_SPARSE_COUNT(w1a.y, w2a.y)
    LDP_SPARSE_COUNT(w1a.z, w2a.z)
    LDP_SPARSE_COUNT(w1a.w, w2a.w)
    LDP_SPARSE_COUNT(w1b.x, w2b.x)
    LDP_SPARSE_COUNT(w1b.y, w2b.y)
    LDP_SPARSE_COUNT(w1b.z, w2b.z)
    LDP_SPARSE_COUNT(w1b.w, w2b.w)
#undef LDP_SPARSE_COUNT
  }
  c2 = __builtin_amdgcn_readfirstlane(wave_reduce_add(c2));  // (the sum lands in lane 0)
  c3 = __builtin_amdgcn_readfirstlane(wave_reduce_add(c3));  // (the sum lands in lane 0)
  c4 = __builtin_amdgcn_readfirstlane(wave_reduce_add(c4));  // (the sum lands in lane 0)
  c5 = __builtin_amdgcn_readfirstlane(wave_reduce_add(c5));  // (the sum lands in lane 0)
  c6 = __builtin_amdgcn_readfirstlane(wave_reduce_add(c6));  // (the sum lands in lane 0)
  ldp_pair_stats_t st;
  st.nm = c2;
  st.ssq2 = c3;
  st.sum2 = static_cast<int32_t>(2 * c4 - c3) * (alt_j ? -1 : 1);
  st.ssq1 = c5;
  st.sum1 = static_cast<int32_t>(2 * c6 - c5) * (alt_i ? -1 : 1);
  st.dot = dot;
  return st;
}

}  // namespace ldp
#endif
