// ldp_pred_csr.hip -- the predicate rows of a launch group, returned as the words that are NOT zero.
//
// The reference's worker writes a removed bit where it decides it (plink2_ld.cc:1093-1097); here the pair kernels decide the predicate
// cov^2 > thr var1 var2 of every candidate pair into dense bit rows in HBM (row j: bit i - 32 (lo[j] >> 5) of its words) and the host replays the
// greedy scan from them.  At config 3's density a share's rows are 270 MB of which ~3e6 words hold a bit: copying them back whole is 5 ms
// of PCIe behind the last kernel of a step.  pred_compact_kernel turns the rows of one launch group into CSR -- per row (offset, count),
// per non-zero word (word index inside the row, bits), ascending -- written STRAIGHT into pinned host memory (the entries stream out in
// coalesced 512-byte runs; nothing is copied afterwards), and the replay (ldp_engine_run.cpp: PredView) walks a row's entries instead of its
// words.  The entries' buffer has a fixed capacity; a run whose rows hold more non-zero words than that raises `overflow` and the host falls
// back to the dense copy.
#include "ldp_device.h"
#include "ldp_pair_device.h"

namespace ldp {

namespace {

constexpr uint32_t kCsrRowsPerBlock = 64;  // 4 waves x 16 rows

__global__ __launch_bounds__(256) void pred_compact_kernel(PredCsrArgs A) {
  __shared__ uint32_t s_cnt[kCsrRowsPerBlock];
  __shared__ uint32_t s_off[kCsrRowsPerBlock];
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t row0 = A.row_first + blockIdx.x * kCsrRowsPerBlock;
  // pass 1: non-zero words per row
  for (uint32_t q = 0; q < kCsrRowsPerBlock / 4; ++q) {
    const uint32_t r = wave * (kCsrRowsPerBlock / 4) + q;
    const uint32_t j = row0 + r;
    uint32_t cnt = 0;
    if (j < A.row_end) {
      const uint64_t w0 = A.row_off[j], w1 = A.row_off[j + 1];
      for (uint64_t wb = w0; wb < w1; wb += 64) {  // (wave-uniform bounds: every lane takes part in the ballot)
        const uint64_t w = wb + lane;
        const bool nz = (w < w1) && (A.pred[w] != 0);
        cnt += static_cast<uint32_t>(__builtin_popcountll(__ballot(nz)));
      }
    }
    if (lane == 0) {
      s_cnt[r] = cnt;
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (uint32_t r = 0; r < kCsrRowsPerBlock; ++r) {
      s_off[r] = run;
      run += s_cnt[r];
    }
    s_base = run ? atomicAdd(A.counter, static_cast<unsigned long long>(run)) : 0ull;
    if (run && (s_base + run > A.capacity)) {
      *A.overflow = 1u;  // (the host falls back to the dense rows: ldp_engine_run.cpp)
    }
  }
  __syncthreads();
  const unsigned long long base = s_base;
  // pass 2: the entries (the rows come out of the L2 this time)
  for (uint32_t q = 0; q < kCsrRowsPerBlock / 4; ++q) {
    const uint32_t r = wave * (kCsrRowsPerBlock / 4) + q;
    const uint32_t j = row0 + r;
    if (j >= A.row_end) {
      break;
    }
    const uint32_t cnt = s_cnt[r];
    const unsigned long long at = base + s_off[r];
    if (lane == 0) {
      A.meta[j] = make_uint2(static_cast<uint32_t>(at), cnt);
    }
    if (!cnt || (at + cnt > A.capacity)) {
      continue;
    }
    const uint64_t w0 = A.row_off[j], w1 = A.row_off[j + 1];
    uint32_t done = 0;
    for (uint64_t wb = w0; wb < w1; wb += 64) {
      const uint64_t w = wb + lane;
      const uint32_t bits = (w < w1) ? A.pred[w] : 0u;
      const unsigned long long mask = __ballot(bits != 0);
      if (bits) {
        const uint32_t before = static_cast<uint32_t>(__builtin_popcountll(mask & ((1ull << lane) - 1ull)));
        A.ent[at + done + before] = make_uint2(static_cast<uint32_t>(w - w0), bits);
      }
      done += static_cast<uint32_t>(__builtin_popcountll(mask));
    }
  }
}

}  // namespace

hipError_t launch_pred_compact(const PredCsrArgs& a, hipStream_t stream) {
  if (a.row_end <= a.row_first) {
    return hipSuccess;
  }
  const uint32_t rows = a.row_end - a.row_first;
  hipLaunchKernelGGL(pred_compact_kernel, dim3((rows + kCsrRowsPerBlock - 1) / kCsrRowsPerBlock), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace ldp
