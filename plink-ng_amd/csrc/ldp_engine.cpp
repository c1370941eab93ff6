// ldp_engine.cpp -- host runtime behind include/ldprune_hip.h.
//
// Host side of the MI355X-native --indep-pairwise path:
//   * planning: subcontig split (LdPruneSubcontigSplitAll, plink2_ld.cc:2165-2268) and the window
//     iterator (LdPruneNextSubcontig / LdPruneNextWindow, :605-689) are run ONCE up front -- they depend
//     only on positions -- to fix, for every variant j, the first partner index lo[j] it is ever
//     compared with.  The candidate pairs form a band {lo[j] <= i < j}.
//   * tile scheduling: the band is cut into (32 seconds) x (<=128 distances) parallelogram work items
//     for pair_tiles_kernel.
//   * replay: the order-dependent greedy scan (:931-1100) consumes only predicate bits and major-allele
//     frequencies, so it is replayed sequentially on the host from the kernel's bit rows.
// There is deliberately no CPU implementation of the pair statistics here: without a HIP device
// ldp_load_genotypes()/ldp_run() fail with LDP_ERR_GPU.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "ldp_device.h"

#include <dlfcn.h>
#include <unistd.h>
#include <cerrno>
#include <rccl/rccl.h>

using namespace ldp;

namespace {


struct Subcontig {
  uint32_t len;
  uint32_t first;        // global variant index
  uint32_t owner;        // rank
  uint32_t local_first;  // valid when owned
};

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Dynamic work queue over [0, n): fn(task) on up to max_threads host threads (the reference spreads
// subcontigs over a ThreadGroup the same way, plink2_ld.cc:2686-2700).
template <class F>
void parallel_for(uint32_t n, uint32_t max_threads, F fn) {
  uint32_t nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(std::min(nt, max_threads), n));
  if (nt <= 1) {
    for (uint32_t t = 0; t < n; ++t) {
      fn(t);
    }
    return;
  }
  std::atomic<uint32_t> next(0);
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (uint32_t w = 0; w < nt; ++w) {
    pool.emplace_back([&]() {
      for (uint32_t t = next.fetch_add(1); t < n; t = next.fetch_add(1)) {
        fn(t);
      }
    });
  }
  for (std::thread& th : pool) {
    th.join();
  }
}

// The same on threads that stay: the file -> pinned-memory copies of ldp_load_genotypes() come as hundreds of short batches (one per
// 16 MiB slot), and spawning sixteen threads for each cost as much as the copy itself.  One pool per process, created at first use;
// run() is called from one thread at a time per pool user (the engines of a multi-device process take turns through the mutex).
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool* pool = new CopyPool();  // (never destroyed: its threads may outlive main()'s statics)
    return *pool;
  }
  template <class F>
  void run(uint32_t n, uint32_t max_threads, F fn) {
    if (n <= 1 || workers_.empty()) {
      for (uint32_t t = 0; t < n; ++t) {
        fn(t);
      }
      return;
    }
    std::lock_guard<std::mutex> user(user_mu_);
    std::function<void(uint32_t)> f = fn;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f;
      n_ = n;
      next_.store(0);
      done_ = 0;
      active_ = std::min<uint32_t>(std::min<uint32_t>(max_threads, static_cast<uint32_t>(workers_.size())), n);
      ++epoch_;
    }
    cv_.notify_all();
    for (uint32_t t = next_.fetch_add(1); t < n; t = next_.fetch_add(1)) {  // (the caller works too)
      f(t);
    }
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&]() { return done_ == active_; });
    fn_ = nullptr;
  }

 private:
  CopyPool() {
    const uint32_t nt = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    for (uint32_t w = 0; w + 1 < nt; ++w) {
      workers_.emplace_back([this, w]() {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(uint32_t)>* f;
          uint32_t n;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&]() { return (epoch_ != seen) && (w < active_); });
            seen = epoch_;
            f = fn_;
            n = n_;
          }
          for (uint32_t t = next_.fetch_add(1); t < n; t = next_.fetch_add(1)) {
            (*f)(t);
          }
          std::lock_guard<std::mutex> lk(mu_);
          if (++done_ == active_) {
            cv_done_.notify_one();
          }
        }
      });
      workers_.back().detach();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, user_mu_;
  std::condition_variable cv_, cv_done_;
  std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0, active_ = 0, done_ = 0;
  uint64_t epoch_ = 0;
  std::atomic<uint32_t> next_{0};
};

}  // namespace

constexpr int kPairStreams = 1;

// Kernel-selection switches of one engine.  Defaults come from the environment when the engine is created (tuning from a
// shell); ldp_debug_set_option() overrides them per engine, which is what the tests use -- no process-global state.
struct EngineOptions {
  bool early_exit = true;     // LDP_EARLY_EXIT=0: exhaustive pair kernels
  bool pair_mfma = true;      // LDP_PAIR_MFMA=0: popcount kernels instead of the matrix pipe
  double sparse_frac = 0.005; // LDP_PAIR_SPARSE=0 -> 0; LDP_DEBUG_SPARSE_FRAC
  uint32_t wide_min_reach = kWdMinReach;  // band reach (row-blocks) from which a subcontig takes the 8 x 8 tile plan; LDP_DEBUG_WIDE_MIN_REACH
  bool pair_four = true;      // LDP_PAIR_FOUR=0: rows with missing calls always take all six products (prune launches otherwise four)
  bool pair_gu = true;        // option "pair_gu" 0: the four-product form multiplies x and n (rounds 2-3) instead of allele counts and missing flags
  bool four_tiles = true;     // LDP_PAIR_FOUR_TILES=0: the four-product form stays on the parallelogram plan in wide bands too
  uint32_t wide_diag_last = 2; // LDP_DEBUG_WIDE_DIAG_LAST=k: tiles fewer than k tile distances from the diagonal run at the end of their XCD stream (0: plain J order)
};

constexpr uint32_t kStageSlots = 4;  // pinned staging ring of host-memory input

struct ldp_engine {
  ldp_params P;
  EngineOptions opt;
  int device = -1;
  bool gpu_ok = false;
  bool gpu_probed = false;  // bind_gpu() ran (it runs at the first device use, not in ldp_create)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // ---- plan (global indices) ----
  bool planned = false;
  bool matrix_mode = false;  // all-pairs plan for --r2-unphased matrices (no band, no predicate rows)
  bool band_r2_mode = false; // windowed plan for the --r2-unphased table (band of r^2 values, no prune run)
  uint32_t variant_ct = 0;
  std::vector<uint32_t> bps;
  std::vector<Subcontig> subs;
  uint32_t window_max = 0;
  std::vector<uint32_t> lo_global;   // window start per variant (== v for variants outside every subcontig)
  std::vector<uint8_t> batch_end;    // 1 = a window batch ends with this variant

  // ---- shard (local indices = owned subcontigs concatenated) ----
  uint32_t rank = 0, world = 1;
  uint32_t local_ct = 0;
  std::vector<uint32_t> owned;            // subcontig ids
  std::vector<uint32_t> local_to_global;
  std::vector<int64_t> global_to_local;   // -1 = not owned
  struct OwnedRun {
    uint32_t g_first, g_end;  // global variants [g_first, g_end): owned, consecutive locally too
  };
  std::vector<OwnedRun> owned_runs;        // sorted; what a load call walks instead of the variants
  std::vector<uint32_t> lo_local;
  std::vector<uint64_t> row_off;          // local_ct + 1
  std::vector<uint64_t> pair_off;         // local_ct + 1
  uint64_t pred_words = 0;
  uint64_t cand_pairs = 0;
  uint64_t computed_pairs = 0;
  std::vector<WorkItem> items;
  uint32_t max_rows = 0;                   // largest LDS row count over the work items
  // Pair-kernel launch groups: runs of J-tiles in item order.  A group is launched (on a side stream) as
  // soon as every variant below need_end has been converted, i.e. while prepare_kernel is still working on the
  // variants behind it: the HBM-bound conversion and the VALU-bound pair kernel overlap.
  struct PairGroup {
    uint32_t item_first = 0, item_ct = 0;
    uint32_t need_end = 0;               // local variants [0, need_end) must be loaded
    uint64_t word_first = 0, word_end = 0;  // predicate words the group's J-tiles own
    uint32_t mf_first = 0, mf_ct = 0;    // the same J range as matrix-pipe workgroups (mf_wgs) ...
    uint32_t mf_diag_ct = 0;             // ... of which the first mf_diag_ct are all-diagonal (partition_diag)
    uint32_t wd_first = 0, wd_ct = 0;    // ... and as wide-band tiles (wd_tiles)
    uint32_t wl_first = 0, wl_ct = 0;    // ... in launch order (wd_launch: eight XCD streams, padded to equal length)
    bool four_tiles = false;             // the group's last launch queued pair_mfma_tile4_kernel for them
    bool launched = false;
    hipEvent_t ev_ready = nullptr;
    hipEvent_t ev_done = nullptr;         // kernels finished and the group's predicate words are back on the host
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // popcount fast / general, matrix pipe complete | general
  };
  std::vector<PairGroup> groups;
  // Matrix-pipe plan of the same band (ldp_pair_mfma.hip): used for complete data, founder_ct <= kMfMaxFounders
  bool mf_enabled = false;
  uint32_t r_signed = 0;                  // ldp_set_r_signed
  std::vector<MfmaWG> mf_wgs;
  std::vector<MfmaTile> wd_tiles;         // the 8 x 8 tile plan of the wide-band subcontigs (ldp_pair_wide.hip), in J order
  std::vector<MfmaTile> wd_launch;        // the same tiles as the device gets them: per launch group eight XCD streams (see build_shard)
  uint64_t mf_products = 0;               // 32 x 32 block products of the plan
  uint32_t next_group = 0;                 // groups before this one are launched for the current load epoch
  uint32_t loaded_prefix = 0;              // local variants [0, loaded_prefix) were loaded in the current epoch
  uint32_t load_epoch = 1;
  std::vector<uint32_t> load_tag;          // local: epoch of the last load

  // ---- data ----
  uint32_t chunks = 0;
  uint64_t row_dwords = 0;
  std::vector<uint8_t> loaded;            // local
  std::vector<ldp_variant_rec> recs;      // local (host mirror)
  bool recs_host_valid = false;
  bool recs_copy_queued = false;
  std::vector<double> maj_freq;           // local
  std::vector<uint8_t> mf_set;            // local: 0 unset, 1 caller-supplied, 2 to be derived from device counts, 3 derived
  std::vector<uint64_t> preferred;        // global bitmap (may be empty)

  // ---- device ----
  // The resident genotype image: 2-bit codes for the matrix-pipe kernels (ldp_device.h; the default), hom / ref2het bit-planes
  // for the popcount kernels (more than kMfMaxFounders founders, or pair_mfma switched off).  Exactly one of the two exists.
  bool codes_format = false;
  uint8_t* d_codes = nullptr;
  uint64_t code_row_bytes = 0;
  uint32_t* d_planes = nullptr;
  ldp_variant_rec* d_recs = nullptr;
  uint32_t* d_lo = nullptr;
  uint64_t* d_row_off = nullptr;
  uint64_t* d_pair_off = nullptr;
  uint32_t* d_pred = nullptr;
  WorkItem* d_items = nullptr;
  uint8_t* d_item_general = nullptr;
  unsigned long long* d_counters = nullptr;
  cp_slot* d_cp_stats = nullptr;           // per-variant checkpoint statistics (early termination)
  cp_gen_slot* d_cp_gen = nullptr;         // ... for tiles with missing calls
  MfmaWG* d_mf_wgs = nullptr;
  MfmaTile* d_wd_tiles = nullptr;          // wd_launch
  MfmaTile* d_wd_tiles_plain = nullptr;    // wd_tiles (J order), when the two differ
  MissStats* d_miss_stats = nullptr;       // [slot of d_route]: missing calls of the resident rows a launch reads (summed from the records when the launch is queued)
  uint32_t* d_route = nullptr;             // [g]: which matrix-pipe kernel owns launch group g (route_kernel, when the group is queued); [groups]: other launches
  uint32_t checkpoint_chunk[kCheckpoints];
  uint32_t n_checkpoints = 0;
  uint32_t* h_pred = nullptr;  // pinned
  unsigned long long* h_counters_pin = nullptr;  // pinned: a pageable destination would make the 'async' copy block the host
  bool plan_uploaded = false;
  bool recs_registered = false;
  hipEvent_t prep_ev0 = nullptr, prep_ev1 = nullptr;
  hipStream_t copy_stream = nullptr;
  // One side stream: groups run back to back (two streams gave the same step time and made every launch's
  // duration overlap its neighbour's, i.e. unreadable in a profile).
  hipStream_t pair_stream[kPairStreams] = {nullptr};
  hipEvent_t pair_tail[kPairStreams] = {nullptr};  // last thing queued on each pair stream
  bool pair_tail_set[kPairStreams] = {false};
  uint8_t* h_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};  // pinned staging ring for host-memory genotype input
  uint8_t* d_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_done[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t h2d_stream[2] = {nullptr, nullptr};   // H2D copies of alternate slots (two SDMA queues: one tops out near 30 GB/s)
  hipEvent_t copied[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  bool prep_pending = false;
  // sample-mapped rows (ldp_set_sample_map): column f <- sample (map & 0x7fffffff), bit 31 = het becomes missing
  std::vector<uint32_t> sample_map;
  uint32_t map_raw_sample_ct = 0;
  uint32_t* d_sample_map = nullptr;
  uint8_t* d_gather = nullptr;      // gathered 2-bit rows of one conversion launch
  size_t gather_bytes = 0;
  uint32_t* d_extra_het = nullptr;  // per variant of that launch
  size_t extra_het_cap = 0;
  // ldp_load_pgen_records(): device scratch of one launch (bytes, record descriptors, decoded rows, per-record outputs) and the
  // most recent non-LD row, kept for a call that continues where this one stopped
  struct DecodeScratch {
    void* ptr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  } dec;
  uint8_t* h_dec_pin = nullptr;  // pinned: the launch's descriptors going up, its per-record results coming down (pageable copies cost ~0.2 ms each)
  size_t dec_pin_cap = 0;
  uint8_t* d_ld_base = nullptr;
  size_t ld_base_cap = 0;
  bool ld_base_valid = false;
  uint32_t dec_next_variant = 0;  // the call that may use d_ld_base starts here
  uint64_t dec_next_offset = 0;        // ... and the file offset right behind that call's last record

  ldp_counters ctr;

  ldp_engine() { memset(&ctr, 0, sizeof(ctr)); }
};

namespace {

// The main stream (conversion) outranks the pair streams: conversion blocks are short and HBM-bound, and the sooner
// they are through the sooner the host has the per-variant records it needs to start replaying finished groups.
hipError_t create_stream(hipStream_t* out, bool high_priority) {
  int lo = 0, hi = 0;
  static const bool flat = (getenv("LDP_STREAM_PRIORITY") != nullptr) && (strcmp(getenv("LDP_STREAM_PRIORITY"), "0") == 0);
  if (flat || (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) || (lo == hi)) {
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  return hipStreamCreateWithPriority(out, hipStreamNonBlocking, high_priority ? hi : lo);
}

int fail(ldp_engine* e, int code, const std::string& msg) {
  if (e) {
    e->err = msg;
  }
  return code;
}

int hipfail(ldp_engine* e, hipError_t rc, const char* what) {
  if (rc == hipErrorOutOfMemory) {
    return fail(e, LDP_ERR_NOMEM, std::string(what) + ": " + hipGetErrorString(rc));
  }
  return fail(e, LDP_ERR_GPU, std::string(what) + ": " + hipGetErrorString(rc));
}

#define HIP_TRY(e, call)                              \
  do {                                                \
    hipError_t rc__ = (call);                         \
    if (rc__ != hipSuccess) {                         \
      return hipfail((e), rc__, #call);               \
    }                                                 \
  } while (0)

constexpr size_t kStageBytes = 16ull << 20;  // per slot: pinning host memory costs ~0.3 ms per MiB on the GPU box, and a 16 MiB copy is 0.3 ms of PCIe

// timing events of one launch, released on every exit path
template <int N>
struct EventSet {
  hipEvent_t ev[N];
  EventSet() {
    for (hipEvent_t& x : ev) {
      x = nullptr;
    }
  }
  ~EventSet() {
    for (hipEvent_t x : ev) {
      if (x) {
        (void)hipEventDestroy(x);
      }
    }
  }
  hipError_t create() {
    for (hipEvent_t& x : ev) {
      const hipError_t rc = hipEventCreate(&x);
      if (rc != hipSuccess) {
        return rc;
      }
    }
    return hipSuccess;
  }
};

// temporary device allocation released on every exit path
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) {
      (void)hipFree(p);
    }
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

void free_device(ldp_engine* e) {
  if (!e->gpu_ok) {
    return;
  }
  (void)hipFree(e->d_planes);
  (void)hipFree(e->d_codes);
  e->d_codes = nullptr;
  (void)hipFree(e->d_recs);
  (void)hipFree(e->d_lo);
  (void)hipFree(e->d_row_off);
  (void)hipFree(e->d_pair_off);
  (void)hipFree(e->d_pred);
  (void)hipFree(e->d_items);
  (void)hipFree(e->d_item_general);
  (void)hipFree(e->d_counters);
  (void)hipFree(e->d_cp_stats);
  (void)hipFree(e->d_cp_gen);
  (void)hipFree(e->d_mf_wgs);
  (void)hipFree(e->d_wd_tiles);
  e->d_wd_tiles = nullptr;
  (void)hipFree(e->d_wd_tiles_plain);
  e->d_wd_tiles_plain = nullptr;
  (void)hipFree(e->d_miss_stats);
  (void)hipFree(e->d_route);
  e->d_mf_wgs = nullptr;
  e->d_miss_stats = nullptr;
  e->d_route = nullptr;
  if (e->h_pred) {
    (void)hipHostFree(e->h_pred);
  }
  if (e->h_counters_pin) {
    (void)hipHostFree(e->h_counters_pin);
    e->h_counters_pin = nullptr;
  }
  if (e->recs_registered) {
    (void)hipHostUnregister(e->recs.data());
    e->recs_registered = false;
  }
  for (int k = 0; k < 2; ++k) {
    if (e->h2d_stream[k]) {
      (void)hipStreamDestroy(e->h2d_stream[k]);
      e->h2d_stream[k] = nullptr;
    }
  }
  for (uint32_t k = 0; k < kStageSlots; ++k) {
    if (e->h_stage[k]) {
      (void)hipHostFree(e->h_stage[k]);
      (void)hipFree(e->d_stage[k]);
      (void)hipEventDestroy(e->stage_done[k]);
      if (e->copied[k]) {
        (void)hipEventDestroy(e->copied[k]);
      }
      e->h_stage[k] = nullptr;
      e->d_stage[k] = nullptr;
      e->stage_done[k] = nullptr;
      e->copied[k] = nullptr;
    }
  }
  for (int k = 0; k < 8; ++k) {
    (void)hipFree(e->dec.ptr[k]);
    e->dec.ptr[k] = nullptr;
    e->dec.cap[k] = 0;
  }
  (void)hipFree(e->d_ld_base);
  e->d_ld_base = nullptr;
  e->ld_base_cap = 0;
  e->ld_base_valid = false;
  if (e->h_dec_pin) {
    (void)hipHostFree(e->h_dec_pin);
    e->h_dec_pin = nullptr;
    e->dec_pin_cap = 0;
  }
  (void)hipFree(e->d_sample_map);
  (void)hipFree(e->d_gather);
  (void)hipFree(e->d_extra_het);
  e->d_sample_map = nullptr;
  e->d_gather = nullptr;
  e->d_extra_het = nullptr;
  e->gather_bytes = 0;
  e->extra_het_cap = 0;
  if (e->prep_ev0) {
    (void)hipEventDestroy(e->prep_ev0);
    (void)hipEventDestroy(e->prep_ev1);
    e->prep_ev0 = nullptr;
    e->prep_ev1 = nullptr;
  }
  for (ldp_engine::PairGroup& g : e->groups) {
    if (g.ev_ready) {
      (void)hipEventDestroy(g.ev_ready);
      (void)hipEventDestroy(g.ev_done);
      g.ev_ready = nullptr;
      g.ev_done = nullptr;
    }
    for (int q = 0; q < 7; ++q) {
      if (g.ev[q]) {
        (void)hipEventDestroy(g.ev[q]);
        g.ev[q] = nullptr;
      }
    }
    g.launched = false;
  }
  for (int k = 0; k < kPairStreams; ++k) {
    if (e->pair_tail[k]) {
      (void)hipEventDestroy(e->pair_tail[k]);
      e->pair_tail[k] = nullptr;
    }
    e->pair_tail_set[k] = false;
  }
  e->next_group = 0;
  e->loaded_prefix = 0;
  e->prep_pending = false;
  e->d_planes = nullptr;
  e->d_recs = nullptr;
  e->d_lo = nullptr;
  e->d_row_off = nullptr;
  e->d_pair_off = nullptr;
  e->d_pred = nullptr;
  e->d_items = nullptr;
  e->d_item_general = nullptr;
  e->d_counters = nullptr;
  e->d_cp_stats = nullptr;
  e->d_cp_gen = nullptr;
  e->h_pred = nullptr;
  e->plan_uploaded = false;
}

// LdPruneSubcontigSplitAll, plink2_ld.cc:2165-2268 (every variant included).
void subcontig_split(const uint32_t* chr_idx, const uint32_t* bps, uint32_t variant_ct, uint32_t window, std::vector<Subcontig>* subs, uint32_t* window_max_out) {
  // What LdPruneSubcontigSplitAll (plink2_ld.cc:2165-2268) produces, derived from its definition rather than its loop:
  //   * a subcontig is a maximal run of variants of one chromosome in which consecutive positions are at most `window`
  //     apart (bp windows; no two variants further apart can ever share a window), or the whole chromosome (count windows);
  //     runs of a single variant are never loaded and do not appear;
  //   * window_max = the largest number of variants any window can hold: for bp windows the longest run
  //     {w .. v} with bps[w] >= bps[v] - window inside one chromosome (at least 1), for count windows the largest
  //     chromosome capped at the window size.
  subs->clear();
  auto push = [&](uint32_t first, uint32_t len) {
    if (len < 2) {
      return;
    }
    Subcontig s;
    s.len = len;
    s.first = first;
    s.owner = 0;
    s.local_first = 0;
    subs->push_back(s);
  };
  uint32_t window_max = bps ? 1 : 0;
  for (uint32_t chr_first = 0; chr_first < variant_ct;) {
    uint32_t chr_end = chr_first + 1;
    while ((chr_end < variant_ct) && (chr_idx[chr_end] == chr_idx[chr_first])) {
      ++chr_end;
    }
    if (!bps) {
      push(chr_first, chr_end - chr_first);
      if (chr_end - chr_first > 1) {
        window_max = std::max(window_max, std::min(chr_end - chr_first, window));
      }
    } else {
      // runs between gaps wider than the window
      uint32_t run_first = chr_first;
      for (uint32_t v = chr_first + 1; v < chr_end; ++v) {
        if (bps[v] - bps[v - 1] > window) {
          push(run_first, v - run_first);
          run_first = v;
        }
      }
      push(run_first, chr_end - run_first);
      // widest window: two pointers over the chromosome (a chromosome of one variant leaves window_max alone)
      if (chr_end - chr_first > 1) {
        uint32_t w = chr_first;
        for (uint32_t v = chr_first; v < chr_end; ++v) {
          const uint32_t reach = (bps[v] > window) ? (bps[v] - window) : 0;
          while (bps[w] < reach) {
            ++w;
          }
          window_max = std::max(window_max, v - w + 1);
        }
      }
    }
    chr_first = chr_end;
  }
  *window_max_out = window_max;
}

// LdPruneNextSubcontig + LdPruneNextWindow (plink2_ld.cc:605-689) reduced to what they decide:
// the sequence of batches [cur, next_end) and the window start each batch is scanned against.
void plan_subcontig(const ldp_engine* e, const Subcontig& s, std::vector<uint32_t>* lo, std::vector<uint8_t>* batch_end) {
  const uint32_t* bps = e->P.window_is_bp ? e->bps.data() : nullptr;
  const uint32_t W = e->P.prune_window_size;
  const uint32_t incr = e->P.prune_window_incr;
  const uint32_t first = s.first;
  const uint32_t end = s.first + s.len;
  uint32_t window_start = first;
  uint32_t winstart_v = first;
  uint32_t winend_v = first;
  uint32_t next_end;
  if (bps) {
    const uint32_t bp_thresh = bps[first] + W;
    uint32_t first_window_len = 1;
    do {
      ++winend_v;
    } while ((bps[winend_v] <= bp_thresh) && (++first_window_len < s.len));
    next_end = first + first_window_len;
  } else {
    next_end = first + std::min(s.len, W);
  }
  uint32_t cur = first;
  while (true) {
    for (uint32_t j = cur; j < next_end; ++j) {
      (*lo)[j] = window_start;
    }
    (*batch_end)[next_end - 1] = 1;
    cur = next_end;
    if (next_end == end) {
      break;
    }
    if (bps) {
      const uint32_t start_min_bp = bps[winend_v] - W;
      uint32_t start_bp;
      do {
        ++window_start;
        ++winstart_v;
        start_bp = bps[winstart_v];
      } while (start_bp < start_min_bp);
      const uint32_t end_thresh = start_bp + W;
      do {
        if (++next_end == end) {
          break;
        }
        ++winend_v;
      } while (bps[winend_v] <= end_thresh);
    } else {
      window_start += incr;
      next_end = std::min(window_start + W, end);
    }
  }
}

// ---- early termination planning (ldp_device.h) -----------------------------------------------------------
// A pair of unrelated variants becomes provably hopeless once the unvisited share of the samples drops below
// ~sqrt(thresh): checkpoint fractions start just past 1 - sqrt(thresh) and spread out from there.
EngineOptions options_from_env() {
  EngineOptions o;
  const char* ee = getenv("LDP_EARLY_EXIT");
  o.early_exit = !(ee && (strcmp(ee, "0") == 0));
  const char* m = getenv("LDP_PAIR_MFMA");
  o.pair_mfma = !(m && (strcmp(m, "0") == 0));
  const char* off = getenv("LDP_PAIR_SPARSE");
  const char* f = getenv("LDP_DEBUG_SPARSE_FRAC");
  o.sparse_frac = (off && (strcmp(off, "0") == 0)) ? 0.0 : (f ? atof(f) : 0.005);
  const char* four = getenv("LDP_PAIR_FOUR");
  o.pair_four = !(four && (strcmp(four, "0") == 0));
  const char* ft = getenv("LDP_PAIR_FOUR_TILES");
  o.four_tiles = !(ft && (strcmp(ft, "0") == 0));
  const char* dl = getenv("LDP_DEBUG_WIDE_DIAG_LAST");
  o.wide_diag_last = dl ? static_cast<uint32_t>(std::max(0, atoi(dl))) : 2u;
  if (const char* w = getenv("LDP_DEBUG_WIDE_MIN_REACH")) {
    o.wide_min_reach = static_cast<uint32_t>(std::max(0, atoi(w)));
  }
  return o;
}

int checkpoint_fractions(double r2_param, double* frac) {
  static const double kStep[kCheckpoints] = {0.012, 0.04, 0.08, 0.16, 0.36};  // (tuned on config 2: an earlier first checkpoint pays, a failed one costs little)
  const double f0 = 1.0 - sqrt(r2_param);
  int n = 0;
  if (const char* dbg = getenv("LDP_DEBUG_CP_FRACS")) {  // tuning aid: comma-separated absolute fractions
    while (*dbg && (n < kCheckpoints)) {
      char* end;
      const double f = strtod(dbg, &end);
      if (end == dbg) {
        break;
      }
      frac[n++] = f;
      dbg = (*end == ',') ? end + 1 : end;
    }
    return n;
  }
  for (int k = 0; k < kCheckpoints; ++k) {
    const double f = f0 + kStep[k];
    if (f < 0.93) {
      frac[n++] = f;
    }
  }
  return n;
}

// ---- matrix-pipe plan (ldp_device.h: MfmaWG) ------------------------------------------------------------------
// Row-blocks of 32 variants aligned to the subcontig start; per pair of second-variant blocks (J0, J1) one wave item
// ("parallelogram") for every four block distances the band reaches; wave items are packed four to a workgroup as
// long as the union of their row-blocks fits the LDS ring (J quads x distance pairs for wide bands, four
// neighbouring J pairs for narrow ones).
// Prune launches whose rows miss on average at most this fraction of their calls stay with pair_mfma_kernel and its interval
// epilogue: 0.5 % by default -- on the benchmark generator the pairs the intervals leave open cost as much as
// the six-product kernel at 0.65 % (profiles/r02_experiments.md).  Rows with more than twice the fraction count as high rows,
// and more than 2 % of those route the launch to the six-product kernel as well (their pairs are mostly open).
// ldp_debug_set_option("pair_sparse", 0) turns the path off, "sparse_frac" sets the fraction (defaults from LDP_PAIR_SPARSE /
// LDP_DEBUG_SPARSE_FRAC at engine creation).
// runs: (first local variant, length) of the row ranges blocks are aligned to (the owned subcontigs; one run over
// everything for the all-pairs plan of --r2-unphased); lo: window start per local variant (nullptr: 0, every earlier
// variant is a partner); only second variants in [j_first, j_end) get products (a row chunk of an r^2 matrix).
// out_tiles (optional): runs whose band reaches wide_min_reach row-blocks ALSO get the wide plan of ldp_pair_wide.hip -- 8 x 8
// block tiles aligned to the run start, J tile by J tile with the V tiles of a J tile consecutive -- and their workgroups here
// are marked (MfmaWG::pad) so that complete-data launches leave them to the tiles.
void plan_mfma_generic(const std::vector<std::pair<uint32_t, uint32_t>>& runs, const uint32_t* lo_of, uint32_t j_first, uint32_t j_end,
                       std::vector<MfmaWG>* out_wgs, uint64_t* out_products, uint32_t i_first = 0, uint32_t i_end = 0xffffffffu,
                       std::vector<MfmaTile>* out_tiles = nullptr, uint32_t wide_min_reach = kWdMinReach) {
  // (i_first / i_end: only first variants in [i_first, i_end) are wanted -- a column block of an r^2 matrix; products whose V
  // block lies outside it are not planned)
  out_wgs->clear();
  *out_products = 0;
  if (out_tiles) {
    out_tiles->clear();
  }
  struct Wave {
    int32_t jv, vv;
    uint32_t jend;
    uint8_t mask;
    uint32_t blocks[7];
    uint8_t used;  // bit u: blocks[u] is read
  };
  std::vector<Wave> pending;
  std::vector<uint32_t> uni;
  auto flush = [&]() {
    if (pending.empty()) {
      return;
    }
    MfmaWG wg;
    memset(&wg, 0, sizeof(wg));
    std::sort(uni.begin(), uni.end());
    wg.n_rb = static_cast<uint32_t>(uni.size());
    for (uint32_t k = 0; k < wg.n_rb; ++k) {
      wg.rb[k] = uni[k];
    }
    for (uint32_t k = wg.n_rb; k < kMfMaxRowBlocks; ++k) {
      wg.rb[k] = uni[0];
    }
    wg.j_lo = 0xffffffffu;
    wg.j_hi = 0;
    bool all_diag = true;
    for (const Wave& pw : pending) {
      all_diag = all_diag && (pw.vv + static_cast<int32_t>(3 * kMfBlock) == pw.jv);
    }
    wg.pad = all_diag ? 2u : 0u;
    for (uint32_t w = 0; w < kMfWaves; ++w) {
      MfmaWaveItem& wi = wg.w[w];
      if (w >= pending.size()) {
        wi.jv = -1;
        wi.vv = 0;
        wi.jend = 0;
        wi.prod_mask = 0;
        continue;
      }
      const Wave& pw = pending[w];
      wi.jv = pw.jv;
      wi.vv = pw.vv;
      wi.jend = pw.jend;
      wi.prod_mask = pw.mask;
      for (int u = 0; u < 7; ++u) {
        wi.slot[u] = 0;
        if (pw.used & (1u << u)) {
          wi.slot[u] = static_cast<uint8_t>(std::lower_bound(uni.begin(), uni.end(), pw.blocks[u]) - uni.begin());
        }
      }
      wg.j_lo = std::min(wg.j_lo, static_cast<uint32_t>(pw.jv));
      wg.j_hi = std::max(wg.j_hi, pw.jend);
      *out_products += static_cast<uint64_t>(__builtin_popcount(pw.mask));
    }
    out_wgs->push_back(wg);
    pending.clear();
    uni.clear();
  };
  auto add = [&](const Wave& w) {
    std::vector<uint32_t> merged = uni;
    for (int u = 0; u < 7; ++u) {
      if ((w.used & (1u << u)) && (std::find(merged.begin(), merged.end(), w.blocks[u]) == merged.end())) {
        merged.push_back(w.blocks[u]);
      }
    }
    static const size_t max_waves = []() {  // tuning aid: fewer wave items per workgroup = fewer row-blocks = a deeper ring
      const char* w = getenv("LDP_DEBUG_MFMA_WAVES");
      const int v = w ? atoi(w) : kMfWaves;
      return static_cast<size_t>(std::min(std::max(v, 1), kMfWaves));
    }();
    if ((pending.size() == max_waves) || (merged.size() > kMfMaxRowBlocks)) {
      flush();
      merged.clear();
      for (int u = 0; u < 7; ++u) {
        if ((w.used & (1u << u)) && (std::find(merged.begin(), merged.end(), w.blocks[u]) == merged.end())) {
          merged.push_back(w.blocks[u]);
        }
      }
    }
    uni.swap(merged);
    pending.push_back(w);
  };
  for (const std::pair<uint32_t, uint32_t>& run : runs) {
    struct {
      uint32_t len;
    } s = {run.second};
    const uint32_t sfirst = run.first;
    const uint32_t nb = (s.len + kMfBlock - 1) / kMfBlock;
    // farthest block distance any second variant of a block reaches (-1: the block holds no candidate pair)
    std::vector<int32_t> reach(nb, -1);
    for (uint32_t v = 0; v < s.len; ++v) {
      const uint32_t j = sfirst + v;
      if ((j < j_first) || (j >= j_end)) {
        continue;
      }
      const uint32_t lo = lo_of ? lo_of[j] : sfirst;
      if (lo < j) {
        const int32_t d = static_cast<int32_t>(v / kMfBlock) - static_cast<int32_t>((lo - sfirst) / kMfBlock);
        reach[v / kMfBlock] = std::max(reach[v / kMfBlock], d);
      }
    }
    auto reach_of = [&](uint32_t b) { return (b < nb) ? reach[b] : -1; };
    const size_t run_wg_first = out_wgs->size();
    bool wide_run = false;
    if (out_tiles) {
      flush();  // (a workgroup never mixes wave items of two runs when some runs are wide: the mark below is per workgroup)
      const int32_t max_reach = reach.empty() ? -1 : *std::max_element(reach.begin(), reach.end());
      wide_run = (max_reach >= static_cast<int32_t>(wide_min_reach));
    }
    const auto block_wanted = [&](uint32_t vblock) {  // does row-block `vblock` of the run hold a wanted first variant?
      const uint64_t v0 = static_cast<uint64_t>(sfirst) + static_cast<uint64_t>(kMfBlock) * vblock;
      return (v0 < i_end) && (v0 + kMfBlock > i_first);
    };
    if (wide_run) {
      const uint32_t nt = (nb + kWdTile - 1) / kWdTile;
      for (uint32_t T = 0; T < nt; ++T) {
        // first V tile any J block of the tile reaches
        int32_t t_lo = static_cast<int32_t>(T);
        for (uint32_t a = 0; a < static_cast<uint32_t>(kWdTile); ++a) {
          const uint32_t ja = T * kWdTile + a;
          if (reach_of(ja) >= 0) {
            t_lo = std::min(t_lo, (static_cast<int32_t>(ja) - reach_of(ja)) / static_cast<int32_t>(kWdTile));
          }
        }
        for (int32_t t = std::max(t_lo, 0); t <= static_cast<int32_t>(T); ++t) {
          MfmaTile tl;
          tl.jv = static_cast<int32_t>(sfirst + kMfBlock * kWdTile * T);
          tl.vv = static_cast<int32_t>(sfirst + kMfBlock * kWdTile * static_cast<uint32_t>(t));
          tl.jend = sfirst + s.len;
          tl.pad = 0;
          tl.mask = 0;
          for (uint32_t a = 0; a < static_cast<uint32_t>(kWdTile); ++a) {
            const uint32_t ja = T * kWdTile + a;
            if (reach_of(ja) < 0) {
              continue;
            }
            for (uint32_t b = 0; b < static_cast<uint32_t>(kWdTile); ++b) {
              const uint32_t vb = static_cast<uint32_t>(t) * kWdTile + b;
              if ((vb <= ja) && (static_cast<int32_t>(ja - vb) <= reach_of(ja)) && block_wanted(vb)) {
                tl.mask |= 1ull << (8 * a + b);
              }
            }
          }
          if (tl.mask) {
            out_tiles->push_back(tl);
          }
        }
      }
    }
    auto make = [&](uint32_t a, uint32_t p, Wave* out) {
      Wave w;
      memset(&w, 0, sizeof(w));
      w.jv = static_cast<int32_t>(sfirst + kMfBlock * a);
      w.vv = w.jv - static_cast<int32_t>(kMfBlock * (4 * p + 3));
      w.jend = std::min(sfirst + s.len, static_cast<uint32_t>(w.jv) + 2 * kMfBlock);
      w.blocks[0] = sfirst + kMfBlock * a;
      w.blocks[1] = sfirst + kMfBlock * (a + 1);
      for (int k = 0; k < 4; ++k) {
        const int32_t off = static_cast<int32_t>(4 * p + 3) - k;  // block distance of both (J0, V_k) and (J1, V_{k+1})
        const auto& v_wanted = block_wanted;
        if ((reach_of(a) >= off) && (static_cast<int32_t>(a) >= off) && v_wanted(a - static_cast<uint32_t>(off))) {
          w.mask |= static_cast<uint8_t>(1u << k);
          w.used |= static_cast<uint8_t>(1u | (1u << (2 + k)));
          w.blocks[2 + k] = sfirst + kMfBlock * (a - static_cast<uint32_t>(off));  // (on the diagonal V3 is J0 itself)
        }
        if ((reach_of(a + 1) >= off) && (static_cast<int32_t>(a + 1) >= off) && v_wanted(a + 1 - static_cast<uint32_t>(off))) {
          w.mask |= static_cast<uint8_t>(1u << (4 + k));
          w.used |= static_cast<uint8_t>(2u | (1u << (3 + k)));
          w.blocks[3 + k] = sfirst + kMfBlock * (a + 1 - static_cast<uint32_t>(off));
        }
      }
      if (p == 0) {
        // on the diagonal the kernel takes V3 / V4 from the J0 / J1 fragments: those blocks must be staged (and have a slot)
        // even when none of their own products is wanted (a column block, or a J0 block whose variants have no partner)
        w.used |= (w.mask & 0x48u) ? 1u : 0u;
        w.used |= (w.mask & 0x80u) ? 2u : 0u;
      }
      *out = w;
      return w.mask != 0;
    };
    for (uint32_t a = 0; a < nb; a += 4) {
      const int32_t r0 = std::max(reach_of(a), reach_of(a + 1));
      const int32_t r1 = std::max(reach_of(a + 2), reach_of(a + 3));
      const uint32_t p0 = (r0 >= 0) ? static_cast<uint32_t>(r0) / 4 + 1 : 0;
      const uint32_t p1 = (r1 >= 0) ? static_cast<uint32_t>(r1) / 4 + 1 : 0;
      for (uint32_t p = 0; p < std::max(p0, p1); p += 2) {
        Wave w;
        for (uint32_t q = 0; q < 2; ++q) {
          if ((p + q < p0) && make(a, p + q, &w)) {
            add(w);
          }
        }
        for (uint32_t q = 0; q < 2; ++q) {
          if ((p + q < p1) && make(a + 2, p + q, &w)) {
            add(w);
          }
        }
      }
    }
    if (out_tiles) {
      flush();
      for (size_t k = run_wg_first; k < out_wgs->size(); ++k) {
        (*out_wgs)[k].pad = ((*out_wgs)[k].pad & 2u) | (wide_run ? 1u : 0u);
      }
    }
  }
  flush();
}

// The workgroups [first, first + ct) with the all-diagonal ones (MfmaWG::pad bit 1) first, J order kept inside either run:
// launch_pair_mfma hands the first run to the single-form instantiation of pair_mfma_kernel and the rest to the general one.
uint32_t partition_diag(std::vector<MfmaWG>* wgs, size_t first, size_t ct) {
  const auto b = wgs->begin() + static_cast<std::ptrdiff_t>(first);
  const auto mid = std::stable_partition(b, b + static_cast<std::ptrdiff_t>(ct), [](const MfmaWG& w) { return (w.pad & 2u) != 0; });
  return static_cast<uint32_t>(mid - b);
}

void plan_mfma(ldp_engine* e) {
  std::vector<std::pair<uint32_t, uint32_t>> runs;
  for (uint32_t sk : e->owned) {
    runs.emplace_back(e->subs[sk].local_first, e->subs[sk].len);
  }
  plan_mfma_generic(runs, e->lo_local.data(), 0, e->local_ct, &e->mf_wgs, &e->mf_products, 0, 0xffffffffu, &e->wd_tiles, e->opt.wide_min_reach);
}

void build_shard(ldp_engine* e) {
  // local index space
  e->owned.clear();
  e->local_to_global.clear();
  e->global_to_local.assign(e->variant_ct, -1);
  e->owned_runs.clear();
  uint32_t local = 0;
  for (uint32_t k = 0; k < e->subs.size(); ++k) {
    Subcontig& s = e->subs[k];
    if (s.owner != e->rank) {
      continue;
    }
    s.local_first = local;
    e->owned.push_back(k);
    if ((!e->owned_runs.empty()) && (e->owned_runs.back().g_end == s.first)) {
      e->owned_runs.back().g_end = s.first + s.len;
    } else if (s.len) {
      e->owned_runs.push_back({s.first, s.first + s.len});
    }
    for (uint32_t v = 0; v < s.len; ++v) {
      e->global_to_local[s.first + v] = local + v;
      e->local_to_global.push_back(s.first + v);
    }
    local += s.len;
  }
  e->local_ct = local;
  e->lo_local.assign(local, 0);
  e->row_off.assign(static_cast<size_t>(local) + 1, 0);
  e->pair_off.assign(static_cast<size_t>(local) + 1, 0);
  uint64_t words = 0, pairs = 0;
  for (uint32_t k : e->owned) {
    if (e->matrix_mode) {
      break;  // lo = 0 everywhere; tiles are generated per row chunk by ldp_r2_unphased_rows()
    }
    const Subcontig& s = e->subs[k];
    for (uint32_t v = 0; v < s.len; ++v) {
      const uint32_t j = s.local_first + v;
      const uint32_t lo = e->lo_global[s.first + v] - s.first + s.local_first;
      e->lo_local[j] = lo;
      e->row_off[j] = words;
      e->pair_off[j] = pairs;
      if (j > lo) {
        words += ((j - 1) >> 5) - (lo >> 5) + 1;
        pairs += j - lo;
      }
    }
  }
  e->row_off[local] = words;
  e->pair_off[local] = pairs;
  e->pred_words = words;
  e->cand_pairs = pairs;

  // work items: 32 seconds x runs of 8-distance units, <= kMaxUnitsPerBlock units per block
  e->items.clear();
  e->max_rows = 0;
  e->computed_pairs = 0;
  for (uint32_t k : e->owned) {
    if (e->matrix_mode) {
      break;
    }
    const Subcontig& s = e->subs[k];
    const uint32_t sfirst = s.local_first;
    const uint32_t send = s.local_first + s.len;
    for (uint32_t j0 = sfirst; j0 < send; j0 += kTileJ) {
      const uint32_t jend = std::min(j0 + kTileJ, send);
      uint32_t dmax = 0;
      for (uint32_t j = j0; j < jend; ++j) {
        dmax = std::max(dmax, j - e->lo_local[j]);
      }
      if (!dmax) {
        continue;
      }
      const uint32_t units = (dmax + 7) / 8;
      static const uint32_t max_units = []() {  // tuning aid: blocks of fewer units than the kernel's limit
        const char* mu = getenv("LDP_DEBUG_MAX_UNITS");
        const int v = mu ? atoi(mu) : kMaxUnitsPerBlock;
        return static_cast<uint32_t>(std::min(std::max(v, 1), kMaxUnitsPerBlock));
      }();
      const uint32_t blocks = (units + max_units - 1) / max_units;
      const uint32_t base = units / blocks;
      const uint32_t extra = units % blocks;
      uint32_t d0 = 1;
      for (uint32_t blk = 0; blk < blocks; ++blk) {
        const uint32_t u = base + ((blk < extra) ? 1 : 0);
        WorkItem it;
        it.j0 = j0;
        it.jend = jend;
        it.d0 = d0;
        it.units = u;
        it.sfirst = sfirst;
        it.send = send;
        e->items.push_back(it);
        e->max_rows = std::max(e->max_rows, tile_rows(d0, u));
        e->computed_pairs += static_cast<uint64_t>(u) * 8 * kTileJ;
        d0 += 8 * u;
      }
    }
  }
  free_device(e);
  e->mf_enabled = e->opt.pair_mfma && (!e->matrix_mode) && (!e->band_r2_mode) && (e->P.founder_ct <= kMfMaxFounders);
  e->mf_wgs.clear();
  e->wd_tiles.clear();
  // second variants at which a launch group may end: every matrix-pipe workgroup (and every wide-band tile) lies on one side
  std::vector<uint32_t> safe_cut;
  if (e->mf_enabled) {
    plan_mfma(e);
    // J ranges of the wide tiles (the V tiles of one J tile share theirs), ascending
    std::vector<std::pair<uint32_t, uint32_t>> tile_j;
    for (const MfmaTile& t : e->wd_tiles) {
      const uint32_t lo = static_cast<uint32_t>(t.jv), hi = std::min(lo + kMfBlock * kWdTile, t.jend);
      if (tile_j.empty() || (tile_j.back().first != lo)) {
        tile_j.emplace_back(lo, hi);
      }
    }
    auto inside_a_tile = [&](uint32_t c) {
      auto it = std::upper_bound(tile_j.begin(), tile_j.end(), std::make_pair(c, 0xffffffffu));
      if (it == tile_j.begin()) {
        return false;
      }
      --it;
      return (it->first < c) && (c < it->second);
    };
    uint32_t hi = 0;
    for (size_t k = 0; k + 1 < e->mf_wgs.size(); ++k) {
      hi = std::max(hi, e->mf_wgs[k].j_hi);
      if ((e->mf_wgs[k + 1].j_lo >= hi) && !inside_a_tile(e->mf_wgs[k + 1].j_lo)) {
        safe_cut.push_back(e->mf_wgs[k + 1].j_lo);
      }
    }
  }
  auto cut_ok = [&](uint32_t prev_j0, uint32_t next_j0) {
    if (!e->mf_enabled) {
      return true;
    }
    const auto it = std::upper_bound(safe_cut.begin(), safe_cut.end(), prev_j0);
    return (it != safe_cut.end()) && (*it <= next_j0);
  };
  // launch groups: ~kTargetGroups runs of whole J-tiles (a J-tile's blocks share predicate rows)
  e->groups.clear();
  {
    // ~4 groups of decreasing size (40/30/20/10 %): what is exposed after the last kernel is that group's copy back
    // and replay, so it should be the small one
    // (the matrix-pipe kernel runs ~4,000 workgroups of four waves at config 2 on 512 resident slots: small groups leave
    // the chip half empty at their tails -- 1 / 2 / 4 groups: 4.25 / 4.6 / 5.0 ms of kernel, 10.84 / 10.87 / 11.17 ms per
    // step with the replay of all but the last group hidden)
    uint32_t kTargetGroups = e->mf_enabled ? 2 : 4;
    if (e->mf_enabled && !e->wd_tiles.empty()) {
      // Wide bands: a tile runs for the whole length of the rows (2.5 ms at config 3's density, a dozen rounds of them in a
      // 120,000-variant slice), so every launch ends in a tail of most of a round, while the replay of such a share takes well
      // under a millisecond per 100,000 variants: one launch (config-3 slice: 30.6 ms of kernel against 32.3 with two, 34.2 with three)
      uint64_t wide_products = 0;
      for (const MfmaTile& t : e->wd_tiles) {
        wide_products += static_cast<uint64_t>(__builtin_popcountll(t.mask));
      }
      if (2 * wide_products >= e->mf_products) {
        kTargetGroups = 1;
      }
    }
    if (const char* tg = getenv("LDP_DEBUG_GROUPS")) {
      kTargetGroups = std::max(1, atoi(tg));
    }
    const uint32_t n_items = static_cast<uint32_t>(e->items.size());
    const double weight_sum = kTargetGroups * (kTargetGroups + 1) / 2.0;
    uint32_t i0 = 0;
    for (uint32_t k = 0; i0 < n_items; ++k) {
      const double share = (k < kTargetGroups) ? (kTargetGroups - k) / weight_sum : 1.0;
      const uint32_t want = std::max<uint32_t>(512, static_cast<uint32_t>(n_items * share));
      uint32_t i1 = ((k + 1 >= kTargetGroups) || (n_items - i0 <= want)) ? n_items : i0 + want;
      while ((i1 < n_items) && ((e->items[i1].j0 == e->items[i1 - 1].j0) || !cut_ok(e->items[i1 - 1].j0, e->items[i1].j0))) {
        ++i1;
      }
      ldp_engine::PairGroup g;
      g.item_first = i0;
      g.item_ct = i1 - i0;
      g.need_end = e->items[i1 - 1].jend;
      g.word_first = e->row_off[e->items[i0].j0];
      g.word_end = e->row_off[e->items[i1 - 1].jend];
      e->groups.push_back(g);
      i0 = i1;
    }
    // the same J ranges as runs of matrix-pipe workgroups
    uint32_t w0 = 0;
    for (size_t gi = 0; gi < e->groups.size(); ++gi) {
      ldp_engine::PairGroup& g = e->groups[gi];
      const uint32_t j_end = (gi + 1 < e->groups.size()) ? e->items[e->groups[gi + 1].item_first].j0 : 0xffffffffu;
      uint32_t w1 = w0;
      while ((w1 < e->mf_wgs.size()) && (e->mf_wgs[w1].j_lo < j_end)) {
        g.need_end = std::max(g.need_end, e->mf_wgs[w1].j_hi);
        ++w1;
      }
      g.mf_first = w0;
      g.mf_ct = w1 - w0;
      w0 = w1;
    }
    for (ldp_engine::PairGroup& g : e->groups) {
      g.mf_diag_ct = partition_diag(&e->mf_wgs, g.mf_first, g.mf_ct);
    }
    uint32_t t0 = 0;
    for (size_t gi = 0; gi < e->groups.size(); ++gi) {
      ldp_engine::PairGroup& g = e->groups[gi];
      const uint32_t j_end = (gi + 1 < e->groups.size()) ? e->items[e->groups[gi + 1].item_first].j0 : 0xffffffffu;
      uint32_t t1 = t0;
      while ((t1 < e->wd_tiles.size()) && (static_cast<uint32_t>(e->wd_tiles[t1].jv) < j_end)) {
        g.need_end = std::max(g.need_end, std::min(static_cast<uint32_t>(e->wd_tiles[t1].jv) + kMfBlock * kWdTile, e->wd_tiles[t1].jend));
        ++t1;
      }
      g.wd_first = t0;
      g.wd_ct = t1 - t0;
      t0 = t1;
    }
    // Launch order of the tiles.  The kernels hand workgroup b to XCD b % 8 and give XCD x the tiles [x per, (x + 1) per) of the
    // launch's array, in order, so that the tiles running together on an XCD are neighbours and share row-blocks through its L2.
    // That only works while they also walk the samples together -- a (row-block, stage) unit lives in the 4 MB L2 for about two
    // stages -- and tiles of equal length started together do: a laggard hits what the leaders fetched and catches up.  DIAGONAL
    // tiles break it: their near products are the pairs in LD, they run to the end of the rows (1.0 against ~0.6 of the others at
    // r^2 0.2), one in eight tiles, and behind the first of them a stream never re-aligns (profiles/r04_pmc_traffic.json: 6.1 x
    // the compulsory bytes over 148 rounds of config 3's share, L2 hit rate 31 %).  So every stream gets its far tiles first, J tile
    // by J tile, and the tiles next to the diagonal at the end -- the diagonal ones and their first neighbours, which hold the rest
    // of the pairs in LD (config 3's share, kernel ms with the last 0 / 1 / 2 / 3 / 4 tile distances deferred: 330 / 302 / 296 / 297 /
    // 319; HBM traffic 6.1 -> 5.3 x compulsory with 1).  Streams are padded to equal length with empty tiles (mask 0).
    e->wd_launch.clear();
    for (ldp_engine::PairGroup& g : e->groups) {
      g.wl_first = static_cast<uint32_t>(e->wd_launch.size());
      g.wl_ct = 0;
      if (!g.wd_ct) {
        continue;
      }
      std::vector<uint32_t> off, diag;
      for (uint32_t t = g.wd_first; t < g.wd_first + g.wd_ct; ++t) {
        const uint32_t dist = static_cast<uint32_t>(e->wd_tiles[t].jv - e->wd_tiles[t].vv) / (kMfBlock * kWdTile);
        ((dist < e->opt.wide_diag_last) ? diag : off).push_back(t);
      }
      auto chunk = [](size_t n, uint32_t x) { return std::make_pair(n * x / 8, n * (x + 1) / 8); };
      size_t per = 0;
      for (uint32_t x = 0; x < 8; ++x) {
        const auto co = chunk(off.size(), x), cd = chunk(diag.size(), x);
        per = std::max(per, (co.second - co.first) + (cd.second - cd.first));
      }
      MfmaTile empty;
      memset(&empty, 0, sizeof(empty));
      for (uint32_t x = 0; x < 8; ++x) {
        const auto co = chunk(off.size(), x), cd = chunk(diag.size(), x);
        size_t k = 0;
        for (size_t q = co.first; q < co.second; ++q, ++k) {
          e->wd_launch.push_back(e->wd_tiles[off[q]]);
        }
        for (size_t q = cd.first; q < cd.second; ++q, ++k) {
          e->wd_launch.push_back(e->wd_tiles[diag[q]]);
        }
        for (; k < per; ++k) {
          e->wd_launch.push_back(empty);
        }
      }
      g.wl_ct = static_cast<uint32_t>(8 * per);
    }
  }
  e->load_tag.assign(local, 0);
  e->load_epoch = 1;
  e->loaded.assign(local, 0);
  e->recs.assign(local, ldp_variant_rec());
  e->recs_host_valid = false;
  e->maj_freq.assign(local, 0.0);
  e->mf_set.assign(local, 0);
}

// Device selection and stream creation, at the first use of the device: ldp_create() and ldp_set_variants() are host
// work, so a caller can plan while the HIP runtime is still starting up on another thread (plink2-hip does).
void bind_gpu(ldp_engine* e) {
  if (e->gpu_probed) {
    return;
  }
  e->gpu_probed = true;
  const int ndev = ldp_device_count();
  if (ndev > 0) {
    int dev = e->P.device;
    if (dev < 0) {
      if (hipGetDevice(&dev) != hipSuccess) {
        dev = 0;
      }
    }
    if ((dev < ndev) && (hipSetDevice(dev) == hipSuccess)) {
      e->device = dev;
      e->gpu_ok = true;
      if (e->P.stream) {
        e->stream = static_cast<hipStream_t>(e->P.stream);
      } else if (create_stream(&e->stream, true) == hipSuccess) {
        e->own_stream = true;
      } else {
        e->gpu_ok = false;
      }
      if (e->gpu_ok && (create_stream(&e->copy_stream, true) != hipSuccess)) {
        e->gpu_ok = false;
      }
      for (int k = 0; (k < kPairStreams) && e->gpu_ok; ++k) {
        if (create_stream(&e->pair_stream[k], false) != hipSuccess) {
          e->gpu_ok = false;
        }
      }
    }
  }
}

int ensure_device_plan(ldp_engine* e) {
  bind_gpu(e);
  if (!e->gpu_ok) {
    return fail(e, LDP_ERR_GPU, "no usable HIP device");
  }
  if (e->plan_uploaded) {
    return LDP_OK;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  static const bool plan_timing = getenv("LDP_DEBUG_LOAD_TIMING") != nullptr;
  double t_mark = now_ms();
  auto mark = [&](const char* what) {
    if (plan_timing) {
      const double t = now_ms();
      fprintf(stderr, "[plan timing] %-28s %.1f ms\n", what, t - t_mark);
      t_mark = t;
    }
  };
  const uint32_t plane_dwords = (e->P.founder_ct + 31) / 32;
  e->chunks = (plane_dwords + kChunkDwords - 1) / kChunkDwords;
  e->row_dwords = static_cast<uint64_t>(e->chunks) * kRowChunkDwords;
  const size_t n = std::max<size_t>(e->local_ct, 1);
  e->codes_format = e->opt.pair_mfma && (e->P.founder_ct <= kMfMaxFounders);
  e->code_row_bytes = code_row_bytes_of(e->P.founder_ct);
  if (e->codes_format) {
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_codes), n * e->code_row_bytes));
  } else {
    HIP_TRY(e, hipMalloc(&e->d_planes, n * e->row_dwords * sizeof(uint32_t)));
  }
  mark("image hipMalloc");
  HIP_TRY(e, hipMalloc(&e->d_recs, n * sizeof(ldp_variant_rec)));
  HIP_TRY(e, hipMalloc(&e->d_lo, n * sizeof(uint32_t)));
  HIP_TRY(e, hipMalloc(&e->d_row_off, (n + 1) * sizeof(uint64_t)));
  HIP_TRY(e, hipMalloc(&e->d_pair_off, (n + 1) * sizeof(uint64_t)));
  HIP_TRY(e, hipMalloc(&e->d_pred, std::max<size_t>(e->pred_words, 1) * sizeof(uint32_t)));
  HIP_TRY(e, hipMalloc(&e->d_items, std::max<size_t>(e->items.size(), 1) * sizeof(WorkItem)));
  HIP_TRY(e, hipMalloc(&e->d_item_general, std::max<size_t>(e->items.size(), 1)));
  HIP_TRY(e, hipMalloc(&e->d_counters, 4 * sizeof(unsigned long long)));
  HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
  HIP_TRY(e, hipMalloc(&e->d_cp_stats, n * kCpStride * sizeof(cp_slot)));
  HIP_TRY(e, hipMalloc(&e->d_cp_gen, n * kCheckpoints * sizeof(cp_gen_slot)));
  HIP_TRY(e, hipMalloc(&e->d_mf_wgs, std::max<size_t>(e->mf_wgs.size(), 1) * sizeof(MfmaWG)));
  HIP_TRY(e, hipMalloc(&e->d_miss_stats, (e->groups.size() + 1) * sizeof(MissStats)));
  HIP_TRY(e, hipMemsetAsync(e->d_miss_stats, 0, (e->groups.size() + 1) * sizeof(MissStats), e->stream));
  HIP_TRY(e, hipMalloc(&e->d_route, (e->groups.size() + 1) * sizeof(uint32_t)));
  HIP_TRY(e, hipMemsetAsync(e->d_route, 0, (e->groups.size() + 1) * sizeof(uint32_t), e->stream));
  if (!e->mf_wgs.empty()) {
    HIP_TRY(e, hipMemcpyAsync(e->d_mf_wgs, e->mf_wgs.data(), e->mf_wgs.size() * sizeof(MfmaWG), hipMemcpyHostToDevice, e->stream));
  }
  HIP_TRY(e, hipMalloc(&e->d_wd_tiles, std::max<size_t>(e->wd_launch.size(), 1) * sizeof(MfmaTile)));
  if (!e->wd_launch.empty()) {
    HIP_TRY(e, hipMemcpyAsync(e->d_wd_tiles, e->wd_launch.data(), e->wd_launch.size() * sizeof(MfmaTile), hipMemcpyHostToDevice, e->stream));
    if (e->opt.wide_diag_last) {
      HIP_TRY(e, hipMalloc(&e->d_wd_tiles_plain, e->wd_tiles.size() * sizeof(MfmaTile)));
      HIP_TRY(e, hipMemcpyAsync(e->d_wd_tiles_plain, e->wd_tiles.data(), e->wd_tiles.size() * sizeof(MfmaTile), hipMemcpyHostToDevice, e->stream));
    }
  }
  mark("other hipMallocs + uploads");
  // checkpoints for early termination
  e->n_checkpoints = 0;
  for (int k = 0; k < kCheckpoints; ++k) {
    e->checkpoint_chunk[k] = 0xffffffffu;
  }
  if ((e->chunks >= 4) && !e->matrix_mode) {
    double frac[kCheckpoints];
    const int nf = checkpoint_fractions(e->P.prune_last_param, frac);
    for (int k = 0; k < nf; ++k) {
      const uint32_t c = static_cast<uint32_t>(ceil(e->chunks * frac[k]));
      if (c && (c < e->chunks) && (!e->n_checkpoints || c > e->checkpoint_chunk[e->n_checkpoints - 1])) {
        e->checkpoint_chunk[e->n_checkpoints++] = c;
      }
    }
  }
  HIP_TRY(e, hipHostMalloc(&e->h_pred, std::max<size_t>(e->pred_words, 1) * sizeof(uint32_t), hipHostMallocDefault));
  // (4 counters, then the route words of the launch groups + the inspection launch)
  HIP_TRY(e, hipHostMalloc(&e->h_counters_pin, 4 * sizeof(unsigned long long) + (e->groups.size() + 1) * sizeof(uint32_t), hipHostMallocDefault));
  mark("pinned host buffers");
  if (e->local_ct) {
    HIP_TRY(e, hipMemcpyAsync(e->d_lo, e->lo_local.data(), e->local_ct * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_row_off, e->row_off.data(), (e->local_ct + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_pair_off, e->pair_off.data(), (e->local_ct + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  }
  if ((!e->items.empty()) && !e->codes_format) {  // (the popcount work items: host-side bookkeeping only when the matrix pipe runs)
    HIP_TRY(e, hipMemcpyAsync(e->d_items, e->items.data(), e->items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, e->stream));
  }
  mark("plan arrays H2D");
  if (e->local_ct) {
    HIP_TRY(e, hipHostRegister(e->recs.data(), e->recs.size() * sizeof(ldp_variant_rec), hipHostRegisterDefault));
    e->recs_registered = true;
  }
  mark("hipHostRegister(recs)");
  HIP_TRY(e, hipEventCreate(&e->prep_ev0));
  HIP_TRY(e, hipEventCreate(&e->prep_ev1));
  for (ldp_engine::PairGroup& g : e->groups) {
    HIP_TRY(e, hipEventCreateWithFlags(&g.ev_ready, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&g.ev_done, hipEventDisableTiming));
    for (int q = 0; q < 7; ++q) {
      HIP_TRY(e, hipEventCreate(&g.ev[q]));
    }
  }
  for (int k = 0; k < kPairStreams; ++k) {
    HIP_TRY(e, hipEventCreateWithFlags(&e->pair_tail[k], hipEventDisableTiming));
  }
  mark("events");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  mark("stream sync");
  e->plan_uploaded = true;
  return LDP_OK;
}

// Major-allele frequency from the device's allele counts, in the reference's arithmetic:
// freq_ref = ref * (1/tot) (plink2_filter.cc:2144-2147), major = REF iff >= 0.5 (plink2_common.h:559-567),
// GetAlleleFreq for the last allele = max(1 - freq_ref, 0) (plink2_common.h:584-593).
bool derive_maj_freq(ldp_engine* e, uint32_t l) {
  const ldp_variant_rec& r = e->recs[l];
  const uint64_t ref_ct = 2ull * r.n_homref + r.n_het;
  const uint64_t alt_ct = 2ull * r.n_homalt + r.n_het;
  const uint64_t tot = ref_ct + alt_ct;
  double ref_freq = 0.5;
  if (tot) {
    const double tot_recip = 1.0 / static_cast<double>(tot);
    ref_freq = static_cast<double>(ref_ct) * tot_recip;
  }
  const bool alt_major = !(ref_freq >= 0.5);
  if (alt_major != static_cast<bool>(r.flags & 1)) {
    return false;
  }
  double mf = ref_freq;
  if (alt_major) {
    mf = 1.0 - ref_freq;
    if (mf < 0.0) {
      mf = 0.0;
    }
  }
  e->maj_freq[l] = mf;
  return true;
}

// Bring the per-variant records to the host (one D2H per ldp_run, not per load call) and derive the
// major-allele frequencies that are still pending.
int ensure_staging(ldp_engine* e) {
  if (e->h_stage[0]) {
    return LDP_OK;
  }
  const size_t row_bytes = std::max<size_t>((static_cast<size_t>(e->P.founder_ct) + 3) / 4, ldp_phased_row_bytes(e->P.founder_ct)) + 4;
  const size_t bytes = std::max(kStageBytes, row_bytes);
  for (uint32_t k = 0; k < kStageSlots; ++k) {
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_stage[k]), bytes, hipHostMallocDefault));
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_stage[k]), bytes));
    HIP_TRY(e, hipEventCreate(&e->stage_done[k]));
    HIP_TRY(e, hipEventRecord(e->stage_done[k], e->stream));
    HIP_TRY(e, hipEventCreateWithFlags(&e->copied[k], hipEventDisableTiming));
  }
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(e, hipStreamCreateWithFlags(&e->h2d_stream[k], hipStreamNonBlocking));
  }
  return LDP_OK;
}

// Queue the copy of the per-variant records (copy stream, ordered after the last prepare kernel only).  Call this
// BEFORE queueing anything else that ends in a device-to-host copy: the copy engine serves its queue in order, so a
// copy that waits for a kernel holds up every copy submitted after it.
int start_fetch_recs(ldp_engine* e) {
  if (e->recs_host_valid || e->recs_copy_queued || (!e->local_ct)) {
    return LDP_OK;
  }
  if (e->prep_pending) {
    HIP_TRY(e, hipStreamWaitEvent(e->copy_stream, e->prep_ev1, 0));
  }
  HIP_TRY(e, hipMemcpyAsync(e->recs.data(), e->d_recs, e->local_ct * sizeof(ldp_variant_rec), hipMemcpyDeviceToHost, e->copy_stream));
  e->recs_copy_queued = true;
  return LDP_OK;
}

int fetch_recs(ldp_engine* e) {
  if (e->recs_host_valid) {
    return LDP_OK;
  }
  if (e->local_ct) {
    const int rc = start_fetch_recs(e);
    if (rc) {
      return rc;
    }
    const double t0 = now_ms();
    HIP_TRY(e, hipStreamSynchronize(e->copy_stream));
    if (getenv("LDP_DEBUG_TIMELINE")) {
      fprintf(stderr, "recs copy: waited %.2f ms\n", now_ms() - t0);
    }
    e->recs_copy_queued = false;
  }
  if (e->prep_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->prep_ev0, e->prep_ev1) == hipSuccess) {
      e->ctr.ms_prepare = ms;
    }
    e->prep_pending = false;
  }
  std::atomic<int> bad(0);
  const uint32_t kBlock = 65536;
  const uint32_t nblk = (e->local_ct + kBlock - 1) / kBlock;
  parallel_for(nblk, 32, [&](uint32_t blk) {
    const uint32_t lend = std::min(e->local_ct, (blk + 1) * kBlock);
    for (uint32_t l = blk * kBlock; l < lend; ++l) {
      if (e->mf_set[l] == 2) {
        if (!derive_maj_freq(e, l)) {
          bad.store(1);
        }
        e->mf_set[l] = 3;
      }
    }
  });
  if (bad.load()) {
    return fail(e, LDP_ERR_GPU, "device and host disagree on the major allele");
  }
  e->recs_host_valid = true;
  return LDP_OK;
}

// Subcontigs are replayed concurrently and neighbouring ones can share a bitmap word, so bits are set
// atomically; reads only ever look at bits of the reader's own subcontig.
inline void set32(std::vector<uint32_t>& bm, uint32_t i) { __atomic_fetch_or(&bm[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED); }
inline uint32_t load32(const std::vector<uint32_t>& bm, uint32_t w) { return __atomic_load_n(&bm[w], __ATOMIC_RELAXED); }

// next index >= from with a clear bit, or `limit` if none below it
inline uint32_t next_clear(const std::vector<uint32_t>& bm, uint32_t from, uint32_t limit) {
  while (from < limit) {
    const uint32_t w = ~load32(bm, from >> 5) >> (from & 31);
    if (w) {
      const uint32_t r = from + __builtin_ctz(w);
      return (r < limit) ? r : limit;
    }
    from = (from | 31) + 1;
  }
  return limit;
}

// The greedy scan of IndepPairwiseThread (plink2_ld.cc:931-1100) for one subcontig, replayed from
// predicate bits.  R = removed bitmap over local indices (u32 words).  pred row j: bit i of word
// (i>>5)-(lo[j]>>5).
// Resumable at batch boundaries (what carries over is R and first_unchecked): *cursor (nullptr: the subcontig's start) is the
// first variant not replayed yet, and only batches whose variants all lie below `covered` -- complete predicate rows -- run.
uint64_t replay_subcontig(const ldp_engine* e, uint32_t k, const uint32_t* pred, const double* mf, std::vector<uint32_t>& R,
                          std::vector<uint32_t>& first_unchecked, uint32_t* cursor = nullptr, uint32_t covered = 0xffffffffu) {
  uint64_t replay_pairs = 0;
  const bool plink1 = e->P.plink1_order != 0;
  const Subcontig& s = e->subs[k];
  const uint32_t sfirst = s.local_first;
  const uint32_t send = s.local_first + s.len;
  uint32_t ns = cursor ? *cursor : sfirst;
  while (ns < send) {
    uint32_t ne = ns;
    while (!e->batch_end[s.first + (ne - sfirst)]) {
      ++ne;
    }
    ++ne;
    if (ne > covered) {
      break;
    }
    const uint32_t lo = e->lo_local[ns];
    // load-time removal of monomorphic variants (:902-904)
    for (uint32_t j = ns; j < ne; ++j) {
      if (e->recs[j].flags & 2u) {
        set32(R, j);
      } else if (plink1) {
        first_unchecked[j] = j + 1;
      }
    }
    if (!plink1) {
      // :1042-1100 -- seconds newest first, firsts descending over live window members.  The second is
      // NOT re-checked for having been removed earlier in this batch (quirk kept on purpose).
      for (uint32_t j = ne; j-- > ns;) {
        if (j <= lo) {
          continue;
        }
        const uint32_t* row = pred + e->row_off[j];
        const uint32_t wbase = lo >> 5;
        const uint32_t nw = ((j - 1) >> 5) - wbase + 1;
        const double mf_j_eps = mf[j] * (1 + kSmallEpsilon);
        bool second_removed = false;
        for (uint32_t w = nw; (w-- > 0) && !second_removed;) {
          uint32_t bits = row[w];
          if (!bits) {
            continue;
          }
          bits &= ~load32(R, wbase + w);
          while (bits) {
            const uint32_t t = 31 - __builtin_clz(bits);
            bits &= ~(1u << t);
            const uint32_t i = ((wbase + w) << 5) + t;
            ++replay_pairs;
            if (mf[i] <= mf_j_eps) {
              set32(R, j);
              second_removed = true;
              break;
            }
            set32(R, i);
          }
        }
      }
    } else {
      // :931-1037 PLINK 1 order
      bool changed;
      do {
        changed = false;
        for (uint32_t first = next_clear(R, lo, ne); first < ne; first = next_clear(R, first + 1, ne)) {
          const uint32_t fu = first_unchecked[first];
          if (fu == ne) {
            continue;
          }
          uint32_t second = next_clear(R, first + 1, ne);
          while ((second < ne) && (second < fu)) {
            second = next_clear(R, second + 1, ne);
          }
          if (second >= ne) {
            first_unchecked[first] = ne;
            continue;
          }
          while (true) {
            const uint32_t lo2 = e->lo_local[second];
            const uint32_t word = pred[e->row_off[second] + ((first >> 5) - (lo2 >> 5))];
            ++replay_pairs;
            if ((word >> (first & 31)) & 1) {
              if (mf[first] > mf[second] * (1 + kSmallEpsilon)) {
                set32(R, first);
              } else {
                set32(R, second);
                const uint32_t nxt = next_clear(R, second + 1, ne);
                first_unchecked[first] = (nxt < ne) ? nxt : ne;
              }
              changed = true;
              break;
            }
            second = next_clear(R, second + 1, ne);
            if (second >= ne) {
              first_unchecked[first] = ne;
              break;
            }
          }
        }
      } while (changed);
    }
    ns = ne;
  }
  if (cursor) {
    *cursor = ns;
  }
  return replay_pairs;
}

// Replay as the launch groups come back: group g is waited for, then every subcontig whose variants all lie below
// its need_end is replayed (concurrently) while the GPU works on the later groups.
int replay_progressive(ldp_engine* e, const uint32_t* pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out, double* busy_ms_out) {
  std::vector<uint32_t> first_unchecked;
  if (e->P.plink1_order) {
    first_unchecked.assign(e->local_ct, 0);
  }
  // One worker per owned subcontig (up to 64), started while the GPU still computes: a worker replays its subcontig batch by
  // batch as far as the predicate rows are complete (`covered`, advanced by this thread as each group's copy lands) and waits
  // for more.  What is left after the last kernel is the last group's share of one subcontig -- no thread start-up, no
  // whole chromosomes.
  const uint32_t n_owned = static_cast<uint32_t>(e->owned.size());
  const uint32_t nt = std::max(1u, std::min(std::min(std::thread::hardware_concurrency(), 64u), n_owned));
  std::atomic<uint32_t> covered(0), next(0);
  std::atomic<uint64_t> total(0);
  std::atomic<bool> give_up(false);
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (uint32_t w = 0; w < nt; ++w) {
    pool.emplace_back([&]() {
      for (uint32_t idx = next.fetch_add(1); idx < n_owned; idx = next.fetch_add(1)) {
        const uint32_t k = e->owned[idx];
        const uint32_t send = e->subs[k].local_first + e->subs[k].len;
        uint32_t cursor = e->subs[k].local_first;
        uint32_t seen = covered.load(std::memory_order_acquire);
        while (true) {
          total.fetch_add(replay_subcontig(e, k, pred, mf, R, first_unchecked, &cursor, seen));
          if (cursor >= send) {
            break;
          }
          uint32_t spins = 0;
          uint32_t now = covered.load(std::memory_order_acquire);
          while ((now == seen) && !give_up.load(std::memory_order_relaxed)) {
            if (++spins > 64) {
              std::this_thread::sleep_for(std::chrono::microseconds(10));
            } else {
              std::this_thread::yield();
            }
            now = covered.load(std::memory_order_acquire);
          }
          if (give_up.load(std::memory_order_relaxed)) {
            return;
          }
          seen = now;
        }
      }
    });
  }
  double t_first = 0.0;
  hipError_t herr = hipSuccess;
  const size_t n_groups = e->groups.size();
  const bool timeline = getenv("LDP_DEBUG_TIMELINE") != nullptr;
  const double t_enter = now_ms();
  for (size_t gi = 0; gi < n_groups; ++gi) {
    herr = hipEventSynchronize(e->groups[gi].ev_done);
    if (herr != hipSuccess) {
      break;
    }
    if (timeline) {
      fprintf(stderr, "replay: group %zu back %.2f ms after the replay threads started\n", gi, now_ms() - t_enter);
    }
    if (!gi) {
      t_first = now_ms();
    }
    covered.store((gi + 1 < n_groups) ? e->groups[gi].need_end : e->local_ct, std::memory_order_release);
  }
  if (herr != hipSuccess) {
    give_up.store(true);
  } else {
    covered.store(e->local_ct, std::memory_order_release);  // (no groups at all: nothing to wait for)
  }
  for (std::thread& th : pool) {
    th.join();
  }
  if (timeline) {
    fprintf(stderr, "replay: workers joined %.2f ms after they started\n", now_ms() - t_enter);
  }
  if (herr != hipSuccess) {
    return hipfail(e, herr, "waiting for a launch group");
  }
  *replay_pairs_out = total.load();
  *busy_ms_out = t_first ? (now_ms() - t_first) : 0.0;
  return LDP_OK;
}

void replay(ldp_engine* e, const uint32_t* pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out) {
  std::vector<uint32_t> first_unchecked;
  if (e->P.plink1_order) {
    first_unchecked.assign(e->local_ct, 0);
  }
  // LDP_DEBUG_REPLAY_STEPS=k (test hook): every subcontig in k instalments, the way the streaming replay of a run advances
  // through it as the launch groups land
  if (const char* st = getenv("LDP_DEBUG_REPLAY_STEPS")) {
    const uint32_t steps = static_cast<uint32_t>(std::max(1, atoi(st)));
    uint64_t total = 0;
    for (uint32_t k : e->owned) {
      const Subcontig& sub = e->subs[k];
      uint32_t cursor = sub.local_first;
      for (uint32_t q = 1; q <= steps; ++q) {
        const uint32_t covered = (q == steps) ? (sub.local_first + sub.len) : (sub.local_first + static_cast<uint32_t>(static_cast<uint64_t>(sub.len) * q / steps));
        total += replay_subcontig(e, k, pred, mf, R, first_unchecked, &cursor, covered);
      }
    }
    *replay_pairs_out = total;
    return;
  }
  // longest subcontig first
  std::vector<uint32_t> order(e->owned);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return e->subs[a].len > e->subs[b].len; });
  std::atomic<uint64_t> total(0);
  parallel_for(static_cast<uint32_t>(order.size()), 64, [&](uint32_t t) {
    total.fetch_add(replay_subcontig(e, order[t], pred, mf, R, first_unchecked));
  });
  *replay_pairs_out = total.load();
}

int finish_removed(ldp_engine* e, const std::vector<uint32_t>& R, uint64_t* removed) {
  memset(removed, 0, ((static_cast<size_t>(e->variant_ct) + 63) / 64) * sizeof(uint64_t));
  for (uint32_t k : e->owned) {
    const Subcontig& s = e->subs[k];
    uint32_t v = 0;
    while (v < s.len) {
      // up to 32 bits at a time: local bits [l, l+n) -> global bits [g, g+n)
      const uint32_t l = s.local_first + v;
      const uint32_t g = s.first + v;
      const uint32_t n = std::min<uint32_t>(std::min<uint32_t>(32 - (l & 31), 64 - (g & 63)), s.len - v);
      uint64_t bits = (R[l >> 5] >> (l & 31));
      if (n < 32) {
        bits &= (1ull << n) - 1;
      }
      if (bits) {
        removed[g >> 6] |= bits << (g & 63);
      }
      v += n;
    }
  }
  return LDP_OK;
}

// frequencies the replay compares: GetAlleleFreq(maj allele), minus 1.0 for --indep-preferred variants
int prepare_mf(ldp_engine* e, std::vector<double>* scratch, const double** mf_out) {
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    if (!e->mf_set[l]) {
      return fail(e, LDP_ERR_STATE, "major-allele frequency missing for an owned variant (ldp_set_maj_freqs)");
    }
  }
  if (e->preferred.empty()) {
    *mf_out = e->maj_freq.data();
    return LDP_OK;
  }
  *scratch = e->maj_freq;
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    const uint32_t g = e->local_to_global[l];
    if ((e->preferred[g >> 6] >> (g & 63)) & 1) {
      (*scratch)[l] -= 1.0;  // plink2_ld.cc:916-918
    }
  }
  *mf_out = scratch->data();
  return LDP_OK;
}

void fill_pair_args(const ldp_engine* e, PairKernelArgs* out, bool with_early_exit) {
  PairKernelArgs& A = *out;
  A.planes = e->d_planes;
  A.codes = e->d_codes;
  A.code_row_bytes = e->code_row_bytes;
  A.row_dwords = e->row_dwords;
  A.chunks = e->chunks;
  A.founder_ct = e->P.founder_ct;
  A.recs = e->d_recs;
  A.lo = e->d_lo;
  A.row_off = e->d_row_off;
  A.pred = e->d_pred;
  A.items = e->d_items;
  A.n_items = e->codes_format ? 0u : static_cast<uint32_t>(e->items.size());
  A.plane_base_variant = 0;
  A.thresh = e->P.prune_last_param * (1 + kSmallEpsilon);  // plink2_ld.cc:1255
  A.stats = nullptr;
  A.pair_off = e->d_pair_off;
  A.counters = e->d_counters;
  A.item_general = e->d_item_general;
  // early termination is off when the caller wants every pair's integers (parity runs) or LDP_EARLY_EXIT=0
  A.cp_stats = (with_early_exit && e->opt.early_exit && e->n_checkpoints) ? e->d_cp_stats : nullptr;
  A.cp_gen = A.cp_stats ? e->d_cp_gen : nullptr;
  for (int k = 0; k < kCheckpoints; ++k) {
    A.checkpoint_chunk[k] = e->checkpoint_chunk[k];
  }
  A.n_checkpoints = A.cp_stats ? e->n_checkpoints : 0;
  A.lds_dwords = 0;
  A.r2_out = nullptr;
  A.r2_hits = nullptr;
  A.r2_hit_capacity = 0;
  A.r2_min = 0.0;
  A.r2_ld = 0;
  A.r2_row_first = 0;
  A.r2_row_end = 0;
  A.r2_col_first = 0;
  A.r2_col_end = 0xffffffffu;
  A.r2_band_base = 0;
  A.r2_float = 0;
  A.r_signed = e->r_signed;
  // matrix-pipe work is attached per launch (launch_group / the inspection run); r^2 launches stay on the popcount kernels
  A.mf_wgs = nullptr;
  A.n_mf_wgs = 0;
  A.mf_diag_ct = 0;
  A.n_local = e->local_ct;
  A.mf_active = 0;
  A.route = nullptr;
  A.sparse_ok = 0;
  A.mf_four = e->opt.pair_four ? 1u : 0u;
  A.mf_gu = (e->opt.pair_gu && (e->P.founder_ct <= kMfGuMaxFounders)) ? 1u : 0u;
  A.wd_general = 0;
  A.wd_tiles = nullptr;
  A.n_wd_tiles = 0;
  A.wd_active = 0;
  A.wd_tiles_plain = nullptr;
  A.n_wd_tiles_plain = 0;
}

// A new load epoch begins (variants are being loaded again): whatever the pair streams still run belongs to the
// old data.  Order the main stream behind it, forget the launches and clear the counters.
int begin_load_epoch(ldp_engine* e) {
  for (int k = 0; k < kPairStreams; ++k) {
    if (e->pair_tail_set[k]) {
      HIP_TRY(e, hipStreamWaitEvent(e->stream, e->pair_tail[k], 0));
      e->pair_tail_set[k] = false;
    }
  }
  HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
  ++e->load_epoch;
  e->loaded_prefix = 0;
  e->next_group = 0;
  for (ldp_engine::PairGroup& g : e->groups) {
    g.launched = false;
  }
  return LDP_OK;
}

// Decide on the device which matrix-pipe kernel owns the launches queued next on `stream` (slot of d_route).  The decision is
// taken from the records of ALL the rows the launches read, local rows [0, row_end), as they are resident when the stream gets
// there -- whichever load call, of whichever load epoch, put them there (a caller may re-load a few rows only; plink2-hip does
// for multiallelic and MT rows, and a record kept per epoch would forget the missing calls of the rows that stayed).
hipError_t queue_route(ldp_engine* e, size_t slot, hipStream_t stream, int allow_sparse, uint32_t row_end) {
  const double rows = static_cast<double>(std::max<uint32_t>(row_end, 1));
  const double frac = allow_sparse ? e->opt.sparse_frac : 0.0;
  const double total_limit = frac * static_cast<double>(e->P.founder_ct) * rows;  // (< 2^64: 16M samples x 2^32 rows)
  const double high_limit = 0.02 * rows;
  const uint32_t miss_high = static_cast<uint32_t>(std::min(2.0 * e->opt.sparse_frac * static_cast<double>(e->P.founder_ct), 4294967295.0));
  MissStats* ms = e->d_miss_stats + slot;
  hipError_t rc = hipMemsetAsync(ms, 0, sizeof(MissStats), stream);
  if (rc != hipSuccess) {
    return rc;
  }
  rc = launch_miss_stats(e->d_recs, row_end, e->P.founder_ct, miss_high, ms, stream);
  if (rc != hipSuccess) {
    return rc;
  }
  return launch_route(ms, static_cast<unsigned long long>(total_limit), static_cast<unsigned long long>(high_limit), allow_sparse && (frac > 0.0),
                      e->d_route + slot, stream);
}

// Queue group gi behind everything the main stream holds right now (the prepare kernels it depends on).
int launch_group(ldp_engine* e, uint32_t gi) {
  ldp_engine::PairGroup& g = e->groups[gi];
  const int k = static_cast<int>(gi % kPairStreams);
  hipStream_t ps = e->pair_stream[k];
  HIP_TRY(e, hipEventRecord(g.ev_ready, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(ps, g.ev_ready, 0));
  if (g.word_end > g.word_first) {
    HIP_TRY(e, hipMemsetAsync(e->d_pred + g.word_first, 0, (g.word_end - g.word_first) * sizeof(uint32_t), ps));
  }
  PairKernelArgs A;
  fill_pair_args(e, &A, true);
  A.items = e->d_items + g.item_first;
  A.item_general = e->d_item_general + g.item_first;
  A.n_items = e->codes_format ? 0u : g.item_ct;
  if (e->mf_enabled) {
    // Which kernel family owns the group is decided on the device, once per group: a snapshot of the missing-calls flag
    // (all of the group's rows are converted by now) that every kernel of the group reads.
    A.mf_active = 2;
    A.sparse_ok = ((A.mf_active == 2) && !A.stats && (e->opt.sparse_frac > 0.0)) ? 1 : 0;
    HIP_TRY(e, queue_route(e, gi, ps, A.sparse_ok, g.need_end));
    A.route = e->d_route + gi;
    A.mf_wgs = e->d_mf_wgs + g.mf_first;
    A.n_mf_wgs = g.mf_ct;
    A.mf_diag_ct = g.mf_diag_ct;
    A.wd_tiles = e->d_wd_tiles + g.wl_first;
    A.n_wd_tiles = g.wl_ct;
    A.wd_active = e->wd_tiles.empty() ? 0u : 1u;
    if (e->d_wd_tiles_plain) {
      A.wd_tiles_plain = e->d_wd_tiles_plain + g.wd_first;
      A.n_wd_tiles_plain = g.wd_ct;
    }
    // prune launches over rows with missing calls: the four-product form takes the tile plan's subcontigs in quarter tiles
    A.wd_general = (A.mf_four && e->opt.four_tiles && !A.stats && !A.r2_out && !A.r2_hits && A.n_wd_tiles) ? 1u : 0u;
    g.four_tiles = (A.wd_general != 0);
  }
  hipError_t krc = launch_pair_tiles(A, e->max_rows, ps, g.ev);
  if (krc != hipSuccess) {
    return hipfail(e, krc, "pair_tiles_kernel launch");
  }
  if (e->mf_enabled) {
    krc = launch_pair_mfma(A, ps, g.ev + 4);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pair_mfma_kernel launch");
    }
  }
  if (g.word_end > g.word_first) {
    HIP_TRY(e, hipMemcpyAsync(e->h_pred + g.word_first, e->d_pred + g.word_first, (g.word_end - g.word_first) * sizeof(uint32_t), hipMemcpyDeviceToHost, ps));
  }
  HIP_TRY(e, hipEventRecord(g.ev_done, ps));
  HIP_TRY(e, hipEventRecord(e->pair_tail[k], ps));
  e->pair_tail_set[k] = true;
  g.launched = true;
  return LDP_OK;
}

// launch every group whose variants are all converted (in order)
int launch_ready_groups(ldp_engine* e) {
  while ((e->loaded_prefix < e->local_ct) && (e->load_tag[e->loaded_prefix] == e->load_epoch)) {
    ++e->loaded_prefix;
  }
  while ((e->next_group < e->groups.size()) && (e->groups[e->next_group].need_end <= e->loaded_prefix)) {
    const int rc = launch_group(e, e->next_group);
    if (rc) {
      return rc;
    }
    ++e->next_group;
  }
  return LDP_OK;
}

int run_impl(ldp_engine* e, uint64_t* removed, ldp_pair_stats_t* stats, uint64_t stats_capacity) {
  if (!e->planned) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants() has not been called");
  }
  if (!removed) {
    return fail(e, LDP_ERR_INVALID, "removed bitmap is NULL");
  }
  if (e->matrix_mode || e->band_r2_mode) {
    return fail(e, LDP_ERR_STATE, "engine is planned for --r2-unphased output (ldp_set_variants_matrix / ldp_set_variants_vcor)");
  }
  const double t_start = now_ms();
  double tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    if (!e->loaded[l]) {
      return fail(e, LDP_ERR_STATE, "genotypes missing for an owned variant (ldp_load_genotypes)");
    }
    if (!e->mf_set[l]) {
      return fail(e, LDP_ERR_STATE, "major-allele frequency missing for an owned variant (ldp_set_maj_freqs)");
    }
  }
  if (stats && (stats_capacity < e->cand_pairs)) {
    return fail(e, LDP_ERR_INVALID, "stats buffer smaller than the candidate pair count");
  }
  HIP_TRY(e, hipSetDevice(e->device));
  DevBuf stats_buf;
  ldp_pair_stats_t* d_stats = nullptr;
  float kms = 0.f, kms_fast = 0.f, kms_general = 0.f, kms_mfma = 0.f, kms_mfma_general = 0.f;
  uint32_t launches = 0;
  std::vector<double> mf_scratch;
  const double* mf = nullptr;
  std::vector<uint32_t> R((static_cast<size_t>(e->local_ct) + 31) / 32 + 1, 0);
  uint64_t replay_pairs = 0;
  double t_replay = now_ms();
  bool replayed = false;
  double replay_busy_ms = 0.0;
  unsigned long long h_counters[4] = {0, 0, 0, 0};
  if (stats) {
    // Inspection run: one launch over every item, every pair's integers stored, no early termination.
    // Whatever the side streams hold is waited for and superseded.
    for (int k = 0; k < kPairStreams; ++k) {
      if (e->pair_tail_set[k]) {
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->pair_tail[k], 0));
        e->pair_tail_set[k] = false;
      }
    }
    if (e->cand_pairs) {
      HIP_TRY(e, hipMalloc(&stats_buf.p, e->cand_pairs * sizeof(ldp_pair_stats_t)));
      d_stats = stats_buf.as<ldp_pair_stats_t>();
      HIP_TRY(e, hipMemsetAsync(d_stats, 0, e->cand_pairs * sizeof(ldp_pair_stats_t), e->stream));
    }
    HIP_TRY(e, hipMemsetAsync(e->d_pred, 0, std::max<size_t>(e->pred_words, 1) * sizeof(uint32_t), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
    PairKernelArgs A;
    fill_pair_args(e, &A, false);
    A.stats = d_stats;
    EventSet<7> evset;
    hipEvent_t* evk = evset.ev;
    HIP_TRY(e, evset.create());
    if (e->mf_enabled) {
      const size_t slot = e->groups.size();
      HIP_TRY(e, queue_route(e, slot, e->stream, 0, e->local_ct));
      A.mf_active = 2;
      A.route = e->d_route + slot;
      A.mf_wgs = e->d_mf_wgs;
      A.n_mf_wgs = static_cast<uint32_t>(e->mf_wgs.size());
      A.wd_tiles = e->d_wd_tiles;
      A.n_wd_tiles = static_cast<uint32_t>(e->wd_launch.size());
      A.wd_active = e->wd_tiles.empty() ? 0u : 1u;
    }
    hipError_t krc = launch_pair_tiles(A, e->max_rows, e->stream, evk);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pair_tiles_kernel launch");
    }
    if (e->mf_enabled) {
      // group by group (each group's workgroups are ordered [all-diagonal | others] for the two instantiations), one route
      for (const ldp_engine::PairGroup& g : e->groups) {
        PairKernelArgs G = A;
        G.mf_wgs = e->d_mf_wgs + g.mf_first;
        G.n_mf_wgs = g.mf_ct;
        G.mf_diag_ct = g.mf_diag_ct;
        G.wd_tiles = e->d_wd_tiles + g.wl_first;
        G.n_wd_tiles = g.wl_ct;
        if (e->d_wd_tiles_plain) {
          G.wd_tiles_plain = e->d_wd_tiles_plain + g.wd_first;
          G.n_wd_tiles_plain = g.wd_ct;
        }
        krc = launch_pair_mfma(G, e->stream, evk + 4);
        if (krc != hipSuccess) {
          return hipfail(e, krc, "pair_mfma_kernel launch");
        }
      }
    }
    if (e->pred_words) {
      HIP_TRY(e, hipMemcpyAsync(e->h_pred, e->d_pred, e->pred_words * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin, e->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    if (e->mf_enabled) {
      HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin + 4, e->d_route, (e->groups.size() + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    if (d_stats) {
      HIP_TRY(e, hipMemcpyAsync(stats, d_stats, e->cand_pairs * sizeof(ldp_pair_stats_t), hipMemcpyDeviceToHost, e->stream));
    }
    rc = fetch_recs(e);
    if (rc) {
      return rc;
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (!e->items.empty()) {
      if (!e->codes_format) {  // (the popcount kernels were launched)
        HIP_TRY(e, hipEventElapsedTime(&kms_fast, evk[0], evk[1]));
        HIP_TRY(e, hipEventElapsedTime(&kms_general, evk[2], evk[3]));
      }
      if (e->mf_enabled && !e->mf_wgs.empty()) {
        HIP_TRY(e, hipEventElapsedTime(&kms_mfma, evk[4], evk[5]));
        HIP_TRY(e, hipEventElapsedTime(&kms_mfma_general, evk[5], evk[6]));
      }
      launches = 1;
    }
    // the next plain run recomputes with the production settings
    for (ldp_engine::PairGroup& g : e->groups) {
      g.launched = false;
    }
    e->next_group = 0;
  } else {
    // 1. Most groups were queued while the genotypes were still being converted (ldp_load_genotypes); queue the
    //    rest, then the copies back, behind the two pair streams.
    rc = start_fetch_recs(e);  // (first in the copy engine's queue, see there)
    if (rc) {
      return rc;
    }
    bool any_launched = false;
    for (const ldp_engine::PairGroup& g : e->groups) {
      any_launched = any_launched || g.launched;
    }
    if (!any_launched) {
      HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));  // (ahead of every ev_ready)
    }
    rc = launch_ready_groups(e);
    if (rc) {
      return rc;
    }
    for (uint32_t gi = 0; gi < e->groups.size(); ++gi) {
      if (!e->groups[gi].launched) {
        rc = launch_group(e, gi);
        if (rc) {
          return rc;
        }
      }
    }
    e->next_group = static_cast<uint32_t>(e->groups.size());
    for (int k = 0; k < kPairStreams; ++k) {
      if (e->pair_tail_set[k]) {
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->pair_tail[k], 0));
      }
    }
    tl[0] = now_ms();
    HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin, e->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    if (e->mf_enabled) {
      HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin + 4, e->d_route, (e->groups.size() + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    // 2. ... meanwhile the per-variant records come back on the copy stream and the host derives the
    //    major-allele frequencies the replay needs ...
    rc = fetch_recs(e);
    if (rc) {
      return rc;
    }
    tl[1] = now_ms();
    rc = prepare_mf(e, &mf_scratch, &mf);
    if (rc) {
      return rc;
    }
    tl[2] = now_ms();
    // 3. ... and replays each group's subcontigs as soon as its predicate words are back.
    t_replay = now_ms();
    rc = replay_progressive(e, e->h_pred, mf, R, &replay_pairs, &replay_busy_ms);
    if (rc) {
      return rc;
    }
    replayed = true;
    tl[3] = now_ms();
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    tl[4] = now_ms();
    for (ldp_engine::PairGroup& g : e->groups) {
      float f = 0.f, gen = 0.f;
      if (!e->codes_format) {  // (the popcount kernels were launched)
        HIP_TRY(e, hipEventElapsedTime(&f, g.ev[0], g.ev[1]));
        HIP_TRY(e, hipEventElapsedTime(&gen, g.ev[2], g.ev[3]));
      }
      if (e->mf_enabled && g.mf_ct) {
        float mf = 0.f, mfg = 0.f;
        HIP_TRY(e, hipEventElapsedTime(&mf, g.ev[4], g.ev[5]));
        HIP_TRY(e, hipEventElapsedTime(&mfg, g.ev[5], g.ev[6]));
        kms_mfma += mf;
        kms_mfma_general += mfg;
      }
      kms_fast += f;
      kms_general += gen;
      ++launches;
      g.launched = false;  // a run consumes its launches: the next one recomputes (or picks up eager launches of new loads)
    }
    e->next_group = 0;
  }
  for (int q = 0; q < 4; ++q) {
    h_counters[q] = e->h_counters_pin[q];  // (the stream that carried the copy has been synchronised in both branches)
  }
  kms = kms_fast + kms_general + kms_mfma + kms_mfma_general;
  // which matrix-pipe kernel route_kernel gave each launch of this run (deterministic evidence of the path taken)
  uint32_t route_ct[3] = {0, 0, 0};
  uint32_t four_tile_launches = 0;
  if (e->mf_enabled && !e->mf_wgs.empty()) {
    const uint32_t* h_route = reinterpret_cast<const uint32_t*>(e->h_counters_pin + 4);
    if (stats) {
      ++route_ct[std::min<uint32_t>(h_route[e->groups.size()], 2)];
    } else {
      for (size_t gi = 0; gi < e->groups.size(); ++gi) {
        if (e->groups[gi].mf_ct) {
          ++route_ct[std::min<uint32_t>(h_route[gi], 2)];
          four_tile_launches += ((h_route[gi] >= 2) && e->groups[gi].four_tiles) ? 1u : 0u;
        }
      }
    }
  }
  if (!replayed) {
    rc = prepare_mf(e, &mf_scratch, &mf);
    if (rc) {
      return rc;
    }
    // greedy replay on the host
    t_replay = now_ms();
    replay(e, e->h_pred, mf, R, &replay_pairs);
  }
  finish_removed(e, R, removed);
  const double t_end = now_ms();
  if (getenv("LDP_DEBUG_TIMELINE")) {
    fprintf(stderr, "run timeline (ms since entry): queued %.2f recs %.2f mf %.2f replayed %.2f synced %.2f end %.2f\n", tl[0] - t_start, tl[1] - t_start,
            tl[2] - t_start, tl[3] - t_start, tl[4] - t_start, t_end - t_start);
  }

  e->ctr.candidate_pairs = e->cand_pairs;
  e->ctr.computed_pairs = e->computed_pairs;
  e->ctr.replay_pairs = replay_pairs;
  e->ctr.pred_true = h_counters[0];
  e->ctr.early_exit_unit_chunks = e->codes_format ? 0 : (h_counters[1] / 4);  // the popcount kernel counts quarter units (one second-variant group)
  e->ctr.tile_unit_chunks = (e->computed_pairs / (8 * kTileJ)) * e->chunks;
  e->ctr.ms_pair_kernel = kms;
  e->ctr.ms_pair_fast = kms_fast;
  e->ctr.ms_pair_general = kms_general;
  e->ctr.ms_pair_mfma = kms_mfma;
  e->ctr.ms_pair_mfma_general = kms_mfma_general;
  e->ctr.mfma_block_products = e->mf_enabled ? e->mf_products : 0;
  e->ctr.mfma_product_stages = e->ctr.mfma_block_products * pair_mfma_ksteps(e->P.founder_ct);
  e->ctr.mfma_skipped_product_stages = h_counters[2];
  e->ctr.mfma_extra_product_stages = e->codes_format ? h_counters[1] : 0;
  e->ctr.wide_tiles = e->mf_enabled ? static_cast<uint32_t>(e->wd_tiles.size()) : 0;
  e->ctr.sparse_exact_pairs = h_counters[3];
  e->ctr.route_complete_launches = route_ct[0];
  e->ctr.route_sparse_launches = route_ct[1];
  e->ctr.route_general_launches = route_ct[2];
  e->ctr.four_tile_launches = four_tile_launches;
  e->ctr.ms_replay = replayed ? replay_busy_ms : (t_end - t_replay);  // (time spent replaying, not waiting for groups)
  e->ctr.ms_run_total = t_end - t_start;
  e->ctr.pair_kernel_launches = launches;
  return LDP_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int ldp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// The first real use of a device creates its context and queues: 40-90 ms on the GPU box (up to half a second on a cold one), which
// an engine otherwise pays inside its first ldp_load_genotypes().  A host that has other start-up work to do (plink2-hip: the variant
// and sample tables) calls this on a side thread first.  Creates a stream and a small pinned allocation and frees both.
int ldp_prewarm(int device) {
  if ((device < 0) || (device >= ldp_device_count())) {
    return LDP_ERR_GPU;
  }
  if (hipSetDevice(device) != hipSuccess) {
    (void)hipGetLastError();
    return LDP_ERR_GPU;
  }
  hipStream_t st = nullptr;
  void* pin = nullptr;
  void* dev = nullptr;
  const bool ok = (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) && (hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault) == hipSuccess) &&
                  (hipMalloc(&dev, 1 << 20) == hipSuccess) && (hipMemcpyAsync(dev, pin, 1 << 20, hipMemcpyHostToDevice, st) == hipSuccess) &&
                  (hipStreamSynchronize(st) == hipSuccess);
  if (dev) {
    (void)hipFree(dev);
  }
  if (pin) {
    (void)hipHostFree(pin);
  }
  if (st) {
    (void)hipStreamDestroy(st);
  }
  if (!ok) {
    (void)hipGetLastError();
    return LDP_ERR_GPU;
  }
  return LDP_OK;
}

uint32_t ldp_matrix_pipe_max_founders(void) { return kMfMaxFounders; }

int ldp_create(const ldp_params* params, ldp_engine** out) {
  if (!params || !out) {
    return LDP_ERR_INVALID;
  }
  *out = nullptr;
  if (params->founder_ct < 2) {  // plink2_ld.cc:2537
    return LDP_ERR_INVALID;
  }
  if (params->founder_ct >= (1u << 30)) {  // plink2_ld.cc:1122
    return LDP_ERR_UNSUPPORTED;
  }
  if (!(params->prune_last_param >= 0.0) || !(params->prune_last_param < 1.0)) {  // plink2.cc:7303
    return LDP_ERR_INVALID;
  }
  if (params->window_is_bp) {
    if ((params->prune_window_incr != 1) || (params->prune_window_size < 2)) {  // plink2.cc:7268,7290
      return LDP_ERR_INVALID;
    }
  } else if ((params->prune_window_size < 1) || (params->prune_window_incr < 1) || (params->prune_window_incr > params->prune_window_size)) {
    return LDP_ERR_INVALID;
  }
  ldp_engine* e = new (std::nothrow) ldp_engine();
  if (!e) {
    return LDP_ERR_NOMEM;
  }
  e->P = *params;
  e->opt = options_from_env();
  e->ctr = ldp_counters();
  *out = e;
  return LDP_OK;
}

void ldp_destroy(ldp_engine* e) {
  if (!e) {
    return;
  }
  if (e->gpu_ok) {
    (void)hipSetDevice(e->device);
    free_device(e);
    if (e->own_stream) {
      (void)hipStreamDestroy(e->stream);
    }
    if (e->copy_stream) {
      (void)hipStreamDestroy(e->copy_stream);
    }
    for (int k = 0; k < kPairStreams; ++k) {
      if (e->pair_stream[k]) {
        (void)hipStreamDestroy(e->pair_stream[k]);
      }
    }
  }
  delete e;
}

const char* ldp_last_error(const ldp_engine* e) { return e ? e->err.c_str() : "null engine"; }

int ldp_set_variants(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (variant_ct && !chr_idx) {
    return fail(e, LDP_ERR_INVALID, "chr_idx is NULL");
  }
  if (e->P.window_is_bp && variant_ct && !bps) {
    return fail(e, LDP_ERR_INVALID, "bp-based window needs variant positions");
  }
  for (uint32_t v = 1; v < variant_ct; ++v) {
    if (chr_idx[v] < chr_idx[v - 1]) {
      return fail(e, LDP_ERR_INVALID, "chr_idx must be nondecreasing");
    }
    if (e->P.window_is_bp && (chr_idx[v] == chr_idx[v - 1]) && (bps[v] < bps[v - 1])) {
      return fail(e, LDP_ERR_INVALID, "positions must be sorted within a chromosome (plink2.cc:2926)");
    }
  }
  e->matrix_mode = false;
  e->band_r2_mode = false;
  e->variant_ct = variant_ct;
  e->bps.clear();
  if (bps) {
    e->bps.assign(bps, bps + variant_ct);
  }
  subcontig_split(chr_idx, e->P.window_is_bp ? e->bps.data() : nullptr, variant_ct, e->P.prune_window_size, &e->subs, &e->window_max);
  e->lo_global.resize(variant_ct);
  for (uint32_t v = 0; v < variant_ct; ++v) {
    e->lo_global[v] = v;
  }
  e->batch_end.assign(variant_ct, 0);
  for (const Subcontig& s : e->subs) {
    plan_subcontig(e, s, &e->lo_global, &e->batch_end);
  }
  e->rank = 0;
  e->world = 1;
  for (Subcontig& s : e->subs) {
    s.owner = 0;
  }
  e->planned = true;
  build_shard(e);
  e->ctr.subcontig_ct = static_cast<uint32_t>(e->subs.size());
  e->ctr.owned_subcontig_ct = static_cast<uint32_t>(e->owned.size());
  e->ctr.window_max = e->window_max;
  return LDP_OK;
}

int ldp_set_variants_matrix(ldp_engine* e, uint32_t variant_ct) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  e->matrix_mode = true;
  e->band_r2_mode = false;
  e->variant_ct = variant_ct;
  e->bps.clear();
  e->subs.clear();
  if (variant_ct) {
    Subcontig s;
    s.len = variant_ct;
    s.first = 0;
    s.owner = 0;
    s.local_first = 0;
    e->subs.push_back(s);
  }
  e->window_max = variant_ct;
  e->lo_global.assign(variant_ct, 0);
  e->batch_end.assign(variant_ct, 0);
  e->rank = 0;
  e->world = 1;
  e->planned = true;
  build_shard(e);
  e->ctr.subcontig_ct = static_cast<uint32_t>(e->subs.size());
  e->ctr.owned_subcontig_ct = e->ctr.subcontig_ct;
  e->ctr.window_max = variant_ct;
  return LDP_OK;
}

// Window of the --r2-unphased table (UpdateVcorWindow, plink2_ld.cc:10984-11023): second variant B is paired with
// the earlier variants A of its chromosome that are at most var_ct_radius variants and bp_radius base pairs away.
int ldp_set_variants_vcor(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps, uint32_t bp_radius, uint32_t var_ct_radius) {
  return ldp_set_variants_vcor_cm(e, variant_ct, chr_idx, bps, nullptr, bp_radius, -1.0, var_ct_radius);
}

int ldp_set_variants_vcor_cm(ldp_engine* e, uint32_t variant_ct, const uint32_t* chr_idx, const uint32_t* bps, const double* cms, uint32_t bp_radius,
                             double cm_radius, uint32_t var_ct_radius) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (variant_ct && (!chr_idx || !bps)) {
    return fail(e, LDP_ERR_INVALID, "chr_idx / bps is NULL");
  }
  for (uint32_t v = 1; v < variant_ct; ++v) {
    if (chr_idx[v] < chr_idx[v - 1]) {
      return fail(e, LDP_ERR_INVALID, "chr_idx must be nondecreasing");
    }
    if ((chr_idx[v] == chr_idx[v - 1]) && (bps[v] < bps[v - 1])) {
      return fail(e, LDP_ERR_INVALID, "positions must be sorted within a chromosome (plink2.cc:2926)");
    }
  }
  e->matrix_mode = false;
  e->band_r2_mode = true;
  e->variant_ct = variant_ct;
  e->bps.assign(bps, bps + variant_ct);
  e->subs.clear();
  e->lo_global.resize(variant_ct);
  e->batch_end.assign(variant_ct, 0);
  uint32_t window_max = 0;
  uint32_t c0 = 0;
  while (c0 < variant_ct) {
    uint32_t c1 = c0 + 1;
    while ((c1 < variant_ct) && (chr_idx[c1] == chr_idx[c0])) {
      ++c1;
    }
    uint32_t lo = c0;
    for (uint32_t j = c0; j < c1; ++j) {
      // (the centimorgan window is open at the far end: B belongs to A's window while cm_B < cm_A + radius, the sum as
      // UpdateVcorWindow forms it, plink2_ld.cc:11010-11013)
      while ((bps[j] - bps[lo] > bp_radius) || (j - lo > var_ct_radius) || (cms && (lo < j) && !(cms[j] < cms[lo] + cm_radius))) {
        ++lo;
      }
      e->lo_global[j] = lo;
      window_max = std::max(window_max, j - lo + 1);
    }
    if (c1 - c0 >= 2) {
      Subcontig s;
      s.len = c1 - c0;
      s.first = c0;
      s.owner = 0;
      s.local_first = 0;
      e->subs.push_back(s);
    } else {
      e->lo_global[c0] = c0;
    }
    c0 = c1;
  }
  e->window_max = window_max;
  e->rank = 0;
  e->world = 1;
  e->planned = true;
  build_shard(e);
  e->ctr.subcontig_ct = static_cast<uint32_t>(e->subs.size());
  e->ctr.owned_subcontig_ct = e->ctr.subcontig_ct;
  e->ctr.window_max = window_max;
  return LDP_OK;
}

// r^2 of every candidate pair whose SECOND variant lies in [row_first, row_first + row_ct), in band order: the
// pairs of second variant j start at sum_{row_first <= j' < j} (j' - lo[j']) and run over i = lo[j] .. j-1
// (lo from ldp_get_band).  Same doubles as ldp_r2_unphased_rows.
}  // extern "C"

namespace {
// --r2-unphased requests on the matrix pipe: plan the requested second variants' block products (ldp_device.h: MfmaWG),
// upload the plan and attach it to the launch.  The r^2 epilogue is emit_pair()'s, shared with the popcount kernels.
bool r2_on_matrix_pipe(const ldp_engine* e) { return e->codes_format; }  // (set by ensure_device_plan: the matrix-pipe kernels read the code image)

// tile_buf (optional): runs whose band is wide ALSO get the 8 x 8 tile plan of ldp_pair_wide.hip -- the all-pairs rows of the r^2
// matrices and of `inter-chr` (BASELINE config 4) are nothing but wide bands --, which owns them on complete-data launches (the
// marked parallelogram workgroups stand by for rows with missing calls, as in the prune).
int attach_mfma_plan(ldp_engine* e, PairKernelArgs* A, const std::vector<std::pair<uint32_t, uint32_t>>& runs, const uint32_t* lo, uint32_t j_first,
                     uint32_t j_end, DevBuf* buf, uint64_t* products, uint32_t i_first = 0, uint32_t i_end = 0xffffffffu, DevBuf* tile_buf = nullptr,
                     uint64_t* tile_products = nullptr) {
  std::vector<MfmaWG> wgs;
  std::vector<MfmaTile> tiles;
  plan_mfma_generic(runs, lo, j_first, j_end, &wgs, products, i_first, i_end, tile_buf ? &tiles : nullptr, e->opt.wide_min_reach);
  A->n_mf_wgs = static_cast<uint32_t>(wgs.size());
  if (wgs.empty()) {
    return LDP_OK;
  }
  A->mf_diag_ct = partition_diag(&wgs, 0, wgs.size());
  HIP_TRY(e, hipMalloc(&buf->p, wgs.size() * sizeof(MfmaWG)));
  HIP_TRY(e, hipMemcpy(buf->p, wgs.data(), wgs.size() * sizeof(MfmaWG), hipMemcpyHostToDevice));
  if (!tiles.empty()) {
    HIP_TRY(e, hipMalloc(&tile_buf->p, tiles.size() * sizeof(MfmaTile)));
    HIP_TRY(e, hipMemcpy(tile_buf->p, tiles.data(), tiles.size() * sizeof(MfmaTile), hipMemcpyHostToDevice));
    A->wd_tiles = tile_buf->as<MfmaTile>();
    A->n_wd_tiles = static_cast<uint32_t>(tiles.size());
    A->wd_active = 1;
    if (tile_products) {
      *tile_products = 0;
      for (const MfmaTile& t : tiles) {
        *tile_products += static_cast<uint64_t>(__builtin_popcountll(t.mask));
      }
    }
  }
  const size_t slot = e->groups.size();  // (the route slot of launches outside the launch groups)
  HIP_TRY(e, queue_route(e, slot, e->stream, 0, e->local_ct));
  A->mf_wgs = buf->as<MfmaWG>();
  A->mf_active = 2;
  A->route = e->d_route + slot;
  return LDP_OK;
}

struct HitRequest {
  double min_r2;
  ldp_r2_hit* out;
  uint64_t capacity;
  uint64_t* count;
};

// band rows: dense into `out` (hits == nullptr) or filtered on the device into hits->out (global variant indices)
int r2_band_impl(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t capacity_elems, const HitRequest* hits) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned || !e->band_r2_mode) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants_vcor() first");
  }
  if (hits) {
    if ((hits->capacity && !hits->out) || !hits->count) {
      return fail(e, LDP_ERR_INVALID, "hit buffer missing");
    }
    *hits->count = 0;
  }
  if (static_cast<uint64_t>(row_first) + row_ct > e->variant_ct) {
    return fail(e, LDP_ERR_INVALID, "row range out of bounds");
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    if (!e->loaded[l]) {
      return fail(e, LDP_ERR_STATE, "genotypes missing for a variant (ldp_load_genotypes)");
    }
  }
  // owned (= paired) variants of the global row range are contiguous in local order
  const uint32_t row_end = row_first + row_ct;
  uint32_t l_first = e->local_ct, l_end = 0;
  for (uint32_t g = row_first; g < row_end; ++g) {
    const int64_t l = e->global_to_local[g];
    if (l >= 0) {
      l_first = std::min<uint32_t>(l_first, static_cast<uint32_t>(l));
      l_end = std::max<uint32_t>(l_end, static_cast<uint32_t>(l) + 1);
    }
  }
  if (l_first >= l_end) {
    return LDP_OK;
  }
  const uint64_t n_elems = e->pair_off[l_end] - e->pair_off[l_first];
  if ((!hits) && (n_elems > capacity_elems)) {
    return fail(e, LDP_ERR_INVALID, "output buffer smaller than the rows' candidate pair count");
  }
  if (!n_elems) {
    return LDP_OK;
  }
  if ((!hits) && !out) {
    return fail(e, LDP_ERR_INVALID, "output buffer is NULL");
  }
  const double t_start = now_ms();
  HIP_TRY(e, hipSetDevice(e->device));
  // items are sorted by J-tile: the ones that touch [l_first, l_end)
  size_t i0 = 0, i1 = e->items.size();
  while ((i0 < i1) && (e->items[i0].jend <= l_first)) {
    ++i0;
  }
  while ((i1 > i0) && (e->items[i1 - 1].j0 >= l_end)) {
    --i1;
  }
  const size_t esz = as_float ? sizeof(float) : sizeof(double);
  DevBuf out_buf;
  if (hits) {
    HIP_TRY(e, hipMalloc(&out_buf.p, std::max<uint64_t>(hits->capacity, 1) * sizeof(ldp_r2_hit)));
    HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
  } else {
    HIP_TRY(e, hipMalloc(&out_buf.p, n_elems * esz));
    HIP_TRY(e, hipMemsetAsync(out_buf.p, 0, n_elems * esz, e->stream));
  }
  PairKernelArgs A;
  fill_pair_args(e, &A, false);  // every r^2 is wanted: no early termination
  A.items = e->d_items + i0;
  A.item_general = e->d_item_general + i0;
  A.n_items = static_cast<uint32_t>(i1 - i0);
  A.thresh = 0.0;
  A.r2_out = hits ? nullptr : out_buf.p;
  A.r2_hits = hits ? out_buf.as<ldp_r2_hit>() : nullptr;
  A.r2_hit_capacity = hits ? hits->capacity : 0;
  A.r2_min = hits ? hits->min_r2 : 0.0;
  A.r2_ld = 0;
  A.r2_row_first = l_first;
  A.r2_row_end = l_end;
  A.r2_band_base = e->pair_off[l_first];
  A.r2_float = as_float ? 1 : 0;
  EventSet<4> evset;
  hipEvent_t* evk = evset.ev;
  HIP_TRY(e, evset.create());
  DevBuf mf_buf;
  uint64_t mf_products = 0;
  const bool on_mfma = r2_on_matrix_pipe(e);
  hipError_t krc;
  if (on_mfma) {
    std::vector<std::pair<uint32_t, uint32_t>> runs;
    for (uint32_t sk : e->owned) {
      runs.emplace_back(e->subs[sk].local_first, e->subs[sk].len);
    }
    rc = attach_mfma_plan(e, &A, runs, e->lo_local.data(), l_first, l_end, &mf_buf, &mf_products);
    if (rc) {
      return rc;
    }
    krc = launch_pair_mfma(A, e->stream, evk);  // evk[0..2]: complete-data kernel | missing-calls kernel
    (void)hipEventRecord(evk[3], e->stream);
  } else {
    krc = launch_pair_tiles(A, e->max_rows, e->stream, evk);
  }
  if (krc != hipSuccess) {
    return hipfail(e, krc, "pair kernel launch");
  }
  if (hits) {
    HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin, e->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint64_t found = e->h_counters_pin[3];
    *hits->count = found;
    const uint64_t stored = std::min<uint64_t>(found, hits->capacity);
    if (stored) {
      HIP_TRY(e, hipMemcpy(hits->out, out_buf.p, stored * sizeof(ldp_r2_hit), hipMemcpyDeviceToHost));
      for (uint64_t q = 0; q < stored; ++q) {  // the kernel works in local (paired-variant) order
        hits->out[q].first = e->local_to_global[hits->out[q].first];
        hits->out[q].second = e->local_to_global[hits->out[q].second];
      }
    }
  } else {
    HIP_TRY(e, hipMemcpyAsync(out, out_buf.p, n_elems * esz, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  float kms_fast = 0.f, kms_general = 0.f;
  if (on_mfma) {
    if (A.n_mf_wgs) {
      HIP_TRY(e, hipEventElapsedTime(&kms_fast, evk[0], evk[1]));
      HIP_TRY(e, hipEventElapsedTime(&kms_general, evk[1], evk[2]));
    }
  } else if (A.n_items) {
    HIP_TRY(e, hipEventElapsedTime(&kms_fast, evk[0], evk[1]));
    HIP_TRY(e, hipEventElapsedTime(&kms_general, evk[2], evk[3]));
  }
  e->ctr.candidate_pairs = n_elems;
  e->ctr.ms_pair_fast = kms_fast;
  e->ctr.ms_pair_general = kms_general;
  e->ctr.ms_pair_kernel = kms_fast + kms_general;
  e->ctr.ms_run_total = now_ms() - t_start;
  e->ctr.pair_kernel_launches = A.n_items ? 1 : 0;
  return LDP_OK;
}

// rows [row_first, row_first+row_ct) of the all-pairs plan: dense into `out` (hits == nullptr) or filtered into hits->out
// as_float 2: the six integers of every pair (ldp_pair_stats_t) instead of their r^2; out_on_device: `out` is device memory of this
// engine's device (left there, no diagonal: the chrX-weighted r^2 below combines two engines' tuples on the device)
int r2_rows_impl(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t ld_elems, const HitRequest* hits,
                 uint32_t col_first = 0, uint32_t col_end = 0xffffffffu, bool out_on_device = false) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned || !e->matrix_mode) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants_matrix() first");
  }
  if (hits) {
    if ((static_cast<uint64_t>(row_first) + row_ct > e->variant_ct) || (hits->capacity && !hits->out) || !hits->count) {
      return fail(e, LDP_ERR_INVALID, "row range / hit buffer out of bounds");
    }
    *hits->count = 0;
  } else if ((static_cast<uint64_t>(row_first) + row_ct > e->variant_ct) || (row_ct && !out) || (col_first > col_end) ||
             (ld_elems + col_first < std::min<uint64_t>(static_cast<uint64_t>(row_first) + row_ct, col_end))) {
    return fail(e, LDP_ERR_INVALID, "row range / leading dimension out of bounds");
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    if (!e->loaded[l]) {
      return fail(e, LDP_ERR_STATE, "genotypes missing for a variant (ldp_load_genotypes)");
    }
  }
  if (!row_ct) {
    return LDP_OK;
  }
  const double t_start = now_ms();
  HIP_TRY(e, hipSetDevice(e->device));
  // tiles of the rows' lower triangle: (32 seconds) x (all distances 1..j), <= 128 distances per block
  std::vector<WorkItem> items;
  uint32_t max_rows = 0;
  uint64_t computed = 0, cand = 0;
  const uint32_t row_end = row_first + row_ct;
  const bool on_mfma = r2_on_matrix_pipe(e);  // (then the popcount work items are never launched: not built, not uploaded)
  for (uint32_t j0 = row_first; j0 < row_end; j0 += kTileJ) {
    const uint32_t jend = std::min(j0 + kTileJ, row_end);
    const uint32_t dmax = jend - 1;
    for (uint32_t j = j0; j < jend; ++j) {
      const uint32_t hi = std::min(j, col_end);
      cand += (hi > col_first) ? (hi - col_first) : 0;
    }
    if ((!dmax) || on_mfma) {
      continue;
    }
    const uint32_t units = (dmax + 7) / 8;
    const uint32_t blocks = (units + kMaxUnitsPerBlock - 1) / kMaxUnitsPerBlock;
    const uint32_t base = units / blocks;
    const uint32_t extra = units % blocks;
    uint32_t d0 = 1;
    for (uint32_t blk = 0; blk < blocks; ++blk) {
      const uint32_t u = base + ((blk < extra) ? 1 : 0);
      WorkItem it;
      it.j0 = j0;
      it.jend = jend;
      it.d0 = d0;
      it.units = u;
      it.sfirst = 0;
      it.send = e->local_ct;
      items.push_back(it);
      max_rows = std::max(max_rows, tile_rows(d0, u));
      computed += static_cast<uint64_t>(u) * 8 * kTileJ;
      d0 += 8 * u;
    }
  }
  const size_t esz = (as_float == 2) ? sizeof(ldp_pair_stats_t) : (as_float ? sizeof(float) : sizeof(double));
  const uint64_t out_elems = hits ? 0 : (static_cast<uint64_t>(row_ct) * ld_elems);
  DevBuf out_buf, items_buf, general_buf;
  void* d_out = nullptr;
  if (hits) {
    HIP_TRY(e, hipMalloc(&out_buf.p, std::max<uint64_t>(hits->capacity, 1) * sizeof(ldp_r2_hit)));
    HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
  } else {
    if (out_on_device) {
      d_out = out;
    } else {
      HIP_TRY(e, hipMalloc(&out_buf.p, out_elems * esz));
      d_out = out_buf.p;
    }
    HIP_TRY(e, hipMemsetAsync(d_out, 0, out_elems * esz, e->stream));
  }
  WorkItem* d_items = nullptr;
  uint8_t* d_general = nullptr;
  if (!items.empty()) {
    HIP_TRY(e, hipMalloc(&items_buf.p, items.size() * sizeof(WorkItem)));
    HIP_TRY(e, hipMalloc(&general_buf.p, items.size()));
    d_items = items_buf.as<WorkItem>();
    d_general = general_buf.as<uint8_t>();
    HIP_TRY(e, hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(WorkItem), hipMemcpyHostToDevice, e->stream));
  }
  PairKernelArgs A;
  fill_pair_args(e, &A, false);  // (defaults, incl. "no matrix-pipe work attached"; the matrix-mode fields follow)
  A.planes = e->d_planes;
  A.row_dwords = e->row_dwords;
  A.chunks = e->chunks;
  A.founder_ct = e->P.founder_ct;
  A.recs = e->d_recs;
  A.lo = e->d_lo;  // all zero in matrix mode
  A.row_off = e->d_row_off;
  A.pred = e->d_pred;
  A.items = d_items;
  A.n_items = static_cast<uint32_t>(items.size());
  A.plane_base_variant = 0;
  A.thresh = 0.0;
  A.stats = nullptr;
  A.pair_off = nullptr;
  A.counters = e->d_counters;
  A.item_general = d_general;
  A.cp_stats = nullptr;  // every r^2 is wanted: no early termination
  A.cp_gen = nullptr;
  for (int k = 0; k < kCheckpoints; ++k) {
    A.checkpoint_chunk[k] = 0xffffffffu;
  }
  A.n_checkpoints = 0;
  A.r2_out = d_out;
  A.r2_hits = hits ? out_buf.as<ldp_r2_hit>() : nullptr;
  A.r2_hit_capacity = hits ? hits->capacity : 0;
  A.r2_min = hits ? hits->min_r2 : 0.0;
  A.r2_ld = ld_elems;
  A.r2_row_first = row_first;
  A.r2_row_end = row_end;
  A.r2_col_first = col_first;
  A.r2_col_end = col_end;
  A.r2_band_base = 0;
  A.r2_float = static_cast<uint32_t>(as_float);
  EventSet<4> evset;
  hipEvent_t* evk = evset.ev;
  HIP_TRY(e, evset.create());
  DevBuf mf_buf, tile_buf;
  uint64_t mf_products = 0, tile_products = 0;
  hipError_t krc;
  if (on_mfma) {
    const std::vector<std::pair<uint32_t, uint32_t>> runs(1, std::make_pair(0u, e->local_ct));
    rc = attach_mfma_plan(e, &A, runs, nullptr, row_first, row_end, &mf_buf, &mf_products, col_first, col_end, &tile_buf, &tile_products);
    if (rc) {
      return rc;
    }
    krc = launch_pair_mfma(A, e->stream, evk);  // evk[0..2]: complete-data kernels (tiles + parallelogram workgroups) | missing-calls kernel
    (void)hipEventRecord(evk[3], e->stream);
    computed = (tile_products ? tile_products : mf_products) * kMfBlock * kMfBlock;  // (what a complete-data launch multiplies: the tiles' products where there are tiles)
  } else {
    krc = launch_pair_tiles(A, std::max<uint32_t>(max_rows, kTileJ + 8), e->stream, evk);
  }
  if (krc != hipSuccess) {
    return hipfail(e, krc, "pair kernel launch");
  }
  if (hits) {
    HIP_TRY(e, hipMemcpyAsync(e->h_counters_pin, e->d_counters, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const uint64_t found = e->h_counters_pin[3];
    *hits->count = found;
    const uint64_t stored = std::min<uint64_t>(found, hits->capacity);
    if (stored) {
      HIP_TRY(e, hipMemcpy(hits->out, out_buf.p, stored * sizeof(ldp_r2_hit), hipMemcpyDeviceToHost));
    }
  } else if (out_on_device) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  } else {
    HIP_TRY(e, hipMemcpyAsync(out, d_out, out_elems * esz, hipMemcpyDeviceToHost, e->stream));
    rc = fetch_recs(e);  // diagonal needs each variant's own variance
    if (rc) {
      return rc;
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  float kms_fast = 0.f, kms_general = 0.f;
  if (on_mfma) {
    if (A.n_mf_wgs) {
      HIP_TRY(e, hipEventElapsedTime(&kms_fast, evk[0], evk[1]));
      HIP_TRY(e, hipEventElapsedTime(&kms_general, evk[1], evk[2]));
    }
  } else if (!items.empty()) {
    HIP_TRY(e, hipEventElapsedTime(&kms_fast, evk[0], evk[1]));
    HIP_TRY(e, hipEventElapsedTime(&kms_general, evk[2], evk[3]));
  }
  // diagonal: r^2(v, v) through the same formula = 1.0, or NaN when the variant has no variance
  for (uint32_t j = row_first; (!hits) && (as_float != 2) && (!out_on_device) && (j < row_end); ++j) {
    if ((j < col_first) || (j >= col_end)) {
      continue;
    }
    const ldp_variant_rec& r = e->recs[j];
    const int64_t var = static_cast<int64_t>(r.ssq) * static_cast<int64_t>(r.nm_ct) - static_cast<int64_t>(r.sum) * static_cast<int64_t>(r.sum);
    const bool defined = r.nm_ct && (static_cast<double>(var) * static_cast<double>(var) != 0.0);
    const uint64_t idx = static_cast<uint64_t>(j - row_first) * ld_elems + (j - col_first);
    if (as_float) {
      const uint32_t bits = defined ? 0x3f800000u : 0xffc00000u;
      memcpy(static_cast<float*>(out) + idx, &bits, 4);
    } else {
      const uint64_t bits = defined ? 0x3ff0000000000000ull : 0xfff8000000000000ull;
      memcpy(static_cast<double*>(out) + idx, &bits, 8);
    }
  }
  e->ctr.candidate_pairs = cand;
  e->ctr.computed_pairs = computed;
  e->ctr.ms_pair_fast = kms_fast;
  e->ctr.ms_pair_general = kms_general;
  e->ctr.ms_pair_kernel = kms_fast + kms_general;
  e->ctr.ms_run_total = now_ms() - t_start;
  return LDP_OK;
}

// chrX pairs of a dense block (ComputeXR2, plink2_ld.cc:7122-7190): both engines' tuples from the pair kernels, combined on the device
// (x_weighted_kernel).  Only the rectangles that hold such pairs are computed: the chrX rows against all columns, the other rows
// against the chrX columns.
int r2_x_block_impl(ldp_engine* e, ldp_engine* male, const uint8_t* is_x, const uint8_t* flip_all, const uint8_t* flip_male, uint32_t row_first, uint32_t row_ct,
                    uint32_t col_first, uint32_t col_ct, int as_float, int unsquared, void* out, uint64_t ld_elems, const HitRequest* hits) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned || !e->matrix_mode || (male && (!male->planned || !male->matrix_mode))) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants_matrix() first (both engines)");
  }
  if (male && ((male->variant_ct != e->variant_ct) || (male->device != e->device))) {
    return fail(e, LDP_ERR_INVALID, "the male founders' engine must hold the same variants on the same device");
  }
  if (!is_x || (static_cast<uint64_t>(row_first) + row_ct > e->variant_ct) || (static_cast<uint64_t>(col_first) + col_ct > e->variant_ct) ||
      (hits ? ((hits->capacity && !hits->out) || !hits->count) : ((row_ct && col_ct && !out) || (ld_elems < col_ct)))) {
    return fail(e, LDP_ERR_INVALID, "is_x missing / block or output out of bounds");
  }
  if (hits) {
    *hits->count = 0;
  }
  if (!row_ct || !col_ct) {
    return LDP_OK;
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  const uint32_t m = e->variant_ct, col_end = col_first + col_ct, row_end = row_first + row_ct;
  // the chrX columns of the block
  uint32_t xc_lo = col_end, xc_hi = col_first;
  for (uint32_t i = col_first; i < col_end; ++i) {
    if (is_x[i]) {
      xc_lo = std::min(xc_lo, i);
      xc_hi = i + 1;
    }
  }
  DevBuf flags_buf, ta_buf, tm_buf, val_buf, hit_buf, ctr_buf;
  HIP_TRY(e, hipMalloc(&flags_buf.p, 3ull * m));
  uint8_t* d_is_x = flags_buf.as<uint8_t>();
  uint8_t* d_flip_all = flip_all ? d_is_x + m : nullptr;
  uint8_t* d_flip_male = (male && flip_male) ? d_is_x + 2ull * m : nullptr;
  HIP_TRY(e, hipMemcpyAsync(d_is_x, is_x, m, hipMemcpyHostToDevice, e->stream));
  if (d_flip_all) {
    HIP_TRY(e, hipMemcpyAsync(d_flip_all, flip_all, m, hipMemcpyHostToDevice, e->stream));
  }
  if (d_flip_male) {
    HIP_TRY(e, hipMemcpyAsync(d_flip_male, flip_male, m, hipMemcpyHostToDevice, e->stream));
  }
  if (hits) {
    HIP_TRY(e, hipMalloc(&hit_buf.p, std::max<uint64_t>(hits->capacity, 1) * sizeof(ldp_r2_hit)));
    HIP_TRY(e, hipMalloc(&ctr_buf.p, sizeof(unsigned long long)));
    HIP_TRY(e, hipMemsetAsync(ctr_buf.p, 0, sizeof(unsigned long long), e->stream));
  }
  // row chunks of at most ~1 GiB of tuples per engine
  const size_t esz = as_float ? sizeof(float) : sizeof(double);
  uint32_t rows_per = static_cast<uint32_t>(std::max<uint64_t>(32, ((1ull << 30) / sizeof(ldp_pair_stats_t)) / col_ct) & ~31ull);
  if (const char* dbg = getenv("LDP_DEBUG_X_ROWS")) {  // (test hook: many small chunks)
    rows_per = static_cast<uint32_t>(std::max(1, atoi(dbg)));
  }
  HIP_TRY(e, hipMalloc(&ta_buf.p, static_cast<uint64_t>(std::min(rows_per, row_ct)) * col_ct * sizeof(ldp_pair_stats_t)));
  if (male) {
    HIP_TRY(e, hipMalloc(&tm_buf.p, static_cast<uint64_t>(std::min(rows_per, row_ct)) * col_ct * sizeof(ldp_pair_stats_t)));
  }
  std::vector<uint8_t> h_val;
  if (!hits) {
    HIP_TRY(e, hipMalloc(&val_buf.p, static_cast<uint64_t>(std::min(rows_per, row_ct)) * col_ct * esz));
  }
  for (uint32_t r0 = row_first; r0 < row_end; r0 += rows_per) {
    const uint32_t rows = std::min(rows_per, row_end - r0);
    bool any_x_row = false;
    for (uint32_t j = r0; j < r0 + rows; ++j) {
      any_x_row = any_x_row || (is_x[j] != 0);
    }
    // (a chunk without a chrX row only needs the chrX columns; pairs are i < j: nothing right of the chunk's last row either)
    const uint32_t c0 = any_x_row ? col_first : xc_lo;
    const uint32_t c1 = std::min(any_x_row ? col_end : xc_hi, r0 + rows - 1);
    if (c0 >= c1) {
      continue;
    }
    const uint32_t cols = c1 - c0;
    if ((rc = r2_rows_impl(e, r0, rows, 2, ta_buf.p, cols, nullptr, c0, c1, true))) {
      return rc;
    }
    if (male && (rc = r2_rows_impl(male, r0, rows, 2, tm_buf.p, cols, nullptr, c0, c1, true))) {
      return fail(e, rc, std::string("male founders' engine: ") + ldp_last_error(male));
    }
    XWeightedArgs X;
    X.all = ta_buf.as<ldp_pair_stats_t>();
    X.male = male ? tm_buf.as<ldp_pair_stats_t>() : nullptr;
    X.rows = rows;
    X.cols = cols;
    X.row_first = r0;
    X.col_first = c0;
    X.is_x = d_is_x;
    X.flip_all = d_flip_all;
    X.flip_male = d_flip_male;
    X.unsquared = unsquared ? 1u : 0u;
    X.as_float = as_float ? 1u : 0u;
    X.out = val_buf.p;
    X.out_ld = cols;
    X.hits = hits ? hit_buf.as<ldp_r2_hit>() : nullptr;
    X.hit_capacity = hits ? hits->capacity : 0;
    X.hit_count = static_cast<unsigned long long*>(ctr_buf.p);
    X.min_r2 = hits ? hits->min_r2 : 0.0;
    const hipError_t krc = launch_x_weighted(X, e->stream);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "x_weighted_kernel launch");
    }
    if (hits) {
      HIP_TRY(e, hipStreamSynchronize(e->stream));  // (the next chunk's tuples overwrite these from the engines' own streams)
      continue;
    }
    // the chunk's values back, and the pairs with a chrX variant into the caller's block (everything else stays as it was)
    h_val.resize(static_cast<size_t>(rows) * cols * esz);
    HIP_TRY(e, hipMemcpyAsync(h_val.data(), val_buf.p, h_val.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (uint32_t q = 0; q < rows; ++q) {
      const uint32_t j = r0 + q;
      const uint32_t i_end = std::min(c1, j);
      uint8_t* dst = static_cast<uint8_t*>(out) + (static_cast<uint64_t>(j - row_first) * ld_elems) * esz;
      const uint8_t* src = h_val.data() + static_cast<size_t>(q) * cols * esz;
      if (is_x[j]) {
        if (i_end > c0) {
          memcpy(dst + static_cast<size_t>(c0 - col_first) * esz, src, static_cast<size_t>(i_end - c0) * esz);
        }
      } else {
        for (uint32_t i = std::max(c0, xc_lo); i < std::min(i_end, xc_hi); ++i) {
          if (is_x[i]) {
            memcpy(dst + static_cast<size_t>(i - col_first) * esz, src + static_cast<size_t>(i - c0) * esz, esz);
          }
        }
      }
    }
  }
  if (hits) {
    unsigned long long found = 0;
    HIP_TRY(e, hipMemcpyAsync(&found, ctr_buf.p, sizeof(found), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    *hits->count = found;
    const uint64_t stored = std::min<uint64_t>(found, hits->capacity);
    if (stored) {
      HIP_TRY(e, hipMemcpy(hits->out, hit_buf.p, stored * sizeof(ldp_r2_hit), hipMemcpyDeviceToHost));
    }
  }
  return LDP_OK;
}
}  // namespace

extern "C" {

int ldp_r2_unphased_block_x(ldp_engine* e, ldp_engine* male, const uint8_t* is_x, const uint8_t* flip_all, const uint8_t* flip_male, uint32_t row_first, uint32_t row_ct,
                            uint32_t col_first, uint32_t col_ct, int as_float, int unsquared, void* out, uint64_t ld_elems) {
  return r2_x_block_impl(e, male, is_x, flip_all, flip_male, row_first, row_ct, col_first, col_ct, as_float, unsquared, out, ld_elems, nullptr);
}

int ldp_r2_unphased_block_x_hits(ldp_engine* e, ldp_engine* male, const uint8_t* is_x, const uint8_t* flip_all, const uint8_t* flip_male, uint32_t row_first,
                                 uint32_t row_ct, uint32_t col_first, uint32_t col_ct, int unsquared, double min_r2, ldp_r2_hit* out, uint64_t capacity, uint64_t* count) {
  HitRequest hr{min_r2, out, capacity, count};
  return r2_x_block_impl(e, male, is_x, flip_all, flip_male, row_first, row_ct, col_first, col_ct, 0, unsquared, nullptr, 0, &hr);
}

int ldp_pair_stats_block(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, ldp_pair_stats_t* out, uint64_t ld_elems) {
  return r2_rows_impl(e, row_first, row_ct, 2, out, ld_elems, nullptr, col_first, col_first + col_ct);
}

int ldp_set_r_signed(ldp_engine* e, int mode) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if ((mode < 0) || (mode > 2)) {
    return fail(e, LDP_ERR_INVALID, "ldp_set_r_signed: mode must be 0 (r^2), 1 (r, major-allele orientation) or 2 (r, REF orientation)");
  }
  e->r_signed = static_cast<uint32_t>(mode);
  return LDP_OK;
}

int ldp_r2_unphased_rows(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t ld_elems) {
  return r2_rows_impl(e, row_first, row_ct, as_float, out, ld_elems, nullptr);
}

int ldp_r2_unphased_block(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, int as_float, void* out, uint64_t ld_elems) {
  return r2_rows_impl(e, row_first, row_ct, as_float, out, ld_elems, nullptr, col_first, col_first + col_ct);
}

int ldp_r2_unphased_block_hits(ldp_engine* e, uint32_t row_first, uint32_t row_ct, uint32_t col_first, uint32_t col_ct, double min_r2, ldp_r2_hit* out,
                               uint64_t capacity, uint64_t* count) {
  HitRequest hr{min_r2, out, capacity, count};
  return r2_rows_impl(e, row_first, row_ct, 0, nullptr, static_cast<uint64_t>(row_first) + row_ct, &hr, col_first, col_first + col_ct);
}

int ldp_r2_unphased_hits(ldp_engine* e, uint32_t row_first, uint32_t row_ct, double min_r2, ldp_r2_hit* out, uint64_t capacity, uint64_t* count) {
  HitRequest hr{min_r2, out, capacity, count};
  if (e && e->planned && e->band_r2_mode) {  // windowed plan (ldp_set_variants_vcor): the band's pairs
    return r2_band_impl(e, row_first, row_ct, 0, nullptr, 0, &hr);
  }
  return r2_rows_impl(e, row_first, row_ct, 0, nullptr, static_cast<uint64_t>(row_first) + row_ct, &hr);
}

int ldp_r2_unphased_band_rows(ldp_engine* e, uint32_t row_first, uint32_t row_ct, int as_float, void* out, uint64_t capacity_elems) {
  return r2_band_impl(e, row_first, row_ct, as_float, out, capacity_elems, nullptr);
}

int ldp_get_subcontigs(const ldp_engine* e, uint32_t* ct, uint32_t* info, uint32_t info_capacity_pairs) {
  if (!e || !e->planned || !ct) {
    return LDP_ERR_STATE;
  }
  *ct = static_cast<uint32_t>(e->subs.size());
  if (info) {
    const uint32_t n = std::min<uint32_t>(*ct, info_capacity_pairs);
    for (uint32_t k = 0; k < n; ++k) {
      info[2 * k] = e->subs[k].len;
      info[2 * k + 1] = e->subs[k].first;
    }
  }
  return LDP_OK;
}

int ldp_set_shard(ldp_engine* e, uint32_t rank, uint32_t world, uint32_t* owner) {
  if (!e || !e->planned) {
    return e ? fail(e, LDP_ERR_STATE, "ldp_set_variants() first") : LDP_ERR_INVALID;
  }
  if (!world || (rank >= world)) {
    return fail(e, LDP_ERR_INVALID, "rank/world out of range");
  }
  // LPT: longest subcontig first onto the least-loaded rank (ties: lower rank; equal lengths: file order)
  std::vector<uint32_t> order(e->subs.size());
  for (uint32_t k = 0; k < order.size(); ++k) {
    order[k] = k;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return e->subs[a].len > e->subs[b].len; });
  std::vector<uint64_t> load(world, 0);
  for (uint32_t k : order) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < world; ++r) {
      if (load[r] < load[best]) {
        best = r;
      }
    }
    e->subs[k].owner = best;
    load[best] += e->subs[k].len;
  }
  if (owner) {
    for (uint32_t k = 0; k < e->subs.size(); ++k) {
      owner[k] = e->subs[k].owner;
    }
  }
  e->rank = rank;
  e->world = world;
  build_shard(e);
  e->ctr.owned_subcontig_ct = static_cast<uint32_t>(e->owned.size());
  return LDP_OK;
}

// ---- the one exchange step of a multi-GPU prune, from the C/C++ host: RCCL, bound at run time -----------------------
// (dlopen: the library also has to load where no RCCL is installed, and inside a process that brought its own copy)
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static const Rccl& rccl() {
  static const Rccl R = []() {
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) {
        break;
      }
    }
    if (r.lib) {
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
      r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
      r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
      r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
      r.ok = r.AllGather && r.CommCount && r.CommUserRank && r.CommInitAll && r.CommDestroy;
    }
    return r;
  }();
  return R;
}
}  // namespace

int ldp_comm_init_all(int n, const int* devices, void** comms) {
  if ((n < 1) || !comms) {
    return LDP_ERR_INVALID;
  }
  const Rccl& R = rccl();
  if (!R.ok) {
    return LDP_ERR_UNSUPPORTED;
  }
  std::vector<ncclComm_t> c(n, nullptr);
  if (R.CommInitAll(c.data(), n, devices) != ncclSuccess) {
    return LDP_ERR_GPU;
  }
  for (int k = 0; k < n; ++k) {
    comms[k] = c[k];
  }
  return LDP_OK;
}

namespace {
// communicators ldp_allgather_removed() had to abort: ncclCommAbort has already released them, a later ldp_comm_destroy() is a no-op
std::mutex g_aborted_mu;
std::set<void*> g_aborted;
}  // namespace

void ldp_comm_destroy(void* comm) {
  if (comm && rccl().ok) {
    {
      std::lock_guard<std::mutex> lk(g_aborted_mu);
      if (g_aborted.erase(comm)) {
        return;
      }
    }
    (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm));
  }
}

// ---- the exchange in three separable steps: pack (host), transport, stitch (host) ---------------------------------------
// Every rank knows every rank's segment: its owned subcontigs in file order (the LPT assignment is deterministic).  A segment
// is padded to the longest one, so that ONE all-gather of equal pieces is the allgatherv.
namespace {
uint64_t shard_segment_words(const ldp_engine* e) {
  std::vector<uint64_t> seg_bits(std::max<uint32_t>(e->world, 1), 0);
  for (const Subcontig& s : e->subs) {
    seg_bits[s.owner] += s.len;
  }
  return std::max<uint64_t>((*std::max_element(seg_bits.begin(), seg_bits.end()) + 63) / 64, 1);
}
bool shard_ready(const ldp_engine* e) { return e && e->planned && !e->matrix_mode && !e->band_r2_mode; }
}  // namespace

int ldp_shard_segment_words(const ldp_engine* e, uint64_t* words) {
  if (!shard_ready(e) || !words) {
    return e ? LDP_ERR_STATE : LDP_ERR_INVALID;
  }
  *words = shard_segment_words(e);
  return LDP_OK;
}

int ldp_pack_removed_segment(const ldp_engine* e, const uint64_t* removed_local, uint64_t* segment) {
  if (!shard_ready(e)) {
    return e ? LDP_ERR_STATE : LDP_ERR_INVALID;
  }
  if (!removed_local || !segment) {
    return LDP_ERR_INVALID;
  }
  const uint64_t words = shard_segment_words(e);
  std::fill(segment, segment + words, 0ull);
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    const uint32_t g = e->local_to_global[l];
    if ((removed_local[g >> 6] >> (g & 63)) & 1ull) {
      segment[l >> 6] |= 1ull << (l & 63);
    }
  }
  return LDP_OK;
}

// the stitch (plink2_ld.cc:1418-1426, CopyBitarrRange per thread): segment bits -> global variant order
int ldp_stitch_removed_segments(const ldp_engine* e, const uint64_t* segments, uint64_t* removed_global) {
  if (!shard_ready(e)) {
    return e ? LDP_ERR_STATE : LDP_ERR_INVALID;
  }
  if (!segments || !removed_global) {
    return LDP_ERR_INVALID;
  }
  const uint64_t words = shard_segment_words(e);
  const size_t gwords = (static_cast<size_t>(e->variant_ct) + 63) / 64;
  std::fill(removed_global, removed_global + gwords, 0ull);
  std::vector<uint64_t> pos(std::max<uint32_t>(e->world, 1), 0);
  for (const Subcontig& s : e->subs) {
    const uint64_t* seg = segments + static_cast<size_t>(s.owner) * words;
    uint64_t& p = pos[s.owner];
    for (uint32_t v = 0; v < s.len; ++v, ++p) {
      if ((seg[p >> 6] >> (p & 63)) & 1ull) {
        const uint32_t g = s.first + v;
        removed_global[g >> 6] |= 1ull << (g & 63);
      }
    }
  }
  return LDP_OK;
}

int ldp_allgather_removed(ldp_engine* e, void* comm, const uint64_t* removed_local, uint64_t* removed_global) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  // A rank that cannot enter the collective must not leave its peers waiting in it: every exit before the ncclAllGather
  // is enqueued aborts the communicator (ncclCommAbort wakes the other ranks with an error instead of a hang).
  const Rccl& R = rccl();
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  auto leave = [&](int code, const std::string& msg) {
    if (c && R.ok && R.CommAbort) {
      (void)R.CommAbort(c);
      std::lock_guard<std::mutex> lk(g_aborted_mu);
      g_aborted.insert(c);
    }
    return fail(e, code, msg);
  };
  if (!shard_ready(e)) {
    return leave(LDP_ERR_STATE, "ldp_set_variants() (+ ldp_set_shard) first");
  }
  if (!comm || !removed_local || !removed_global) {
    return leave(LDP_ERR_INVALID, "null argument");
  }
  if (!R.ok) {
    return fail(e, LDP_ERR_UNSUPPORTED, "RCCL (librccl.so.1) is not available");
  }
  bind_gpu(e);
  if (!e->gpu_ok) {
    return leave(LDP_ERR_GPU, "no usable HIP device");
  }
  int count = 0, urank = -1;
  if ((R.CommCount(c, &count) != ncclSuccess) || (R.CommUserRank(c, &urank) != ncclSuccess)) {
    return leave(LDP_ERR_GPU, "ncclCommCount / ncclCommUserRank failed");
  }
  if ((static_cast<uint32_t>(count) != e->world) || (static_cast<uint32_t>(urank) != e->rank)) {
    return leave(LDP_ERR_INVALID, "the communicator's size / rank differ from ldp_set_shard()'s");
  }
  const uint64_t words = shard_segment_words(e);
  std::vector<uint64_t> mine(words, 0);
  (void)ldp_pack_removed_segment(e, removed_local, mine.data());
  DevBuf send, recv;
  hipError_t hrc = hipSetDevice(e->device);
  if (hrc == hipSuccess) {
    hrc = hipMalloc(&send.p, words * sizeof(uint64_t));
  }
  if (hrc == hipSuccess) {
    hrc = hipMalloc(&recv.p, words * sizeof(uint64_t) * e->world);
  }
  if (hrc == hipSuccess) {
    hrc = hipMemcpyAsync(send.p, mine.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream);
  }
  if (hrc != hipSuccess) {
    return leave(LDP_ERR_GPU, std::string("setup of the all-gather buffers: ") + hipGetErrorString(hrc));
  }
  const ncclResult_t nrc = R.AllGather(send.p, recv.p, words, ncclUint64, c, e->stream);
  if (nrc != ncclSuccess) {
    return leave(LDP_ERR_GPU, std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(nrc) : "failed"));
  }
  std::vector<uint64_t> all(words * e->world);
  HIP_TRY(e, hipMemcpyAsync(all.data(), recv.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return ldp_stitch_removed_segments(e, all.data(), removed_global);
}

int ldp_get_band(const ldp_engine* e, uint32_t* lo, uint64_t* candidate_pairs) {
  if (!e || !e->planned) {
    return LDP_ERR_STATE;
  }
  if (lo) {
    memcpy(lo, e->lo_global.data(), e->variant_ct * sizeof(uint32_t));
  }
  if (candidate_pairs) {
    uint64_t tot = 0;
    for (uint32_t v = 0; v < e->variant_ct; ++v) {
      tot += v - e->lo_global[v];
    }
    *candidate_pairs = tot;
  }
  return LDP_OK;
}

uint64_t ldp_phased_phase_offset(uint32_t hap_ct) { return ((static_cast<uint64_t>(hap_ct / 2) + 3) / 4 + 3) & ~static_cast<uint64_t>(3); }
uint64_t ldp_phased_row_bytes(uint32_t hap_ct) { return ldp_phased_phase_offset(hap_ct) + (static_cast<uint64_t>(hap_ct / 2) + 7) / 8; }

namespace {
// ldp_load_genotypes(); d_row_inverse / h_row_inverse (optional, device and host copies of the same n bytes): rows that are
// LDP_GENO_INVERSE whatever `encoding` says (the collapsed multiallelic rows of ldp_load_pgen_records)
// src_fd >= 0: the rows are read from that file descriptor at src_off + (variant - first_variant) * stride_bytes (pread straight into
// the pinned ring; `geno` is unused and location is LDP_MEM_HOST)
int load_rows_impl(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* geno, uint64_t stride_bytes, int location, int encoding,
                   const uint8_t* d_row_inverse, const uint8_t* h_row_inverse, int src_fd = -1, uint64_t src_off = 0) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants() first");
  }
  const bool phased = (encoding & LDP_GENO_PHASED) != 0;
  const bool mapped = (encoding & LDP_GENO_MAPPED) != 0;
  const int base_encoding = encoding & ~(LDP_GENO_PHASED | LDP_GENO_MAPPED);
  if ((base_encoding < LDP_GENO_INVERSE) || (base_encoding > LDP_GENO_BED) || (phased && (base_encoding == LDP_GENO_BED)) ||
      ((location != LDP_MEM_HOST) && (location != LDP_MEM_DEVICE))) {
    return fail(e, LDP_ERR_INVALID, "bad encoding/location");
  }
  if (mapped && (phased || e->sample_map.empty())) {
    return fail(e, LDP_ERR_INVALID, "LDP_GENO_MAPPED needs ldp_set_sample_map() and unphased rows");
  }
  if (phased && (e->P.founder_ct & 1)) {
    return fail(e, LDP_ERR_INVALID, "LDP_GENO_PHASED rows need an even founder_ct (haplotype count = 2 x samples)");
  }
  if ((static_cast<uint64_t>(first_variant) + n > e->variant_ct) || (n && !geno && (src_fd < 0))) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds");
  }
  const uint64_t row_bytes = mapped ? ((static_cast<uint64_t>(e->map_raw_sample_ct) + 3) / 4)
                                    : (phased ? ldp_phased_row_bytes(e->P.founder_ct) : ((static_cast<uint64_t>(e->P.founder_ct) + 3) / 4));
  if (stride_bytes < row_bytes) {
    return fail(e, LDP_ERR_INVALID, "stride smaller than a genotype row");
  }
  const double t_entry = now_ms();
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  const double t_planned = now_ms();
  HIP_TRY(e, hipSetDevice(e->device));
  const uint8_t* src = static_cast<const uint8_t*>(geno);
  // Host input goes through a 3-deep ring of pinned staging buffers: host threads gather rows into
  // pinned memory (packed to row_bytes rounded up to a dword, so the conversion kernel's wide loads stay aligned)
  // while the previous slot's H2D copy and prepare kernel are in flight.
  const uint64_t pack_stride = (row_bytes + 3) & ~static_cast<uint64_t>(3);
  size_t stage_rows = 0;
  if (location == LDP_MEM_HOST) {
    rc = ensure_staging(e);
    if (rc) {
      return rc;
    }
    stage_rows = std::max<size_t>(1, kStageBytes / pack_stride);
  }
  const uint64_t gather_stride = ((static_cast<uint64_t>(e->P.founder_ct) + 3) / 4 + 3) & ~static_cast<uint64_t>(3);
  size_t gather_rows = 0;
  if (mapped) {
    if (!e->d_sample_map) {  // (the device copy went with a re-plan or ldp_release_device(): the host copy is the master)
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_sample_map), e->sample_map.size() * sizeof(uint32_t)));
      HIP_TRY(e, hipMemcpy(e->d_sample_map, e->sample_map.data(), e->sample_map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    gather_rows = std::max<size_t>(1, std::max<size_t>(stage_rows, kStageBytes / gather_stride));
    if (location == LDP_MEM_HOST) {
      gather_rows = stage_rows;
    }
    if (e->gather_bytes < gather_rows * gather_stride) {
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      (void)hipFree(e->d_gather);
      e->d_gather = nullptr;
      e->gather_bytes = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_gather), gather_rows * gather_stride));
      e->gather_bytes = gather_rows * gather_stride;
    }
    if (e->extra_het_cap < gather_rows) {
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      (void)hipFree(e->d_extra_het);
      e->d_extra_het = nullptr;
      e->extra_het_cap = 0;
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_extra_het), gather_rows * sizeof(uint32_t)));
      e->extra_het_cap = gather_rows;
    }
  }
  // loading a variant a second time since the last epoch began starts a new epoch (see begin_load_epoch)
  for (uint32_t q = first_variant; q < first_variant + n; ++q) {
    const int64_t l = e->global_to_local[q];
    if ((l >= 0) && (e->load_tag[l] == e->load_epoch)) {
      rc = begin_load_epoch(e);
      if (rc) {
        return rc;
      }
      break;
    }
  }
  static const bool eager_always = (getenv("LDP_EAGER_PAIRS") != nullptr) && (strcmp(getenv("LDP_EAGER_PAIRS"), "1") == 0);
  const bool eager = (!e->matrix_mode) && (!e->band_r2_mode) && ((location == LDP_MEM_HOST) || eager_always);
  static const uint32_t copy_threads = []() {
    const char* c = getenv("LDP_DEBUG_COPY_THREADS");
    return (c && atoi(c) > 0) ? static_cast<uint32_t>(atoi(c)) : 16u;
  }();
  static const uint64_t copy_task_bytes = []() {
    const char* c = getenv("LDP_DEBUG_COPY_TASK_KB");
    return (c && atoi(c) > 0) ? (static_cast<uint64_t>(atoi(c)) << 10) : (1ull << 20);
  }();
  static const bool load_timing = getenv("LDP_DEBUG_LOAD_TIMING") != nullptr;  // host side of the file -> HBM leg, on stderr
  double t_wait_slot = 0.0, t_copy = 0.0;
  uint32_t n_slots = 0;
  const double t_call0 = now_ms();
  uint32_t slot = 0;
  uint32_t g = first_variant;
  const uint32_t gend = first_variant + n;
  while (g < gend) {
    // maximal run of owned variants consecutive both globally and locally (owned_runs: a million-variant call is one or a few
    // of them, and walking the variants instead kept the count pass from starting for most of a millisecond)
    auto it = std::upper_bound(e->owned_runs.begin(), e->owned_runs.end(), g, [](uint32_t v, const ldp_engine::OwnedRun& r) { return v < r.g_first; });
    if ((it == e->owned_runs.begin()) || (g >= (it - 1)->g_end)) {
      g = (it == e->owned_runs.end()) ? gend : std::min(gend, it->g_first);  // not owned: on to the next run
      continue;
    }
    const uint32_t run = std::min(gend, (it - 1)->g_end) - g;
    uint32_t done = 0;
    while (done < run) {
      uint32_t cnt = run - done;
      if (mapped) {
        cnt = static_cast<uint32_t>(std::min<size_t>(cnt, gather_rows));  // (one gather buffer, reused in stream order)
      }
      const uint32_t l0 = static_cast<uint32_t>(e->global_to_local[g + done]);
      // end the conversion launch where the next pair group becomes ready, so that group starts behind it
      if (eager && (e->next_group < e->groups.size())) {
        const uint32_t need = e->groups[e->next_group].need_end;
        if ((l0 < need) && (l0 + cnt > need)) {
          cnt = need - l0;
        }
      }
      const uint8_t* d_src;
      uint64_t d_stride = stride_bytes;
      if (location == LDP_MEM_HOST) {
        cnt = static_cast<uint32_t>(std::min<size_t>(cnt, stage_rows));
        const double tw0 = load_timing ? now_ms() : 0.0;
        HIP_TRY(e, hipEventSynchronize(e->stage_done[slot]));  // slot free again?
        const double tw1 = load_timing ? now_ms() : 0.0;
        t_wait_slot += tw1 - tw0;
        uint8_t* pin = e->h_stage[slot];
        const uint8_t* from = src + static_cast<uint64_t>(g + done - first_variant) * stride_bytes;
        const uint32_t kRowsPerTask = std::max<uint32_t>(1, static_cast<uint32_t>((copy_task_bytes) / pack_stride));
        const uint32_t tasks = (cnt + kRowsPerTask - 1) / kRowsPerTask;
        std::atomic<int> read_failed(0);
        static const bool use_pool = !(getenv("LDP_DEBUG_COPY_POOL") && (atoi(getenv("LDP_DEBUG_COPY_POOL")) == 0));
        auto copy_task = [&](uint32_t t) {
          const uint32_t r0 = t * kRowsPerTask;
          const uint32_t r1 = std::min(cnt, r0 + kRowsPerTask);
          if (src_fd >= 0) {
            // file -> pinned memory with pread: the kernel copies out of the page cache in large runs and no page of a 12 GB mapping
            // has to be faulted in first (a memcpy out of an mmap pays one minor fault per 4 KiB: ~28 GB/s on 16 threads, measured)
            const uint64_t base = src_off + static_cast<uint64_t>(g + done - first_variant) * stride_bytes;
            auto read_all = [&](uint8_t* dst, uint64_t off, uint64_t len) {
              while (len) {
                const ssize_t got = pread(src_fd, dst, len, static_cast<off_t>(off));
                if (got <= 0) {
                  if ((got < 0) && (errno == EINTR)) {
                    continue;
                  }
                  read_failed.store(1);
                  return;
                }
                dst += got;
                off += static_cast<uint64_t>(got);
                len -= static_cast<uint64_t>(got);
              }
            };
            if (stride_bytes == pack_stride) {
              read_all(pin + static_cast<uint64_t>(r0) * pack_stride, base + static_cast<uint64_t>(r0) * stride_bytes, static_cast<uint64_t>(r1 - r0 - 1) * pack_stride + row_bytes);
            } else {
              for (uint32_t r = r0; r < r1; ++r) {
                read_all(pin + static_cast<uint64_t>(r) * pack_stride, base + static_cast<uint64_t>(r) * stride_bytes, row_bytes);
              }
            }
            return;
          }
          if (stride_bytes == pack_stride) {
            const uint64_t len = static_cast<uint64_t>(r1 - r0 - 1) * pack_stride + row_bytes;  // the last row may end at the caller's buffer end
            memcpy(pin + static_cast<uint64_t>(r0) * pack_stride, from + static_cast<uint64_t>(r0) * pack_stride, len);
          } else {
            for (uint32_t r = r0; r < r1; ++r) {
              memcpy(pin + static_cast<uint64_t>(r) * pack_stride, from + static_cast<uint64_t>(r) * stride_bytes, row_bytes);
            }
          }
        };
        if (use_pool) {
          CopyPool::get().run(tasks, copy_threads, copy_task);
        } else {
          parallel_for(tasks, copy_threads, copy_task);
        }
        if (read_failed.load()) {
          return fail(e, LDP_ERR_INVALID, "reading the genotype rows from the file descriptor failed (short file or I/O error)");
        }
        if (load_timing) {
          t_copy += now_ms() - tw1;
          ++n_slots;
        }
        // file -> HBM: the pinned slot crosses PCIe (a) by an SDMA copy on one of TWO copy streams, alternating, which the engine's
        // stream then waits for (one copy queue tops out near 30 GB/s on this host; the staged rows are read by the count pass), or
        // (b) LDP_DEBUG_H2D_MODE=2: not at all -- the count pass reads the pinned rows over PCIe itself (host memory is
        // device-accessible), or (c) =0: the single in-order copy of rounds 1-3.
        static const int h2d_mode = []() {
          const char* m = getenv("LDP_DEBUG_H2D_MODE");
          return m ? atoi(m) : 1;
        }();
        if (h2d_mode == 2) {
          d_src = pin;
        } else if (h2d_mode == 1) {
          hipStream_t cs = e->h2d_stream[slot & 1];
          HIP_TRY(e, hipMemcpyAsync(e->d_stage[slot], pin, static_cast<size_t>(cnt) * pack_stride, hipMemcpyHostToDevice, cs));
          HIP_TRY(e, hipEventRecord(e->copied[slot], cs));
          HIP_TRY(e, hipStreamWaitEvent(e->stream, e->copied[slot], 0));
          d_src = e->d_stage[slot];
        } else {
          HIP_TRY(e, hipMemcpyAsync(e->d_stage[slot], pin, static_cast<size_t>(cnt) * pack_stride, hipMemcpyHostToDevice, e->stream));
          d_src = e->d_stage[slot];
        }
        d_stride = pack_stride;
      } else {
        d_src = src + static_cast<uint64_t>(g + done - first_variant) * stride_bytes;
      }
      PrepareArgs PA;
      PA.extra_het = nullptr;
      int prep_encoding = encoding;
      if (mapped) {
        // the engine's columns out of the file's rows: gather_rows_kernel, then the ordinary conversion on its output
        const hipError_t grc = launch_gather_rows(d_src, d_stride, cnt, base_encoding == LDP_GENO_BED, e->d_sample_map, e->P.founder_ct, e->d_gather, gather_stride,
                                                   e->d_extra_het, e->stream);
        if (grc != hipSuccess) {
          return hipfail(e, grc, "gather_rows_kernel launch");
        }
        d_src = e->d_gather;
        d_stride = gather_stride;
        if (base_encoding == LDP_GENO_INVERSE) {
          prep_encoding = LDP_GENO_INVERSE;  // (the caller decided the major allele and supplies maj_freq: nothing to count)
        } else {
          prep_encoding = LDP_GENO_REF;
          PA.extra_het = e->d_extra_het;
        }
      }
      PA.row_inverse = d_row_inverse ? (d_row_inverse + (g + done - first_variant)) : nullptr;
      PA.geno = d_src;
      PA.stride_bytes = d_stride;
      PA.n_variants = cnt;
      PA.founder_ct = e->P.founder_ct;
      PA.encoding = prep_encoding;
      PA.codes_out = nullptr;
      PA.code_row_bytes = e->code_row_bytes;
      if (e->codes_format) {
        PA.codes_out = e->d_codes + static_cast<uint64_t>(l0) * e->code_row_bytes;
        // Rows the caller filled in the image itself (ldp_map_rows) are counted where they are; any other pointer into the image
        // would be read while it is being written.
        const uint8_t* img_end = e->d_codes + static_cast<uint64_t>(e->local_ct) * e->code_row_bytes;
        if ((location == LDP_MEM_DEVICE) && (!mapped) && (d_src + static_cast<uint64_t>(cnt - 1) * d_stride + row_bytes > e->d_codes) && (d_src < img_end)) {
          if ((d_src != PA.codes_out) || (d_stride != e->code_row_bytes) || phased || (base_encoding == LDP_GENO_BED)) {
            return fail(e, LDP_ERR_INVALID, "rows inside the engine's image must be the mapped rows themselves (ldp_map_rows: same variants, same stride, REF or INVERSE codes)");
          }
        }
      }
      PA.planes = e->codes_format ? nullptr : (e->d_planes + static_cast<uint64_t>(l0) * e->row_dwords);
      PA.row_dwords = e->row_dwords;
      PA.chunks = e->chunks;
      PA.recs = e->d_recs + l0;
      PA.cp_stats = e->d_cp_stats + static_cast<uint64_t>(l0) * kCpStride;
      PA.cp_gen = e->d_cp_gen + static_cast<uint64_t>(l0) * kCheckpoints;
      PA.cp_tv_scale = sqrt(sqrt(e->P.prune_last_param * (1 + kSmallEpsilon)) * (1.0 - 1e-6));
      for (int k = 0; k < kCheckpoints; ++k) {
        PA.checkpoint_chunk[k] = e->checkpoint_chunk[k];
      }
      PA.n_checkpoints = e->n_checkpoints;
      PA.miss_stats = nullptr;  // (the route is taken from the records when a launch is queued: queue_route)
      PA.miss_high = static_cast<uint32_t>(std::min(2.0 * e->opt.sparse_frac * static_cast<double>(e->P.founder_ct), 4294967295.0));
      PA.fix_cp_gen = !e->mf_enabled;  // (only the popcount kernel's interval bound reads cp_gen)
      if (!e->prep_pending) {
        HIP_TRY(e, hipEventRecord(e->prep_ev0, e->stream));
        e->prep_pending = true;
      }
      hipError_t krc = e->codes_format ? launch_codes(PA, e->stream) : launch_prepare(PA, e->stream);
      if (krc != hipSuccess) {
        return hipfail(e, krc, e->codes_format ? "codes_kernel launch" : "prepare_kernel launch");
      }
      HIP_TRY(e, hipEventRecord(e->prep_ev1, e->stream));
      if (location == LDP_MEM_HOST) {
        HIP_TRY(e, hipEventRecord(e->stage_done[slot], e->stream));
        slot = (slot + 1) % kStageSlots;
      }
      std::fill(e->loaded.begin() + l0, e->loaded.begin() + l0 + cnt, static_cast<uint8_t>(1));
      std::fill(e->load_tag.begin() + l0, e->load_tag.begin() + l0 + cnt, e->load_epoch);
      if (base_encoding != LDP_GENO_INVERSE) {
        // derived from the device's allele counts at the next ldp_run()
        if (!h_row_inverse) {
          std::fill(e->mf_set.begin() + l0, e->mf_set.begin() + l0 + cnt, static_cast<uint8_t>(2));
        } else {
          for (uint32_t q = 0; q < cnt; ++q) {
            if (!h_row_inverse[g + done + q - first_variant]) {
              e->mf_set[l0 + q] = 2;
            }
          }
        }
      }
      // Pair tiles whose variants are all converted start right away when the input comes over PCIe (the GPU is
      // mostly idle then).  With device-resident input the conversion is HBM-bound and gains nothing from sharing
      // the CUs (measured), it only finishes later -- and with it the records the host replay is waiting for.
      if (eager) {
        rc = launch_ready_groups(e);
        if (rc) {
          return rc;
        }
      }
      done += cnt;
    }
    g += run;
  }
  e->recs_host_valid = false;
  e->recs_copy_queued = false;
  if (location == LDP_MEM_HOST) {
    const double ts0 = now_ms();
    // (No stream synchronisation here: the caller's rows have been copied into the engine's own pinned slots by the time the copy
    // threads return, so its buffer is free, and the slots' reuse waits on their events.  Draining the ring at the end of every call
    // cost 3.4 ms per 1 GB call of plink2-hip: profiles/r04_experiments.md.)
    if (load_timing) {
      fprintf(stderr, "[load timing] %u rows in %u slots: %.1f ms in all = %.1f waiting for a free slot + %.1f copying into pinned memory (%s) + %.1f other; before that %.1f ms "
                      "device plan / image allocation + %.1f ms staging ring\n", n,
              n_slots, now_ms() - t_call0, t_wait_slot, t_copy, (src_fd >= 0) ? "pread" : "memcpy", (now_ms() - t_call0) - t_wait_slot - t_copy, t_planned - t_entry,
              t_call0 - t_planned);
      (void)ts0;
    }
  }
  return LDP_OK;
}
}  // namespace

int ldp_load_genotypes(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* geno, uint64_t stride_bytes, int location, int encoding) {
  return load_rows_impl(e, first_variant, n, geno, stride_bytes, location, encoding, nullptr, nullptr);
}

int ldp_load_genotypes_fd(ldp_engine* e, uint32_t first_variant, uint32_t n, int fd, uint64_t file_offset, uint64_t stride_bytes, int encoding) {
  if (fd < 0) {
    return e ? fail(e, LDP_ERR_INVALID, "bad file descriptor") : LDP_ERR_INVALID;
  }
  return load_rows_impl(e, first_variant, n, nullptr, stride_bytes, LDP_MEM_HOST, encoding, nullptr, nullptr, fd, file_offset);
}

namespace {
// slot k of the decode scratch, at least `bytes` large (contents are not kept when it grows)
int dec_reserve(ldp_engine* e, int k, size_t bytes, void** out) {
  if (e->dec.cap[k] < bytes) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    (void)hipFree(e->dec.ptr[k]);
    e->dec.ptr[k] = nullptr;
    e->dec.cap[k] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(e, hipMalloc(&e->dec.ptr[k], want));
    e->dec.cap[k] = want;
  }
  *out = e->dec.ptr[k];
  return LDP_OK;
}
}  // namespace

// Variant records of a variable-width .pgen, decoded on the device (ldp_pgen_decode.hip) into rows of the FILE's samples and
// loaded from there like any device-resident rows.  What the reference does per variant on its one reader thread
// (PgrGetInv1 -> ReadGenovecSubsetUnsafe, plink2_ld.cc:1345-1390 / pgenlib_read.cc:2849-2912, 5417-5563).
namespace {
int load_pgen_records_impl(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                           const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* major_allele_out, bool phased, uint32_t* unphased_variant);
}  // namespace

int ldp_load_pgen_records(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                          const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* major_allele_out) {
  return load_pgen_records_impl(e, first_variant, n, bytes, n_bytes, location, recs, ld_base, raw_sample_ct, major_allele_out, false, nullptr);
}

int ldp_load_pgen_records_phased(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                                 const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* unphased_variant) {
  if (unphased_variant) {
    *unphased_variant = UINT32_MAX;
  }
  return load_pgen_records_impl(e, first_variant, n, bytes, n_bytes, location, recs, ld_base, raw_sample_ct, nullptr, true, unphased_variant);
}

namespace {
int load_pgen_records_impl(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                           const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* major_allele_out, bool phased, uint32_t* unphased_variant) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants() first");
  }
  if ((static_cast<uint64_t>(first_variant) + n > e->variant_ct) || (n && (!bytes || !recs)) || ((location != LDP_MEM_HOST) && (location != LDP_MEM_DEVICE))) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds / null input / bad location");
  }
  const bool mapped = !e->sample_map.empty();
  if (phased) {
    // --indep-pairphase: the engine's founder_ct is the haplotype count, two per sample of the file (LDP_GENO_PHASED)
    if (mapped || (static_cast<uint64_t>(raw_sample_ct) * 2 != e->P.founder_ct)) {
      return fail(e, LDP_ERR_INVALID, "phased records: the engine's founder_ct must be twice the file's sample count, without a sample map");
    }
  } else if (mapped ? (raw_sample_ct != e->map_raw_sample_ct) : (raw_sample_ct != e->P.founder_ct)) {
    return fail(e, LDP_ERR_INVALID, "the records' sample count is neither the engine's founder count nor the sample map's raw count");
  }
  if (!n) {
    return LDP_OK;
  }
  // every record inside the buffer
  auto cover = [&](const ldp_pgen_rec& r) { return (r.offset <= n_bytes) && (r.length <= n_bytes - r.offset); };
  bool any_multi = false, any_ld = false;
  for (uint32_t q = 0; q < n; ++q) {
    if (!cover(recs[q])) {
      return fail(e, LDP_ERR_INVALID, "a record lies outside the byte buffer");
    }
    if ((recs[q].allele_ct < 2) || (recs[q].allele_ct > 255)) {
      return fail(e, LDP_ERR_INVALID, "allele_ct must lie in [2, 255]");
    }
    any_multi = any_multi || (recs[q].allele_ct > 2);
    const uint32_t type = recs[q].vrtype & 7u;
    any_ld = any_ld || (type == 2) || (type == 3);
  }
  if (any_multi && phased) {
    return fail(e, LDP_ERR_UNSUPPORTED, "phased records with more than one ALT allele: their phase refers to allele pairs (Get1MP, pgenlib_read.cc:6962): build those rows on the host");
  }
  if (any_multi && mapped) {
    return fail(e, LDP_ERR_UNSUPPORTED, "variants with more than one ALT allele are collapsed over the file's samples: not with a sample map (collapse them on the host, LDP_GENO_INVERSE)");
  }
  if (ld_base) {
    const uint32_t type = ld_base->vrtype & 7u;
    if ((type == 2) || (type == 3) || !cover(*ld_base)) {
      return fail(e, LDP_ERR_INVALID, "ld_base must be a record that stands alone, inside the byte buffer");
    }
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  // (phased: a row is codes, padding to a dword, then 16 phase bits per code dword -- the LDP_GENO_PHASED layout the count pass splits)
  const uint64_t phase_off = phased ? ldp_phased_phase_offset(2 * raw_sample_ct) : 0;
  const uint64_t stride = phased ? ((phase_off + 2ull * ((static_cast<uint64_t>(raw_sample_ct) + 15) / 16) + 15) & ~static_cast<uint64_t>(15))
                                 : (((static_cast<uint64_t>(raw_sample_ct) + 3) / 4 + 15) & ~static_cast<uint64_t>(15));
  // ---- the bytes
  // (host bytes go to the device launch by launch, below: only the span of that launch's records, so that the first decode does not wait
  // for the whole call's upload and a call over a file with long dosage tracks does not need one allocation for all of them)
  const uint8_t* d_bytes = (location == LDP_MEM_HOST) ? nullptr : static_cast<const uint8_t*>(bytes);
  if (e->ld_base_cap < stride) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    (void)hipFree(e->d_ld_base);
    e->d_ld_base = nullptr;
    e->ld_base_cap = 0;
    e->ld_base_valid = false;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_ld_base), stride));
    e->ld_base_cap = stride;
  }
  // (the carried base is the record that PRECEDES this call's first one in the file: same engine position AND same file position --
  // a caller that loads non-adjacent file ranges into adjacent engine indices gets LDP_ERR_INVALID below instead of a wrong row)
  bool have_carried = e->ld_base_valid && (e->dec_next_variant == first_variant) && (e->dec_next_offset == recs[0].offset);
  // ---- in launches of at most ~256 MiB of rows
  uint32_t rows_per_launch = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / stride)));
  if (const char* dbg = getenv("LDP_DEBUG_DECODE_ROWS")) {  // (test hook: many small launches, LD chains cut everywhere)
    rows_per_launch = static_cast<uint32_t>(std::max(1, atoi(dbg)));
  }
  std::vector<uint32_t> multi;
  std::vector<uint8_t> h_inverse;
  int status = LDP_OK;
  // pinned staging for one launch: descriptors | multiallelic record indices | major-allele frequencies | major alleles | error word
  {
    const size_t rows_max = static_cast<size_t>(rows_per_launch) + 1;
    const size_t want = rows_max * (sizeof(ldp::PgenRecDesc) + sizeof(uint32_t) + sizeof(double) + sizeof(uint32_t)) + 64;  // (+ the error word and the unphased record behind them)
    if (e->dec_pin_cap < want) {
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      if (e->h_dec_pin) {
        (void)hipHostFree(e->h_dec_pin);
        e->h_dec_pin = nullptr;
        e->dec_pin_cap = 0;
      }
      HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_dec_pin), want, hipHostMallocDefault));
      e->dec_pin_cap = want;
    }
  }
  const size_t rows_cap = static_cast<size_t>(rows_per_launch) + 1;
  ldp::PgenRecDesc* descs = reinterpret_cast<ldp::PgenRecDesc*>(e->h_dec_pin);
  double* h_maj_freq = reinterpret_cast<double*>(e->h_dec_pin + rows_cap * sizeof(ldp::PgenRecDesc));
  uint32_t* h_multi = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(h_maj_freq) + rows_cap * sizeof(double));
  uint32_t* h_maj_idx = h_multi + rows_cap;
  int* h_err_pin = reinterpret_cast<int*>(h_maj_idx + rows_cap);
  for (uint32_t q0 = 0; (q0 < n) && (status == LDP_OK); q0 += rows_per_launch) {
    const double t_call = now_ms();
    const uint32_t cnt = std::min(rows_per_launch, n - q0);
    const bool with_base_rec = (q0 == 0) && (ld_base != nullptr);
    const uint32_t rows = cnt + (with_base_rec ? 1u : 0u);  // (the caller's ld_base record is decoded as an extra row behind the others)
    for (uint32_t q = 0; q < rows; ++q) {
      descs[q] = ldp::PgenRecDesc();
    }
    multi.clear();
    uint32_t last_alone = with_base_rec ? cnt : (have_carried ? kPgenBaseCarried : kPgenNoBase);
    int64_t last_alone_row = -1;
    for (uint32_t q = 0; q < cnt; ++q) {
      const ldp_pgen_rec& r = recs[q0 + q];
      ldp::PgenRecDesc& d = descs[q];
      d.off = r.offset;
      d.len = r.length;
      d.allele_ct = r.allele_ct;
      d.vrtype = r.vrtype;
      d.base = kPgenNoBase;
      const uint32_t type = r.vrtype & 7u;
      if ((type == 2) || (type == 3)) {
        if (last_alone == kPgenNoBase) {
          return fail(e, LDP_ERR_INVALID, "an LD-compressed record whose base is neither in this call, nor ld_base, nor the last record of the previous call");
        }
        d.base = last_alone;
      } else {
        last_alone = q;
        last_alone_row = q;
      }
      if (r.allele_ct > 2) {
        multi.push_back(q);
      }
    }
    if (with_base_rec) {
      ldp::PgenRecDesc& d = descs[cnt];
      d.off = ld_base->offset;
      d.len = ld_base->length;
      d.allele_ct = 2;
      d.vrtype = ld_base->vrtype;
      d.base = kPgenNoBase;
    }
    if (location == LDP_MEM_HOST) {
      uint64_t lo = UINT64_MAX, hi = 0;
      for (uint32_t q = 0; q < rows; ++q) {
        lo = std::min<uint64_t>(lo, descs[q].off);
        hi = std::max<uint64_t>(hi, descs[q].off + descs[q].len);
      }
      void* p = nullptr;
      if ((rc = dec_reserve(e, 0, hi - lo + 16, &p))) {
        return rc;
      }
      HIP_TRY(e, hipMemcpyAsync(p, static_cast<const uint8_t*>(bytes) + lo, hi - lo, hipMemcpyHostToDevice, e->stream));
      d_bytes = static_cast<const uint8_t*>(p) - lo;  // (record offsets stay as the caller gave them)
    }
    void *p_recs = nullptr, *p_rows = nullptr, *p_end = nullptr, *p_multi = nullptr, *p_mf = nullptr, *p_mi = nullptr, *p_inv = nullptr;
    if ((rc = dec_reserve(e, 1, rows * sizeof(ldp::PgenRecDesc), &p_recs)) || (rc = dec_reserve(e, 2, static_cast<size_t>(rows) * stride, &p_rows)) ||
        (rc = dec_reserve(e, 3, rows * sizeof(uint64_t), &p_end)) || (rc = dec_reserve(e, 4, (multi.size() + 1) * sizeof(uint32_t), &p_multi)) ||
        (rc = dec_reserve(e, 5, (multi.size() + 1) * sizeof(double), &p_mf)) || (rc = dec_reserve(e, 6, (multi.size() + 1) * sizeof(uint32_t) + sizeof(int), &p_mi)) ||
        (rc = dec_reserve(e, 7, rows + 8, &p_inv))) {
      return rc;
    }
    int* d_err = reinterpret_cast<int*>(static_cast<uint32_t*>(p_mi) + multi.size() + 1);
    uint32_t* d_unphased = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(p_inv) + ((static_cast<size_t>(rows) + 3) & ~static_cast<size_t>(3)));  // (behind the row flags)
    HIP_TRY(e, hipMemcpyAsync(p_recs, descs, rows * sizeof(ldp::PgenRecDesc), hipMemcpyHostToDevice, e->stream));
    if (!multi.empty()) {
      memcpy(h_multi, multi.data(), multi.size() * sizeof(uint32_t));
      HIP_TRY(e, hipMemcpyAsync(p_multi, h_multi, multi.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(e, hipMemsetAsync(d_err, 0, sizeof(int), e->stream));
    HIP_TRY(e, hipMemsetAsync(p_inv, 0, rows, e->stream));
    HIP_TRY(e, hipMemsetAsync(d_unphased, 0xff, sizeof(uint32_t), e->stream));
    ldp::PgenDecodeArgs DA;
    DA.bytes = d_bytes;
    DA.recs = static_cast<const ldp::PgenRecDesc*>(p_recs);
    DA.n = rows;
    DA.sample_ct = raw_sample_ct;
    DA.rows = static_cast<uint8_t*>(p_rows);
    DA.stride = stride;
    DA.carried_base = have_carried ? e->d_ld_base : nullptr;
    DA.main_end = static_cast<uint64_t*>(p_end);
    DA.error = d_err;
    DA.pass = 0;
    DA.any_ld = any_ld ? 1 : 0;
    DA.multi_rec = static_cast<const uint32_t*>(p_multi);
    DA.n_multi = static_cast<uint32_t>(multi.size());
    DA.maj_freq = static_cast<double*>(p_mf);
    DA.maj_idx = static_cast<uint32_t*>(p_mi);
    DA.row_inverse = static_cast<uint8_t*>(p_inv);
    DA.phase_off = phase_off;
    DA.unphased = d_unphased;
    hipError_t krc = launch_pgen_main(DA, e->stream);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pgen_main_kernel launch");
    }
    if (phased) {
      krc = launch_pgen_phase(DA, cnt, e->stream);  // (not the caller's ld_base row behind them: only its codes are a base)
      if (krc != hipSuccess) {
        return hipfail(e, krc, "pgen_phase_kernel launch");
      }
    }
    // the row the next launch's LD-compressed records may build on (taken BEFORE the multiallelic collapse rewrites rows:
    // an LD base is the main track as stored)
    if (last_alone_row >= 0) {
      HIP_TRY(e, hipMemcpyAsync(e->d_ld_base, DA.rows + static_cast<uint64_t>(last_alone_row) * stride, stride, hipMemcpyDeviceToDevice, e->stream));
      have_carried = true;
    } else if (with_base_rec) {
      HIP_TRY(e, hipMemcpyAsync(e->d_ld_base, DA.rows + static_cast<uint64_t>(cnt) * stride, stride, hipMemcpyDeviceToDevice, e->stream));
      have_carried = true;
    }
    krc = launch_pgen_aux1(DA, e->stream);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pgen_aux1_kernel launch");
    }
    h_inverse.assign(cnt, 0);
    *h_err_pin = 0;
    HIP_TRY(e, hipMemcpyAsync(h_err_pin, d_err, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    h_err_pin[1] = -1;
    if (phased) {
      HIP_TRY(e, hipMemcpyAsync(h_err_pin + 1, d_unphased, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    if (!multi.empty()) {
      HIP_TRY(e, hipMemcpyAsync(h_maj_freq, p_mf, multi.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(e, hipMemcpyAsync(h_maj_idx, p_mi, multi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    }
    const double t_q = now_ms();
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    const double t_s = now_ms();
    const int h_err = *h_err_pin;
    if (h_err) {
      e->ld_base_valid = false;
      const uint32_t bad = static_cast<uint32_t>(h_err - 1);
      return fail(e, LDP_ERR_INVALID, "malformed variant record in .pgen data (variant " + std::to_string((bad < cnt) ? (first_variant + q0 + bad) : first_variant) + ((bad < cnt) ? ")" : ": its LD base)"));
    }
    if (phased && (static_cast<uint32_t>(h_err_pin[1]) != UINT32_MAX)) {
      // a het call without phase: the reference's "variant #k is not fully phased" (plink2_ld.cc:2045-2049); nothing of this launch is loaded
      e->ld_base_valid = false;
      if (unphased_variant) {
        *unphased_variant = first_variant + q0 + static_cast<uint32_t>(h_err_pin[1]);
      }
      return fail(e, LDP_ERR_UNPHASED, "a heterozygous call has no phase (variant " + std::to_string(first_variant + q0 + static_cast<uint32_t>(h_err_pin[1])) + ")");
    }
    for (size_t k = 0; k < multi.size(); ++k) {
      h_inverse[multi[k]] = 1;
      if (major_allele_out) {
        major_allele_out[q0 + multi[k]] = h_maj_idx[k];
      }
    }
    if (major_allele_out) {
      for (uint32_t q = 0; q < cnt; ++q) {
        if (!h_inverse[q]) {
          major_allele_out[q0 + q] = UINT32_MAX;  // one ALT allele: the count pass decides (ldp_get_variant_recs: flags bit 0)
        }
      }
    }
    status = load_rows_impl(e, first_variant + q0, cnt, DA.rows, stride, LDP_MEM_DEVICE, LDP_GENO_REF | (mapped ? LDP_GENO_MAPPED : 0) | (phased ? LDP_GENO_PHASED : 0),
                            multi.empty() ? nullptr : DA.row_inverse, multi.empty() ? nullptr : h_inverse.data());
    if (getenv("LDP_DEBUG_TIMELINE")) {
      fprintf(stderr, "decode launch of %u rows: queued in %.3f ms, device done %.3f ms later, rows loaded %.3f ms after that\n", rows, t_q - t_call, t_s - t_q, now_ms() - t_s);
    }
    if (status == LDP_OK) {
      for (size_t k = 0; k < multi.size(); ++k) {
        const int64_t l = e->global_to_local[first_variant + q0 + multi[k]];
        if (l >= 0) {
          e->maj_freq[l] = h_maj_freq[k];
          e->mf_set[l] = 1;
        }
      }
    }
  }
  e->ld_base_valid = have_carried && (status == LDP_OK);
  e->dec_next_variant = first_variant + n;
  e->dec_next_offset = recs[n - 1].offset + recs[n - 1].length;
  return status;
}
}  // namespace

int ldp_map_rows(ldp_engine* e, uint32_t first_variant, uint32_t n, void** device_rows, uint64_t* stride_bytes) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!device_rows || !stride_bytes) {
    return fail(e, LDP_ERR_INVALID, "null output pointer");
  }
  *device_rows = nullptr;
  *stride_bytes = 0;
  if (!e->planned) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants() first");
  }
  if ((!n) || (static_cast<uint64_t>(first_variant) + n > e->variant_ct)) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds");
  }
  const int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  if (!e->codes_format) {
    return fail(e, LDP_ERR_UNSUPPORTED, "this engine keeps bit-planes (more founders than the matrix pipe takes, or pair_mfma off): load from your own buffer");
  }
  const int64_t l0 = e->global_to_local[first_variant];
  if (l0 < 0) {
    return fail(e, LDP_ERR_INVALID, "the first variant is not owned by this engine (ldp_set_shard / a subcontig of length one)");
  }
  for (uint32_t q = 1; q < n; ++q) {
    if (e->global_to_local[first_variant + q] != l0 + q) {
      return fail(e, LDP_ERR_INVALID, "the variants are not consecutive rows of this engine (map one owned run at a time: ldp_get_subcontigs)");
    }
  }
  *device_rows = e->d_codes + static_cast<uint64_t>(l0) * e->code_row_bytes;
  *stride_bytes = e->code_row_bytes;
  return LDP_OK;
}

int ldp_release_device(ldp_engine* e) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->gpu_ok || !e->plan_uploaded) {
    return LDP_OK;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  for (int k = 0; k < kPairStreams; ++k) {
    HIP_TRY(e, hipStreamSynchronize(e->pair_stream[k]));
  }
  HIP_TRY(e, hipStreamSynchronize(e->copy_stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  free_device(e);
  // the rows are gone with the image: everything has to be loaded again before the next run
  std::fill(e->loaded.begin(), e->loaded.end(), 0);
  std::fill(e->load_tag.begin(), e->load_tag.end(), 0);
  for (uint8_t& m : e->mf_set) {
    m = (m == 1) ? 1 : 0;  // (caller-supplied frequencies stay)
  }
  e->load_epoch = 1;
  e->recs_host_valid = false;
  e->recs_copy_queued = false;
  return LDP_OK;
}

int ldp_set_sample_map(ldp_engine* e, uint32_t raw_sample_ct, const uint32_t* src_sample, const uint8_t* het_to_missing) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!raw_sample_ct || !src_sample || (raw_sample_ct > 0x7fffffffu)) {
    return fail(e, LDP_ERR_INVALID, "sample map: raw_sample_ct / src_sample");
  }
  bind_gpu(e);
  if (!e->gpu_ok) {
    return fail(e, LDP_ERR_GPU, "no usable HIP device");
  }
  HIP_TRY(e, hipSetDevice(e->device));
  std::vector<uint32_t> m(e->P.founder_ct);
  for (uint32_t f = 0; f < e->P.founder_ct; ++f) {
    if (src_sample[f] >= raw_sample_ct) {
      return fail(e, LDP_ERR_INVALID, "sample map: source sample out of range");
    }
    m[f] = src_sample[f] | ((het_to_missing && het_to_missing[f]) ? 0x80000000u : 0u);
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (!e->d_sample_map) {
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_sample_map), m.size() * sizeof(uint32_t)));
  }
  HIP_TRY(e, hipMemcpy(e->d_sample_map, m.data(), m.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  e->sample_map.swap(m);
  e->map_raw_sample_ct = raw_sample_ct;
  return LDP_OK;
}

int ldp_set_maj_freqs(ldp_engine* e, uint32_t first_variant, uint32_t n, const double* maj_freqs) {
  if (!e || !e->planned) {
    return e ? fail(e, LDP_ERR_STATE, "ldp_set_variants() first") : LDP_ERR_INVALID;
  }
  if ((static_cast<uint64_t>(first_variant) + n > e->variant_ct) || (n && !maj_freqs)) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds");
  }
  for (uint32_t q = 0; q < n; ++q) {
    const int64_t l = e->global_to_local[first_variant + q];
    if (l >= 0) {
      e->maj_freq[l] = maj_freqs[q];
      e->mf_set[l] = 1;
    }
  }
  return LDP_OK;
}

int ldp_set_preferred(ldp_engine* e, const uint64_t* preferred_bitmap) {
  if (!e || !e->planned) {
    return e ? fail(e, LDP_ERR_STATE, "ldp_set_variants() first") : LDP_ERR_INVALID;
  }
  if (!preferred_bitmap) {
    e->preferred.clear();
    return LDP_OK;
  }
  e->preferred.assign(preferred_bitmap, preferred_bitmap + (static_cast<size_t>(e->variant_ct) + 63) / 64);
  return LDP_OK;
}

int ldp_run(ldp_engine* e, uint64_t* removed) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  return run_impl(e, removed, nullptr, 0);
}

int ldp_run_with_stats(ldp_engine* e, uint64_t* removed, ldp_pair_stats_t* stats, uint64_t stats_capacity) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!stats) {
    return fail(e, LDP_ERR_INVALID, "stats is NULL");
  }
  return run_impl(e, removed, stats, stats_capacity);
}

int ldp_pair_stats(ldp_engine* e, uint32_t n_pairs, const uint32_t* first, const uint32_t* second, ldp_pair_stats_t* out) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->planned) {
    return fail(e, LDP_ERR_STATE, "ldp_set_variants() first");
  }
  if (n_pairs && (!first || !second || !out)) {
    return fail(e, LDP_ERR_INVALID, "NULL argument");
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  std::vector<uint32_t> lf(n_pairs), ls(n_pairs);
  for (uint32_t k = 0; k < n_pairs; ++k) {
    if ((first[k] >= e->variant_ct) || (second[k] >= e->variant_ct)) {
      return fail(e, LDP_ERR_INVALID, "variant index out of range");
    }
    const int64_t a = e->global_to_local[first[k]];
    const int64_t b = e->global_to_local[second[k]];
    if ((a < 0) || (b < 0) || !e->loaded[a] || !e->loaded[b]) {
      return fail(e, LDP_ERR_STATE, "pair refers to a variant that is not owned/loaded");
    }
    lf[k] = static_cast<uint32_t>(a);
    ls[k] = static_cast<uint32_t>(b);
  }
  if (!n_pairs) {
    return LDP_OK;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  DevBuf idx_buf, out_buf;
  HIP_TRY(e, hipMalloc(&idx_buf.p, 2ull * n_pairs * sizeof(uint32_t)));
  HIP_TRY(e, hipMalloc(&out_buf.p, static_cast<size_t>(n_pairs) * sizeof(ldp_pair_stats_t)));
  uint32_t* d_idx = idx_buf.as<uint32_t>();
  ldp_pair_stats_t* d_out = out_buf.as<ldp_pair_stats_t>();
  HIP_TRY(e, hipMemcpyAsync(d_idx, lf.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(d_idx + n_pairs, ls.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  hipError_t krc = e->codes_format ? launch_pair_stats_ref_codes(e->d_codes, e->code_row_bytes, e->d_recs, d_idx, d_idx + n_pairs, n_pairs, d_out, e->stream)
                                   : launch_pair_stats_ref(e->d_planes, e->row_dwords, e->chunks, 0, d_idx, d_idx + n_pairs, n_pairs, d_out, e->stream);
  if (krc != hipSuccess) {
    return hipfail(e, krc, "pair_stats_ref launch");
  }
  HIP_TRY(e, hipMemcpyAsync(out, d_out, static_cast<size_t>(n_pairs) * sizeof(ldp_pair_stats_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return LDP_OK;
}

int ldp_debug_set_option(ldp_engine* e, const char* name, double value) {
  if (!e || !name) {
    return LDP_ERR_INVALID;
  }
  const std::string n(name);
  if (n == "early_exit") {
    e->opt.early_exit = (value != 0.0);
  } else if (n == "pair_mfma") {
    if (e->planned) {
      return fail(e, LDP_ERR_STATE, "pair_mfma must be set before ldp_set_variants()");
    }
    e->opt.pair_mfma = (value != 0.0);
  } else if (n == "wide_min_reach") {
    if (e->planned) {
      return fail(e, LDP_ERR_STATE, "wide_min_reach must be set before ldp_set_variants()");
    }
    e->opt.wide_min_reach = (value >= 4294967295.0) ? 0xffffffffu : static_cast<uint32_t>(std::max(0.0, value));
  } else if (n == "pair_four") {
    e->opt.pair_four = (value != 0.0);
  } else if (n == "pair_gu") {
    e->opt.pair_gu = (value != 0.0);
  } else if (n == "wide_diag_last") {
    if (e->planned) {
      return fail(e, LDP_ERR_STATE, "wide_diag_last must be set before ldp_set_variants()");
    }
    e->opt.wide_diag_last = static_cast<uint32_t>(std::max(0.0, value));
  } else if (n == "pair_four_tiles") {
    e->opt.four_tiles = (value != 0.0);
  } else if (n == "pair_sparse") {
    if (value == 0.0) {
      e->opt.sparse_frac = 0.0;
    } else if (e->opt.sparse_frac == 0.0) {
      e->opt.sparse_frac = 0.005;
    }
  } else if (n == "sparse_frac") {
    if (!(value >= 0.0) || !(value <= 1.0)) {
      return fail(e, LDP_ERR_INVALID, "sparse_frac must lie in [0, 1]");
    }
    e->opt.sparse_frac = value;
  } else {
    return fail(e, LDP_ERR_INVALID, "unknown option: " + n);
  }
  return LDP_OK;
}

int ldp_debug_set_variant_recs(ldp_engine* e, const ldp_variant_rec* recs) {
  if (!e || !e->planned || !recs) {
    return LDP_ERR_STATE;
  }
  for (uint32_t l = 0; l < e->local_ct; ++l) {
    e->recs[l] = recs[e->local_to_global[l]];
  }
  e->recs_host_valid = true;
  return LDP_OK;
}

int ldp_debug_replay_pairs(ldp_engine* e, uint64_t n_true, const uint32_t* first, const uint32_t* second, uint64_t* removed) {
  if (!e || !e->planned || !removed || (n_true && (!first || !second))) {
    return LDP_ERR_STATE;
  }
  if (!e->recs_host_valid) {
    return fail(e, LDP_ERR_STATE, "variant records not set");
  }
  std::vector<double> mf_scratch;
  const double* mf = nullptr;
  int rc = prepare_mf(e, &mf_scratch, &mf);
  if (rc) {
    return rc;
  }
  std::vector<uint32_t> pred(std::max<size_t>(e->pred_words, 1), 0);
  for (uint64_t k = 0; k < n_true; ++k) {
    if ((first[k] >= e->variant_ct) || (second[k] >= e->variant_ct)) {
      return fail(e, LDP_ERR_INVALID, "variant index out of range");
    }
    const int64_t i = e->global_to_local[first[k]];
    const int64_t j = e->global_to_local[second[k]];
    if ((i < 0) || (j < 0)) {
      continue;  // not this shard's pair
    }
    const uint32_t lo = e->lo_local[j];
    if ((i >= j) || (i < lo)) {
      return fail(e, LDP_ERR_INVALID, "pair outside the candidate band");
    }
    pred[e->row_off[j] + ((static_cast<uint32_t>(i) >> 5) - (lo >> 5))] |= 1u << (i & 31);
  }
  std::vector<uint32_t> R((static_cast<size_t>(e->local_ct) + 31) / 32 + 1, 0);
  uint64_t replay_pairs = 0;
  replay(e, pred.data(), mf, R, &replay_pairs);
  e->ctr.replay_pairs = replay_pairs;
  return finish_removed(e, R, removed);
}

int ldp_debug_mfma_plan(const ldp_engine* e, uint32_t* wg_count, uint32_t* words, uint64_t capacity_words, uint32_t* lo_local, uint32_t* local_ct) {
  if (!e || !e->planned || !wg_count) {
    return LDP_ERR_INVALID;
  }
  *wg_count = static_cast<uint32_t>(e->mf_wgs.size());
  if (local_ct) {
    *local_ct = e->local_ct;
  }
  if (lo_local) {
    std::copy(e->lo_local.begin(), e->lo_local.end(), lo_local);
  }
  if (!words) {
    return LDP_OK;
  }
  constexpr uint32_t kWords = 3 + kMfMaxRowBlocks + kMfWaves * 11;
  if (capacity_words < static_cast<uint64_t>(kWords) * e->mf_wgs.size()) {
    return LDP_ERR_INVALID;
  }
  for (const MfmaWG& wg : e->mf_wgs) {
    *words++ = wg.n_rb | ((wg.pad & 1u) ? 0x80000000u : 0u) | ((wg.pad & 2u) ? 0x40000000u : 0u);  // (bit 31: the subcontig also has the wide plan; bit 30: all-diagonal)
    *words++ = wg.j_lo;
    *words++ = wg.j_hi;
    for (uint32_t k = 0; k < kMfMaxRowBlocks; ++k) {
      *words++ = wg.rb[k];
    }
    for (uint32_t w = 0; w < kMfWaves; ++w) {
      *words++ = static_cast<uint32_t>(wg.w[w].jv);
      *words++ = static_cast<uint32_t>(wg.w[w].vv);
      *words++ = wg.w[w].jend;
      *words++ = wg.w[w].prod_mask;
      for (int u = 0; u < 7; ++u) {
        *words++ = wg.w[w].slot[u];
      }
    }
  }
  return LDP_OK;
}

int ldp_debug_wide_plan(const ldp_engine* e, uint32_t* tile_count, uint32_t* words, uint64_t capacity_words) {
  if (!e || !e->planned || !tile_count) {
    return LDP_ERR_INVALID;
  }
  *tile_count = static_cast<uint32_t>(e->wd_tiles.size());
  if (!words) {
    return LDP_OK;
  }
  if (capacity_words < 5ull * e->wd_tiles.size()) {
    return LDP_ERR_INVALID;
  }
  for (const MfmaTile& t : e->wd_tiles) {
    *words++ = static_cast<uint32_t>(t.jv);
    *words++ = static_cast<uint32_t>(t.vv);
    *words++ = t.jend;
    *words++ = static_cast<uint32_t>(t.mask);
    *words++ = static_cast<uint32_t>(t.mask >> 32);
  }
  return LDP_OK;
}

int ldp_get_variant_recs(ldp_engine* e, uint32_t first_variant, uint32_t n, ldp_variant_rec* out) {
  if (!e || !e->planned || !out) {
    return LDP_ERR_STATE;
  }
  if (static_cast<uint64_t>(first_variant) + n > e->variant_ct) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds");
  }
  if (!e->recs_host_valid) {
    int rc = ensure_device_plan(e);
    if (rc) {
      return rc;
    }
    rc = fetch_recs(e);
    if (rc) {
      return rc;
    }
  }
  for (uint32_t q = 0; q < n; ++q) {
    const int64_t l = e->global_to_local[first_variant + q];
    if (l >= 0) {
      out[q] = e->recs[l];
    } else {
      memset(&out[q], 0, sizeof(ldp_variant_rec));
    }
  }
  return LDP_OK;
}

int ldp_get_maj_freqs(ldp_engine* e, uint32_t first_variant, uint32_t n, double* out) {
  if (!e || !e->planned || !out) {
    return LDP_ERR_STATE;
  }
  if (static_cast<uint64_t>(first_variant) + n > e->variant_ct) {
    return fail(e, LDP_ERR_INVALID, "variant range out of bounds");
  }
  if (!e->recs_host_valid && e->plan_uploaded) {
    const int rc = fetch_recs(e);  // derives the frequencies still pending from the device's allele counts
    if (rc) {
      return rc;
    }
  }
  for (uint32_t q = 0; q < n; ++q) {
    const int64_t l = e->global_to_local[first_variant + q];
    out[q] = ((l >= 0) && e->mf_set[l]) ? e->maj_freq[l] : 0.0;
  }
  return LDP_OK;
}

int ldp_get_planes(ldp_engine* e, uint32_t variant, uint32_t* hom, uint32_t* ref2het) {
  if (!e || !e->planned || !hom || !ref2het) {
    return LDP_ERR_STATE;
  }
  if (variant >= e->variant_ct) {
    return fail(e, LDP_ERR_INVALID, "variant out of range");
  }
  const int64_t l = e->global_to_local[variant];
  if ((l < 0) || !e->loaded[l]) {
    return fail(e, LDP_ERR_STATE, "variant not owned/loaded");
  }
  int rc = ensure_device_plan(e);
  if (rc) {
    return rc;
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const uint32_t plane_dwords = (e->P.founder_ct + 31) / 32;
  if (e->codes_format) {
    // the planes SplitHomRef2het (pgenlib_misc.cc:1797) would make of the major-allele-oriented row, from the image's codes:
    // hom = !b0, ref2het = !b1, and ref2het ^= hom when the row is ALT-major (the 0 <-> 2 inversion the image does not carry)
    std::vector<uint32_t> codes(e->code_row_bytes / 4);
    ldp_variant_rec rec;
    HIP_TRY(e, hipMemcpy(codes.data(), e->d_codes + static_cast<uint64_t>(l) * e->code_row_bytes, e->code_row_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(&rec, e->d_recs + l, sizeof(rec), hipMemcpyDeviceToHost));
    for (uint32_t p = 0; p < plane_dwords; ++p) {
      uint32_t h = 0, r = 0;
      for (uint32_t half = 0; half < 2; ++half) {
        const uint32_t w = codes[2 * p + half];
        for (uint32_t t = 0; t < 16; ++t) {
          const uint32_t c = (w >> (2 * t)) & 3u;
          h |= ((c & 1u) ^ 1u) << (16 * half + t);
          r |= ((c >> 1) ^ 1u) << (16 * half + t);
        }
      }
      hom[p] = h;
      ref2het[p] = (rec.flags & 1u) ? (r ^ h) : r;
    }
    return LDP_OK;
  }
  std::vector<uint32_t> row(e->row_dwords);
  HIP_TRY(e, hipMemcpy(row.data(), e->d_planes + static_cast<uint64_t>(l) * e->row_dwords, e->row_dwords * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t p = 0; p < plane_dwords; ++p) {
    const uint32_t off = (p / kChunkDwords) * kRowChunkDwords + (p % kChunkDwords);
    hom[p] = row[off];
    ref2het[p] = row[off + kChunkDwords];
  }
  return LDP_OK;
}

int ldp_get_counters(const ldp_engine* e, ldp_counters* out) {
  if (!e || !out) {
    return LDP_ERR_INVALID;
  }
  *out = e->ctr;
  return LDP_OK;
}

}  // extern "C"
