// ldp_engine_internal.h -- what the translation units of the host runtime share (ldp_engine*.cpp): the engine object, its options,
// small RAII helpers and the functions one unit calls in another.  Not part of the boundary (include/ldprune_hip.h).
#ifndef LDP_ENGINE_INTERNAL_H
#define LDP_ENGINE_INTERNAL_H
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

// (the units' shared helpers stay out of the library's dynamic symbol table: the boundary is the C functions of ldprune_hip.h)
#define LDP_HIDDEN __attribute__((visibility("hidden")))

namespace ldph LDP_HIDDEN {


struct Subcontig {
  uint32_t len;
  uint32_t first;        // global variant index
  uint32_t owner;        // rank
  uint32_t local_first;  // valid when owned
};

inline double now_ms() {
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
// An engine whose host feeds several engines at once, each from a thread of its own next to its device (plink2-hip --gpus N: the
// reference's main thread fills every worker's slot of a batch and the workers run together, plink2_ld.cc:1292-1417), gets a pool of
// ITS OWN through ldp_use_private_copy_threads(): its workers are created by -- and inherit the CPU affinity of -- the calling thread.
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool* pool = new CopyPool();  // (never destroyed: its threads may outlive main()'s statics)
    return *pool;
  }
  CopyPool() { start(); }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      ++epoch_;
      active_ = static_cast<uint32_t>(workers_.size());
    }
    cv_.notify_all();
    for (std::thread& t : workers_) {
      t.join();
    }
  }
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;
  template <class F>
  void run(uint32_t n, uint32_t max_threads, F fn) {
    // (a forked child inherits the object but none of its threads: it works on its own)
    if (n <= 1 || workers_.empty() || (getpid() != pid_)) {
      for (uint32_t t = 0; t < n; ++t) {
        fn(t);
      }
      return;
    }
    // one user at a time; an engine that finds the pool busy (several engines loading from their own host threads: plink2-hip --gpus N)
    // does not queue up behind the others' copies but spawns threads for this batch, as every call did before the pool existed
    std::unique_lock<std::mutex> user(user_mu_, std::try_to_lock);
    if (!user.owns_lock()) {
      parallel_for(n, max_threads, fn);
      return;
    }
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
  void start() {
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
            if (stop_) {
              return;
            }
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
    }
  }
  std::vector<std::thread> workers_;
  bool stop_ = false;
  const pid_t pid_ = getpid();
  std::mutex mu_, user_mu_;
  std::condition_variable cv_, cv_done_;
  std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0, active_ = 0, done_ = 0;
  uint64_t epoch_ = 0;
  std::atomic<uint32_t> next_{0};
};

}  // namespace ldph
using namespace ldph;

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
  uint64_t csr_capacity = 0;   // test hook "csr_capacity" k: the CSR buffer holds k entries (0: a quarter of the predicate words), to force the dense fallback
  bool orient_rows = true;     // option "orient_rows" 0: ALT-major rows stay in the image as the input had them (rounds 2-5); default: the count pass stores them
                               // inverted, so that every row of the image is major-allele-oriented (ldp_device.h)
  bool pred_csr = true;        // option "pred_csr" 0: prune runs copy their dense predicate rows back (rounds 1-5) instead of the non-zero words (ldp_pred_csr.hip)
  bool wide_diag_kernel = true;  // option "wide_diag_kernel" 0: diagonal tiles stay with the 2 x 4 kernel (rounds 2-5)
  bool wide_sparse = true;     // LDP_WIDE_SPARSE=0 / option "wide_sparse" 0: launches with a few missing calls leave the 8 x 8 tiles for the parallelogram plan (rounds 2-5)
  bool wide_async = false;     // option "wide_async": the 8 x 8 tiles on pair_mfma_wide_async_kernel (flags instead of a workgroup barrier per stage)
  // test hooks (ldp_debug_set_option only; 0 = off): results never depend on them
  uint32_t replay_steps = 0;   // "replay_steps" k: ldp_debug_replay_pairs() walks every subcontig in k instalments, as the streaming replay of a run does
  uint32_t decode_rows = 0;    // "decode_rows" k: record decode in launches of k rows (LD chains cut everywhere)
  bool decode_no_lds = false;  // "decode_no_lds" 1: decoded rows are assembled in global memory (what rows beyond 128 KiB take) instead of LDS
  uint32_t x_rows = 0;         // "x_rows" k: the chrX-weighted blocks in chunks of k rows
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
    uint32_t row_first = 0, row_end = 0;    // ... = the predicate rows of these second variants
    uint32_t mf_first = 0, mf_ct = 0;    // the same J range as matrix-pipe workgroups (mf_wgs) ...
    uint32_t mf_diag_ct = 0;             // ... of which the first mf_diag_ct are all-diagonal (partition_diag)
    uint32_t wd_first = 0, wd_ct = 0;    // ... and as wide-band tiles (wd_tiles)
    uint32_t wl_first = 0, wl_ct = 0;    // ... in launch order (wd_launch: eight XCD streams, padded to equal length)
    bool four_tiles = false;             // the group's last launch queued pair_mfma_tile4_kernel for them
    bool sparse_tiles = false;           // ... and pair_mfma_wide_kernel<., SPARSE> (the tiles on the route of rows with a few missing calls)
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
  uint32_t* h_pred = nullptr;  // pinned; allocated when a run needs the dense rows (inspection runs, engines without pred_csr, the overflow fallback)
  // the predicate rows as CSR (ldp_pred_csr.hip), written by the device straight into these pinned buffers; h_csr_flag[0] = overflow
  uint2* h_csr_meta = nullptr;
  uint2* h_csr_ent = nullptr;
  uint32_t* h_csr_flag = nullptr;
  uint64_t csr_capacity = 0;
  unsigned long long* d_csr_counter = nullptr;
  uint8_t* d_stored_inv = nullptr;  // per local row: 1 = the image row is stored inverted relative to the input (codes_kernel, ldp_device.h)
  bool wd_diag_lower = false;       // every diagonal tile's live products lie inside the 2 x 3 rectangles of pair_mfma_wide_kernel<., false, 3> (build_shard)
  bool any_stored_inv = false;      // some load may have inverted rows since the flags were last cleared
  uint32_t ctr_csr_overflows = 0;  // runs that fell back to the dense rows (test hook: option "csr_capacity")
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
  size_t stage_bytes = 0;              // bytes of one slot of the ring (ensure_staging)
  std::unique_ptr<CopyPool> own_pool;  // ldp_use_private_copy_threads(): this engine's own copy threads (else the process-wide pool)
  uint8_t* d_stage[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_done[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t h2d_stream[2] = {nullptr, nullptr};   // H2D copies of alternate slots (two SDMA queues: one tops out near 30 GB/s)
  hipEvent_t copied[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  bool prep_pending = false;
  // sample-mapped rows (ldp_set_sample_map): column f <- sample (map & 0x7fffffff), bit 31 = het becomes missing
  std::vector<uint32_t> sample_map;
  uint32_t map_raw_sample_ct = 0;
  uint32_t* d_sample_map = nullptr;
  // the map is a plain SUBSET of the file's samples (each at most once, no het -> missing: founders among non-founders): then the
  // device-side multiallelic collapse counts alleles over exactly these samples (bitmap over the file's samples)
  bool map_is_subset = false;
  uint32_t* d_map_mask = nullptr;
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

namespace ldph LDP_HIDDEN {
int fail(ldp_engine* e, int code, const std::string& msg);
int hipfail(ldp_engine* e, hipError_t rc, const char* what);

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

// ---- defined in one unit, called in another
// ldp_engine.cpp: planning and the device-side plan
void plan_mfma_generic(const std::vector<std::pair<uint32_t, uint32_t>>& runs, const uint32_t* lo_of, uint32_t j_first, uint32_t j_end,
                       std::vector<MfmaWG>* out_wgs, uint64_t* out_products, uint32_t i_first = 0, uint32_t i_end = 0xffffffffu,
                       std::vector<MfmaTile>* out_tiles = nullptr, uint32_t wide_min_reach = kWdMinReach);
uint32_t partition_diag(std::vector<MfmaWG>* wgs, size_t first, size_t ct);
void build_shard(ldp_engine* e);
void bind_gpu(ldp_engine* e);
int ensure_device_plan(ldp_engine* e);
int ensure_staging(ldp_engine* e);
int start_fetch_recs(ldp_engine* e);
int fetch_recs(ldp_engine* e);
// ldp_engine_run.cpp: launches and replay
void replay(ldp_engine* e, const uint32_t* pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out);
int finish_removed(ldp_engine* e, const std::vector<uint32_t>& R, uint64_t* removed);
int prepare_mf(ldp_engine* e, std::vector<double>* scratch, const double** mf_out);
void fill_pair_args(const ldp_engine* e, PairKernelArgs* out, bool with_early_exit);
int begin_load_epoch(ldp_engine* e);
hipError_t queue_route(ldp_engine* e, size_t slot, hipStream_t stream, int allow_sparse, uint32_t row_end);
int launch_ready_groups(ldp_engine* e);
}  // namespace ldph

using namespace ldph;
#endif
