// ldp_engine_run.cpp -- the pair launches of a prune, the greedy replay of their predicate bits, ldp_run*
// (host runtime behind include/ldprune_hip.h; ldp_engine.cpp has the overview)
#include "ldp_engine_internal.h"

namespace ldph LDP_HIDDEN {
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
// The predicate rows the replay reads: dense bit rows (`dense` != nullptr), or their non-zero words as CSR (ldp_pred_csr.hip: meta[j] = first
// entry / entries of row j, an entry = (word index inside the row, bits), ascending).
struct PredView {
  const uint32_t* dense;
  const uint2* meta;
  const uint2* ent;
  uint64_t capacity;
};
// word `widx` of row j
inline uint32_t pred_word(const ldp_engine* e, const PredView& pv, uint32_t j, uint32_t widx) {
  if (pv.dense) {
    return pv.dense[e->row_off[j] + widx];
  }
  const uint2 mt = pv.meta[j];
  if (static_cast<uint64_t>(mt.x) + mt.y > pv.capacity) {
    return 0;  // (an overflowed run: the caller replays from the dense rows afterwards)
  }
  for (uint32_t q = 0; q < mt.y; ++q) {
    if (pv.ent[mt.x + q].x == widx) {
      return pv.ent[mt.x + q].y;
    }
  }
  return 0;
}

uint64_t replay_subcontig(const ldp_engine* e, uint32_t k, const PredView& pv, const double* mf, std::vector<uint32_t>& R,
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
        const uint32_t wbase = lo >> 5;
        const uint32_t nw = ((j - 1) >> 5) - wbase + 1;
        const double mf_j_eps = mf[j] * (1 + kSmallEpsilon);
        bool second_removed = false;
        auto visit = [&](uint32_t w, uint32_t bits) {  // word w of the row, highest first
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
        };
        if (pv.dense) {
          const uint32_t* row = pv.dense + e->row_off[j];
          for (uint32_t w = nw; (w-- > 0) && !second_removed;) {
            if (row[w]) {
              visit(w, row[w]);
            }
          }
        } else {
          const uint2 mt = pv.meta[j];
          if (static_cast<uint64_t>(mt.x) + mt.y <= pv.capacity) {
            const uint2* en = pv.ent + mt.x;
            for (uint32_t q = mt.y; (q-- > 0) && !second_removed;) {
              if (en[q].x < nw) {
                visit(en[q].x, en[q].y);
              }
            }
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
            const uint32_t word = pred_word(e, pv, second, (first >> 5) - (lo2 >> 5));
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
int replay_progressive(ldp_engine* e, const PredView& pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out, double* busy_ms_out) {
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
  const bool timeline = LDP_ENV("LDP_DEBUG_TIMELINE") != nullptr;
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

void replay_view(ldp_engine* e, const PredView& pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out) {
  std::vector<uint32_t> first_unchecked;
  if (e->P.plink1_order) {
    first_unchecked.assign(e->local_ct, 0);
  }
  // option "replay_steps" k (test hook): every subcontig in k instalments, the way the streaming replay of a run advances
  // through it as the launch groups land
  if (e->opt.replay_steps) {
    const uint32_t steps = e->opt.replay_steps;
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

void replay(ldp_engine* e, const uint32_t* pred, const double* mf, std::vector<uint32_t>& R, uint64_t* replay_pairs_out) {
  const PredView pv = {pred, nullptr, nullptr, 0};
  replay_view(e, pv, mf, R, replay_pairs_out);
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
  A.wd_async = e->opt.wide_async ? 1u : 0u;
  A.wd_sparse = 0;
  A.wd_diag_split = 0;
}

// the dense predicate rows on the host: pinned, allocated the first time a run wants them
int ensure_h_pred(ldp_engine* e) {
  if (!e->h_pred) {
    HIP_TRY(e, hipHostMalloc(&e->h_pred, std::max<size_t>(e->pred_words, 1) * sizeof(uint32_t), hipHostMallocDefault));
  }
  return LDP_OK;
}
// prune runs return their predicate rows as CSR (ldp_pred_csr.hip) unless the engine says otherwise
inline bool use_pred_csr(const ldp_engine* e) { return e->opt.pred_csr && (e->h_csr_meta != nullptr); }
// a new series of pair launches begins: the device-side counters they add to
int reset_launch_counters(ldp_engine* e) {
  HIP_TRY(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(unsigned long long), e->stream));
  if (e->d_csr_counter) {
    HIP_TRY(e, hipMemsetAsync(e->d_csr_counter, 0, sizeof(unsigned long long), e->stream));
    HIP_TRY(e, hipMemsetAsync(e->h_csr_flag, 0, sizeof(uint32_t), e->stream));  // (pinned host memory, cleared in stream order)
  }
  return LDP_OK;
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
  {
    const int rrc = reset_launch_counters(e);
    if (rrc) {
      return rrc;
    }
  }
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
    // ... and launches whose rows have only a few: the tiles' SPARSE instantiation
    A.wd_sparse = (A.sparse_ok && e->opt.wide_sparse && A.n_wd_tiles) ? 1u : 0u;
    // complete data: the diagonal tiles in 2 x 3 rectangles, by a kernel of their own (prune launches: their live products lie on and below the diagonal)
    A.wd_diag_split = (e->opt.wide_diag_kernel && !e->opt.wide_async && A.n_wd_tiles && !A.stats && !A.r2_out && !A.r2_hits && e->wd_diag_lower) ? 1u : 0u;
    g.sparse_tiles = (A.wd_sparse != 0);
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
  if (use_pred_csr(e)) {
    // the group's rows go back as their non-zero words, written by the device straight into pinned host memory (plink2_ld.cc:1093-1097 writes a
    // removed bit where it is decided; here 270 MB of predicate rows shrink to ~30 MB before they cross PCIe)
    ldp::PredCsrArgs C;
    C.pred = e->d_pred;
    C.row_off = e->d_row_off;
    C.row_first = g.row_first;
    C.row_end = g.row_end;
    C.meta = e->h_csr_meta;
    C.ent = e->h_csr_ent;
    C.counter = e->d_csr_counter;
    C.capacity = e->csr_capacity;
    C.overflow = e->h_csr_flag;
    krc = launch_pred_compact(C, ps);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pred_compact_kernel launch");
    }
  } else if (g.word_end > g.word_first) {
    const int prc = ensure_h_pred(e);
    if (prc) {
      return prc;
    }
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
    rc = ensure_h_pred(e);  // (an inspection run reads the dense rows)
    if (rc) {
      return rc;
    }
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
      rc = reset_launch_counters(e);  // (ahead of every ev_ready)
      if (rc) {
        return rc;
      }
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
    const bool csr = use_pred_csr(e);
    const PredView pview = csr ? PredView{nullptr, e->h_csr_meta, e->h_csr_ent, e->csr_capacity} : PredView{e->h_pred, nullptr, nullptr, 0};
    rc = replay_progressive(e, pview, mf, R, &replay_pairs, &replay_busy_ms);
    if (rc) {
      return rc;
    }
    replayed = true;
    tl[3] = now_ms();
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (csr && e->h_csr_flag[0]) {
      // more non-zero predicate words than the CSR buffer holds (a quarter of all words): the dense rows are still in HBM -- copy them
      // back and replay from those
      rc = ensure_h_pred(e);
      if (rc) {
        return rc;
      }
      HIP_TRY(e, hipMemcpyAsync(e->h_pred, e->d_pred, std::max<size_t>(e->pred_words, 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      std::fill(R.begin(), R.end(), 0u);
      replay_pairs = 0;
      t_replay = now_ms();
      replay(e, e->h_pred, mf, R, &replay_pairs);
      replay_busy_ms = now_ms() - t_replay;
      e->ctr_csr_overflows += 1;
    }
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
  uint32_t four_tile_launches = 0, sparse_tile_launches = 0;
  if (e->mf_enabled && !e->mf_wgs.empty()) {
    const uint32_t* h_route = reinterpret_cast<const uint32_t*>(e->h_counters_pin + 4);
    if (stats) {
      ++route_ct[std::min<uint32_t>(h_route[e->groups.size()], 2)];
    } else {
      for (size_t gi = 0; gi < e->groups.size(); ++gi) {
        if (e->groups[gi].mf_ct) {
          ++route_ct[std::min<uint32_t>(h_route[gi], 2)];
          four_tile_launches += ((h_route[gi] >= 2) && e->groups[gi].four_tiles) ? 1u : 0u;
          sparse_tile_launches += ((h_route[gi] == 1) && e->groups[gi].sparse_tiles) ? 1u : 0u;
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
  if (LDP_ENV("LDP_DEBUG_TIMELINE")) {
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
  e->ctr.sparse_tile_launches = sparse_tile_launches;
  e->ctr.ms_replay = replayed ? replay_busy_ms : (t_end - t_replay);  // (time spent replaying, not waiting for groups)
  e->ctr.ms_run_total = t_end - t_start;
  e->ctr.pair_kernel_launches = launches;
  return LDP_OK;
}

}  // namespace ldph

extern "C" {

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

}  // extern "C"
