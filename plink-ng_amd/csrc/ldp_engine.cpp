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
#include "ldp_engine_internal.h"

namespace ldph LDP_HIDDEN {

// The main stream (conversion) outranks the pair streams: conversion blocks are short and HBM-bound, and the sooner
// they are through the sooner the host has the per-variant records it needs to start replaying finished groups.
hipError_t create_stream(hipStream_t* out, bool high_priority) {
  int lo = 0, hi = 0;
  static const bool flat = (LDP_ENV("LDP_STREAM_PRIORITY") != nullptr) && (strcmp(LDP_ENV("LDP_STREAM_PRIORITY"), "0") == 0);
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

void free_device(ldp_engine* e) {
  if (!e->gpu_ok) {
    return;
  }
  (void)hipFree(e->d_planes);
  (void)hipFree(e->d_codes);
  e->d_codes = nullptr;
  (void)hipFree(e->d_map_mask);
  e->d_map_mask = nullptr;
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
  if (e->h_csr_meta) {
    (void)hipHostFree(e->h_csr_meta);
    e->h_csr_meta = nullptr;
  }
  if (e->h_csr_ent) {
    (void)hipHostFree(e->h_csr_ent);
    e->h_csr_ent = nullptr;
  }
  if (e->h_csr_flag) {
    (void)hipHostFree(e->h_csr_flag);
    e->h_csr_flag = nullptr;
  }
  (void)hipFree(e->d_csr_counter);
  e->d_csr_counter = nullptr;
  (void)hipFree(e->d_stored_inv);
  e->d_stored_inv = nullptr;
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
  const char* ee = LDP_ENV("LDP_EARLY_EXIT");
  o.early_exit = !(ee && (strcmp(ee, "0") == 0));
  const char* m = LDP_ENV("LDP_PAIR_MFMA");
  o.pair_mfma = !(m && (strcmp(m, "0") == 0));
  const char* off = LDP_ENV("LDP_PAIR_SPARSE");
  const char* f = LDP_ENV("LDP_DEBUG_SPARSE_FRAC");
  o.sparse_frac = (off && (strcmp(off, "0") == 0)) ? 0.0 : (f ? atof(f) : 0.005);
  const char* four = LDP_ENV("LDP_PAIR_FOUR");
  o.pair_four = !(four && (strcmp(four, "0") == 0));
  const char* ft = LDP_ENV("LDP_PAIR_FOUR_TILES");
  o.four_tiles = !(ft && (strcmp(ft, "0") == 0));
  const char* ws = LDP_ENV("LDP_WIDE_SPARSE");
  o.wide_sparse = !(ws && (strcmp(ws, "0") == 0));
  const char* dl = LDP_ENV("LDP_DEBUG_WIDE_DIAG_LAST");
  o.wide_diag_last = dl ? static_cast<uint32_t>(std::max(0, atoi(dl))) : 2u;
  if (const char* wa = LDP_ENV("LDP_DEBUG_WIDE_ASYNC")) {
    o.wide_async = (atoi(wa) != 0);
  }
  if (const char* w = LDP_ENV("LDP_DEBUG_WIDE_MIN_REACH")) {
    o.wide_min_reach = static_cast<uint32_t>(std::max(0, atoi(w)));
  }
  return o;
}

int checkpoint_fractions(double r2_param, double* frac) {
  static const double kStep[kCheckpoints] = {0.012, 0.04, 0.08, 0.16, 0.36};  // (tuned on config 2: an earlier first checkpoint pays, a failed one costs little)
  const double f0 = 1.0 - sqrt(r2_param);
  int n = 0;
  if (const char* dbg = LDP_ENV("LDP_DEBUG_CP_FRACS")) {  // tuning aid: comma-separated absolute fractions
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
                       std::vector<MfmaWG>* out_wgs, uint64_t* out_products, uint32_t i_first, uint32_t i_end, std::vector<MfmaTile>* out_tiles,
                       uint32_t wide_min_reach) {
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
      const char* w = LDP_ENV("LDP_DEBUG_MFMA_WAVES");
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
        const char* mu = LDP_ENV("LDP_DEBUG_MAX_UNITS");
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
    if (const char* tg = LDP_ENV("LDP_DEBUG_GROUPS")) {
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
      g.row_first = e->items[i0].j0;
      g.row_end = e->items[i1 - 1].jend;
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
    // the diagonal tiles' own kernel covers J blocks (0,1) x V 0-2, (2,3) and (4,5) x V 0-5, (6,7) x V 0-7: every prune plan's diagonal tile lies
    // inside (its live products are on and below the diagonal); a plan that does not keeps the 2 x 4 kernel for them
    e->wd_diag_lower = !e->wd_tiles.empty();
    for (const MfmaTile& t : e->wd_tiles) {
      if ((t.jv == t.vv) && (t.mask & ~0xffff3f3f3f3f0707ull)) {
        e->wd_diag_lower = false;
      }
    }
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
  static const bool plan_timing = LDP_ENV("LDP_DEBUG_LOAD_TIMING") != nullptr;
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
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_stored_inv), n));
    HIP_TRY(e, hipMemsetAsync(e->d_stored_inv, 0, n, e->stream));
    e->any_stored_inv = false;
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
  // (the dense predicate rows on the host -- 270 MB of pinned memory for a config-3 share -- only when a run wants them: ensure_h_pred)
  if (e->opt.pred_csr && e->local_ct) {
    e->csr_capacity = e->opt.csr_capacity ? e->opt.csr_capacity : std::min<uint64_t>(std::max<uint64_t>(1u << 16, e->pred_words / 4), 0xffffffffull);
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_csr_meta), static_cast<size_t>(e->local_ct) * sizeof(uint2), hipHostMallocDefault));
    memset(e->h_csr_meta, 0, static_cast<size_t>(e->local_ct) * sizeof(uint2));  // (rows no launch group owns hold no pair: no entries, ever)
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_csr_ent), static_cast<size_t>(e->csr_capacity) * sizeof(uint2), hipHostMallocDefault));
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_csr_flag), 64, hipHostMallocDefault));
    e->h_csr_flag[0] = 0;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_csr_counter), sizeof(unsigned long long)));
    HIP_TRY(e, hipMemsetAsync(e->d_csr_counter, 0, sizeof(unsigned long long), e->stream));
  }
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
  size_t slot_bytes = kStageBytes;
  if (const char* mb = LDP_ENV("LDP_DEBUG_STAGE_MB")) {  // (measurement build: the slot size of the pinned ring)
    slot_bytes = static_cast<size_t>(std::max(1, atoi(mb))) << 20;
  }
  const size_t bytes = std::max(slot_bytes, row_bytes);
  e->stage_bytes = bytes;
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
    if (LDP_ENV("LDP_DEBUG_TIMELINE")) {
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

}  // namespace ldph

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
// (see include/ldprune_hip.h: the copy threads of ONE engine, created by and bound like the thread that feeds it)
int ldp_use_private_copy_threads(ldp_engine* e) {
  if (!e) {
    return LDP_ERR_INVALID;
  }
  if (!e->own_pool) {
    e->own_pool.reset(new CopyPool());
  }
  return LDP_OK;
}

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

}  // extern "C"

extern "C" {

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
  // a plain subset?  (then records with several ALT alleles can be collapsed on the device: ldp_load_pgen_records)
  std::vector<uint32_t> mask((static_cast<size_t>(raw_sample_ct) + 31) / 32 + 1, 0);
  bool subset = true;
  for (uint32_t f = 0; f < e->P.founder_ct; ++f) {
    const uint32_t sidx = m[f] & 0x7fffffffu;
    if ((m[f] >> 31) || ((mask[sidx >> 5] >> (sidx & 31)) & 1u)) {
      subset = false;
      break;
    }
    mask[sidx >> 5] |= 1u << (sidx & 31);
  }
  (void)hipFree(e->d_map_mask);
  e->d_map_mask = nullptr;
  e->map_is_subset = subset;
  if (subset) {
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_map_mask), mask.size() * sizeof(uint32_t)));
    HIP_TRY(e, hipMemcpy(e->d_map_mask, mask.data(), mask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
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
  } else if (n == "wide_async") {
    e->opt.wide_async = (value != 0.0);
  } else if (n == "orient_rows") {
    e->opt.orient_rows = (value != 0.0);
  } else if (n == "pred_csr") {
    if (e->planned && e->gpu_ok && (value != 0.0) && !e->h_csr_meta) {
      return fail(e, LDP_ERR_STATE, "pred_csr can only be switched ON before the engine's device buffers exist (ldp_set_variants)");
    }
    e->opt.pred_csr = (value != 0.0);
  } else if (n == "csr_capacity") {
    if (e->planned) {
      return fail(e, LDP_ERR_STATE, "csr_capacity must be set before ldp_set_variants()");
    }
    e->opt.csr_capacity = static_cast<uint64_t>(std::max(0.0, value));
  } else if (n == "wide_diag_kernel") {
    e->opt.wide_diag_kernel = (value != 0.0);
  } else if (n == "wide_sparse") {
    e->opt.wide_sparse = (value != 0.0);
  } else if (n == "replay_steps") {
    e->opt.replay_steps = static_cast<uint32_t>(std::max(0.0, value));
  } else if (n == "decode_rows") {
    e->opt.decode_rows = static_cast<uint32_t>(std::max(0.0, value));
  } else if (n == "decode_no_lds") {
    e->opt.decode_no_lds = (value != 0.0);
  } else if (n == "x_rows") {
    e->opt.x_rows = static_cast<uint32_t>(std::max(0.0, value));
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
      ref2het[p] = ldp::img_differs(rec.flags) ? (r ^ h) : r;  // (planes in major-allele orientation, whichever way the row is stored)
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
