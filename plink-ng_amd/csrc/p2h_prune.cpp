// p2h_prune.cpp -- plink2-hip: --indep-pairwise / --indep-pairphase (one translation unit of the front-end; plink2_hip_cli.cpp has the overview)
#include "p2h_cli.h"

namespace p2h {

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
  std::vector<uint32_t> sub_info, sub_owner;  // several engines: (length, first variant) and owning engine of every subcontig
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
        if (r == 0) {
          sub_info.assign(2 * static_cast<size_t>(subcontig_ct), 0);
          sub_owner.assign(subcontig_ct, 0);
          ldp_get_subcontigs(eng[r], &subcontig_ct, sub_info.data(), subcontig_ct);
        }
        rc = ldp_set_shard(eng[r], r, world, (r == 0) ? sub_owner.data() : nullptr);
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
    // --indep-pairphase: main AND phase track on the device (ldp_load_pgen_records_phased) when every sample is a founder -- variants with
    // more than one ALT allele included since round 5 (collapse and phase bits in one kernel); otherwise the host decoder.
    const bool device_phase = A.pairphase && all_founders && (storage_mode != 0x01) && (storage_mode != 0x02) && (!g_dbg.host_decode);
    const bool device_decode = (!direct) && ((!A.pairphase) || device_phase) && (storage_mode != 0x01) && (storage_mode != 0x02) && (!g_dbg.host_decode);
    // (records with several ALT alleles are collapsed on the device as well: over the file's samples, or over the founders when the
    // engines pick those through a subset sample map)
    device_multi = device_decode && (all_founders || device_subset);
    uint64_t file_size = 0;
    const void* file_bytes = device_decode ? ldp_pgen_file_bytes(pg, &file_size) : nullptr;
    std::vector<ldp_pgen_rec> rec_index;
    std::thread decoder;
    // The decoder runs beside the engine's copy threads (ldp_load_genotypes: 16 of them feeding the pinned ring); a record
    // takes microseconds, so a few dozen threads keep ahead of PCIe and more only get in the copies' way.
    const uint32_t decode_threads = g_dbg.decode_threads ? g_dbg.decode_threads : 32;
    double t_wait_decode = 0.0, t_load_calls = 0.0;
    // Several engines: each is fed by a host thread of its own, all at once -- the reference's main thread fills EVERY worker's slot of a
    // batch and the workers run together (plink2_ld.cc:1292-1417).  A feeding thread walks only the file ranges of its engine's shard,
    // binds itself next to ITS device (NUMA node of the device's PCI function) and gives the engine copy threads and a pinned ring of
    // its own there (ldp_use_private_copy_threads), so N PCIe links work at a time instead of one.  (Rows that reach the engines through
    // the host decoder or the host-side founder subset keep the one-after-the-other loop below: --indep-pairphase, --debug-host-decode.)
    if ((world > 1) && (!A.pairphase) && (!g_dbg.serial_feed) && (direct || device_decode) && (all_founders || device_subset)) {
      struct FeedStat {
        double t0 = -1.0, t1 = -1.0;
        uint64_t rows = 0, calls = 0;
        int node = -1, rc = 0;
        std::string err;
      };
      std::vector<FeedStat> stat(world);
      // variant ranges [first, end) of every engine's subcontigs, in variant order
      std::vector<std::vector<std::pair<uint32_t, uint32_t>>> mine(world);
      for (uint32_t s = 0; s < subcontig_ct; ++s) {
        mine[sub_owner[s]].push_back({sub_info[2 * s + 1], sub_info[2 * s + 1] + sub_info[2 * s]});
      }
      for (auto& v : mine) {
        std::sort(v.begin(), v.end());
      }
      std::vector<std::thread> feeders;
      for (int r = 0; r < world; ++r) {
        feeders.emplace_back([&, r]() {
          FeedStat& st = stat[r];
          AffinityMask before;
          if (!g_dbg.no_bind) {
            st.node = bind_near_device(r % n_devices, &before);
          }
          (void)ldp_use_private_copy_threads(eng[r]);
          std::vector<ldp_pgen_rec> idx;
          size_t at = 0;  // first range of mine[r] that may still reach the current run
          for (const Run& rn : runs) {
            while ((at < mine[r].size()) && (mine[r][at].second <= rn.q)) {
              ++at;
            }
            uint64_t owned_rows = 0;
            for (size_t k = at; (k < mine[r].size()) && (mine[r][k].first < rn.q + rn.n); ++k) {
              owned_rows += std::min(mine[r][k].second, rn.q + rn.n) - std::max(mine[r][k].first, rn.q);
            }
            if (!owned_rows) {
              continue;  // another engine's part of the file
            }
            const double tc = now_s();
            if (st.t0 < 0.0) {
              st.t0 = tc;
            }
            int rc;
            if (device_decode) {
              idx.resize(rn.n);
              uint32_t base_v = UINT32_MAX;
              ldp_pgen_rec base_rec;
              if (ldp_pgen_record_index(pg, rn.raw0, rn.n, idx.data(), &base_v) || ((base_v != UINT32_MAX) && ldp_pgen_record_index(pg, base_v, 1, &base_rec, nullptr))) {
                st.rc = 6;
                st.err = gpath + ": malformed variant record index.";
                return;
              }
              if (device_multi) {
                for (uint32_t t = 0; t < rn.n; ++t) {
                  const uint32_t alts = V.alt_ct[rn.raw0 + t];
                  if ((alts > 1) && (vcls[mk[rn.q + t]] != 5)) {
                    if (alts > 254) {
                      st.rc = 63;
                      st.err = "variant '" + V.id[rn.raw0 + t] + "' has more than 254 ALT alleles: not supported by plink2-hip.";
                      return;
                    }
                    idx[t].allele_ct = static_cast<uint8_t>(alts + 1);
                  }
                }
              }
              rc = ldp_load_pgen_records(eng[r], rn.q, rn.n, file_bytes, file_size, LDP_MEM_HOST, idx.data(), (base_v != UINT32_MAX) ? &base_rec : nullptr, raw_sample_ct, nullptr);
              if (rc) {
                st.rc = (rc == LDP_ERR_INVALID) ? 6 : 16;
                st.err = gpath + ": " + ldp_last_error(eng[r]);
                return;
              }
            } else {
              const int enc = load_encoding | (device_subset ? LDP_GENO_MAPPED : 0);
              rc = (direct_fd >= 0) ? ldp_load_genotypes_fd(eng[r], rn.q, rn.n, direct_fd, direct_off + static_cast<uint64_t>(rn.raw0) * rec_bytes, in_rec, enc)
                                    : ldp_load_genotypes(eng[r], rn.q, rn.n, direct + static_cast<uint64_t>(rn.raw0) * rec_bytes, in_rec, LDP_MEM_HOST, enc);
              if (rc) {
                st.rc = 16;
                st.err = ldp_last_error(eng[r]);
                return;
              }
            }
            st.t1 = now_s();
            st.rows += owned_rows;
            ++st.calls;
          }
          restore_affinity(before);
        });
      }
      for (std::thread& t : feeders) {
        t.join();
      }
      for (int r = 0; r < world; ++r) {
        if (stat[r].rc) {
          die(stat[r].rc, "\nError: %s\n", stat[r].err.c_str());
        }
      }
      if (A.timing) {
        // the engines' load intervals (seconds since main()): they overlap when the engines are fed at the same time
        double busy_sum = 0.0, span0 = 1e300, span1 = 0.0;
        for (int r = 0; r < world; ++r) {
          const FeedStat& st = stat[r];
          if (st.t0 < 0.0) {
            logprintf("\n[timing] engine %d (device %d): no rows of its own in this file", r, r % n_devices);
            continue;
          }
          logprintf("\n[timing] engine %d (device %d, NUMA node %d): fed %.3f - %.3f s (%.3f s), %llu rows in %llu calls, %.1f GB/s", r, r % n_devices, st.node,
                    st.t0 - t_begin, st.t1 - t_begin, st.t1 - st.t0, static_cast<unsigned long long>(st.rows), static_cast<unsigned long long>(st.calls),
                    (st.t1 > st.t0) ? (static_cast<double>(st.rows) * static_cast<double>(direct ? rec_bytes : out_rec) / (st.t1 - st.t0) / 1e9) : 0.0);
          busy_sum += st.t1 - st.t0;
          span0 = std::min(span0, st.t0);
          span1 = std::max(span1, st.t1);
        }
        if (span1 > span0) {
          logprintf("\n[timing] %d engines fed concurrently: load span %.3f s, sum of the engines' own intervals %.3f s (overlap factor %.2f)\n", world, span1 - span0, busy_sum,
                    busy_sum / (span1 - span0));
        }
      }
      return;
    }
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
      if (is_mt && (alts >= 2)) {
        multiallelic_sex_row(pg, raw_v, alts, mt_plan, &lo, &hi, inv_row.data(), out_rec, &mf);
        ++mt_ct;
      } else if (is_mt) {
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
    // which pair kernels the device-side route gave engine 0's launches (ldp_counters; the dispatch of plink2_ld.cc:699-723, per launch)
    logprintf("[timing] pair launches by route: complete data %u | a few missing calls %u (%u on the 8 x 8 tiles) | missing calls %u (%u on quarter tiles) | %u tiles planned, %llu pairs recounted exactly\n",
              c.route_complete_launches, c.route_sparse_launches, c.sparse_tile_launches, c.route_general_launches, c.four_tile_launches, c.wide_tiles,
              static_cast<unsigned long long>(c.sparse_exact_pairs));
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
        // variants with several ALT alleles: their rows (collapsed on the major allele of the chromosome's own allele-frequency rule) are
        // built here and replace the mapped main tracks
        std::vector<uint8_t> lo_a, hi_a, m_row;
        for (uint32_t w = 0; w < ks.size(); ++w) {
          const uint32_t raw_v = inc[ks[w]];
          if (V.alt_ct[raw_v] < 2) {
            continue;
          }
          if (lo_a.empty()) {
            lo_a.resize(raw_sample_ct);
            hi_a.resize(raw_sample_ct);
            m_row.resize(s_rec);
          }
          double mf = 0.0;
          multiallelic_sex_row(pg, raw_v, V.alt_ct[raw_v], sp, &lo_a, &hi_a, m_row.data(), s_rec, &mf);
          if (ldp_load_genotypes(se, w, 1, m_row.data(), s_rec, LDP_MEM_HOST, LDP_GENO_INVERSE) || ldp_set_maj_freqs(se, w, 1, &mf)) {
            die(16, "\nError: %s\n", ldp_last_error(se));
          }
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
            std::vector<uint8_t> raw_row(in_rec + 8), lo_a, hi_a, ph_a;
            for (uint32_t w = t; w < cnt; w += nthreads) {
              const uint32_t raw_v = inc[ks[w0 + w]];
              if (x_phased && (V.alt_ct[raw_v] > 1)) {
                // several ALT alleles: collapsed on the major allele of chrX's allele-frequency rule, the non-males' haplotypes split in Get1MP's reading
                if (lo_a.empty()) {
                  lo_a.resize(raw_sample_ct);
                  hi_a.resize(raw_sample_ct);
                  ph_a.resize(2 * ((static_cast<size_t>(raw_sample_ct) + 7) / 8));
                }
                bool unph = false;
                multiallelic_sex_row_phased(pg, raw_v, V.alt_ct[raw_v], sp, &lo_a, &hi_a, ph_a.data(), ph_a.size() / 2, rows.data() + static_cast<uint64_t>(w) * s_rec, s_rec,
                                            &mfs[w], &unph);
                if (unph) {
                  uint32_t cur = x_unphased.load();
                  while ((raw_v < cur) && !x_unphased.compare_exchange_weak(cur, raw_v)) {
                  }
                }
                continue;
              }
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
        S.file_to_hbm_done();  // (this thread moved next to device 0 for the file -> HBM leg, Session::join_hip: back to where it was)
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



}  // namespace p2h
