// ldp_engine_load.cpp -- rows on their way into the engine: host / device / file-descriptor rows, .pgen records decoded on the device, mapped rows
// (host runtime behind include/ldprune_hip.h; ldp_engine.cpp has the overview)
#include "ldp_engine_internal.h"

extern "C" {

namespace ldph LDP_HIDDEN {
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
    stage_rows = std::max<size_t>(1, e->stage_bytes / pack_stride);
  }
  const uint64_t gather_stride = ((static_cast<uint64_t>(e->P.founder_ct) + 3) / 4 + 3) & ~static_cast<uint64_t>(3);
  size_t gather_rows = 0;
  if (mapped) {
    if (!e->d_sample_map) {  // (the device copy went with a re-plan or ldp_release_device(): the host copy is the master)
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_sample_map), e->sample_map.size() * sizeof(uint32_t)));
      HIP_TRY(e, hipMemcpy(e->d_sample_map, e->sample_map.data(), e->sample_map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    gather_rows = std::max<size_t>(1, std::max<size_t>(stage_rows, std::max(kStageBytes, e->stage_bytes) / gather_stride));
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
  static const bool eager_always = (LDP_ENV("LDP_EAGER_PAIRS") != nullptr) && (strcmp(LDP_ENV("LDP_EAGER_PAIRS"), "1") == 0);
  const bool eager = (!e->matrix_mode) && (!e->band_r2_mode) && ((location == LDP_MEM_HOST) || eager_always);
  // host threads per 16 MiB slot and bytes per task: a memcpy out of a mapping runs at its best on 16 threads; pread() calls (the file
  // descriptor form, what plink2-hip uses for fixed-width rows) want more, smaller ones -- 32 x 256 KiB: 0.36-0.38 s for config 2's
  // 12.5 GB on a host where the mapping took 0.35-0.97 s from run to run (profiles/r04_experiments.md)
  static const uint32_t copy_threads_env = []() {
    const char* c = LDP_ENV("LDP_DEBUG_COPY_THREADS");
    return (c && atoi(c) > 0) ? static_cast<uint32_t>(atoi(c)) : 0u;
  }();
  static const uint64_t copy_task_env = []() {
    const char* c = LDP_ENV("LDP_DEBUG_COPY_TASK_KB");
    return (c && atoi(c) > 0) ? (static_cast<uint64_t>(atoi(c)) << 10) : 0ull;
  }();
  const uint32_t copy_threads = copy_threads_env ? copy_threads_env : ((src_fd >= 0) ? 32u : 16u);
  const uint64_t copy_task_bytes = copy_task_env ? copy_task_env : ((src_fd >= 0) ? (256ull << 10) : (1ull << 20));
  static const bool load_timing = LDP_ENV("LDP_DEBUG_LOAD_TIMING") != nullptr;  // host side of the file -> HBM leg, on stderr
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
        static const bool use_pool = !(LDP_ENV("LDP_DEBUG_COPY_POOL") && (atoi(LDP_ENV("LDP_DEBUG_COPY_POOL")) == 0));
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
          (e->own_pool ? *e->own_pool : CopyPool::get()).run(tasks, copy_threads, copy_task);
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
          const char* m = LDP_ENV("LDP_DEBUG_H2D_MODE");
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
      PA.stored_inv = (e->codes_format && e->d_stored_inv) ? (e->d_stored_inv + l0) : nullptr;
      PA.orient = (e->codes_format && e->opt.orient_rows) ? 1 : 0;
      e->any_stored_inv = e->any_stored_inv || (PA.orient != 0);
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
}  // namespace ldph

int ldp_load_genotypes(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* geno, uint64_t stride_bytes, int location, int encoding) {
  return load_rows_impl(e, first_variant, n, geno, stride_bytes, location, encoding, nullptr, nullptr);
}

int ldp_load_genotypes_fd(ldp_engine* e, uint32_t first_variant, uint32_t n, int fd, uint64_t file_offset, uint64_t stride_bytes, int encoding) {
  if (fd < 0) {
    return e ? fail(e, LDP_ERR_INVALID, "bad file descriptor") : LDP_ERR_INVALID;
  }
  return load_rows_impl(e, first_variant, n, nullptr, stride_bytes, LDP_MEM_HOST, encoding, nullptr, nullptr, fd, file_offset);
}

namespace ldph LDP_HIDDEN {
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
}  // namespace ldph

// Variant records of a variable-width .pgen, decoded on the device (ldp_pgen_decode.hip) into rows of the FILE's samples and
// loaded from there like any device-resident rows.  What the reference does per variant on its one reader thread
// (PgrGetInv1 -> ReadGenovecSubsetUnsafe, plink2_ld.cc:1345-1390 / pgenlib_read.cc:2849-2912, 5417-5563).
namespace ldph LDP_HIDDEN {
int load_pgen_records_impl(ldp_engine* e, uint32_t first_variant, uint32_t n, const void* bytes, uint64_t n_bytes, int location, const ldp_pgen_rec* recs,
                           const ldp_pgen_rec* ld_base, uint32_t raw_sample_ct, uint32_t* major_allele_out, bool phased, uint32_t* unphased_variant);
}  // namespace ldph

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

namespace ldph LDP_HIDDEN {
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
  if (any_multi && mapped && !e->map_is_subset) {
    return fail(e, LDP_ERR_UNSUPPORTED, "variants with more than one ALT allele are collapsed over the file's samples or a plain subset of them: not with a sample map that repeats samples or turns het calls missing (collapse them on the host, LDP_GENO_INVERSE)");
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
  // (a launch that decodes straight into the image -- below -- reads and writes rows at the image's pitch: the carried base too)
  const uint64_t base_cap = std::max<uint64_t>(stride, e->codes_format ? e->code_row_bytes : 0);
  if (e->ld_base_cap < base_cap) {
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    (void)hipFree(e->d_ld_base);
    e->d_ld_base = nullptr;
    e->ld_base_cap = 0;
    e->ld_base_valid = false;
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_ld_base), base_cap));
    HIP_TRY(e, hipMemsetAsync(e->d_ld_base, 0, base_cap, e->stream));
    e->ld_base_cap = base_cap;
  }
  // (the carried base is the record that PRECEDES this call's first one in the file: same engine position AND same file position --
  // a caller that loads non-adjacent file ranges into adjacent engine indices gets LDP_ERR_INVALID below instead of a wrong row)
  bool have_carried = e->ld_base_valid && (e->dec_next_variant == first_variant) && (e->dec_next_offset == recs[0].offset);
  // ---- in launches of at most ~256 MiB of rows
  uint32_t rows_per_launch = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(n, (256ull << 20) / stride)));
  if (e->opt.decode_rows) {  // (test hook, option "decode_rows": many small launches, LD chains cut everywhere)
    rows_per_launch = e->opt.decode_rows;
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
      // the span of THIS launch's own records goes up in one copy; the caller's ld_base record -- which may lie anywhere before them
      // (variant filters, a run that starts late in a 64k-variant block: the gap can be gigabytes) -- in a copy of its own behind it
      uint64_t lo = UINT64_MAX, hi = 0;
      for (uint32_t q = 0; q < cnt; ++q) {
        lo = std::min<uint64_t>(lo, descs[q].off);
        hi = std::max<uint64_t>(hi, descs[q].off + descs[q].len);
      }
      const uint64_t span = (hi - lo + 15) & ~static_cast<uint64_t>(15);
      const uint64_t base_len = with_base_rec ? ld_base->length : 0;
      void* p = nullptr;
      if ((rc = dec_reserve(e, 0, span + base_len + 32, &p))) {
        return rc;
      }
      HIP_TRY(e, hipMemcpyAsync(p, static_cast<const uint8_t*>(bytes) + lo, hi - lo, hipMemcpyHostToDevice, e->stream));
      if (with_base_rec) {
        HIP_TRY(e, hipMemcpyAsync(static_cast<uint8_t*>(p) + span, static_cast<const uint8_t*>(bytes) + ld_base->offset, base_len, hipMemcpyHostToDevice, e->stream));
        descs[cnt].off = lo + span;  // (where the kernels find it: d_bytes + off)
      }
      // record offsets stay as the caller gave them: the kernels add them to this base (an address computed as an integer: `p - lo` need
      // not lie inside the allocation)
      d_bytes = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) - static_cast<uintptr_t>(lo));
    }
    // Rows of variants this engine owns back to back are decoded STRAIGHT INTO ITS IMAGE (round 5) and counted there by the load below --
    // what ReadGenovecSubsetUnsafe does when it writes its caller's buffer (pgenlib_read.cc:2849-2912): no scratch row, no second copy.
    // Not with a sample map or phase bits (those rows are re-laid out on the way), not for the launch that carries the caller's ld_base
    // record as an extra row, not on bit-plane engines.
    int64_t l_first = -1;
    bool into_image = e->codes_format && (!mapped) && (!phased) && (!with_base_rec) && (stride <= e->code_row_bytes);
    if (into_image) {
      l_first = e->global_to_local[first_variant + q0];
      into_image = (l_first >= 0);
      for (uint32_t q = 1; into_image && (q < cnt); ++q) {
        into_image = (e->global_to_local[first_variant + q0 + q] == l_first + static_cast<int64_t>(q));
      }
    }
    if (into_image) {
      // (rows about to be overwritten may belong to launches of the current load epoch: the same rule as in load_rows_impl)
      for (uint32_t q = 0; q < cnt; ++q) {
        if (e->load_tag[static_cast<size_t>(l_first) + q] == e->load_epoch) {
          if ((rc = begin_load_epoch(e))) {
            return rc;
          }
          break;
        }
      }
      // the decode overwrites these rows before their records have been checked: from here on they hold nothing -- whatever happens below
      // (a malformed record, a failed copy, launch or sync) -- until load_rows_impl has counted them again
      std::fill(e->loaded.begin() + l_first, e->loaded.begin() + l_first + cnt, static_cast<uint8_t>(0));
      if (e->d_stored_inv) {
        HIP_TRY(e, hipMemsetAsync(e->d_stored_inv + l_first, 0, cnt, e->stream));  // (... and they will be in the file's orientation)
      }
    }
    const uint64_t lstride = into_image ? e->code_row_bytes : stride;
    void *p_recs = nullptr, *p_rows = nullptr, *p_end = nullptr, *p_multi = nullptr, *p_mf = nullptr, *p_mi = nullptr, *p_inv = nullptr;
    if ((rc = dec_reserve(e, 1, rows * sizeof(ldp::PgenRecDesc), &p_recs)) ||
        (into_image ? 0 : (rc = dec_reserve(e, 2, static_cast<size_t>(rows) * stride, &p_rows))) ||
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
    DA.rows = into_image ? (e->d_codes + static_cast<uint64_t>(l_first) * e->code_row_bytes) : static_cast<uint8_t*>(p_rows);
    DA.stride = lstride;
    DA.carried_base = have_carried ? e->d_ld_base : nullptr;
    DA.main_end = static_cast<uint64_t*>(p_end);
    DA.error = d_err;
    DA.pass = 0;
    DA.any_ld = any_ld ? 1 : 0;
    DA.no_lds = e->opt.decode_no_lds ? 1 : 0;
    DA.multi_rec = static_cast<const uint32_t*>(p_multi);
    DA.n_multi = static_cast<uint32_t>(multi.size());
    DA.maj_freq = static_cast<double*>(p_mf);
    DA.maj_idx = static_cast<uint32_t*>(p_mi);
    DA.row_inverse = static_cast<uint8_t*>(p_inv);
    if (mapped && e->map_is_subset && (!e->d_map_mask) && !multi.empty()) {
      // (the device copy went with a re-plan or ldp_release_device(): the host's sample map is the master)
      std::vector<uint32_t> mask((static_cast<size_t>(e->map_raw_sample_ct) + 31) / 32 + 1, 0);
      for (uint32_t sm : e->sample_map) {
        mask[(sm & 0x7fffffffu) >> 5] |= 1u << (sm & 31);
      }
      HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&e->d_map_mask), mask.size() * sizeof(uint32_t)));
      HIP_TRY(e, hipMemcpy(e->d_map_mask, mask.data(), mask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    DA.sample_mask = (mapped && e->map_is_subset) ? e->d_map_mask : nullptr;  // (a subset map: the alleles of its samples decide the major allele)
    DA.mask_ct = e->P.founder_ct;
    DA.phase_off = phase_off;
    DA.unphased = d_unphased;
    hipError_t krc = launch_pgen_main(DA, e->stream);
    if (krc != hipSuccess) {
      return hipfail(e, krc, "pgen_main_kernel launch");
    }
    if (phased) {
      // (not the caller's ld_base row behind them: only its codes are a base; records with several ALT alleles get their phase bits from
      // pgen_aux1_kernel below, which knows their het calls)
      krc = launch_pgen_phase(DA, cnt, e->stream);
      if (krc != hipSuccess) {
        return hipfail(e, krc, "pgen_phase_kernel launch");
      }
    }
    // the row the next launch's LD-compressed records may build on (taken BEFORE the multiallelic collapse rewrites rows:
    // an LD base is the main track as stored)
    if (last_alone_row >= 0) {
      HIP_TRY(e, hipMemcpyAsync(e->d_ld_base, DA.rows + static_cast<uint64_t>(last_alone_row) * lstride, stride, hipMemcpyDeviceToDevice, e->stream));
      have_carried = true;
    } else if (with_base_rec) {
      HIP_TRY(e, hipMemcpyAsync(e->d_ld_base, DA.rows + static_cast<uint64_t>(cnt) * lstride, stride, hipMemcpyDeviceToDevice, e->stream));
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
      e->ld_base_valid = false;  // (rows decoded in place were marked not loaded before the launch)
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
    status = load_rows_impl(e, first_variant + q0, cnt, DA.rows, lstride, LDP_MEM_DEVICE, LDP_GENO_REF | (mapped ? LDP_GENO_MAPPED : 0) | (phased ? LDP_GENO_PHASED : 0),
                            multi.empty() ? nullptr : DA.row_inverse, multi.empty() ? nullptr : h_inverse.data());
    if (LDP_ENV("LDP_DEBUG_TIMELINE")) {
      fprintf(stderr, "decode launch of %u rows: queued in %.3f ms, device done %.3f ms later, rows loaded %.3f ms after that\n", rows, t_q - t_call, t_s - t_q, now_ms() - t_s);
    }
    if (status == LDP_OK) {
      if (into_image) {
        e->ctr.decoded_in_place_rows += cnt;
      }
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
}  // namespace ldph

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
  if (e->any_stored_inv && e->d_stored_inv) {
    // rows an earlier load stored inverted (major-allele-oriented image) go back to the input's orientation before the caller sees or
    // rewrites them; the call returns when that is done (the caller writes from a stream of its own)
    HIP_TRY(e, hipSetDevice(e->device));
    const hipError_t urc = ldp::launch_unflip_rows(e->d_codes + static_cast<uint64_t>(l0) * e->code_row_bytes, e->code_row_bytes, e->d_stored_inv + l0, n, e->stream);
    if (urc != hipSuccess) {
      return hipfail(e, urc, "unflip_rows_kernel launch");
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  *device_rows = e->d_codes + static_cast<uint64_t>(l0) * e->code_row_bytes;
  *stride_bytes = e->code_row_bytes;
  return LDP_OK;
}

}  // extern "C"
