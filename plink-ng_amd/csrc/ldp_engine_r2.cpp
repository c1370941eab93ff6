// ldp_engine_r2.cpp -- the r^2 outputs: band rows, dense rows and blocks, hit filters, chrX-weighted blocks
// (host runtime behind include/ldprune_hip.h; ldp_engine.cpp has the overview)
#include "ldp_engine_internal.h"

namespace ldph LDP_HIDDEN {
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
  // row chunks of at most ~1 GiB of tuples per engine; a chunk without a chrX row only needs the chrX columns (pairs are i < j: nothing
  // right of the chunk's last row either).  Blocks without any chrX pair -- most chunks of a genome-wide table -- cost nothing.
  const size_t esz = as_float ? sizeof(float) : sizeof(double);
  uint32_t rows_per = static_cast<uint32_t>(std::max<uint64_t>(32, ((1ull << 30) / sizeof(ldp_pair_stats_t)) / col_ct) & ~31ull);
  if (e->opt.x_rows) {  // (test hook, option "x_rows": many small chunks)
    rows_per = e->opt.x_rows;
  }
  struct Chunk {
    uint32_t r0, rows, c0, c1;
  };
  std::vector<Chunk> chunks;
  uint64_t max_elems = 0;
  for (uint32_t r0 = row_first; r0 < row_end; r0 += rows_per) {
    const uint32_t rows = std::min(rows_per, row_end - r0);
    bool any_x_row = false;
    for (uint32_t j = r0; j < r0 + rows; ++j) {
      any_x_row = any_x_row || (is_x[j] != 0);
    }
    const uint32_t c0 = any_x_row ? col_first : xc_lo;
    const uint32_t c1 = std::min(any_x_row ? col_end : xc_hi, r0 + rows - 1);
    if (c0 < c1) {
      chunks.push_back({r0, rows, c0, c1});
      max_elems = std::max(max_elems, static_cast<uint64_t>(rows) * (c1 - c0));
    }
  }
  if (chunks.empty()) {
    return LDP_OK;
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
  HIP_TRY(e, hipMalloc(&ta_buf.p, max_elems * sizeof(ldp_pair_stats_t)));
  if (male) {
    HIP_TRY(e, hipMalloc(&tm_buf.p, max_elems * sizeof(ldp_pair_stats_t)));
  }
  std::vector<uint8_t> h_val;
  if (!hits) {
    HIP_TRY(e, hipMalloc(&val_buf.p, max_elems * esz));
  }
  for (const Chunk& ch : chunks) {
    const uint32_t r0 = ch.r0, rows = ch.rows, c0 = ch.c0, c1 = ch.c1;
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
}  // namespace ldph

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

}  // extern "C"
