// ldp_engine_shard.cpp -- several engines: the LPT shard of the subcontigs and the one exchange of removed bits (RCCL all-gather or pack / stitch)
// (host runtime behind include/ldprune_hip.h; ldp_engine.cpp has the overview)
#include "ldp_engine_internal.h"

extern "C" {

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
namespace ldph LDP_HIDDEN {
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
}  // namespace ldph

namespace ldph LDP_HIDDEN {
// communicators ldp_allgather_removed() had to abort: ncclCommAbort has already released them, a later ldp_comm_destroy() is a no-op
std::mutex g_aborted_mu;
std::set<void*> g_aborted;
}  // namespace ldph

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
  {
    // (a fresh communicator may live at the address of one that was aborted and never passed to ldp_comm_destroy)
    std::lock_guard<std::mutex> lk(g_aborted_mu);
    for (int k = 0; k < n; ++k) {
      g_aborted.erase(c[k]);
    }
  }
  for (int k = 0; k < n; ++k) {
    comms[k] = c[k];
  }
  return LDP_OK;
}

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
namespace ldph LDP_HIDDEN {
uint64_t shard_segment_words(const ldp_engine* e) {
  std::vector<uint64_t> seg_bits(std::max<uint32_t>(e->world, 1), 0);
  for (const Subcontig& s : e->subs) {
    seg_bits[s.owner] += s.len;
  }
  return std::max<uint64_t>((*std::max_element(seg_bits.begin(), seg_bits.end()) + 63) / 64, 1);
}
bool shard_ready(const ldp_engine* e) { return e && e->planned && !e->matrix_mode && !e->band_r2_mode; }
}  // namespace ldph

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
  // A communicator an earlier failed call aborted is gone (ncclCommAbort freed it): nothing may touch the handle again -- not even
  // ncclCommCount -- so the list of aborted handles is consulted before anything else
  if (c) {
    std::lock_guard<std::mutex> lk(g_aborted_mu);
    if (g_aborted.count(c)) {
      return fail(e, LDP_ERR_STATE, "this communicator was aborted by an earlier failed ldp_allgather_removed(): create a new one");
    }
  }
  // (only a handle RCCL itself has answered for is aborted: ncclCommCount on it comes first, before any other reason to leave)
  int count = 0, urank = -1;
  const bool handle_ok = c && R.ok && (R.CommCount(c, &count) == ncclSuccess) && (R.CommUserRank(c, &urank) == ncclSuccess);
  auto leave = [&](int code, const std::string& msg) {
    if (handle_ok && R.CommAbort) {
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
  if (!handle_ok) {
    return fail(e, LDP_ERR_GPU, "ncclCommCount / ncclCommUserRank failed: not a live communicator");
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
  // (from here on the collective is in flight: the buffers it reads and writes are released only behind a synchronised stream, or --
  // when that fails too -- behind an aborted communicator)
  std::vector<uint64_t> all(words * e->world);
  hrc = hipMemcpyAsync(all.data(), recv.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream);
  const hipError_t src = hipStreamSynchronize(e->stream);
  if ((hrc != hipSuccess) || (src != hipSuccess)) {
    if (src != hipSuccess) {
      return leave(LDP_ERR_GPU, std::string("all-gather of the removed bits: ") + hipGetErrorString(src));
    }
    return fail(e, LDP_ERR_GPU, std::string("copy of the gathered segments: ") + hipGetErrorString(hrc));
  }
  return ldp_stitch_removed_segments(e, all.data(), removed_global);
}

}  // extern "C"
