#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "ldprune_hip_debug.h"
int main() {
  std::mt19937 rng(7);
  for (int iter = 0; iter < 60; ++iter) {
    const uint32_t m = 50 + rng() % 3000;
    const bool is_bp = rng() & 1;
    ldp_params P; memset(&P, 0, sizeof(P));
    P.founder_ct = 50 + rng() % 1000;
    P.window_is_bp = is_bp;
    P.prune_window_size = is_bp ? (500 + rng() % 50000) : (2 + rng() % 300);
    P.prune_window_incr = is_bp ? 1 : (1 + rng() % P.prune_window_size);
    P.prune_last_param = 0.5; P.plink1_order = rng() & 1; P.device = -1;
    ldp_engine* e = nullptr;
    if (ldp_create(&P, &e)) { printf("create failed\n"); return 1; }
    std::vector<uint32_t> chr(m), bp(m);
    uint32_t c = 0, pos = 1;
    for (uint32_t v = 0; v < m; ++v) { if (rng() % 400 == 0) { ++c; pos = 1; } pos += 1 + rng() % 300 + ((rng() % 100 == 0) ? 100000 : 0); chr[v] = c; bp[v] = pos; }
    int rc;
    const int mode = iter % 3;
    if (mode == 0) rc = ldp_set_variants(e, m, chr.data(), is_bp ? bp.data() : nullptr);
    else if (mode == 1) rc = ldp_set_variants_vcor(e, m, chr.data(), bp.data(), 1000 + rng() % 100000, 1 + rng() % 500);
    else rc = ldp_set_variants_matrix(e, m);
    if (rc) { printf("plan rc %d: %s\n", rc, ldp_last_error(e)); return 1; }
    std::vector<uint32_t> lo(m); uint64_t cand = 0;
    ldp_get_band(e, lo.data(), &cand);
    uint32_t sct = 0; ldp_get_subcontigs(e, &sct, nullptr, 0);
    if (mode == 0 && sct > 1 && (iter & 1)) { std::vector<uint32_t> owner(sct); ldp_set_shard(e, rng() % 3, 3, owner.data()); }
    if (mode == 0) {
      // replay over random true pairs inside the band
      std::vector<ldp_variant_rec> recs(m);
      memset(recs.data(), 0, recs.size() * sizeof(ldp_variant_rec));
      for (uint32_t v = 0; v < m; ++v) { recs[v].nm_ct = P.founder_ct; recs[v].ssq = 10; recs[v].sum = 1; }
      if (ldp_debug_set_variant_recs(e, recs.data()) == 0) {
        std::vector<double> mf(m); for (auto& x : mf) x = 0.5 + (rng() % 1000) / 2001.0;
        ldp_set_maj_freqs(e, 0, m, mf.data());
        std::vector<uint32_t> f, s2;
        for (uint32_t j = 1; j < m; ++j) if (lo[j] < j && rng() % 3 == 0) { f.push_back(lo[j] + rng() % (j - lo[j])); s2.push_back(j); }
        std::vector<uint64_t> removed((m + 63) / 64 + 1);
        ldp_debug_replay_pairs(e, f.size(), f.data(), s2.data(), removed.data());
      }
    }
    ldp_destroy(e);
  }
  printf("engine host logic: done\n");
  return 0;
}
