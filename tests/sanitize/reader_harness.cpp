#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ldprune_hip.h"
extern "C" int ldp_pgen_debug_force_portable(int on);

int main(int argc, char** argv) {
  if (getenv("LDTEST_PGEN_PORTABLE")) {  // (the harness is test infrastructure: the library itself reads no environment)
    ldp_pgen_debug_force_portable(1);
  }
  for (int a = 1; a + 1 < argc; a += 2) {
    ldp_pgen* pg = nullptr;
    if (ldp_pgen_open(argv[a], 0, 0, &pg)) {
      printf("open failed %s: %s\n", argv[a], ldp_pgen_last_error(pg));
      ldp_pgen_close(pg);  // (a failed open keeps the handle alive for the message)
      return 1;
    }
    const uint32_t alt_max = atoi(argv[a + 1]);
    uint32_t m, n; int mode, enc, multi;
    ldp_pgen_info(pg, &m, &n, &mode, &enc, &multi);
    const uint64_t rec = (n + 3) / 4;
    std::vector<uint8_t> rows((size_t)m * rec);
    for (uint32_t th : {1u, 0u}) {
      int rc = ldp_pgen_read(pg, 0, m, rows.data(), rec, th);
      if (rc) { printf("read rc %d\n", rc); }
    }
    // the record index ldp_load_pgen_records is fed from: whole file, ragged pieces, every record inside the mapping
    {
      uint64_t nbytes = 0;
      const uint8_t* bytes = static_cast<const uint8_t*>(ldp_pgen_file_bytes(pg, &nbytes));
      std::vector<ldp_pgen_rec> idx(m);
      uint32_t base = 0;
      if (ldp_pgen_record_index(pg, 0, m, idx.data(), &base) == 0) {
        uint64_t sum = 0;
        for (uint32_t v = 0; v < m; ++v) {
          if (idx[v].offset + idx[v].length > nbytes) { printf("record %u outside the file\n", v); return 1; }
          sum += bytes[idx[v].offset] + bytes[idx[v].offset + idx[v].length - 1];
        }
        for (uint32_t v = 0; v < m; v += 5) {
          ldp_pgen_rec one_rec;
          ldp_pgen_record_index(pg, v, 1, &one_rec, &base);
          if ((base != 0xffffffffu) && (base >= v)) { printf("LD base %u of %u\n", base, v); return 1; }
        }
        (void)sum;
      }
    }
    // ragged single reads
    std::vector<uint8_t> one(rec);
    for (uint32_t v = 0; v < m; v += 7) ldp_pgen_read(pg, v, 1, one.data(), rec, 1);
    const uint64_t prow = ldp_phased_row_bytes(2 * n);
    std::vector<uint8_t> prows((size_t)m * prow);
    std::vector<uint8_t> mask((n + 7) / 8, 0xff);
    uint32_t bad = 0;
    int rc = ldp_pgen_read_phased(pg, 0, m, prows.data(), prow, mask.data(), 0, &bad);
    printf("%s: m=%u n=%u mode=%d multi=%d read_phased rc=%d bad=%u\n", argv[a], m, n, mode, multi, rc, bad);
    for (uint32_t v = 0; v < m; v += 3) ldp_pgen_read_phased(pg, v, 1, prows.data(), prow, nullptr, 1, &bad);
    std::vector<uint8_t> lo(n), hi(n), pp((n + 7) / 8), pi((n + 7) / 8);
    for (uint32_t v = 0; v < m; ++v) {
      for (uint32_t alts = 1; alts <= alt_max; ++alts) {
        ldp_pgen_read_alleles_phased(pg, v, alts, lo.data(), hi.data(), pp.data(), pi.data());
      }
    }
    // dosage tracks: sums over everybody and over a subset
    {
      std::vector<uint8_t> third((n + 7) / 8, 0);
      for (uint32_t s = 0; s < n; s += 3) third[s >> 3] |= 1u << (s & 7);
      uint64_t rd = 0, ad = 0, with_track = 0, failed = 0;
      for (uint32_t v = 0; v < m; ++v) {
        with_track += ldp_pgen_variant_has_dosage(pg, v);
        failed += ldp_pgen_dosage_sums(pg, v, nullptr, &rd, &ad) != 0;
        failed += ldp_pgen_dosage_sums(pg, v, third.data(), &rd, &ad) != 0;
      }
      printf("dosage: file %d, records with a track %llu, failed calls %llu\n", ldp_pgen_has_dosage(pg), (unsigned long long)with_track, (unsigned long long)failed);
    }
    // subset
    std::vector<uint8_t> keep((n + 7) / 8, 0);
    uint32_t kept = 0;
    for (uint32_t s = 0; s < n; ++s) if (s % 3) { keep[s >> 3] |= 1u << (s & 7); ++kept; }
    const uint64_t orow = ldp_phased_row_bytes(2 * kept);
    std::vector<uint8_t> out((size_t)m * orow);
    rc = ldp_subset_samples(prows.data(), prow, m, n, keep.data(), out.data(), orow, 1, 0);
    std::vector<uint8_t> out2((size_t)m * ((kept + 3) / 4));
    rc |= ldp_subset_samples(rows.data(), rec, m, n, keep.data(), out2.data(), (kept + 3) / 4, 0, 1);
    printf("subset rc %d\n", rc);
    ldp_pgen_close(pg);
  }
  return 0;
}
