// ldp_pgen.cpp -- from-scratch reader for the genotype main track of PLINK binary files: .bed (storage
// mode 0x01), fixed-width .pgen (0x02) and standard variable-width .pgen (0x10).  Host-side I/O edge of the
// --indep-pairwise path: it produces the 2-bit rows ldp_load_genotypes() consumes (what
// ReadGenovecSubsetUnsafe, 2.0/include/pgenlib_read.cc:2849-2912, produces before sample subsetting).
//
// Written from the format specification (pgen_spec/pgen_spec.tex:87-235 header, :321-467 difflists and main
// track record types) and checked against files the reference binary writes.  Only the main track is decoded;
// phase/dosage/multiallelic auxiliary tracks that may follow it inside a record are skipped.  Variable-width
// records inside one 65,536-variant block are decoded in order (LD-compressed records patch the most recent
// non-LD record of the same block, pgenlib_read.cc:1848 GetLdbaseVidx); blocks decode concurrently.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldprune_hip.h"
#include "ldp_env.h"

struct ldp_pgen {
  std::string err;
  std::mutex err_mutex;
  int fd = -1;
  const uint8_t* map = nullptr;
  uint64_t size = 0;
  int mode = 0;
  int file_mode = 0;            // the storage-mode byte of the file when it differs from `mode` (0x20: external index)
  int index_fd = -1;            // .pgen.pgi of the external-index modes
  const uint8_t* index_map = nullptr;
  uint64_t index_size = 0;
  uint32_t variant_ct = 0;
  uint32_t sample_ct = 0;
  uint64_t rec_bytes = 0;       // ceil(sample_ct / 4)
  uint64_t data_off = 0;        // fixed-width modes
  // variable-width index
  std::vector<uint8_t> vrtype;  // per variant
  std::vector<uint64_t> fpos;   // variant_ct + 1 record offsets
  bool any_multiallelic = false;
  bool any_dosage = false;        // some record carries a dosage track (skipped by this reader)
  int nonref_storage = 0;         // control bits 6-7: 0 see the .pvar, 1 every REF trusted, 2 every REF provisional, 3 nonref_bits
  std::vector<uint8_t> nonref_bits;  // storage 3: bit v = variant v's REF allele is provisional
};

namespace {

constexpr uint32_t kBlockVariants = 65536;
std::atomic<bool> g_force_portable(false);  // ldp_pgen_debug_force_portable(): the bit-deposit loops instead of pext / pdep (a test hook)

// (readers may call the per-record functions on one handle from several threads: the message is the first failure's)
int pfail(ldp_pgen* p, int code, const std::string& msg) {
  std::lock_guard<std::mutex> lock(p->err_mutex);
  if (p->err.empty()) {
    p->err = msg;
  }
  return code;
}

struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  uint32_t varint() {
    uint32_t v = 0;
    for (int shift = 0; shift < 35; shift += 7) {
      if (p >= end) {
        ok = false;
        return 0;
      }
      const uint8_t b = *p++;
      v |= static_cast<uint32_t>(b & 0x7f) << shift;
      if (!(b & 0x80)) {
        return v;
      }
    }
    ok = false;
    return 0;
  }
  bool skip(uint64_t n) {
    if (static_cast<uint64_t>(end - p) < n) {
      ok = false;
      return false;
    }
    p += n;
    return true;
  }
};

inline void set_code(uint8_t* row, uint32_t s, uint32_t code) {
  uint8_t& b = row[s >> 2];
  const uint32_t sh = 2 * (s & 3);
  b = static_cast<uint8_t>((b & ~(3u << sh)) | (code << sh));
}

// Difflist (pgen_spec.tex:367-430) of (sample id, 2-bit value) pairs applied onto `row`.
bool apply_difflist(Cursor& c, uint32_t sample_ct, uint8_t* row) {
  const uint32_t L = c.varint();
  if (!c.ok) {
    return false;
  }
  if (!L) {
    return true;
  }
  if (L > sample_ct) {
    return false;
  }
  const uint32_t G = (L + 63) / 64;
  const uint32_t idw = (sample_ct <= 256) ? 1 : ((sample_ct <= 65536) ? 2 : ((sample_ct <= 16777216) ? 3 : 4));
  const uint8_t* first_ids = c.p;
  if (!c.skip(static_cast<uint64_t>(G) * idw)) {
    return false;
  }
  if (!c.skip(G - 1)) {  // group byte sizes: only needed for random access
    return false;
  }
  const uint8_t* vals = c.p;
  if (!c.skip((L + 3) / 4)) {
    return false;
  }
  for (uint32_t g = 0; g < G; ++g) {
    uint32_t id = 0;
    memcpy(&id, first_ids + static_cast<uint64_t>(g) * idw, idw);
    const uint32_t kend = std::min(L, (g + 1) * 64);
    for (uint32_t k = g * 64; k < kend; ++k) {
      if (k != g * 64) {
        id += c.varint();
        if (!c.ok) {
          return false;
        }
      }
      if (id >= sample_ct) {
        return false;
      }
      set_code(row, id, (vals[k >> 2] >> (2 * (k & 3))) & 3);
    }
  }
  return true;
}

// 0 <-> 2 (GenovecInvertUnsafe semantics) over a whole row
inline void invert_row(uint8_t* row, uint64_t nbytes) {
  for (uint64_t b = 0; b < nbytes; ++b) {
    const uint8_t g = row[b];
    row[b] = static_cast<uint8_t>(g ^ (((~static_cast<uint32_t>(g)) << 1) & 0xaau));
  }
}

// Decode one main-track record into `row` (rec_bytes, trailing bits zero).  ldbase = most recent non-LD row.
bool decode_record(const ldp_pgen* P, uint32_t v, const uint8_t* ldbase, uint8_t* row, const uint8_t** main_track_end = nullptr) {
  const uint32_t type = P->vrtype[v] & 7;
  Cursor c{P->map + P->fpos[v], P->map + P->fpos[v + 1]};
  const uint64_t nb = P->rec_bytes;
  const uint32_t n = P->sample_ct;
  switch (type) {
    case 0:
      if (static_cast<uint64_t>(c.end - c.p) < nb) {
        return false;
      }
      memcpy(row, c.p, nb);
      break;
    case 1: {
      // one-bit representation: byte = low*4 + (high-low); bit set -> high category; then exceptions
      if (c.p >= c.end) {
        return false;
      }
      const uint32_t code = *c.p++;
      const uint32_t low = code >> 2;
      const uint32_t high = low + (code & 3);
      if (high > 3 || high == low) {
        return false;
      }
      const uint64_t bit_bytes = (static_cast<uint64_t>(n) + 7) / 8;
      if (static_cast<uint64_t>(c.end - c.p) < bit_bytes) {
        return false;
      }
      const uint8_t* bits = c.p;
      c.p += bit_bytes;
      // expand 4 samples (4 bits) -> one output byte
      uint8_t lut[16];
      for (uint32_t q = 0; q < 16; ++q) {
        uint32_t o = 0;
        for (uint32_t k = 0; k < 4; ++k) {
          o |= (((q >> k) & 1) ? high : low) << (2 * k);
        }
        lut[q] = static_cast<uint8_t>(o);
      }
      for (uint64_t b = 0; b < nb; ++b) {
        const uint8_t src = bits[b >> 1];
        row[b] = lut[(b & 1) ? (src >> 4) : (src & 15)];
      }
      if (!apply_difflist(c, n, row)) {
        return false;
      }
      break;
    }
    case 2:
    case 3:
      if (!ldbase) {
        return false;
      }
      memcpy(row, ldbase, nb);
      if (!apply_difflist(c, n, row)) {
        return false;
      }
      if (type == 3) {
        invert_row(row, nb);
      }
      break;
    case 4:
    case 6:
    case 7: {
      const uint32_t fill = (type == 4) ? 0 : ((type == 6) ? 2 : 3);
      memset(row, static_cast<int>(fill * 0x55), nb);
      if (!apply_difflist(c, n, row)) {
        return false;
      }
      break;
    }
    default:  // 5: reserved; the reference decodes it as all hom-REF (pgenlib_read.cc:2740-2742)
      memset(row, 0, nb);
      break;
  }
  const uint32_t rem = n & 3;
  if (rem) {
    row[nb - 1] &= static_cast<uint8_t>((1u << (2 * rem)) - 1);
  }
  if (main_track_end) {
    if (type == 0) {
      c.p += nb;
    }
    *main_track_end = c.p;
  }
  return true;
}

// Difflist that carries sample IDs only (no 2-bit values component), pgen_spec.tex:367-398.
bool read_id_difflist(Cursor& c, uint32_t sample_ct, std::vector<uint32_t>* ids) {
  ids->clear();
  const uint32_t L = c.varint();
  if (!c.ok || L > sample_ct) {
    return false;
  }
  if (!L) {
    return true;
  }
  const uint32_t G = (L + 63) / 64;
  const uint32_t idw = (sample_ct <= 256) ? 1 : ((sample_ct <= 65536) ? 2 : ((sample_ct <= 16777216) ? 3 : 4));
  const uint8_t* first_ids = c.p;
  if (!c.skip(static_cast<uint64_t>(G) * idw) || !c.skip(G - 1)) {
    return false;
  }
  ids->reserve(L);
  for (uint32_t g = 0; g < G; ++g) {
    uint32_t id = 0;
    memcpy(&id, first_ids + static_cast<uint64_t>(g) * idw, idw);
    const uint32_t kend = std::min(L, (g + 1) * 64);
    for (uint32_t k = g * 64; k < kend; ++k) {
      if (k != g * 64) {
        id += c.varint();
        if (!c.ok) {
          return false;
        }
      }
      if (id >= sample_ct) {
        return false;
      }
      ids->push_back(id);
    }
  }
  return true;
}

inline uint32_t packed_get(const uint8_t* base, uint64_t idx, uint32_t width_bits) {
  if (!width_bits) {
    return 0;
  }
  if (width_bits >= 8) {
    uint32_t v = 0;
    memcpy(&v, base + idx * (width_bits / 8), width_bits / 8);
    return v;
  }
  const uint64_t bit = idx * width_bits;
  return (base[bit >> 3] >> (bit & 7)) & ((1u << width_bits) - 1);
}

}  // namespace

extern "C" {

int ldp_pgen_debug_force_portable(int on) {
  g_force_portable.store(on != 0);
  return LDP_OK;
}

int ldp_pgen_open(const char* path, uint32_t sample_ct_hint, uint32_t variant_ct_hint, ldp_pgen** out) {
  return ldp_pgen_open_indexed(path, nullptr, sample_ct_hint, variant_ct_hint, out);
}

int ldp_pgen_open_indexed(const char* path, const char* pgi_path, uint32_t sample_ct_hint, uint32_t variant_ct_hint, ldp_pgen** out) {
  if (!path || !out) {
    return LDP_ERR_INVALID;
  }
  ldp_pgen* P = new (std::nothrow) ldp_pgen();
  if (!P) {
    return LDP_ERR_NOMEM;
  }
  *out = P;
  P->fd = open(path, O_RDONLY);
  if (P->fd < 0) {
    return pfail(P, LDP_ERR_INVALID, std::string("Failed to open ") + path + ".");
  }
  struct stat st;
  if (fstat(P->fd, &st) || st.st_size < 3) {
    return pfail(P, LDP_ERR_INVALID, std::string(path) + " is too small to be a PLINK genotype file.");
  }
  P->size = static_cast<uint64_t>(st.st_size);
  void* m = mmap(nullptr, P->size, PROT_READ, MAP_PRIVATE | (LDP_ENV("LDP_DEBUG_MAP_POPULATE") ? MAP_POPULATE : 0), P->fd, 0);
  if (m == MAP_FAILED) {
    return pfail(P, LDP_ERR_NOMEM, std::string("Failed to map ") + path + ".");
  }
  P->map = static_cast<const uint8_t*>(m);
  if (P->map[0] != 0x6c || P->map[1] != 0x1b) {
    return pfail(P, LDP_ERR_INVALID, std::string(path) + " is not a PLINK binary genotype file.");
  }
  P->mode = P->map[2];
  if (P->mode == 0x01) {
    // PLINK 1 variant-major .bed: dimensions come from .fam/.bim (pgenlib_read.cc:767-789)
    P->sample_ct = sample_ct_hint;
    P->variant_ct = variant_ct_hint;
    P->rec_bytes = (static_cast<uint64_t>(P->sample_ct) + 3) / 4;
    P->data_off = 3;
    if (P->size != 3 + P->rec_bytes * P->variant_ct) {
      return pfail(P, LDP_ERR_INVALID, "Unexpected .bed file size (expected " + std::to_string(3 + P->rec_bytes * P->variant_ct) + " bytes).");
    }
    return LDP_OK;
  }
  // The header: the file's own first bytes, or for the external-index modes the .pgen.pgi beside it (pgen_spec.tex:149-170:
  // "formatted just like a PGEN header", third byte 0x30; block offsets still point into the .pgen)
  const uint8_t* H = P->map;
  uint64_t Hsize = P->size;
  // 0x11 / 0x21 are 0x10 / 0x20 with "ignorable extensions" (pgen_spec.tex:237-270): bytes behind the header body and
  // possibly behind the last record.  Every record is located through the block offsets and record lengths, so ignoring
  // them takes no parsing at all.
  if (P->mode == 0x11) {
    P->file_mode = P->mode;
    P->mode = 0x10;
  }
  if ((P->mode & 0xfe) == 0x20) {
    const std::string ipath = (pgi_path && pgi_path[0]) ? std::string(pgi_path) : (std::string(path) + ".pgi");
    P->index_fd = open(ipath.c_str(), O_RDONLY);
    if (P->index_fd < 0) {
      return pfail(P, LDP_ERR_INVALID, "Failed to open " + ipath + ".");
    }
    struct stat ist;
    if (fstat(P->index_fd, &ist) || ist.st_size < 12) {
      return pfail(P, LDP_ERR_INVALID, ipath + " is too small to be a .pgen.pgi file.");
    }
    P->index_size = static_cast<uint64_t>(ist.st_size);
    void* im = mmap(nullptr, P->index_size, PROT_READ, MAP_PRIVATE, P->index_fd, 0);
    if (im == MAP_FAILED) {
      return pfail(P, LDP_ERR_NOMEM, "Failed to map " + ipath + ".");
    }
    P->index_map = static_cast<const uint8_t*>(im);
    if ((P->index_map[0] != 0x6c) || (P->index_map[1] != 0x1b) || ((P->index_map[2] & 0xfe) != 0x30)) {  // (either index flavour for either .pgen mode, as PgfiInitPhase1 accepts: pgenlib_read.cc:836)
      return pfail(P, LDP_ERR_INVALID, ipath + " is not a .pgen.pgi file (first three bytes don't match the magic number).");
    }
    H = P->index_map;
    Hsize = P->index_size;
    P->file_mode = P->mode;
    P->mode = 0x10;  // (from here on an ordinary variable-width file whose header lives elsewhere)
  }
  if (Hsize < 12) {
    return pfail(P, LDP_ERR_INVALID, std::string(path) + " is too small to be a .pgen file.");
  }
  memcpy(&P->variant_ct, H + 3, 4);
  memcpy(&P->sample_ct, H + 7, 4);
  if ((sample_ct_hint && sample_ct_hint != P->sample_ct) || (variant_ct_hint && variant_ct_hint != P->variant_ct)) {
    return pfail(P, LDP_ERR_INVALID, ".pgen header (" + std::to_string(P->variant_ct) + " variants, " + std::to_string(P->sample_ct) +
                                         " samples) does not match the variant/sample files.");
  }
  P->rec_bytes = (static_cast<uint64_t>(P->sample_ct) + 3) / 4;
  const uint8_t ctrl = H[11];
  const uint32_t nonref_storage = ctrl >> 6;
  P->nonref_storage = static_cast<int>(nonref_storage);
  if (nonref_storage == 3) {
    P->nonref_bits.assign((static_cast<size_t>(P->variant_ct) + 7) / 8, 0);
  }
  if (P->mode == 0x02) {
    if (ctrl & 63) {
      return pfail(P, LDP_ERR_INVALID, "fixed-width .pgen with a variable-width control byte.");
    }
    P->data_off = 12 + ((nonref_storage == 3) ? (static_cast<uint64_t>(P->variant_ct) + 7) / 8 : 0);
    if ((nonref_storage == 3) && (P->size >= P->data_off)) {
      memcpy(P->nonref_bits.data(), P->map + 12, P->nonref_bits.size());
    }
    if (P->size != P->data_off + P->rec_bytes * P->variant_ct) {
      return pfail(P, LDP_ERR_INVALID, "Unexpected .pgen file size (expected " + std::to_string(P->data_off + P->rec_bytes * P->variant_ct) + " bytes).");
    }
    return LDP_OK;
  }
  if ((P->mode == 0x03) || (P->mode == 0x04)) {
    // Fixed-width records with dosages (PgfiInitPhase1, pgenlib_read.cc:885-913): every record is the 2-bit hardcalls followed by
    // one 16-bit dosage per sample (record type 0x40, "unconditional dosage") and, in mode 4, one 16-bit phased-dosage
    // difference per sample (0xc0).  Represented here as a variable-width file whose records happen to have one length.
    if (ctrl & 63) {
      return pfail(P, LDP_ERR_INVALID, "fixed-width .pgen with a variable-width control byte.");
    }
    const uint64_t off = 12 + ((nonref_storage == 3) ? (static_cast<uint64_t>(P->variant_ct) + 7) / 8 : 0);
    const uint64_t width = P->rec_bytes + static_cast<uint64_t>(P->sample_ct) * ((P->mode == 0x03) ? 2 : 4);
    if ((nonref_storage == 3) && (P->size >= off)) {
      memcpy(P->nonref_bits.data(), P->map + 12, P->nonref_bits.size());
    }
    if (P->size != off + width * P->variant_ct) {
      return pfail(P, LDP_ERR_INVALID, "Unexpected .pgen file size (expected " + std::to_string(off + width * P->variant_ct) + " bytes).");
    }
    P->vrtype.assign(P->variant_ct, static_cast<uint8_t>((P->mode == 0x03) ? 0x40 : 0xc0));
    P->fpos.resize(static_cast<size_t>(P->variant_ct) + 1);
    for (uint64_t v = 0; v <= P->variant_ct; ++v) {
      P->fpos[v] = off + v * width;
    }
    P->any_dosage = (P->variant_ct != 0);
    P->file_mode = P->mode;
    P->mode = 0x10;
    return LDP_OK;
  }
  if (P->mode != 0x10) {
    char buf[200];
    snprintf(buf, sizeof(buf), ".pgen storage mode 0x%02x is not supported (supported: 0x01 .bed, 0x02-0x04 fixed-width, 0x10 / 0x11 standard, 0x20 / 0x21 standard with an external index).", P->mode);
    return pfail(P, LDP_ERR_UNSUPPORTED, buf);
  }
  // ---- standard variable-width header (pgen_spec.tex:160-235)
  const uint32_t tl = ctrl & 15;
  if (tl > 7) {
    return pfail(P, LDP_ERR_UNSUPPORTED, "reserved record-type/length storage (control bits 0-3 > 7).");
  }
  const uint32_t type_bits = (tl < 4) ? 4 : 8;
  const uint32_t len_bytes = (tl & 3) + 1;
  const uint32_t ac_bytes = (ctrl >> 4) & 3;
  const uint32_t M = P->variant_ct;
  const uint32_t B = (M + kBlockVariants - 1) / kBlockVariants;
  uint64_t pos = 12 + 8ull * B;
  if (pos > Hsize) {
    return pfail(P, LDP_ERR_INVALID, "truncated .pgen header.");
  }
  P->vrtype.resize(M);
  P->fpos.resize(static_cast<size_t>(M) + 1);
  for (uint32_t b = 0; b < B; ++b) {
    const uint32_t cnt = std::min(kBlockVariants, M - b * kBlockVariants);
    uint64_t block_off;
    memcpy(&block_off, H + 12 + 8ull * b, 8);
    const uint64_t types_bytes = (type_bits == 4) ? (cnt + 1) / 2 : cnt;
    const uint64_t need = types_bytes + static_cast<uint64_t>(cnt) * len_bytes + static_cast<uint64_t>(cnt) * ac_bytes +
                          ((nonref_storage == 3) ? (cnt + 7) / 8 : 0);
    if (pos + need > Hsize) {
      return pfail(P, LDP_ERR_INVALID, "truncated .pgen header.");
    }
    const uint8_t* types = H + pos;
    const uint8_t* lens = types + types_bytes;
    uint64_t rec = block_off;
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t v = b * kBlockVariants + k;
      const uint8_t t = (type_bits == 4) ? ((types[k >> 1] >> (4 * (k & 1))) & 15) : types[k];
      P->vrtype[v] = t;
      if (t & 8) {
        P->any_multiallelic = true;
      }
      if (t & 0x60) {
        P->any_dosage = true;  // (bits 5-6: a dosage track follows the hardcalls; bit 7, phased dosage, implies one)
      }
      uint32_t len = 0;
      memcpy(&len, lens + static_cast<uint64_t>(k) * len_bytes, len_bytes);
      P->fpos[v] = rec;
      rec += len;
    }
    if (rec > P->size) {
      return pfail(P, LDP_ERR_INVALID, "variant records run past the end of the .pgen file.");
    }
    if (b + 1 == B) {
      P->fpos[M] = rec;
    }
    if (nonref_storage == 3) {  // (the block's flags close its header section; 65,536 variants per block: byte-aligned)
      memcpy(P->nonref_bits.data() + static_cast<size_t>(b) * (kBlockVariants / 8), H + pos + need - (cnt + 7) / 8, (cnt + 7) / 8);
    }
    pos += need;
  }
  if (!M) {
    P->fpos[0] = (H == P->map) ? pos : 3;
  }
  return LDP_OK;
}

int ldp_pgen_provisional_ref(const ldp_pgen* P, uint8_t* bits, uint64_t bits_bytes) {
  if (!P) {
    return -1;
  }
  if (P->mode == 0x01) {
    return 2;  // a .bed has no notion of REF: every A2 allele is provisional (pgenlib_read.cc:790)
  }
  if ((P->nonref_storage == 3) && bits) {
    memcpy(bits, P->nonref_bits.data(), std::min<uint64_t>(bits_bytes, P->nonref_bits.size()));
  }
  return P->nonref_storage;
}

int ldp_pgen_info(const ldp_pgen* P, uint32_t* variant_ct, uint32_t* sample_ct, int* storage_mode, int* row_encoding, int* has_multiallelic) {
  if (!P) {
    return LDP_ERR_INVALID;
  }
  if (variant_ct) *variant_ct = P->variant_ct;
  if (sample_ct) *sample_ct = P->sample_ct;
  if (storage_mode) *storage_mode = P->mode;  // (0x10 for an external-index file as well: the records are the same)
  if (row_encoding) *row_encoding = (P->mode == 0x01) ? LDP_GENO_BED : LDP_GENO_REF;
  if (has_multiallelic) *has_multiallelic = P->any_multiallelic ? 1 : 0;
  return LDP_OK;
}

int ldp_pgen_has_dosage(const ldp_pgen* P) { return (P && P->any_dosage) ? 1 : 0; }

const void* ldp_pgen_file_bytes(const ldp_pgen* P, uint64_t* n_bytes) {
  if (!P || !P->map) {
    return nullptr;
  }
  if (n_bytes) {
    *n_bytes = P->size;
  }
  return P->map;
}

int ldp_pgen_record_index(const ldp_pgen* P, uint32_t first_variant, uint32_t n, ldp_pgen_rec* out, uint32_t* ld_base_variant) {
  if (!P || (n && !out) || (static_cast<uint64_t>(first_variant) + n > P->variant_ct)) {
    return LDP_ERR_INVALID;
  }
  if (ld_base_variant) {
    *ld_base_variant = UINT32_MAX;
  }
  if (P->mode == 0x01) {
    return LDP_ERR_UNSUPPORTED;
  }
  const bool fixed = (P->mode == 0x02);
  for (uint32_t q = 0; q < n; ++q) {
    const uint32_t v = first_variant + q;
    out[q].offset = fixed ? (P->data_off + static_cast<uint64_t>(v) * P->rec_bytes) : P->fpos[v];
    out[q].length = static_cast<uint32_t>(fixed ? P->rec_bytes : (P->fpos[v + 1] - P->fpos[v]));
    out[q].vrtype = fixed ? 0 : P->vrtype[v];
    out[q].allele_ct = 2;
    out[q].reserved = 0;
  }
  if (ld_base_variant && n && (!fixed) && ((P->vrtype[first_variant] & 6) == 2)) {
    const uint32_t blk_first = (first_variant / kBlockVariants) * kBlockVariants;
    uint32_t b = first_variant;
    while ((b > blk_first) && ((P->vrtype[b] & 6) == 2)) {
      --b;
    }
    if ((P->vrtype[b] & 6) == 2) {
      return LDP_ERR_INVALID;  // (a block that opens with an LD-compressed record)
    }
    *ld_base_variant = b;
  }
  return LDP_OK;
}

int ldp_pgen_direct_fd(const ldp_pgen* P, uint64_t* first_row_offset, uint64_t* stride_bytes) {
  if (!P || (P->mode != 0x01 && P->mode != 0x02) || (P->fd < 0)) {
    return -1;
  }
  if (first_row_offset) {
    *first_row_offset = P->data_off;
  }
  if (stride_bytes) {
    *stride_bytes = P->rec_bytes;
  }
  return P->fd;
}

const void* ldp_pgen_direct_rows(const ldp_pgen* P, uint64_t* stride_bytes) {
  if (!P || (P->mode != 0x01 && P->mode != 0x02)) {
    return nullptr;
  }
  if (stride_bytes) {
    *stride_bytes = P->rec_bytes;
  }
  return P->map + P->data_off;
}

}  // extern "C"

namespace {

// gather the bits of x selected by mask / scatter the low bits of x to the positions set in mask (pext / pdep);
// the BMI2 instructions when the host has them (decided once), a loop over the set bits otherwise
inline uint64_t pext64_loop(uint64_t x, uint64_t mask) {
  uint64_t out = 0;
  for (uint32_t k = 0; mask; mask &= mask - 1, ++k) {
    out |= ((x >> __builtin_ctzll(mask)) & 1) << k;
  }
  return out;
}
inline uint64_t pdep64_loop(uint64_t x, uint64_t mask) {
  uint64_t out = 0;
  for (uint32_t k = 0; mask; mask &= mask - 1, ++k) {
    out |= ((x >> k) & 1) << __builtin_ctzll(mask);
  }
  return out;
}
#if defined(__x86_64__)
// (out of line on purpose: only these two functions may contain BMI2 instructions)
__attribute__((target("bmi2"), noinline)) uint64_t pext64_hw(uint64_t x, uint64_t mask) { return __builtin_ia32_pext_di(x, mask); }
__attribute__((target("bmi2"), noinline)) uint64_t pdep64_hw(uint64_t x, uint64_t mask) { return __builtin_ia32_pdep_di(x, mask); }
#else
inline uint64_t pext64_hw(uint64_t x, uint64_t mask) { return pext64_loop(x, mask); }
inline uint64_t pdep64_hw(uint64_t x, uint64_t mask) { return pdep64_loop(x, mask); }
#endif

// up to 57 bits starting at bit `bit` of a byte stream that ends at `end` (bits past the end read as zero)
inline uint64_t read_bits(const uint8_t* base, const uint8_t* end, uint64_t bit, uint32_t count) {
  const uint8_t* p = base + (bit >> 3);
  uint64_t w = 0;
  if (p + 8 <= end) {
    memcpy(&w, p, 8);
  } else if (p < end) {
    memcpy(&w, p, static_cast<size_t>(end - p));
  }
  w >>= (bit & 7);
  return (count >= 64) ? w : (w & ((1ull << count) - 1));
}

template <bool HW>
bool decode_phase_impl(const ldp_pgen* P, uint32_t v, const uint8_t* row, const uint8_t* aux2, const uint8_t* sample_mask, uint8_t* phase, bool* unphased) {
  const uint32_t n = P->sample_ct;
  const uint64_t phase_bytes = (static_cast<uint64_t>(n) + 7) / 8;
  *unphased = false;
  const bool has_track = (P->vrtype[v] & 0x10) != 0;
  const uint8_t* end = P->map + P->fpos[v + 1];
  const uint64_t m5 = 0x5555555555555555ull;
  uint32_t het_ct = 0;
  if (has_track) {
    uint64_t b = 0;
    for (; b + 8 <= P->rec_bytes; b += 8) {
      uint64_t g;
      memcpy(&g, row + b, 8);
      het_ct += static_cast<uint32_t>(__builtin_popcountll(g & ~(g >> 1) & m5));
    }
    for (; b < P->rec_bytes; ++b) {
      const uint32_t x = row[b];
      het_ct += __builtin_popcount(x & ~(x >> 1) & 0x55u);
    }
    if (static_cast<uint64_t>(end - aux2) < 1 + het_ct / 8) {
      return false;
    }
  }
  const bool explicit_present = has_track && (aux2[0] & 1);
  const uint8_t* info = aux2;  // implicit: phaseinfo = bits 1..het_ct of the first part
  uint64_t info_bit = 1;
  uint64_t present_bit = 1;    // explicit: phasepresent = bits 1..het_ct of the first part
  if (explicit_present) {
    uint32_t present_ct = 0;
    for (uint32_t b = 0; b < 1 + het_ct / 8; ++b) {
      present_ct += __builtin_popcount(aux2[b]);
    }
    present_ct -= 1;
    info = aux2 + 1 + het_ct / 8;
    info_bit = 0;
    if ((!present_ct) || (static_cast<uint64_t>(end - info) < (present_ct + 7) / 8)) {
      return false;
    }
  }
  // 32 samples (one 64-bit word of codes) at a time
  bool any_unphased = false;
  for (uint32_t s0 = 0; s0 < n; s0 += 32) {
    const uint64_t byte0 = s0 >> 2;
    uint64_t g = 0;
    const uint64_t avail = P->rec_bytes - byte0;
    memcpy(&g, row + byte0, avail < 8 ? avail : 8);
    const uint64_t het64 = g & ~(g >> 1) & m5;
    const uint32_t out_bytes = static_cast<uint32_t>(std::min<uint64_t>(4, phase_bytes - (s0 >> 3)));
    uint32_t ph32 = 0;
    if (het64) {
      const uint32_t het32 = static_cast<uint32_t>(HW ? pext64_hw(het64, m5) : pext64_loop(het64, m5));
      const uint32_t k = static_cast<uint32_t>(__builtin_popcount(het32));
      uint32_t present32 = has_track ? het32 : 0;
      if (explicit_present) {
        const uint64_t pres = read_bits(aux2, end, present_bit, k);
        present_bit += k;
        present32 = static_cast<uint32_t>(HW ? pdep64_hw(pres, het32) : pdep64_loop(pres, het32));
      }
      const uint32_t kk = static_cast<uint32_t>(__builtin_popcount(present32));
      if (kk) {
        const uint64_t bits = read_bits(info, end, info_bit, kk);
        info_bit += kk;
        ph32 = static_cast<uint32_t>(HW ? pdep64_hw(bits, present32) : pdep64_loop(bits, present32));
      }
      uint32_t missing_phase = het32 & ~present32;
      if (missing_phase && sample_mask) {
        uint32_t mk = 0;
        memcpy(&mk, sample_mask + (s0 >> 3), out_bytes);
        missing_phase &= mk;
      }
      any_unphased |= (missing_phase != 0);
    }
    memcpy(phase + (s0 >> 3), &ph32, out_bytes);
  }
  *unphased = any_unphased;
  return true;
}

// Hardcall-phase track (pgen_spec.tex:541-562; parsed as ParseAux2Subset does, pgenlib_read.cc:6774-6836) of a record
// whose main track has just been decoded into `row`: writes the phaseinfo bit of every phased het call to `phase`
// (ceil(sample_ct/8) bytes).  Returns false on a malformed track; *unphased = some het call of a masked sample
// carries no phase.
bool decode_phase(const ldp_pgen* P, uint32_t v, const uint8_t* row, const uint8_t* aux2, const uint8_t* sample_mask, uint8_t* phase, bool* unphased) {
#if defined(__x86_64__)
  static const bool cpu_bmi2 = __builtin_cpu_supports("bmi2");
  const bool have_bmi2 = cpu_bmi2 && !g_force_portable.load(std::memory_order_relaxed);  // (ldp_pgen_debug_force_portable: tests of the portable path)
  if (have_bmi2) {
    return decode_phase_impl<true>(P, v, row, aux2, sample_mask, phase, unphased);
  }
#endif
  return decode_phase_impl<false>(P, v, row, aux2, sample_mask, phase, unphased);
}

int read_impl(ldp_pgen* P, uint32_t first_variant, uint32_t n, void* out_rows, uint64_t stride_bytes, uint32_t threads,
              bool phased, const uint8_t* sample_mask, uint32_t* unphased_variant) {
  if (!P || (n && !out_rows)) {
    return LDP_ERR_INVALID;
  }
  const uint64_t phase_off = (P->rec_bytes + 3) & ~static_cast<uint64_t>(3);
  const uint64_t need_bytes = phased ? (phase_off + (static_cast<uint64_t>(P->sample_ct) + 7) / 8) : P->rec_bytes;
  if ((static_cast<uint64_t>(first_variant) + n > P->variant_ct) || (stride_bytes < need_bytes)) {
    return pfail(P, LDP_ERR_INVALID, "variant range / stride out of bounds");
  }
  uint8_t* out = static_cast<uint8_t*>(out_rows);
  if (phased && (P->mode == 0x01 || P->mode == 0x02)) {
    // no phase track in the fixed-width modes: any het call is unphased
    std::atomic<uint32_t> lowest(UINT32_MAX);
    for (uint32_t k = 0; k < n; ++k) {
      uint8_t* row = out + k * stride_bytes;
      memcpy(row, P->map + P->data_off + (static_cast<uint64_t>(first_variant) + k) * P->rec_bytes, P->rec_bytes);
      memset(row + P->rec_bytes, 0, need_bytes - P->rec_bytes);
      if (P->mode == 0x01) {
        // .bed -> .pgen codes (PgrPlink1ToPlink2InplaceUnsafe, pgenlib_read.cc:2157): phased rows are always REF-coded
        static const uint8_t conv[4] = {2, 3, 1, 0};
        for (uint64_t b = 0; b < P->rec_bytes; ++b) {
          const uint32_t x = row[b];
          row[b] = static_cast<uint8_t>(conv[x & 3] | (conv[(x >> 2) & 3] << 2) | (conv[(x >> 4) & 3] << 4) | (conv[x >> 6] << 6));
        }
        const uint32_t rem = P->sample_ct & 3;
        if (rem) {
          row[P->rec_bytes - 1] &= static_cast<uint8_t>((1u << (2 * rem)) - 1);
        }
      }
      for (uint32_t s = 0; s < P->sample_ct; ++s) {
        if ((((row[s >> 2] >> (2 * (s & 3))) & 3) == 1) && ((!sample_mask) || ((sample_mask[s >> 3] >> (s & 7)) & 1))) {
          lowest.store(std::min(lowest.load(), first_variant + k));
          break;
        }
      }
    }
    if (lowest.load() != UINT32_MAX) {
      if (unphased_variant) {
        *unphased_variant = lowest.load();
      }
      return pfail(P, LDP_ERR_UNPHASED, "a heterozygous call has no phase");
    }
    return LDP_OK;
  }
  if (P->mode == 0x01 || P->mode == 0x02) {
    for (uint32_t k = 0; k < n; ++k) {
      memcpy(out + k * stride_bytes, P->map + P->data_off + (static_cast<uint64_t>(first_variant) + k) * P->rec_bytes, P->rec_bytes);
    }
    return LDP_OK;
  }
  if (!n) {
    return LDP_OK;
  }
  // Tasks of kTaskVariants consecutive variants (never across a 65,536-variant block: LD-compressed records patch the
  // latest non-LD record of their own block).  A task walks back to that base record first, so tasks are independent
  // and a call that spans only one or two blocks still keeps every host thread busy.
  // ~2 MiB of decoded rows per task: 256 variants at 50,000 samples, 16 at 500,000 (a 256 MiB chunk of the caller is
  // then still >100 tasks)
  const uint32_t kTaskVariants = static_cast<uint32_t>(std::min<uint64_t>(256, std::max<uint64_t>(16, (2ull << 20) / std::max<uint64_t>(P->rec_bytes, 1))));
  struct Task {
    uint32_t first, end;
  };
  std::vector<Task> tasks;
  for (uint32_t v = first_variant; v < first_variant + n;) {
    const uint32_t blk_end = std::min(P->variant_ct, (v / kBlockVariants + 1) * kBlockVariants);
    const uint32_t e = std::min({first_variant + n, blk_end, v + kTaskVariants});
    tasks.push_back({v, e});
    v = e;
  }
  std::atomic<uint32_t> next(0);
  std::atomic<int> bad(0);
  std::atomic<uint32_t> lowest_unphased(UINT32_MAX);
  auto worker = [&]() {
    std::vector<uint8_t> ldbase(P->rec_bytes), scratch(P->rec_bytes);
    for (uint32_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) {
      const uint32_t want_first = tasks[t].first;
      const uint32_t want_end = tasks[t].end;
      const uint32_t blk_first = (want_first / kBlockVariants) * kBlockVariants;
      // the LD base of the first wanted record: latest non-LD record at or before it
      uint32_t start = want_first;
      while (start > blk_first && ((P->vrtype[start] & 6) == 2)) {
        --start;
      }
      bool have_base = false;
      for (uint32_t v = start; v < want_end; ++v) {
        const bool is_ld = ((P->vrtype[v] & 6) == 2);
        if (v < want_first && is_ld) {
          continue;  // only the base row matters before the wanted range
        }
        uint8_t* dst = (v >= want_first) ? (out + static_cast<uint64_t>(v - first_variant) * stride_bytes) : scratch.data();
        const uint8_t* track_end = nullptr;
        if (!decode_record(P, v, have_base ? ldbase.data() : nullptr, dst, &track_end)) {
          bad.store(1);
          return;
        }
        if (phased && (v >= want_first)) {
          memset(dst + P->rec_bytes, 0, phase_off - P->rec_bytes);
          if (P->vrtype[v] & 8) {
            // multiallelic record: the phase track sits behind aux track 1 and refers to allele pairs, not to the
            // main track's codes -- ldp_pgen_read_alleles_phased() is the reader for these; here: codes only
            memset(dst + phase_off, 0, need_bytes - phase_off);
          }
          bool unphased = false;
          if ((!(P->vrtype[v] & 8)) && !decode_phase(P, v, dst, track_end, sample_mask, dst + phase_off, &unphased)) {
            bad.store(1);
            return;
          }
          if (unphased) {
            uint32_t cur = lowest_unphased.load();
            while ((v < cur) && !lowest_unphased.compare_exchange_weak(cur, v)) {
            }
          }
        }
        if (!is_ld) {
          memcpy(ldbase.data(), dst, P->rec_bytes);
          have_base = true;
        }
      }
    }
  };
  uint32_t nt = std::max(1u, std::min({threads ? threads : std::thread::hardware_concurrency(), static_cast<uint32_t>(tasks.size()), 64u}));
  if (nt == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < nt; ++t) {
      pool.emplace_back(worker);
    }
    for (std::thread& t : pool) {
      t.join();
    }
  }
  if (bad.load()) {
    return pfail(P, LDP_ERR_INVALID, "malformed variant record in .pgen file");
  }
  if (lowest_unphased.load() != UINT32_MAX) {
    if (unphased_variant) {
      *unphased_variant = lowest_unphased.load();
    }
    return pfail(P, LDP_ERR_UNPHASED, "a heterozygous call has no phase");
  }
  return LDP_OK;
}

}  // namespace

extern "C" {

int ldp_pgen_read(ldp_pgen* P, uint32_t first_variant, uint32_t n, void* out_rows, uint64_t stride_bytes, uint32_t threads) {
  return read_impl(P, first_variant, n, out_rows, stride_bytes, threads, false, nullptr, nullptr);
}

int ldp_pgen_read_phased(ldp_pgen* P, uint32_t first_variant, uint32_t n, void* out_rows, uint64_t stride_bytes,
                         const uint8_t* sample_mask, uint32_t threads, uint32_t* unphased_variant) {
  return read_impl(P, first_variant, n, out_rows, stride_bytes, threads, true, sample_mask, unphased_variant);
}

int ldp_pgen_variant_has_dosage(const ldp_pgen* P, uint32_t variant) {
  if (!P || variant >= P->variant_ct || P->mode != 0x10) {
    return 0;
  }
  return (P->vrtype[variant] & 0x60) ? 1 : 0;
}

// The two allele dosage sums of a biallelic variant over the samples of sample_mask, as GetBasicGenotypeCountsAndDosage16s
// computes them (pgenlib_read.cc:7917-8190): a sample with a dosage contributes it (16384 per ALT copy, 32768 - that to REF),
// any other sample its hardcall, a sample with neither nothing.
int ldp_pgen_dosage_sums(ldp_pgen* P, uint32_t variant, const uint8_t* sample_mask, uint64_t* ref_dosage, uint64_t* alt_dosage) {
  if (!P || !ref_dosage || !alt_dosage) {
    return LDP_ERR_INVALID;
  }
  if (variant >= P->variant_ct) {
    return pfail(P, LDP_ERR_INVALID, "variant index out of range");
  }
  const uint32_t n = P->sample_ct;
  std::vector<uint8_t> row(P->rec_bytes + 8, 0);
  const uint8_t* aux = nullptr;
  uint32_t vrtype = 0;
  if (P->mode == 0x01 || P->mode == 0x02) {
    const int rc = ldp_pgen_read(P, variant, 1, row.data(), P->rec_bytes, 1);
    if (rc) {
      return rc;
    }
    if (P->mode == 0x01) {  // .bed -> pgen codes
      static const uint8_t conv[4] = {2, 3, 1, 0};
      for (uint32_t s = 0; s < n; ++s) {
        set_code(row.data(), s, conv[(row[s >> 2] >> (2 * (s & 3))) & 3]);
      }
    }
  } else {
    vrtype = P->vrtype[variant];
    if (vrtype & 8) {
      return pfail(P, LDP_ERR_UNSUPPORTED, "dosage sums of a multiallelic record (the reference has none either: pgenlib_read.cc:8036)");
    }
    const uint32_t blk_first = (variant / kBlockVariants) * kBlockVariants;
    uint32_t start = variant;
    while (start > blk_first && ((P->vrtype[start] & 6) == 2)) {
      --start;
    }
    std::vector<uint8_t> base(P->rec_bytes + 8, 0);
    bool have_base = false;
    for (uint32_t v = start; v <= variant; ++v) {
      const bool is_ld = ((P->vrtype[v] & 6) == 2);
      if (v < variant && is_ld) {
        continue;
      }
      if (!decode_record(P, v, have_base ? base.data() : nullptr, row.data(), (v == variant) ? &aux : nullptr)) {
        return pfail(P, LDP_ERR_INVALID, "malformed variant record in .pgen file");
      }
      if (!is_ld) {
        memcpy(base.data(), row.data(), P->rec_bytes);
        have_base = true;
      }
    }
  }
  // 32 samples at a time: one 64-bit word of codes, the sample mask and the dosage-presence bits spread to the codes' even bit
  // positions, category counts as popcounts (a record is ~1 MB at 500,000 samples: this pass has to stream)
  const uint64_t m5 = 0x5555555555555555ull;
  const uint32_t nblk = (n + 31) / 32;
  const uint64_t nbits_bytes = (static_cast<uint64_t>(n) + 7) / 8;
  auto spread32 = [](uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & m5;
    return v;
  };
  auto valid32 = [&](uint32_t blk) { return (n - 32 * blk >= 32) ? 0xffffffffu : ((1u << (n - 32 * blk)) - 1u); };
  auto bits32 = [&](const uint8_t* bits, uint32_t blk) {  // bits == nullptr: everybody
    uint32_t w = 0xffffffffu;
    if (bits) {
      w = 0;
      memcpy(&w, bits + 4ull * blk, static_cast<size_t>(std::min<uint64_t>(4, nbits_bytes - 4ull * blk)));
    }
    return w & valid32(blk);
  };
  struct Cats {
    uint64_t homref, het, homalt, missing;
  };
  auto cats_of = [&](uint32_t blk) {
    uint64_t g = 0;
    memcpy(&g, row.data() + 8ull * blk, 8);  // (the buffer is rec_bytes + 8 long, zero behind the row)
    const uint64_t lo = g & m5, hi = (g >> 1) & m5;
    return Cats{~(lo | hi) & m5, lo & ~hi, hi & ~lo, lo & hi};
  };
  typedef uint16_t u16u __attribute__((aligned(1)));
  // hardcall counts of the subset; of the raw file (the phase track's length is a function of every sample's het calls)
  uint64_t geno[4] = {0, 0, 0, 0};
  uint32_t raw_het_ct = 0;
  for (uint32_t blk = 0; blk < nblk; ++blk) {
    const Cats c = cats_of(blk);
    const uint64_t V = spread32(valid32(blk)), M = spread32(bits32(sample_mask, blk));
    raw_het_ct += static_cast<uint32_t>(__builtin_popcountll(c.het & V));
    geno[0] += static_cast<uint64_t>(__builtin_popcountll(c.homref & M));
    geno[1] += static_cast<uint64_t>(__builtin_popcountll(c.het & M));
    geno[2] += static_cast<uint64_t>(__builtin_popcountll(c.homalt & M));
    geno[3] += static_cast<uint64_t>(__builtin_popcountll(c.missing & M));
  }
  uint64_t alt = 0, dosage_ct = 0;
  uint64_t replaced[4] = {0, 0, 0, 0};
  if (vrtype & 0x60) {
    Cursor c{aux, P->map + P->fpos[variant + 1]};
    if (vrtype & 0x10) {
      // the phase track (aux 2, pgen_spec.tex:541-562) sits in front: 1 + het_ct bits, then -- when bit 0 says the phase of some
      // het calls is absent -- one phaseinfo bit per het call that has one
      const uint64_t first = 1 + raw_het_ct / 8;
      if (static_cast<uint64_t>(c.end - c.p) < first) {
        return pfail(P, LDP_ERR_INVALID, "truncated phase track");
      }
      uint64_t skip = first;
      if (c.p[0] & 1) {
        uint32_t present = 0;
        for (uint64_t b = 0; b < first; ++b) {
          present += static_cast<uint32_t>(__builtin_popcount(c.p[b]));
        }
        skip += (present - 1 + 7) / 8;
      }
      if (!c.skip(skip)) {
        return pfail(P, LDP_ERR_INVALID, "truncated phase track");
      }
    }
    if ((vrtype & 0x60) == 0x40) {
      // one value per sample, 65535 = none (and then no hardcall either: pgen_spec.tex:601-604)
      const u16u* vals = reinterpret_cast<const u16u*>(c.p);
      if (!c.skip(2ull * n)) {
        return pfail(P, LDP_ERR_INVALID, "truncated dosage track");
      }
      for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t mw = bits32(sample_mask, blk), s0 = 32 * blk;
        if (mw == valid32(blk)) {
          const uint32_t cnt = std::min(32u, n - s0);
          uint32_t sum = 0, have = 0;
          for (uint32_t k = 0; k < cnt; ++k) {
            const uint32_t d = vals[s0 + k];
            sum += (d != 65535u) ? d : 0u;
            have += (d != 65535u) ? 1u : 0u;
          }
          alt += sum;
          dosage_ct += have;
        } else {
          for (uint32_t w = mw; w; w &= w - 1) {
            const uint32_t d = vals[s0 + static_cast<uint32_t>(__builtin_ctz(w))];
            if (d != 65535u) {
              alt += d;
              ++dosage_ct;
            }
          }
        }
      }
      // (every called sample has a dosage: the hardcalls are all replaced)
      for (int q = 0; q < 3; ++q) {
        replaced[q] = geno[q];
      }
    } else if ((vrtype & 0x60) == 0x60) {
      // track 3: one presence bit per sample (pgen_spec.tex:605-606); track 4: the values of the samples that have one, in order
      const uint8_t* bits = c.p;
      if (!c.skip(nbits_bytes)) {
        return pfail(P, LDP_ERR_INVALID, "truncated dosage track");
      }
      uint64_t total = 0;
      for (uint32_t blk = 0; blk < nblk; ++blk) {
        total += static_cast<uint64_t>(__builtin_popcount(bits32(bits, blk)));
      }
      const u16u* vals = reinterpret_cast<const u16u*>(c.p);
      if (!c.skip(2ull * total)) {
        return pfail(P, LDP_ERR_INVALID, "truncated dosage track");
      }
      uint64_t k = 0;
      for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t pw = bits32(bits, blk);
        if (!pw) {
          continue;
        }
        const uint32_t mw = bits32(sample_mask, blk), cnt = static_cast<uint32_t>(__builtin_popcount(pw));
        if (!(pw & ~mw)) {
          uint32_t sum = 0;
          for (uint32_t q = 0; q < cnt; ++q) {
            sum += vals[k + q];
          }
          alt += sum;
          dosage_ct += cnt;
        } else {
          uint32_t q = 0;
          for (uint32_t w = pw; w; w &= w - 1, ++q) {
            if ((mw >> __builtin_ctz(w)) & 1u) {
              alt += vals[k + q];
              ++dosage_ct;
            }
          }
        }
        k += cnt;
        const Cats cc = cats_of(blk);
        const uint64_t PM = spread32(pw & mw);
        replaced[0] += static_cast<uint64_t>(__builtin_popcountll(cc.homref & PM));
        replaced[1] += static_cast<uint64_t>(__builtin_popcountll(cc.het & PM));
        replaced[2] += static_cast<uint64_t>(__builtin_popcountll(cc.homalt & PM));
      }
    } else {
      // track 3: the ids of the samples that have a dosage (pgen_spec.tex:598-600)
      std::vector<uint32_t> ids;
      if (!read_id_difflist(c, n, &ids)) {
        return pfail(P, LDP_ERR_INVALID, "malformed dosage list");
      }
      const u16u* vals = reinterpret_cast<const u16u*>(c.p);
      if (!c.skip(2ull * ids.size())) {
        return pfail(P, LDP_ERR_INVALID, "truncated dosage track");
      }
      for (size_t q = 0; q < ids.size(); ++q) {
        const uint32_t sx = ids[q];
        if (sample_mask && !((sample_mask[sx >> 3] >> (sx & 7)) & 1)) {
          continue;
        }
        alt += vals[q];
        ++dosage_ct;
        ++replaced[(row[sx >> 2] >> (2 * (sx & 3))) & 3u];
      }
    }
  }
  const uint64_t replaced_ct = replaced[0] + replaced[1] + replaced[2];
  const uint64_t remaining_het = geno[1] - replaced[1], remaining_homalt = geno[2] - replaced[2];
  alt += (2 * remaining_homalt + remaining_het) * 16384ull;
  const uint64_t nondosage_nm = (geno[0] + geno[1] + geno[2]) - replaced_ct;
  *alt_dosage = alt;
  *ref_dosage = (dosage_ct + nondosage_nm) * 32768ull - alt;
  return LDP_OK;
}

int ldp_pgen_variant_is_multiallelic(const ldp_pgen* P, uint32_t variant) {
  if (!P || variant >= P->variant_ct || P->mode != 0x10) {
    return 0;
  }
  return (P->vrtype[variant] & 8) ? 1 : 0;
}

// Per-sample allele pairs of one variant (multiallelic hard-call track, pgen_spec.tex:469-540): allele_lo[s] <=
// allele_hi[s] are allele indices (0 = REF, k = ALTk), 255/255 = missing.  alt_ct = number of ALT alleles the
// variant has in the .pvar (<= 254 supported).
}  // extern "C"

namespace {
// *track_end (optional): first byte behind the main track and, if present, aux track 1 of the record
int read_alleles_impl(ldp_pgen* P, uint32_t variant, uint32_t alt_ct, uint8_t* allele_lo, uint8_t* allele_hi, const uint8_t** track_end) {
  if (!P || !allele_lo || !allele_hi) {
    return LDP_ERR_INVALID;
  }
  if (track_end) {
    *track_end = nullptr;
  }
  if (variant >= P->variant_ct || alt_ct < 1 || alt_ct > 254) {
    return pfail(P, LDP_ERR_INVALID, "variant index / ALT allele count out of range");
  }
  const uint32_t n = P->sample_ct;
  std::vector<uint8_t> row(P->rec_bytes + 8, 0);
  const uint8_t* aux = nullptr;
  bool multi = false;
  if (P->mode == 0x01 || P->mode == 0x02) {
    int rc = ldp_pgen_read(P, variant, 1, row.data(), P->rec_bytes, 1);
    if (rc) {
      return rc;
    }
    if (P->mode == 0x01) {  // .bed -> pgen codes
      static const uint8_t conv[4] = {2, 3, 1, 0};
      for (uint32_t s = 0; s < n; ++s) {
        const uint32_t code = (row[s >> 2] >> (2 * (s & 3))) & 3;
        set_code(row.data(), s, conv[code]);
      }
    }
  } else {
    // walk from the LD base like ldp_pgen_read does, keeping the end of the main track of `variant`
    const uint32_t blk_first = (variant / kBlockVariants) * kBlockVariants;
    uint32_t start = variant;
    while (start > blk_first && ((P->vrtype[start] & 6) == 2)) {
      --start;
    }
    std::vector<uint8_t> base(P->rec_bytes + 8, 0);
    bool have_base = false;
    for (uint32_t v = start; v <= variant; ++v) {
      const bool is_ld = ((P->vrtype[v] & 6) == 2);
      if (v < variant && is_ld) {
        continue;
      }
      if (!decode_record(P, v, have_base ? base.data() : nullptr, row.data(), (v == variant) ? &aux : nullptr)) {
        return pfail(P, LDP_ERR_INVALID, "malformed variant record in .pgen file");
      }
      if (!is_ld) {
        memcpy(base.data(), row.data(), P->rec_bytes);
        have_base = true;
      }
    }
    multi = (P->vrtype[variant] & 8) != 0;
  }
  // main track: 0 = REF/REF, 1 = REF/ALT1, 2 = ALT1/ALT1, 3 = missing
  std::vector<uint32_t> cat1, cat2;
  for (uint32_t s = 0; s < n; ++s) {
    const uint32_t code = (row[s >> 2] >> (2 * (s & 3))) & 3;
    switch (code) {
      case 0: allele_lo[s] = 0; allele_hi[s] = 0; break;
      case 1: allele_lo[s] = 0; allele_hi[s] = 1; cat1.push_back(s); break;
      case 2: allele_lo[s] = 1; allele_hi[s] = 1; cat2.push_back(s); break;
      default: allele_lo[s] = 255; allele_hi[s] = 255; break;
    }
  }
  if (!multi) {
    if (track_end) {
      *track_end = aux;
    }
    return LDP_OK;
  }
  if (alt_ct < 2) {
    return pfail(P, LDP_ERR_INVALID, "record carries multiallelic hard-calls but the variant has one ALT allele");
  }
  Cursor c{aux, P->map + P->fpos[variant + 1]};
  if (c.p >= c.end) {
    return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
  }
  const uint32_t fmt = *c.p++;
  const uint32_t fmt1 = fmt & 15, fmt2 = fmt >> 4;
  std::vector<uint32_t> ids;
  // ---- category 1 patch set: REF/ALTx with x >= 2
  if (fmt1 != 15) {
    std::vector<uint32_t> patched;  // sample ids
    if (fmt1 == 0) {
      const uint64_t nbytes = (cat1.size() + 7) / 8;
      const uint8_t* bits = c.p;
      if (!c.skip(nbytes)) {
        return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
      }
      for (size_t k = 0; k < cat1.size(); ++k) {
        if ((bits[k >> 3] >> (k & 7)) & 1) {
          patched.push_back(cat1[k]);
        }
      }
    } else if (fmt1 == 1) {
      if (!read_id_difflist(c, n, &patched)) {
        return pfail(P, LDP_ERR_INVALID, "malformed multiallelic difflist");
      }
    } else {
      return pfail(P, LDP_ERR_UNSUPPORTED, "reserved multiallelic patch format");
    }
    const uint32_t w = (alt_ct == 2) ? 0 : ((alt_ct == 3) ? 1 : ((alt_ct <= 5) ? 2 : ((alt_ct <= 17) ? 4 : 8)));
    const uint8_t* vals = c.p;
    if (!c.skip((patched.size() * w + 7) / 8)) {
      return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
    }
    for (size_t k = 0; k < patched.size(); ++k) {
      const uint32_t s = patched[k];
      if (s >= n || allele_hi[s] != 1 || allele_lo[s] != 0) {
        return pfail(P, LDP_ERR_INVALID, "multiallelic patch does not match the main track");
      }
      allele_hi[s] = static_cast<uint8_t>(2 + packed_get(vals, k, w));
    }
  }
  // ---- category 2 patch set: ALTx/ALTy other than ALT1/ALT1
  if (fmt2 != 15) {
    std::vector<uint32_t> patched;
    if (fmt2 == 0) {
      const uint64_t nbytes = (cat2.size() + 7) / 8;
      const uint8_t* bits = c.p;
      if (!c.skip(nbytes)) {
        return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
      }
      for (size_t k = 0; k < cat2.size(); ++k) {
        if ((bits[k >> 3] >> (k & 7)) & 1) {
          patched.push_back(cat2[k]);
        }
      }
    } else if (fmt2 == 1) {
      if (!read_id_difflist(c, n, &patched)) {
        return pfail(P, LDP_ERR_INVALID, "malformed multiallelic difflist");
      }
    } else {
      return pfail(P, LDP_ERR_UNSUPPORTED, "reserved multiallelic patch format");
    }
    if (alt_ct == 2) {
      const uint8_t* bits = c.p;
      if (!c.skip((patched.size() + 7) / 8)) {
        return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
      }
      for (size_t k = 0; k < patched.size(); ++k) {
        const uint32_t s = patched[k];
        if (s >= n || allele_lo[s] != 1 || allele_hi[s] != 1) {
          return pfail(P, LDP_ERR_INVALID, "multiallelic patch does not match the main track");
        }
        if ((bits[k >> 3] >> (k & 7)) & 1) {
          allele_lo[s] = 2;
          allele_hi[s] = 2;
        } else {
          allele_hi[s] = 2;
        }
      }
    } else {
      const uint32_t w = (alt_ct <= 4) ? 2 : ((alt_ct <= 16) ? 4 : 8);
      const uint8_t* vals = c.p;
      if (!c.skip((patched.size() * 2 * w + 7) / 8)) {
        return pfail(P, LDP_ERR_INVALID, "truncated multiallelic track");
      }
      for (size_t k = 0; k < patched.size(); ++k) {
        const uint32_t s = patched[k];
        if (s >= n || allele_lo[s] != 1 || allele_hi[s] != 1) {
          return pfail(P, LDP_ERR_INVALID, "multiallelic patch does not match the main track");
        }
        allele_lo[s] = static_cast<uint8_t>(1 + packed_get(vals, 2 * k, w));
        allele_hi[s] = static_cast<uint8_t>(1 + packed_get(vals, 2 * k + 1, w));
      }
    }
  }
  if (track_end) {
    *track_end = c.p;
  }
  return LDP_OK;
}
}  // namespace

extern "C" {

int ldp_pgen_read_alleles(ldp_pgen* P, uint32_t variant, uint32_t alt_ct, uint8_t* allele_lo, uint8_t* allele_hi) {
  return read_alleles_impl(P, variant, alt_ct, allele_lo, allele_hi, nullptr);
}

int ldp_pgen_read_alleles_phased(ldp_pgen* P, uint32_t variant, uint32_t alt_ct, uint8_t* allele_lo, uint8_t* allele_hi, uint8_t* phasepresent,
                                 uint8_t* phaseinfo) {
  if (!phasepresent || !phaseinfo) {
    return LDP_ERR_INVALID;
  }
  const uint8_t* aux2 = nullptr;
  const int rc = read_alleles_impl(P, variant, alt_ct, allele_lo, allele_hi, &aux2);
  if (rc) {
    return rc;
  }
  const uint32_t n = P->sample_ct;
  const uint64_t nbytes = (static_cast<uint64_t>(n) + 7) / 8;
  memset(phasepresent, 0, nbytes);
  memset(phaseinfo, 0, nbytes);
  if ((P->mode != 0x10) || !(P->vrtype[variant] & 0x10)) {
    return LDP_OK;  // no hardcall-phase track: nothing is phased
  }
  // pgen_spec.tex:541-562 over ALL heterozygous calls, multiallelic ones included (ReadGenovecHphaseSubsetUnsafe /
  // Get1Multiallelic: all_hets |= aux1b hets, pgenlib_read.cc:5497-5510)
  const uint8_t* end = P->map + P->fpos[variant + 1];
  uint32_t het_ct = 0;
  for (uint32_t s = 0; s < n; ++s) {
    het_ct += (allele_lo[s] != allele_hi[s]) ? 1 : 0;
  }
  if ((!aux2) || (static_cast<uint64_t>(end - aux2) < 1 + het_ct / 8)) {
    return pfail(P, LDP_ERR_INVALID, "truncated hardcall-phase track");
  }
  const bool explicit_present = aux2[0] & 1;
  const uint8_t* info = aux2;
  uint64_t info_bit = 1;
  if (explicit_present) {
    uint32_t present_ct = 0;
    for (uint32_t b = 0; b < 1 + het_ct / 8; ++b) {
      present_ct += __builtin_popcount(aux2[b]);
    }
    present_ct -= 1;
    info = aux2 + 1 + het_ct / 8;
    info_bit = 0;
    if ((!present_ct) || (static_cast<uint64_t>(end - info) < (present_ct + 7) / 8)) {
      return pfail(P, LDP_ERR_INVALID, "truncated hardcall-phase track");
    }
  }
  uint64_t het_idx = 0;
  for (uint32_t s = 0; s < n; ++s) {
    if (allele_lo[s] == allele_hi[s]) {
      continue;
    }
    bool present = true;
    if (explicit_present) {
      const uint64_t pb = 1 + het_idx;
      present = (aux2[pb >> 3] >> (pb & 7)) & 1;
    }
    ++het_idx;
    if (!present) {
      continue;
    }
    phasepresent[s >> 3] |= static_cast<uint8_t>(1u << (s & 7));
    if ((info[info_bit >> 3] >> (info_bit & 7)) & 1) {
      phaseinfo[s >> 3] |= static_cast<uint8_t>(1u << (s & 7));
    }
    ++info_bit;
  }
  return LDP_OK;
}

// CopyNyparrNonemptySubset (include/pgenlib_misc.cc:32,185) -- and CopyBitarrSubset for the phase bits of
// LDP_GENO_PHASED rows -- over a block of rows: per 64-bit input word one pext (bit-gather) with a mask prepared once,
// appended to the output bit stream; rows in parallel on host threads.
int ldp_subset_samples(const void* in_rows, uint64_t in_stride, uint32_t n_rows, uint32_t raw_sample_ct, const uint8_t* sample_mask,
                       void* out_rows, uint64_t out_stride, int phased, uint32_t threads) {
  if ((n_rows && (!in_rows || !out_rows)) || !sample_mask || !raw_sample_ct) {
    return LDP_ERR_INVALID;
  }
  const uint64_t m5 = 0x5555555555555555ull;
  uint32_t kept = 0;
  for (uint32_t s0 = 0; s0 < raw_sample_ct; ++s0) {
    kept += (sample_mask[s0 >> 3] >> (s0 & 7)) & 1;
  }
  const uint64_t in_code_bytes = (static_cast<uint64_t>(raw_sample_ct) + 3) / 4;
  const uint64_t out_code_bytes = (static_cast<uint64_t>(kept) + 3) / 4;
  const uint64_t in_phase_off = (in_code_bytes + 3) & ~static_cast<uint64_t>(3);
  const uint64_t out_phase_off = (out_code_bytes + 3) & ~static_cast<uint64_t>(3);
  const uint64_t in_phase_bytes = (static_cast<uint64_t>(raw_sample_ct) + 7) / 8;
  const uint64_t out_phase_bytes = (static_cast<uint64_t>(kept) + 7) / 8;
  const uint64_t in_need = phased ? (in_phase_off + in_phase_bytes) : in_code_bytes;
  const uint64_t out_need = phased ? (out_phase_off + out_phase_bytes) : out_code_bytes;
  if ((in_stride < in_need) || (out_stride < out_need)) {
    return LDP_ERR_INVALID;
  }
  // per input word of 32 codes: the 2-bit-expanded mask and the output bit offset; per word of 64 phase bits likewise
  const uint32_t code_words = static_cast<uint32_t>((in_code_bytes + 7) / 8);
  const uint32_t bit_words = static_cast<uint32_t>((in_phase_bytes + 7) / 8);
  std::vector<uint64_t> mask2(code_words, 0), mask1(bit_words, 0);
  std::vector<uint64_t> off2(code_words + 1, 0), off1(bit_words + 1, 0);
  for (uint32_t s0 = 0; s0 < raw_sample_ct; ++s0) {
    if ((sample_mask[s0 >> 3] >> (s0 & 7)) & 1) {
      mask2[s0 >> 5] |= 3ull << (2 * (s0 & 31));
      mask1[s0 >> 6] |= 1ull << (s0 & 63);
    }
  }
  for (uint32_t w = 0; w < code_words; ++w) {
    off2[w + 1] = off2[w] + static_cast<uint64_t>(__builtin_popcountll(mask2[w]));
  }
  for (uint32_t w = 0; w < bit_words; ++w) {
    off1[w + 1] = off1[w] + static_cast<uint64_t>(__builtin_popcountll(mask1[w]));
  }
  (void)m5;
#if defined(__x86_64__)
  static const bool cpu_bmi2 = __builtin_cpu_supports("bmi2");
  const bool have_bmi2 = cpu_bmi2 && !g_force_portable.load(std::memory_order_relaxed);
#else
  static const bool have_bmi2 = false;
#endif
  const uint8_t* in = static_cast<const uint8_t*>(in_rows);
  uint8_t* out = static_cast<uint8_t*>(out_rows);
  auto gather_bits = [&](const uint8_t* src, uint64_t src_bytes, const std::vector<uint64_t>& mask, const std::vector<uint64_t>& off, std::vector<uint64_t>& acc,
                         uint8_t* dst, uint64_t dst_bytes) {
    std::fill(acc.begin(), acc.end(), 0);
    const uint32_t words = static_cast<uint32_t>(mask.size());
    for (uint32_t w = 0; w < words; ++w) {
      if (!mask[w]) {
        continue;
      }
      uint64_t v = 0;
      const uint64_t byte0 = 8ull * w;
      memcpy(&v, src + byte0, std::min<uint64_t>(8, src_bytes - byte0));
      v = have_bmi2 ? pext64_hw(v, mask[w]) : pext64_loop(v, mask[w]);
      const uint64_t o = off[w];
      acc[o >> 6] |= v << (o & 63);
      if ((o & 63) && ((o & 63) + (off[w + 1] - o) > 64)) {
        acc[(o >> 6) + 1] |= v >> (64 - (o & 63));
      }
    }
    memcpy(dst, acc.data(), dst_bytes);
  };
  constexpr uint32_t kRowsPerTask = 64;
  const uint32_t tasks = (n_rows + kRowsPerTask - 1) / kRowsPerTask;
  std::atomic<uint32_t> next(0);
  auto worker = [&]() {
    std::vector<uint64_t> acc2((out_code_bytes + 7) / 8 + 2, 0), acc1((out_phase_bytes + 7) / 8 + 2, 0);
    for (uint32_t t = next.fetch_add(1); t < tasks; t = next.fetch_add(1)) {
      const uint32_t r1 = std::min(n_rows, (t + 1) * kRowsPerTask);
      for (uint32_t r = t * kRowsPerTask; r < r1; ++r) {
        const uint8_t* src = in + static_cast<uint64_t>(r) * in_stride;
        uint8_t* dst = out + static_cast<uint64_t>(r) * out_stride;
        gather_bits(src, in_code_bytes, mask2, off2, acc2, dst, out_code_bytes);
        if (phased) {
          memset(dst + out_code_bytes, 0, out_phase_off - out_code_bytes);
          gather_bits(src + in_phase_off, in_phase_bytes, mask1, off1, acc1, dst + out_phase_off, out_phase_bytes);
        }
      }
    }
  };
  const uint32_t nt = std::max(1u, std::min({threads ? threads : std::thread::hardware_concurrency(), tasks, 64u}));
  if (nt == 1) {
    worker();
  } else {
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < nt; ++t) {
      pool.emplace_back(worker);
    }
    for (std::thread& th : pool) {
      th.join();
    }
  }
  return LDP_OK;
}

const char* ldp_pgen_last_error(const ldp_pgen* P) { return P ? P->err.c_str() : "null reader"; }

void ldp_pgen_close(ldp_pgen* P) {
  if (!P) {
    return;
  }
  if (P->map) {
    munmap(const_cast<uint8_t*>(P->map), P->size);
  }
  if (P->fd >= 0) {
    close(P->fd);
  }
  if (P->index_map) {
    munmap(const_cast<uint8_t*>(P->index_map), P->index_size);
  }
  if (P->index_fd >= 0) {
    close(P->index_fd);
  }
  delete P;
}

}  // extern "C"
