"""Test-side helpers: ctypes binding of the CPU oracle (oracle/ldoracle.c), genotype packing,
PLINK file writers and a runner for the reference binary (oracle/_ref/plink2, when present).

TEST INFRASTRUCTURE ONLY -- the product (plink-ng_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libldoracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "plink2")

K_SMALL_EPSILON = 2.0 ** -44


class LdoVaggs(ctypes.Structure):
    _fields_ = [("nm_ct", ctypes.c_uint32), ("sum", ctypes.c_int32), ("ssq", ctypes.c_uint32),
                ("plusone_ct", ctypes.c_uint32), ("minusone_ct", ctypes.c_uint32)]


class LdoPairStats(ctypes.Structure):
    _fields_ = [("nm", ctypes.c_uint32), ("sum1", ctypes.c_int32), ("ssq1", ctypes.c_uint32),
                ("sum2", ctypes.c_int32), ("ssq2", ctypes.c_uint32), ("dot", ctypes.c_int32)]

    def astuple(self):
        return (self.nm, self.sum1, self.ssq1, self.sum2, self.ssq2, self.dot)


class LdoVhaggs(ctypes.Structure):
    _fields_ = [("nm_ct", ctypes.c_uint32), ("sum", ctypes.c_uint32)]


class LdoHapPairStats(ctypes.Structure):
    _fields_ = [("nm", ctypes.c_uint32), ("sum1", ctypes.c_uint32), ("sum2", ctypes.c_uint32), ("dot", ctypes.c_uint32)]

    def astuple(self):
        return (self.nm, self.sum1, self.sum2, self.dot)


def build_oracle():
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("ldoracle.c", "ldoracle.h")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = ctypes.CDLL(build_oracle())
        u64p = ctypes.POINTER(ctypes.c_uint64)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.ldo_split_hom_ref2het.argtypes = [u64p, ctypes.c_uint32, u64p, u64p]
        lib.ldo_fill_vaggs.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.POINTER(LdoVaggs)]
        lib.ldo_is_monomorphic.argtypes = [ctypes.POINTER(LdoVaggs)]
        lib.ldo_pair_stats.argtypes = [u64p, u64p, ctypes.POINTER(LdoVaggs), u64p, u64p, ctypes.POINTER(LdoVaggs),
                                       ctypes.c_uint32, ctypes.POINTER(LdoPairStats)]
        lib.ldo_exceeds.argtypes = [ctypes.POINTER(LdoPairStats), ctypes.c_double]
        lib.ldo_cov_vars.argtypes = [ctypes.POINTER(LdoPairStats), f64p, f64p, f64p]
        lib.ldo_prune_thresh.argtypes = [ctypes.c_double]
        lib.ldo_prune_thresh.restype = ctypes.c_double
        lib.ldo_major_allele.argtypes = [u64p, ctypes.c_uint32, u32p, f64p]
        lib.ldo_invert_geno.argtypes = [u64p, ctypes.c_uint32, u64p]
        lib.ldo_subcontig_split.argtypes = [u32p, u32p, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p]
        lib.ldo_subcontig_split.restype = ctypes.c_uint32
        lib.ldo_indep_pairwise.argtypes = [u64p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, f64p,
                                           ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_double,
                                           ctypes.c_int, u64p, u64p]
        lib.ldo_indep_pairwise.restype = ctypes.c_int
        lib.ldo_hapsplit_must_phased.argtypes = [u64p, u64p, u64p, ctypes.c_uint32, u64p, u64p]
        lib.ldo_hapsplit_must_phased.restype = ctypes.c_int
        lib.ldo_hapsplit_haploid.argtypes = [u64p, ctypes.c_uint32, u64p, u64p]
        lib.ldo_fill_vhaggs.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.POINTER(LdoVhaggs)]
        lib.ldo_fill_vhaggs.restype = ctypes.c_int
        lib.ldo_hap_pair_stats.argtypes = [u64p, u64p, ctypes.POINTER(LdoVhaggs), u64p, u64p, ctypes.POINTER(LdoVhaggs),
                                           ctypes.c_uint32, ctypes.POINTER(LdoHapPairStats)]
        lib.ldo_hap_exceeds.argtypes = [ctypes.POINTER(LdoHapPairStats), ctypes.c_double]
        lib.ldo_indep_pairphase.argtypes = lib.ldo_indep_pairwise.argtypes
        lib.ldo_indep_pairphase.restype = ctypes.c_int
        _oracle = lib
    return _oracle


def _p(arr, ctype):
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


# ---------------------------------------------------------------- genotype packing
def pack_2bit(codes):
    """codes: (M, N) uint8 array of 2-bit codes -> (M, stride_words) uint64, 32 genotypes per word,
    little-endian bit order (sample s -> bits 2*(s%32) of word s//32).  Trailing bits are zero."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    m, n = codes.shape
    stride_words = (n + 31) // 32
    padded = np.zeros((m, stride_words * 32), dtype=np.uint8)
    padded[:, :n] = codes & 3
    quads = padded.reshape(m, stride_words * 8, 4)
    bytes_ = (quads[:, :, 0] | (quads[:, :, 1] << 2) | (quads[:, :, 2] << 4) | (quads[:, :, 3] << 6)).astype(np.uint8)
    return np.ascontiguousarray(bytes_).view(np.uint64).reshape(m, stride_words)


def unpack_2bit(words, n):
    m = words.shape[0]
    b = np.ascontiguousarray(words).view(np.uint8).reshape(m, -1)
    out = np.empty((m, b.shape[1] * 4), dtype=np.uint8)
    for k in range(4):
        out[:, k::4] = (b >> (2 * k)) & 3
    return out[:, :n]


def bitmap_to_bool(bm, n):
    return np.unpackbits(np.ascontiguousarray(bm).view(np.uint8), bitorder="little")[:n].astype(bool)


# ---------------------------------------------------------------- oracle wrappers
def oracle_prepare(raw_codes):
    """raw REF-based codes (0 hom-REF,1 het,2 hom-ALT,3 missing), (M,N) -> (inv packed words, maj_freq, alt_major)"""
    lib = oracle()
    raw = pack_2bit(raw_codes)
    m, n = raw_codes.shape
    inv = np.empty_like(raw)
    mf = np.empty(m, dtype=np.float64)
    altmaj = np.empty(m, dtype=np.uint32)
    for v in range(m):
        am = ctypes.c_uint32()
        f = ctypes.c_double()
        lib.ldo_major_allele(_p(raw[v], ctypes.c_uint64), n, ctypes.byref(am), ctypes.byref(f))
        mf[v] = f.value
        altmaj[v] = am.value
        if am.value:
            lib.ldo_invert_geno(_p(raw[v], ctypes.c_uint64), n, _p(inv[v], ctypes.c_uint64))
        else:
            inv[v] = raw[v]
    return inv, mf, altmaj


def oracle_split(inv_words, n):
    """(M, stride) packed inverse codes -> hom, r2h planes (M, ceil(n/64)) uint64 + list of vaggs"""
    lib = oracle()
    m = inv_words.shape[0]
    wc = (n + 63) // 64
    gw = (n + 31) // 32
    hom = np.zeros((m, wc), dtype=np.uint64)
    r2h = np.zeros((m, wc), dtype=np.uint64)
    vaggs = (LdoVaggs * m)()
    for v in range(m):
        row = np.zeros(gw + 1, dtype=np.uint64)
        row[:gw] = inv_words[v, :gw]
        lib.ldo_split_hom_ref2het(_p(row, ctypes.c_uint64), n, _p(hom[v], ctypes.c_uint64), _p(r2h[v], ctypes.c_uint64))
        lib.ldo_fill_vaggs(_p(hom[v], ctypes.c_uint64), _p(r2h[v], ctypes.c_uint64), wc, ctypes.byref(vaggs[v]))
    return hom, r2h, vaggs


def oracle_pair_stats(hom, r2h, vaggs, n, first, second):
    lib = oracle()
    st = LdoPairStats()
    lib.ldo_pair_stats(_p(hom[first], ctypes.c_uint64), _p(r2h[first], ctypes.c_uint64), ctypes.byref(vaggs[first]),
                       _p(hom[second], ctypes.c_uint64), _p(r2h[second], ctypes.c_uint64), ctypes.byref(vaggs[second]),
                       n, ctypes.byref(st))
    return st


def oracle_r2(st):
    lib = oracle()
    c, v1, v2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.ldo_cov_vars(ctypes.byref(st), ctypes.byref(c), ctypes.byref(v1), ctypes.byref(v2))
    return c.value, v1.value, v2.value


def oracle_indep_pairwise(inv_words, n, chr_idx, bps, maj_freqs, window, step, is_bp, r2, order=2):
    """Returns (removed bool array, pair evaluation count)."""
    lib = oracle()
    m = inv_words.shape[0]
    inv_words = np.ascontiguousarray(inv_words, dtype=np.uint64)
    chr_idx = np.ascontiguousarray(chr_idx, dtype=np.uint32)
    bps = np.ascontiguousarray(bps, dtype=np.uint32)
    maj_freqs = np.ascontiguousarray(maj_freqs, dtype=np.float64)
    removed = np.zeros((m + 63) // 64 + 1, dtype=np.uint64)
    evals = ctypes.c_uint64()
    rc = lib.ldo_indep_pairwise(_p(inv_words, ctypes.c_uint64), inv_words.shape[1], m, n, _p(chr_idx, ctypes.c_uint32),
                                _p(bps, ctypes.c_uint32), _p(maj_freqs, ctypes.c_double), window, step, int(is_bp),
                                r2, int(order == 1), _p(removed, ctypes.c_uint64), ctypes.byref(evals))
    assert rc == 0
    return bitmap_to_bool(removed, m), evals.value


def pack_bits(bits):
    """(M, N) 0/1 array -> (M, ceil(N/64)) uint64, sample s -> bit s%64 of word s//64"""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    m, n = bits.shape
    wc = (n + 63) // 64
    padded = np.zeros((m, wc * 64), dtype=np.uint8)
    padded[:, :n] = bits & 1
    return np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little")).view(np.uint64).reshape(m, wc)


def oracle_hapsplit(raw_codes, phasepresent, phaseinfo, haploid=False):
    """REF-based codes (M, N) + per-sample phase bits (0/1 arrays, phaseinfo 1 = ALT on the first haplotype) ->
    (hap_nm rows (M, 2*wc) uint64 in the major-allele-inverse coding the pruner sees, maj_freq, unphased flags, hap_ct).
    Mirrors what the reference's loader hands to IndepPairphaseThread (plink2_ld.cc:2040-2052): PgrGetInv1P
    output through HapsplitMustPhased / HapsplitHaploid."""
    lib = oracle()
    m, n = raw_codes.shape
    inv, mf, altmaj = oracle_prepare(raw_codes)
    hap_ct = n if haploid else 2 * n
    wc = (hap_ct + 63) // 64
    rows = np.zeros((m, 2 * wc), dtype=np.uint64)
    unphased = np.zeros(m, dtype=bool)
    pp = pack_bits(phasepresent)
    gw = (n + 31) // 32
    for v in range(m):
        g = np.zeros(gw + 2, dtype=np.uint64)
        g[:gw] = inv[v, :gw]
        if haploid:
            lib.ldo_hapsplit_haploid(_p(g, ctypes.c_uint64), n, _p(rows[v, :wc], ctypes.c_uint64), _p(rows[v, wc:], ctypes.c_uint64))
            continue
        # inverting the counted allele swaps which haplotype carries it (IMPLPgrGetInv1P, pgenlib_read.cc:7016)
        info = phaseinfo[v:v + 1] ^ 1 if altmaj[v] else phaseinfo[v:v + 1]
        pi = pack_bits(info & phasepresent[v:v + 1])
        hap = np.zeros(gw + 1, dtype=np.uint64)
        nm = np.zeros(gw + 1, dtype=np.uint64)
        ppv = np.zeros(pp.shape[1] + 1, dtype=np.uint64)
        ppv[:pp.shape[1]] = pp[v]
        piv = np.zeros(pp.shape[1] + 1, dtype=np.uint64)
        piv[:pp.shape[1]] = pi[0]
        unphased[v] = bool(lib.ldo_hapsplit_must_phased(_p(g, ctypes.c_uint64), _p(ppv, ctypes.c_uint64), _p(piv, ctypes.c_uint64), n,
                                                        _p(hap, ctypes.c_uint64), _p(nm, ctypes.c_uint64)))
        rows[v, :wc] = hap[:wc]
        rows[v, wc:] = nm[:wc]
    return rows, mf, unphased, hap_ct


def oracle_hap_pair_stats(rows, hap_ct, first, second):
    lib = oracle()
    wc = (hap_ct + 63) // 64
    vh = [LdoVhaggs(), LdoVhaggs()]
    for k, v in enumerate((first, second)):
        lib.ldo_fill_vhaggs(_p(rows[v, :wc], ctypes.c_uint64), _p(rows[v, wc:], ctypes.c_uint64), wc, ctypes.byref(vh[k]))
    st = LdoHapPairStats()
    lib.ldo_hap_pair_stats(_p(rows[first, :wc], ctypes.c_uint64), _p(rows[first, wc:], ctypes.c_uint64), ctypes.byref(vh[0]),
                           _p(rows[second, :wc], ctypes.c_uint64), _p(rows[second, wc:], ctypes.c_uint64), ctypes.byref(vh[1]),
                           hap_ct, ctypes.byref(st))
    return st


def oracle_indep_pairphase(hap_nm_rows, hap_ct, chr_idx, bps, maj_freqs, window, step, is_bp, r2, order=2):
    """Returns (removed bool array, pair evaluation count)."""
    lib = oracle()
    m = hap_nm_rows.shape[0]
    rows = np.ascontiguousarray(hap_nm_rows, dtype=np.uint64)
    chr_idx = np.ascontiguousarray(chr_idx, dtype=np.uint32)
    bps = np.ascontiguousarray(bps, dtype=np.uint32)
    maj_freqs = np.ascontiguousarray(maj_freqs, dtype=np.float64)
    removed = np.zeros((m + 63) // 64 + 1, dtype=np.uint64)
    evals = ctypes.c_uint64()
    rc = lib.ldo_indep_pairphase(_p(rows, ctypes.c_uint64), rows.shape[1], m, hap_ct, _p(chr_idx, ctypes.c_uint32),
                                 _p(bps, ctypes.c_uint32), _p(maj_freqs, ctypes.c_double), window, step, int(is_bp),
                                 r2, int(order == 1), _p(removed, ctypes.c_uint64), ctypes.byref(evals))
    assert rc == 0
    return bitmap_to_bool(removed, m), evals.value


def oracle_subcontig_split(chr_idx, bps, window):
    lib = oracle()
    chr_idx = np.ascontiguousarray(chr_idx, dtype=np.uint32)
    m = len(chr_idx)
    info = np.zeros(2 * max(m, 1), dtype=np.uint32)
    wmax = ctypes.c_uint32()
    bp_ptr = None
    if bps is not None:
        bps = np.ascontiguousarray(bps, dtype=np.uint32)
        bp_ptr = _p(bps, ctypes.c_uint32)
    ct = lib.ldo_subcontig_split(_p(chr_idx, ctypes.c_uint32), bp_ptr, m, window, _p(info, ctypes.c_uint32), ctypes.byref(wmax))
    return [(int(info[2 * k]), int(info[2 * k + 1])) for k in range(ct)], wmax.value


# ---------------------------------------------------------------- synthetic data (numpy; small sizes)
def synth_raw_codes(m, n, seed, missing_rate=0.0, ld_copy_prob=0.5, redraw=0.05, maf_lo=0.01):
    """REF-based codes with planted LD between consecutive variants (same spirit as the reference's
    --dummy generator, plink2_import.cc:16387-16432)."""
    rng = np.random.default_rng(seed)
    out = np.empty((m, n), dtype=np.uint8)
    prev = None
    for v in range(m):
        maf = rng.uniform(maf_lo, 0.5)
        if rng.random() < 0.5:
            maf = 1.0 - maf  # ALT may be the major allele
        fresh = (rng.random(n) < maf).astype(np.uint8) + (rng.random(n) < maf).astype(np.uint8)
        if prev is not None and rng.random() < ld_copy_prob:
            keep = rng.random(n) >= redraw
            cur = np.where(keep, prev, fresh)
        else:
            cur = fresh
        prev = cur.copy()
        if missing_rate > 0:
            cur = np.where(rng.random(n) < missing_rate, 3, cur).astype(np.uint8)
        out[v] = cur
    return out


# ---------------------------------------------------------------- PLINK file writers / reference runner
def write_pgen_fixed(prefix, raw_codes, chroms, bps, ids=None, sexes=None):
    """Fixed-width .pgen (storage mode 0x02: pgenlib_read.cc:881-911) + .pvar + .psam."""
    m, n = raw_codes.shape
    rec = (n + 3) // 4
    packed = pack_2bit(raw_codes).view(np.uint8).reshape(m, -1)[:, :rec]
    with open(prefix + ".pgen", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x02]))
        f.write(np.uint32(m).tobytes())
        f.write(np.uint32(n).tobytes())
        f.write(bytes([0x40]))
        f.write(np.ascontiguousarray(packed).tobytes())
    if ids is None:
        ids = ["snp%d" % i for i in range(m)]
    with open(prefix + ".pvar", "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for i in range(m):
            f.write("%s\t%d\t%s\tA\tC\n" % (chroms[i], bps[i], ids[i]))
    with open(prefix + ".psam", "w") as f:
        f.write("#IID\tSEX\n")
        for s in range(n):
            f.write("s%d\t%s\n" % (s, "2" if sexes is None else str(sexes[s])))
    return ids


def phased_pgen_records(raw_codes, phaseinfo):
    """Variable-width .pgen records with the hardcall-phase track (pgen_spec.tex:541-562), every het phased: record =
    raw 2-bit main track (type 0) + aux track 2 = 1 + H bits (bit 0 clear: no explicit phasepresent; then the H
    phaseinfo bits of the het calls in sample order).  Returns (list of record bytes, vrtype array)."""
    m, n = raw_codes.shape
    rec = (n + 3) // 4
    packed = pack_2bit(raw_codes).view(np.uint8).reshape(m, -1)[:, :rec]
    records = []
    vrtypes = np.zeros(m, dtype=np.uint8)
    for v in range(m):
        het = np.flatnonzero(raw_codes[v] == 1)
        body = packed[v].tobytes()
        if len(het):
            bits = np.zeros(1 + len(het), dtype=np.uint8)
            bits[1:] = phaseinfo[v, het] & 1
            body += np.packbits(bits, bitorder="little").tobytes()
            vrtypes[v] = 0x10
        records.append(body)
    return records, vrtypes


def write_pgen_phased(prefix, raw_codes, phaseinfo, chroms, bps, ids=None, sexes=None, parents=None):
    """Standard variable-width .pgen (storage mode 0x10, 8-bit record types, 3-byte record lengths;
    pgen_spec.tex:160-235) whose records carry the hardcall-phase track, + .pvar + .psam."""
    records, vrtypes = phased_pgen_records(raw_codes, phaseinfo)
    return write_pgen_records(prefix, records, vrtypes, raw_codes.shape[1], chroms, bps, ids, sexes, parents)


def write_pgen_records(prefix, records, vrtypes, n, chroms, bps, ids=None, sexes=None, parents=None):
    """... from ready-made record bytes and their 8-bit record types."""
    m = len(records)
    vrtypes = np.asarray(vrtypes, dtype=np.uint8)
    blocks = (m + 65535) // 65536
    header_len = 12 + 8 * blocks + sum(min(65536, m - b * 65536) * 4 for b in range(blocks))
    with open(prefix + ".pgen", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x10]))
        f.write(np.uint32(m).tobytes())
        f.write(np.uint32(n).tobytes())
        f.write(bytes([0x40 | 6]))
        off = header_len
        for b in range(blocks):
            f.write(np.uint64(off).tobytes())
            off += sum(len(r) for r in records[b * 65536:(b + 1) * 65536])
        for b in range(blocks):
            lo, hi = b * 65536, min(m, (b + 1) * 65536)
            f.write(vrtypes[lo:hi].tobytes())
            for r in records[lo:hi]:
                f.write(len(r).to_bytes(3, "little"))
        for r in records:
            f.write(r)
    if ids is None:
        ids = ["snp%d" % i for i in range(m)]
    with open(prefix + ".pvar", "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for i in range(m):
            f.write("%s\t%d\t%s\tA\tC\n" % (chroms[i], bps[i], ids[i]))
    with open(prefix + ".psam", "w") as f:
        f.write("#IID\tPAT\tMAT\tSEX\n")
        for s in range(n):
            pat, mat = parents[s] if parents is not None else ("0", "0")
            f.write("s%d\t%s\t%s\t%s\n" % (s, pat, mat, "NA" if (sexes is None or sexes[s] == 0) else str(sexes[s])))
    return ids


def write_bed(prefix, raw_codes, chroms, bps, ids=None):
    """PLINK 1 .bed/.bim/.fam (pgen_spec.tex:425-428: 00 hom-A1(ALT) 01 missing 10 het 11 hom-A2(REF))."""
    m, n = raw_codes.shape
    lut = np.array([3, 2, 0, 1], dtype=np.uint8)  # pgen code -> bed code
    rec = (n + 3) // 4
    packed = pack_2bit(lut[raw_codes]).view(np.uint8).reshape(m, -1)[:, :rec]
    with open(prefix + ".bed", "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(np.ascontiguousarray(packed).tobytes())
    if ids is None:
        ids = ["snp%d" % i for i in range(m)]
    with open(prefix + ".bim", "w") as f:
        for i in range(m):
            f.write("%s\t%s\t0\t%d\tC\tA\n" % (chroms[i], ids[i], bps[i]))
    with open(prefix + ".fam", "w") as f:
        for s in range(n):
            f.write("s%d s%d 0 0 2 -9\n" % (s, s))
    return ids


def synth_phase(raw_codes, seed, unphased_rate=0.0):
    """Random phase for every het call: (phasepresent, phaseinfo) 0/1 arrays; phaseinfo 1 = "1|0"."""
    rng = np.random.default_rng(seed)
    het = (raw_codes == 1)
    present = het & (rng.random(raw_codes.shape) >= unphased_rate)
    info = present & (rng.random(raw_codes.shape) < 0.5)
    return present.astype(np.uint8), info.astype(np.uint8)


def synth_phased(m, n, seed, missing_rate=0.0, ld_copy_prob=0.5, redraw=0.05, maf_lo=0.01):
    """Phased data with LD planted between consecutive variants on the haplotype level: returns (raw REF-based codes
    (M, N), phasepresent, phaseinfo); every het is phased, phaseinfo 1 = ALT on the first haplotype ("1|0")."""
    rng = np.random.default_rng(seed)
    raw = np.empty((m, n), dtype=np.uint8)
    info = np.zeros((m, n), dtype=np.uint8)
    prev = None
    for v in range(m):
        maf = rng.uniform(maf_lo, 0.5)
        if rng.random() < 0.5:
            maf = 1.0 - maf
        fresh = (rng.random((2, n)) < maf).astype(np.uint8)
        if prev is not None and rng.random() < ld_copy_prob:
            cur = np.where(rng.random((2, n)) >= redraw, prev, fresh)
        else:
            cur = fresh
        prev = cur.copy()
        code = (cur[0] + cur[1]).astype(np.uint8)
        info[v] = ((code == 1) & (cur[0] == 1)).astype(np.uint8)
        if missing_rate > 0:
            code = np.where(rng.random(n) < missing_rate, 3, code).astype(np.uint8)
        raw[v] = code
    present = (raw == 1).astype(np.uint8)
    return raw, present, info & present


def write_vcf(path, raw_codes, chroms, bps, phasepresent=None, phaseinfo=None, ids=None):
    """VCF 4.2 with GT only; phased hets written 0|1 / 1|0, everything else with '/'."""
    m, n = raw_codes.shape
    if ids is None:
        ids = ["snp%d" % i for i in range(m)]
    table = {0: "0/0", 1: "0/1", 2: "1/1", 3: "./."}
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n")
        for c in sorted(set(chroms), key=lambda x: (len(x), x)):
            f.write("##contig=<ID=%s>\n" % c)
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            gts = []
            for s in range(n):
                c = int(raw_codes[v, s])
                if c == 1 and phasepresent is not None and phasepresent[v, s]:
                    gts.append("1|0" if phaseinfo[v, s] else "0|1")
                else:
                    gts.append(table[c])
            f.write("%s\t%d\t%s\tA\tC\t.\t.\t.\tGT\t%s\n" % (chroms[v], bps[v], ids[v], "\t".join(gts)))
    return ids


def synth_multiallelic_haps(m, n, seed, max_alt=3, multi_rate=0.3, missing_rate=0.02, ld_copy_prob=0.5, redraw=0.08):
    """Phased multiallelic data: (first, second) haplotype allele indices (M, N) int arrays (-1 = missing call),
    alt_ct per variant.  LD planted by copying the previous variant's haplotypes (allele indices clipped)."""
    rng = np.random.default_rng(seed)
    first = np.zeros((m, n), dtype=np.int16)
    second = np.zeros((m, n), dtype=np.int16)
    alt_ct = np.where(rng.random(m) < multi_rate, rng.integers(2, max_alt + 1, size=m), 1)
    prev = None
    for v in range(m):
        k = int(alt_ct[v]) + 1
        w = rng.dirichlet(np.ones(k) * rng.uniform(0.3, 2.0))
        fresh = rng.choice(k, size=(2, n), p=w)
        if prev is not None and rng.random() < ld_copy_prob:
            cur = np.where(rng.random((2, n)) >= redraw, np.minimum(prev, k - 1), fresh)
        else:
            cur = fresh
        prev = cur.copy()
        miss = rng.random(n) < missing_rate
        first[v] = np.where(miss, -1, cur[0])
        second[v] = np.where(miss, -1, cur[1])
    return first, second, alt_ct


def write_vcf_haps(path, first, second, alt_ct, chroms, bps, ids=None, unphased=None):
    """VCF with phased GTs a|b from haplotype allele indices (-1 = ./.); unphased (optional bool (M, N)): write a/b"""
    m, n = first.shape
    if ids is None:
        ids = ["snp%d" % i for i in range(m)]
    alts = ["C", "G", "T", "AC", "AG", "AT", "CA", "CC", "CG", "CT"]
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n")
        for c in sorted(set(chroms), key=lambda x: (len(x), x)):
            f.write("##contig=<ID=%s>\n" % c)
        f.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n')
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            gts = []
            for s in range(n):
                a, b = int(first[v, s]), int(second[v, s])
                if a < 0:
                    gts.append("./.")
                elif unphased is not None and unphased[v, s]:
                    gts.append("%d/%d" % (min(a, b), max(a, b)))
                else:
                    gts.append("%d|%d" % (a, b))
            f.write("%s\t%d\t%s\tA\t%s\t.\t.\t.\tGT\t%s\n" % (chroms[v], bps[v], ids[v], ",".join(alts[:int(alt_ct[v])]), "\t".join(gts)))
    return ids


def major_allele_multi(cnt):
    """Allele counts of one variant -> (major allele index, its frequency) the way the reference picks them
    (ComputeAlleleFreqs plink2_filter.cc:2113-2153, GetMajIdxMulti plink2_common.cc:1042-1070, GetAlleleFreq
    plink2_common.h:584-593): frequencies of all alleles but the last are count * (1/total)."""
    k = len(cnt)
    tot = int(np.sum(cnt))
    if tot == 0:
        freq = [1.0 / k] * (k - 1)
    else:
        recip = 1.0 / tot
        freq = [float(cnt[a]) * recip for a in range(k - 1)]
    if freq[0] >= 0.5:
        maj = 0
    elif k == 2:
        maj = 1
    elif freq[1] >= 0.5:
        maj = 1
    else:
        maj, mx = 1, freq[1]
        if freq[0] >= freq[1]:
            maj, mx = 0, freq[0]
        tot_nonlast = freq[0] + freq[1]
        for a in range(2, k - 1):
            if freq[a] > mx:
                maj, mx = a, freq[a]
            tot_nonlast += freq[a]
        if mx + tot_nonlast < 1.0 - K_SMALL_EPSILON:
            maj = k - 1
    if maj + 1 < k:
        return maj, freq[maj]
    last = 1.0 - freq[0]
    for a in range(1, k - 1):
        last -= freq[a]
    return maj, max(last, 0.0)


def pairphase_hap_rows_multiallelic(lo, hi, phasepresent, phaseinfo, alt_ct, quirk=True):
    """What PgrGetInv1P -> HapsplitMustPhased hand to the pairphase scan for possibly multiallelic variants, from allele
    pairs (lo <= hi, 255 missing) and the file's phase bits (phaseinfo 1 = the higher allele on the first haplotype):
    (hap_nm rows for oracle_indep_pairphase, maj_freq, unphased flags).  quirk=True reproduces the reference, whose
    Get1MP (pgenlib_read.cc:6962) passes phaseinfo through unchanged, i.e. reads it as "the counted allele is on the
    first haplotype" even when the major allele is the LOWER allele of a multiallelic het."""
    m, n = lo.shape
    codes = np.full((m, 2 * n), 3, dtype=np.uint8)
    mf = np.zeros(m)
    unphased = np.zeros(m, dtype=bool)
    for v in range(m):
        k = int(alt_ct[v]) + 1
        nm = lo[v] != 255
        cnt = [int((lo[v][nm] == a).sum() + (hi[v][nm] == a).sum()) for a in range(k)]
        maj, mf[v] = major_allele_multi(cnt)
        a, b = lo[v].astype(int), hi[v].astype(int)
        sw = phaseinfo[v].astype(bool)
        first = np.where(sw, b, a)
        second = np.where(sw, a, b)
        if quirk and maj >= 1:
            flip = (a == maj) & (b != maj)
            first, second = np.where(flip, second, first), np.where(flip, first, second)
        codes[v, 1::2] = np.where(nm, np.where(first != maj, 2, 0), 3)
        codes[v, 0::2] = np.where(nm, np.where(second != maj, 2, 0), 3)
        unphased[v] = bool((nm & ((a == maj) != (b == maj)) & ~phasepresent[v].astype(bool)).any())
    rows = np.concatenate([pack_bits((codes == 2).astype(np.uint8)), pack_bits((codes != 3).astype(np.uint8))], axis=1)
    return rows, mf, unphased


def _haploid_style_freqs(codes_subset):
    """diploid-style counts over the given samples: (ALT-is-major flags, major allele frequencies), a*(1/t) arithmetic"""
    cnt = np.stack([(codes_subset == k).sum(1) for k in range(3)], 1).astype(np.int64)
    refc = 2 * cnt[:, 0] + cnt[:, 1]
    tot = refc + 2 * cnt[:, 2] + cnt[:, 1]
    ref_freq = np.where(tot > 0, refc * (1.0 / np.maximum(tot, 1)), 0.5)
    altmaj = ~(ref_freq >= 0.5)
    return altmaj, np.where(altmaj, np.maximum(1.0 - ref_freq, 0.0), ref_freq)


def sex_chromosome_rows(raw, founder, sex, kind, phaseinfo=None):
    """What the reference's loaders hand to the scan on chrX / chrY / MT, as 2-bit codes of 'virtual samples' + the major
    allele frequencies (LoadAlleleAndGenoCountsThread, plink2_data.cc:2421-2700):
      kind "Y": non-female founders, hets -> missing;  "MT": every founder, hets -> missing  (haploid);
      kind "X", --indep-pairwise (phaseinfo None): male founders (hets -> missing) followed by the non-male founders
        TWICE (their statistics count double, plink2_ld.cc:890-901,1066-1082);
      kind "X", --indep-pairphase (phaseinfo given): male founders one haplotype each (hets missing), non-male founders
        two haplotypes split by phase (plink2_ld.cc:2060-2097); a haplotype h is the code 2h.
    raw: REF-based codes (M, N); founder: bool (N,); sex: 1 male / 2 female / 0 unknown."""
    founder = np.asarray(founder, dtype=bool)
    sex = np.asarray(sex)
    if kind in ("Y", "MT"):
        smp = np.where(founder & ((sex != 2) if kind == "Y" else True))[0]
        rs = raw[:, smp]
        _, mf = _haploid_style_freqs(rs)
        return np.where(rs == 1, 3, rs).astype(np.uint8), mf
    males = np.where(founder & (sex == 1))[0]
    non = np.where(founder & (sex != 1))[0]
    rm, rn = raw[:, males], raw[:, non]
    g = np.stack([(raw[:, founder] == k).sum(1) for k in range(3)], 1).astype(np.int64)
    mm = np.stack([(rm == k).sum(1) for k in range(3)], 1).astype(np.int64)
    alt = 4 * g[:, 2] + 2 * g[:, 1] - 2 * mm[:, 2] - mm[:, 1]  # plink2_data.cc:2641
    tot = 2 * (2 * g.sum(1) - mm.sum(1))
    ref_freq = np.where(tot > 0, (tot - alt) * (1.0 / np.maximum(tot, 1)), 0.5)
    altmaj = ~(ref_freq >= 0.5)
    mf = np.where(altmaj, np.maximum(1.0 - ref_freq, 0.0), ref_freq)
    hm = np.where(rm == 1, 3, rm)
    if phaseinfo is None:
        return np.concatenate([hm, rn, rn], axis=1).astype(np.uint8), mf
    pin = phaseinfo[:, non]
    ha = np.where(rn == 3, 3, np.where((rn == 2) | ((rn == 1) & (pin == 1)), 2, 0))
    hb = np.where(rn == 3, 3, np.where((rn == 2) | ((rn == 1) & (pin == 0)), 2, 0))
    return np.concatenate([hm, np.stack([ha, hb], 2).reshape(raw.shape[0], -1)], axis=1).astype(np.uint8), mf


def haploid_codes_to_hap_rows(codes):
    """codes in {0, 2, 3} (haplotype h as 2h, 3 missing) -> hap_nm rows for oracle_indep_pairphase"""
    return np.concatenate([pack_bits((codes == 2).astype(np.uint8)), pack_bits((codes != 3).astype(np.uint8))], axis=1)


def ref_import_vcf(vcf_path, prefix, extra=()):
    """reference: --vcf -> variable-width .pgen (with the hardcall-phase track when the VCF has phased hets)"""
    cp = run_ref(["--vcf", os.path.basename(vcf_path), "--make-pgen", "--out", os.path.basename(prefix)] + list(extra), os.path.dirname(prefix))
    if cp.returncode != 0:
        raise RuntimeError("reference plink2 --vcf failed:\n" + cp.stdout)
    return cp.stdout


def ref_pairphase_chrx_is_unreliable(sexes, founder_mask):
    """The reference's --indep-pairphase chrX loader ORs the male remainder word into a word it never initialised
    when 2 x (non-male founders) is a multiple of 64 and the male founder count is not (plink2_ld.cc:2087-2091:
    word_idx is then one past what HapsplitMustPhased wrote) -- its output then depends on stale memory (observed:
    --threads 1..8 vs 64 vs the default give three different prune lists on the same input).  Comparisons against the
    reference have to stay clear of that sample layout."""
    sexes = np.asarray(sexes)
    founder_mask = np.asarray(founder_mask, dtype=bool)
    males = int(((sexes == 1) & founder_mask).sum())
    nonmales = int(founder_mask.sum()) - males
    return (males % 64 != 0) and ((2 * nonmales) % 64 == 0)


def have_ref():
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def run_ref(args, cwd, timeout=600):
    """Run the reference binary; returns CompletedProcess (stdout+stderr captured as text)."""
    return subprocess.run([REF_BIN] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=timeout)


def read_id_list(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip()]


def ref_indep_pairwise(prefix, window_args, r2, order=2, threads=2, bad_ld=True, fmt="pfile", extra=(), mode="wise"):
    """Run reference --indep-pairwise (mode="phase": --indep-pairphase) on <prefix>; returns (kept ids, removed ids, log text)."""
    cwd = os.path.dirname(prefix)
    out = prefix + ".ref"
    args = ["--" + fmt, os.path.basename(prefix), "--indep-pair" + mode] + [str(a) for a in window_args] + [repr(float(r2))]
    if order == 1:
        args += ["--indep-order", "1"]
    if bad_ld:
        args += ["--bad-ld"]
    args += ["--threads", str(threads), "--out", os.path.basename(out)] + list(extra)
    cp = run_ref(args, cwd)
    if cp.returncode != 0:
        raise RuntimeError("reference plink2 failed:\n" + cp.stdout)
    return read_id_list(out + ".prune.in"), read_id_list(out + ".prune.out"), cp.stdout


def split_pgen_index(pgen_path, out_pgen, out_pgi):
    """A standard (0x10) .pgen rewritten as an external-index pair (pgen_spec.tex:149-170): the header becomes the .pgen.pgi
    (third byte 0x30, block offsets re-based), the records follow a bare three-byte .pgen header (mode 0x20)."""
    data = open(pgen_path, "rb").read()
    assert data[:3] == bytes([0x6C, 0x1B, 0x10])
    m = int.from_bytes(data[3:7], "little")
    n_blocks = (m + 65535) // 65536
    offs = [int.from_bytes(data[12 + 8 * b:20 + 8 * b], "little") for b in range(n_blocks)]
    header_len = offs[0]
    hdr = bytearray(data[:header_len])
    hdr[2] = 0x30
    for b, o in enumerate(offs):
        hdr[12 + 8 * b:20 + 8 * b] = (o - header_len + 3).to_bytes(8, "little")
    with open(out_pgi, "wb") as f:
        f.write(bytes(hdr))
    with open(out_pgen, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x20]) + data[header_len:])


def add_pgen_header_extension(pgen_path, out_path, writer_id=b"ldtools test writer"):
    """A standard (0x10) .pgen rewritten with an ignorable header extension (mode 0x11, pgen_spec.tex:237-270): header flag
    varint 0x2 (writer identifier), no footer extensions, the body's length, the body; block offsets shifted accordingly."""
    data = open(pgen_path, "rb").read()
    assert data[:3] == bytes([0x6C, 0x1B, 0x10]) and len(writer_id) < 128
    m = int.from_bytes(data[3:7], "little")
    n_blocks = (m + 65535) // 65536
    offs = [int.from_bytes(data[12 + 8 * b:20 + 8 * b], "little") for b in range(n_blocks)]
    header_len = offs[0]
    ext = bytes([0x02, 0x00, len(writer_id)]) + writer_id
    hdr = bytearray(data[:header_len])
    hdr[2] = 0x11
    for b, o in enumerate(offs):
        hdr[12 + 8 * b:20 + 8 * b] = (o + len(ext)).to_bytes(8, "little")
    with open(out_path, "wb") as f:
        f.write(bytes(hdr) + ext + data[header_len:])
