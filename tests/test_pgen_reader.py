"""The .pgen/.bed main-track reader (ldp_pgen_*).  Round trips through the REFERENCE's writer: data is
written as fixed-width .pgen by the test, re-encoded by oracle/_ref/plink2 `--make-pgen` into the standard
variable-width format (LD-compressed, one-bit and difflist records), and must decode back to the same codes.
A small reference-written variable-width file is also committed under tests/golden/ so the decoder is pinned
where the reference binary is absent."""
import os
import subprocess

import numpy as np
import pytest

import ldtools as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pgen")


def codes(rows, n):
    """(m, ceil(n/4)) uint8 packed rows -> (m, n) codes"""
    pad = (-rows.shape[1]) % 8
    return T.unpack_2bit(np.ascontiguousarray(np.pad(rows, ((0, 0), (0, pad)))).view(np.uint64), n)


def structured_codes(m, n, seed):
    """Rows that make the reference's writer pick every main-track record type."""
    rng = np.random.default_rng(seed)
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.01, ld_copy_prob=0.6, redraw=0.01)
    for v in range(0, m, 7):       # sparse vs all-0: difflist type 4
        raw[v] = 0
        raw[v, rng.integers(0, n, size=max(1, n // 50))] = rng.integers(1, 4)
    for v in range(3, m, 11):      # sparse vs all-2: type 6
        raw[v] = 2
        raw[v, rng.integers(0, n, size=max(1, n // 60))] = rng.integers(0, 2)
    for v in range(5, m, 13):      # mostly two categories: one-bit type 1
        raw[v] = np.where(rng.random(n) < 0.4, 1, 0)
        raw[v, rng.integers(0, n, size=3)] = 3
    for v in range(6, m, 17):      # inverted copy of the previous row: LD-compressed inverted (type 3)
        inv = raw[v - 1].copy()
        inv[raw[v - 1] == 0] = 2
        inv[raw[v - 1] == 2] = 0
        raw[v] = inv
    raw[1] = 3                     # all missing: type 7
    return raw


def test_fixed_width_and_bed_rows(pkg, tmp_path):
    m, n = 50, 77
    raw = T.synth_raw_codes(m, n, 3, missing_rate=0.05)
    prefix = str(tmp_path / "f")
    T.write_pgen_fixed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    T.write_bed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    f = pkg.PgenFile(prefix + ".pgen")
    assert (f.variant_ct, f.sample_ct, f.mode, f.encoding) == (m, n, 2, pkg.LDP_GENO_REF)
    assert np.array_equal(codes(f.read(), n), raw)
    f.close()
    b = pkg.PgenFile(prefix + ".bed", n, m)
    assert (b.mode, b.encoding) == (1, pkg.LDP_GENO_BED)
    lut = np.array([2, 3, 1, 0], dtype=np.uint8)  # bed -> pgen
    assert np.array_equal(lut[codes(b.read(5, 20), n)], raw[5:25])
    b.close()
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(prefix + ".bed", n + 4, m)   # wrong dimensions -> size check (pgenlib_read.cc:767-789)
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(prefix + ".pgen", n, m + 1)


def test_committed_variable_width_file(pkg):
    path = os.path.join(GOLD, "varwidth_small.pgen")
    z = np.load(os.path.join(GOLD, "varwidth_small_codes.npz"))
    f = pkg.PgenFile(path)
    assert f.mode == 0x10 and (f.variant_ct, f.sample_ct) == (int(z["m"]), int(z["n"]))
    got = codes(f.read(), f.sample_ct)
    assert np.array_equal(got, T.unpack_2bit(z["raw_packed"].reshape(int(z["m"]), -1).view(np.uint64), int(z["n"])))
    # every main-track record type the writer produced is exercised
    assert set(int(t) for t in z["vrtypes_present"]) >= {0, 1, 2, 4, 6}
    # random access into the middle of LD-compressed runs
    for first, cnt in [(17, 5), (100, 1), (299, 40)]:
        part = codes(f.read(first, cnt), f.sample_ct)
        assert np.array_equal(part, got[first:first + cnt])
    f.close()


def test_external_index_file(pkg, tmp_path):
    """Storage mode 0x20: the records in the .pgen, the header in the .pgen.pgi beside it (or wherever --pgi says)."""
    import ctypes
    z = np.load(os.path.join(GOLD, "varwidth_small_codes.npz"))
    want = T.unpack_2bit(z["raw_packed"].reshape(int(z["m"]), -1).view(np.uint64), int(z["n"]))
    T.split_pgen_index(os.path.join(GOLD, "varwidth_small.pgen"), str(tmp_path / "x.pgen"), str(tmp_path / "x.pgen.pgi"))
    f = pkg.PgenFile(str(tmp_path / "x.pgen"))
    assert f.mode == 0x10 and (f.variant_ct, f.sample_ct) == (int(z["m"]), int(z["n"]))
    assert np.array_equal(codes(f.read(), f.sample_ct), want)
    assert np.array_equal(codes(f.read(120, 33), f.sample_ct), want[120:153])
    f.close()
    # the index under another name
    os.rename(str(tmp_path / "x.pgen.pgi"), str(tmp_path / "elsewhere.idx"))
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(str(tmp_path / "x.pgen"))
    L = pkg.lib()
    h = ctypes.c_void_p()
    L.ldp_pgen_open_indexed.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
    assert L.ldp_pgen_open_indexed(str(tmp_path / "x.pgen").encode(), str(tmp_path / "elsewhere.idx").encode(), 0, 0, ctypes.byref(h)) == 0
    L.ldp_pgen_close.argtypes = [ctypes.c_void_p]
    L.ldp_pgen_close(h)
    # a .pgi is not a .pgen; an index with the wrong magic is refused
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(str(tmp_path / "elsewhere.idx"))
    bad = bytearray(open(str(tmp_path / "elsewhere.idx"), "rb").read())
    bad[2] = 0x10
    open(str(tmp_path / "x.pgen.pgi"), "wb").write(bytes(bad))
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(str(tmp_path / "x.pgen"))


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
def test_external_index_file_is_what_the_reference_reads(pkg, tmp_path):
    """The split file is a valid external-index fileset by the reference's own reader: it prunes it to the same lists."""
    m, n = 600, 80
    raw = structured_codes(m, n, 5)
    T.write_pgen_fixed(str(tmp_path / "f"), raw, ["1"] * m, np.arange(m) * 100 + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    T.split_pgen_index(str(tmp_path / "v.pgen"), str(tmp_path / "x.pgen"), str(tmp_path / "x.pgen.pgi"))
    for ext in (".pvar", ".psam"):
        os.link(str(tmp_path / ("v" + ext)), str(tmp_path / ("x" + ext)))
    a = T.run_ref(["--pfile", "v", "--indep-pairwise", "50", "5", "0.3", "--out", "a"], str(tmp_path))
    b = T.run_ref(["--pfile", "x", "--indep-pairwise", "50", "5", "0.3", "--out", "b"], str(tmp_path))
    assert a.returncode == 0 and b.returncode == 0, b.stdout
    assert open(str(tmp_path / "a.prune.in")).read() == open(str(tmp_path / "b.prune.in")).read()
    f = pkg.PgenFile(str(tmp_path / "x.pgen"))
    assert np.array_equal(codes(f.read(threads=3), n), raw)
    f.close()
    # ignorable header extensions (0x11), alone and with the external index (0x21 / 0x31)
    T.add_pgen_header_extension(str(tmp_path / "v.pgen"), str(tmp_path / "e.pgen"))
    for ext in (".pvar", ".psam"):
        os.link(str(tmp_path / ("v" + ext)), str(tmp_path / ("e" + ext)))
    c = T.run_ref(["--pfile", "e", "--indep-pairwise", "50", "5", "0.3", "--out", "c"], str(tmp_path))
    assert c.returncode == 0, c.stdout
    assert open(str(tmp_path / "a.prune.in")).read() == open(str(tmp_path / "c.prune.in")).read()
    f = pkg.PgenFile(str(tmp_path / "e.pgen"))
    assert np.array_equal(codes(f.read(threads=2), n), raw)
    f.close()
    pgi = bytearray(open(str(tmp_path / "x.pgen.pgi"), "rb").read())
    pgi[2] = 0x31
    open(str(tmp_path / "y.pgen.pgi"), "wb").write(bytes(pgi) + bytes([0x00, 0x00]))   # (no extensions present)
    body = bytearray(open(str(tmp_path / "x.pgen"), "rb").read())
    body[2] = 0x21
    open(str(tmp_path / "y.pgen"), "wb").write(bytes(body))
    f = pkg.PgenFile(str(tmp_path / "y.pgen"))
    assert np.array_equal(codes(f.read(), n), raw)
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed", [(400, 50, 1), (900, 300, 2), (70000, 40, 3), (300, 70000, 4)])
def test_roundtrip_through_reference_writer(pkg, tmp_path, m, n, seed):
    raw = structured_codes(m, n, seed)
    prefix = str(tmp_path / "f")
    T.write_pgen_fixed(prefix, raw, ["1"] * m, np.arange(m) + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "v.pgen"))
    assert f.mode == 0x10 and (f.variant_ct, f.sample_ct) == (m, n)
    got = codes(f.read(threads=4), n)
    assert np.array_equal(got, raw)
    part = codes(f.read(m // 3, min(50, m - m // 3)), n)
    assert np.array_equal(part, raw[m // 3:m // 3 + min(50, m - m // 3)])
    f.close()


def test_malformed_files_are_rejected(pkg, tmp_path):
    src = open(os.path.join(GOLD, "varwidth_small.pgen"), "rb").read()
    bad = str(tmp_path / "t.pgen")
    open(bad, "wb").write(src[:len(src) // 2])            # truncated records
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(bad).read()
    open(bad, "wb").write(src[:2] + bytes([0x21]) + src[3:])  # external-index mode without its index
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(bad)
    open(bad, "wb").write(src[:2] + bytes([0x05]) + src[3:])  # no such storage mode
    with pytest.raises(pkg.LdpError) as ei:
        pkg.PgenFile(bad)
    assert ei.value.code == pkg.LDP_ERR_UNSUPPORTED
    open(bad, "wb").write(src[:2] + bytes([0x03]) + src[3:])  # fixed-width dosage mode over a variable-width header
    with pytest.raises(pkg.LdpError) as ei:
        pkg.PgenFile(bad)
    assert ei.value.code == pkg.LDP_ERR_INVALID
    open(bad, "wb").write(b"\x00\x01\x02")
    with pytest.raises(pkg.LdpError):
        pkg.PgenFile(bad)


# ---------------------------------------------------------------- multiallelic hard-call track
def make_multiallelic_vcf(path, m, n, seed, max_alt=5, missing=0.03):
    """VCF with 1..max_alt ALT alleles per site; returns (alt_ct[m], lo[m,n], hi[m,n]) with 255 = missing."""
    rng = np.random.default_rng(seed)
    alt_ct = rng.integers(1, max_alt + 1, size=m)
    alt_ct[:4] = [1, 2, 3, min(max_alt, 5)]
    lo = np.zeros((m, n), dtype=np.uint8)
    hi = np.zeros((m, n), dtype=np.uint8)
    bases = ["C", "G", "T", "AC", "AG", "AT", "ACC", "AGG", "ATT", "ACCC", "AGGG", "ATTT", "ACCCC", "AGGGG", "ATTTT", "ACCCCC", "AGGGGG", "ATTTTT"]
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=1>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"GT\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("s%d" % s for s in range(n)) + "\n")
        for v in range(m):
            k = int(alt_ct[v])
            # allele frequencies: sometimes an ALT allele is the most common one
            w = rng.dirichlet(np.ones(k + 1) * (0.6 if v % 3 else 2.0))
            a = rng.choice(k + 1, size=n, p=w)
            b = rng.choice(k + 1, size=n, p=w)
            l, h = np.minimum(a, b), np.maximum(a, b)
            miss = rng.random(n) < missing
            l = np.where(miss, 255, l).astype(np.uint8)
            h = np.where(miss, 255, h).astype(np.uint8)
            if v == 5 and k >= 2:      # a declared-multiallelic site where only REF and ALT1 occur
                l = np.where(l == 255, 255, np.minimum(l, 1)).astype(np.uint8)
                h = np.where(h == 255, 255, np.minimum(h, 1)).astype(np.uint8)
            lo[v], hi[v] = l, h
            gts = ["./." if l[s] == 255 else "%d/%d" % (l[s], h[s]) for s in range(n)]
            f.write("1\t%d\tsnp%d\tA\t%s\t.\t.\t.\tGT\t%s\n" % (1000 + 37 * v, v, ",".join(bases[:k]), "\t".join(gts)))
    return alt_ct, lo, hi


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed,max_alt", [(120, 90, 1, 2), (150, 300, 2, 5), (80, 70, 3, 17), (60, 40, 4, 18)])
def test_multiallelic_track_against_vcf(pkg, tmp_path, m, n, seed, max_alt):
    alt_ct, lo, hi = make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed, max_alt=max_alt)
    cp = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "mv.pgen"))
    assert f.mode == 0x10 and f.has_multiallelic
    for v in range(m):
        glo, ghi = f.read_alleles(v, int(alt_ct[v]))
        assert np.array_equal(glo, lo[v]) and np.array_equal(ghi, hi[v]), v
    f.close()


@pytest.mark.parametrize("raw_n,keep_rate,phased", [(1, 1.0, False), (5, 0.5, True), (63, 0.9, False), (64, 0.3, True), (129, 0.97, True),
                                                     (1000, 0.99, False), (4097, 0.6, True), (257, 0.0, False)])
def test_subset_samples_matches_numpy(pkg, raw_n, keep_rate, phased):
    """ldp_subset_samples = CopyNyparrNonemptySubset (+ CopyBitarrSubset for phase bits) on row blocks"""
    import ldtools as T
    rng = np.random.default_rng(raw_n)
    m = 37
    codes = rng.integers(0, 4, size=(m, raw_n)).astype(np.uint8)
    phase = rng.integers(0, 2, size=(m, raw_n)).astype(np.uint8)
    keep = rng.random(raw_n) < keep_rate
    if not keep.any():
        keep[raw_n // 2] = True
    packed = T.pack_2bit(codes).view(np.uint8).reshape(m, -1)
    rows = pkg.pack_phased_rows(packed, phase, raw_n) if phased else np.ascontiguousarray(packed[:, :(raw_n + 3) // 4])
    # an input stride with slack, like a caller's buffer
    wide = np.full((m, rows.shape[1] + 5), 0xAB, dtype=np.uint8)
    wide[:, :rows.shape[1]] = rows
    for threads in (1, 0):
        got = pkg.subset_samples(wide, raw_n, keep, phased=phased, threads=threads)
        kept = int(keep.sum())
        want_codes = T.pack_2bit(codes[:, keep]).view(np.uint8).reshape(m, -1)
        want = pkg.pack_phased_rows(want_codes, phase[:, keep], kept) if phased else want_codes[:, :(kept + 3) // 4]
        assert np.array_equal(got, want), (raw_n, phased, threads)


def test_subset_samples_property(pkg):
    """hypothesis: arbitrary sample counts / masks / strides against the numpy definition"""
    from hypothesis import given, settings, strategies as st
    import ldtools as T

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 300), st.integers(0, 2 ** 32 - 1), st.booleans(), st.integers(0, 9))
    def check(raw_n, seed, phased, slack):
        rng = np.random.default_rng(seed)
        m = int(rng.integers(1, 6))
        codes = rng.integers(0, 4, size=(m, raw_n)).astype(np.uint8)
        phase = rng.integers(0, 2, size=(m, raw_n)).astype(np.uint8)
        keep = rng.random(raw_n) < rng.random()
        if not keep.any():
            keep[int(rng.integers(0, raw_n))] = True
        packed = T.pack_2bit(codes).view(np.uint8).reshape(m, -1)
        rows = pkg.pack_phased_rows(packed, phase, raw_n) if phased else np.ascontiguousarray(packed[:, :(raw_n + 3) // 4])
        wide = np.full((m, rows.shape[1] + slack), 0x5A, dtype=np.uint8)
        wide[:, :rows.shape[1]] = rows
        got = pkg.subset_samples(wide, raw_n, keep, phased=phased, threads=1)
        kept = int(keep.sum())
        want_codes = T.pack_2bit(codes[:, keep]).view(np.uint8).reshape(m, -1)
        want = pkg.pack_phased_rows(want_codes, phase[:, keep], kept) if phased else want_codes[:, :(kept + 3) // 4]
        assert np.array_equal(got, want)

    check()


def test_record_index_and_file_bytes(pkg, tmp_path):
    """What ldp_load_pgen_records is fed from: the reader's mapping and its index entries (no GPU needed).  Records tile the file's
    record area in order, types are the header's, and the LD base of an LD-compressed record is the closest earlier record that is
    not LD-compressed; a fixed-width file yields plain records, a .bed is refused."""
    path = os.path.join(GOLD, "varwidth_small.pgen")
    f = pkg.PgenFile(path)
    m = f.variant_ct
    raw = open(path, "rb").read()
    ptr, nbytes = f.file_bytes()
    assert nbytes == len(raw)
    assert bytes((pkg.ctypes.c_uint8 * 16).from_address(ptr)) == raw[:16]
    recs, base = f.record_index()
    assert base is None
    offs = [recs[q].offset for q in range(m)]
    lens = [recs[q].length for q in range(m)]
    assert all(offs[q] + lens[q] == offs[q + 1] for q in range(m - 1)) and offs[-1] + lens[-1] == len(raw)
    assert all(recs[q].allele_ct == 2 for q in range(m))
    types = [recs[q].vrtype & 7 for q in range(m)]
    assert types[0] not in (2, 3)
    ld = [q for q in range(m) if types[q] in (2, 3)]
    assert ld
    for q in ld[:: max(1, len(ld) // 20)]:
        _, b = f.record_index(q, 1)
        want = max(k for k in range(q) if types[k] not in (2, 3))
        assert b == want
        assert f.record_index(want, 1)[1] is None
    # allele counts from the caller
    recs2, _ = f.record_index(3, 4, allele_cts=[2, 3, 5, 2])
    assert [recs2[q].allele_ct for q in range(4)] == [2, 3, 5, 2] and recs2[0].offset == offs[3]
    f.close()
    # fixed-width .pgen: plain records; .bed: refused
    rawc = T.synth_raw_codes(30, 41, 3, missing_rate=0.05)
    T.write_pgen_fixed(str(tmp_path / "fx"), rawc, ["1"] * 30, np.arange(30) + 1)
    g = pkg.PgenFile(str(tmp_path / "fx.pgen"))
    rg, bg = g.record_index()
    assert bg is None and all((rg[q].vrtype, rg[q].length) == (0, (41 + 3) // 4) for q in range(30)) and rg[1].offset - rg[0].offset == 11
    g.close()
    T.write_bed(str(tmp_path / "b"), rawc, ["1"] * 30, np.arange(30) + 1)
    h = pkg.PgenFile(str(tmp_path / "b.bed"), sample_ct_hint=41, variant_ct_hint=30)
    with pytest.raises(pkg.LdpError) as ei:
        h.record_index()
    assert ei.value.code == pkg.LDP_ERR_UNSUPPORTED
    h.close()
