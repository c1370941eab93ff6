"""ldp_load_pgen_records: variant records of a variable-width .pgen decoded on the device (ldp_pgen_decode.hip) -- every
main-track record type, LD-compressed chains across calls, auxiliary track 1 with the major-vs-rest collapse -- against the
host reader (itself pinned to the reference's writer and to VCF input in test_pgen_reader.py) and against numpy restatements of
Get1Multiallelic / GetMajIdxMulti.  Rows are compared bit for bit (ldp_get_planes), records and major-allele frequencies
exactly, and the prune set with the one of the same rows loaded from host memory."""
import os

import numpy as np
import pytest

import ldtools as T
from test_pgen_reader import GOLD, codes, make_multiallelic_vcf, structured_codes

pytestmark = pytest.mark.gpu


def positions(m, spacing=1000):
    return np.zeros(m, dtype=np.uint32), (np.arange(m, dtype=np.uint32) + 1) * spacing


def engine(pkg, n, m, window=40, r2=0.3):
    eng = pkg.LdPruneEngine(n, window, 1, False, r2, order=2, device=0)
    chr_idx, bps = positions(m)
    eng.set_variants(chr_idx, bps)
    return eng


def assert_same_rows(a, b, m):
    ra, rb = a.variant_recs(), b.variant_recs()
    for name in ("nm_ct", "sum", "ssq", "flags", "n_homref", "n_het", "n_homalt"):
        assert np.array_equal(ra[name], rb[name]), name
    for v in range(m):
        ha, xa = a.planes(v)
        hb, xb = b.planes(v)
        assert np.array_equal(ha, hb) and np.array_equal(xa, xb), v
    assert np.array_equal(a.maj_freqs(), b.maj_freqs())


def test_committed_variable_width_file(gpu_pkg):
    pkg = gpu_pkg
    f = pkg.PgenFile(os.path.join(GOLD, "varwidth_small.pgen"))
    z = np.load(os.path.join(GOLD, "varwidth_small_codes.npz"))
    m, n = f.variant_ct, f.sample_ct
    want = T.unpack_2bit(z["raw_packed"].reshape(int(z["m"]), -1).view(np.uint64), int(z["n"]))
    rows = f.read()
    assert np.array_equal(codes(rows, n), want)
    host = engine(pkg, n, m)
    host.load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    dev = engine(pkg, n, m)
    maj = dev.load_pgen_records(0, f)
    assert np.all(maj == 0xffffffff)
    assert dev.counters()["decoded_in_place_rows"] == m   # (straight into the engine's image: no scratch row, no copy)
    assert_same_rows(host, dev, m)
    assert np.array_equal(host.run(), dev.run())
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed", [(400, 50, 1), (900, 300, 2), (3000, 1237, 5), (300, 70000, 4)])
def test_every_record_type_of_the_reference_writer(gpu_pkg, tmp_path, m, n, seed):
    pkg = gpu_pkg
    raw = structured_codes(m, n, seed)
    T.write_pgen_fixed(str(tmp_path / "f"), raw, ["1"] * m, np.arange(m) + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "v.pgen"))
    assert f.mode == 0x10
    recs, _ = f.record_index()
    types = {int(recs[q].vrtype) & 7 for q in range(m)}
    assert types & {2, 3} and types & {4, 6, 7} and (1 in types), types   # (LD-compressed, difflist and one-bit records are all there)
    rows = f.read(threads=4)
    host = engine(pkg, n, m)
    host.load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    # one call
    dev = engine(pkg, n, m)
    dev.load_pgen_records(0, f)
    assert dev.counters()["decoded_in_place_rows"] == m
    assert_same_rows(host, dev, m)
    # several calls that cut LD chains: the engine carries the base over
    dev2 = engine(pkg, n, m)
    cuts = [0, 1, 7, m // 3, m // 3 + 1, (2 * m) // 3, m]
    for a, b in zip(cuts[:-1], cuts[1:]):
        dev2.load_pgen_records(a, f, a, b - a)
    assert_same_rows(host, dev2, m)
    # a call that starts inside a chain without its predecessor: ld_base names the record
    ld = [q for q in range(1, m) if (int(recs[q].vrtype) & 6) == 2 and (int(recs[q - 1].vrtype) & 6) == 2]
    if ld:
        q = ld[len(ld) // 2]
        dev3 = engine(pkg, n, m)
        dev3.load_pgen_records(q, f, q, m - q)
        dev3.load_pgen_records(0, f, 0, q)
        assert_same_rows(host, dev3, m)
        assert 0 < dev3.counters()["decoded_in_place_rows"] <= m   # (the launch that carries the ld_base record as an extra row takes the scratch)
    # the file's bytes already on the device
    import torch
    ptr, nbytes = f.file_bytes()
    buf = torch.from_numpy(np.ctypeslib.as_array((pkg.ctypes.c_uint8 * nbytes).from_address(ptr)).copy()).cuda()
    dev4 = engine(pkg, n, m)
    dev4.load_pgen_records(0, f, location=pkg.LDP_MEM_DEVICE, device_bytes=buf.data_ptr())
    assert_same_rows(host, dev4, m)
    assert np.array_equal(host.run(), dev4.run())
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
def test_small_launches_and_sharded_engines(gpu_pkg, tmp_path, monkeypatch):
    """The call's own split into launches (here forced to 5 rows: every LD chain is cut) and engines that own only some of the
    subcontigs (LD chains run through records the engine does not keep): the union of two shards equals the unsharded result."""
    pkg = gpu_pkg
    m, n = 1200, 700
    raw = structured_codes(m, n, 21)
    T.write_pgen_fixed(str(tmp_path / "f"), raw, ["1"] * m, np.arange(m) + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "v.pgen"))
    rows = f.read()
    host = engine(pkg, n, m)
    host.load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    want = host.run()
    dev = engine(pkg, n, m)
    dev.set_option("decode_rows", 5)
    dev.load_pgen_records(0, f)
    assert_same_rows(host, dev, m)
    # rows assembled in global memory (what rows beyond 128 KiB take) instead of LDS
    dev_g = engine(pkg, n, m)
    dev_g.set_option("decode_no_lds", 1)
    dev_g.load_pgen_records(0, f)
    assert_same_rows(host, dev_g, m)
    # four chromosomes, two shards on the same device
    chr_idx = (np.arange(m) * 4 // m).astype(np.uint32)
    bps = (np.arange(m, dtype=np.uint32) + 1) * 1000
    whole = pkg.LdPruneEngine(n, 40, 1, False, 0.3, order=2, device=0)
    whole.set_variants(chr_idx, bps)
    whole.load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    want4 = whole.run()
    got = np.zeros(m, dtype=bool)
    for rank in range(2):
        eng = pkg.LdPruneEngine(n, 40, 1, False, 0.3, order=2, device=0)
        eng.set_variants(chr_idx, bps)
        eng.set_shard(rank, 2)
        for a, b in [(0, 333), (333, 901), (901, m)]:
            eng.load_pgen_records(a, f, a, b - a)
        got |= eng.run()
    assert np.array_equal(got, want4) and want4.sum() > 0 and want.sum() > 0
    f.close()


def test_records_into_an_engine_that_keeps_bit_planes(gpu_pkg):
    """pair_mfma 0 (what engines beyond 16,000,000 founders do): the decoded rows go through prepare_kernel instead of the count
    pass of the code image; same records, planes and prune set as host-loaded rows."""
    pkg = gpu_pkg
    f = pkg.PgenFile(os.path.join(GOLD, "varwidth_small.pgen"))
    m, n = f.variant_ct, f.sample_ct
    rows = f.read()
    engines = []
    for _ in range(2):
        eng = pkg.LdPruneEngine(n, 40, 1, False, 0.3, order=2, device=0)
        eng.set_option("pair_mfma", 0)
        chr_idx, bps = positions(m)
        eng.set_variants(chr_idx, bps)
        engines.append(eng)
    engines[0].load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    engines[1].load_pgen_records(0, f)
    assert_same_rows(engines[0], engines[1], m)
    assert np.array_equal(engines[0].run(), engines[1].run())
    f.close()


def collapse(lo, hi, alt_ct):
    """Get1Multiallelic + GetMajIdxMulti in numpy: (INVERSE-coded codes, major allele, its frequency) of one variant."""
    called = lo != 255
    cnt = np.zeros(alt_ct + 1, dtype=np.int64)
    np.add.at(cnt, lo[called], 1)
    np.add.at(cnt, hi[called], 1)
    maj, mf = T.major_allele_multi(cnt)
    c = np.where(called, (lo != maj).astype(np.uint8) + (hi != maj).astype(np.uint8), 3).astype(np.uint8)
    return c, maj, mf


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed,max_alt", [(120, 90, 1, 2), (150, 300, 2, 5), (80, 70, 3, 17), (60, 40, 4, 18), (60, 5000, 6, 4), (40, 3000, 7, 2)])
def test_multiallelic_records_are_collapsed_on_the_device(gpu_pkg, tmp_path, m, n, seed, max_alt):
    pkg = gpu_pkg
    alt_ct, lo, hi = make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed, max_alt=max_alt, missing=0.03 if n < 1000 else 0.002)
    if n >= 1000:
        # rare third alleles: the patch sets become sample-id lists instead of bit arrays
        rng = np.random.default_rng(seed)
        with open(str(tmp_path / "m.vcf")) as fh:
            lines = fh.read().split("\n")
        out = []
        v = 0
        for line in lines:
            if line.startswith("#") or not line:
                out.append(line)
                continue
            parts = line.split("\t")
            k = int(alt_ct[v])
            if k >= 2 and v % 2 == 0:
                l = np.where(rng.random(n) < 0.3, 1, 0).astype(np.uint8)
                h = np.maximum(l, np.where(rng.random(n) < 0.3, 1, 0)).astype(np.uint8)
                l, h = np.minimum(l, h), np.maximum(l, h)
                rare = rng.choice(n, size=7, replace=False)
                h[rare[:4]] = k
                l[rare[4:]] = np.minimum(l[rare[4:]], 1)
                h[rare[4:]] = k
                l[rare[2:4]] = k if k >= 2 else l[rare[2:4]]
                l, h = np.minimum(l, h), np.maximum(l, h)
                lo[v], hi[v] = l, h
                parts[9:] = ["%d/%d" % (l[s], h[s]) for s in range(n)]
            out.append("\t".join(parts))
            v += 1
        open(str(tmp_path / "m.vcf"), "w").write("\n".join(out))
    cp = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "mv.pgen"))
    assert f.has_multiallelic
    # host path: main track for one ALT allele, numpy collapse of the reader's allele pairs for the others
    rows = f.read()
    host = engine(pkg, n, m)
    host.load_genotypes_host(0, rows, pkg.LDP_GENO_REF)
    want_maj = np.full(m, 0xffffffff, dtype=np.uint32)
    for v in range(m):
        if alt_ct[v] > 1:
            glo, ghi = f.read_alleles(v, int(alt_ct[v]))
            assert np.array_equal(glo, lo[v]) and np.array_equal(ghi, hi[v])
            c, maj, mf = collapse(glo, ghi, int(alt_ct[v]))
            want_maj[v] = maj
            packed = np.ascontiguousarray(T.pack_2bit(c[None, :]).view(np.uint8).reshape(1, -1)[:, :(n + 3) // 4])
            host.load_genotypes_host(v, packed, pkg.LDP_GENO_INVERSE)
            host.set_maj_freqs(v, np.array([mf]))
    dev = engine(pkg, n, m)
    got_maj = dev.load_pgen_records(0, f, allele_cts=alt_ct + 1)
    assert np.array_equal(got_maj, want_maj)
    assert len(set(want_maj[alt_ct > 1].tolist())) >= 2   # (REF, ALT1 and later alleles all occur as the major one over the cases)
    assert_same_rows(host, dev, m)
    assert np.array_equal(host.run(), dev.run())
    # in pieces (the collapse must not disturb LD bases: a base is the main track as stored)
    dev2 = engine(pkg, n, m)
    for a, b in [(0, 5), (5, m // 2), (m // 2, m)]:
        dev2.load_pgen_records(a, f, a, b - a, allele_cts=(alt_ct + 1)[a:b])
    assert_same_rows(host, dev2, m)
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
def test_records_of_a_file_with_more_samples_than_founders(gpu_pkg, tmp_path):
    """With a sample map the decoded rows are the file's and the engine gathers its columns (LDP_GENO_MAPPED)."""
    pkg = gpu_pkg
    m, raw_n = 500, 333
    raw = structured_codes(m, raw_n, 11)
    T.write_pgen_fixed(str(tmp_path / "f"), raw, ["1"] * m, np.arange(m) + 1)
    cp = T.run_ref(["--pfile", "f", "--make-pgen", "--out", "v"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "v.pgen"))
    keep = np.flatnonzero(np.random.default_rng(3).random(raw_n) < 0.7).astype(np.uint32)
    n = len(keep)
    sub = np.ascontiguousarray(T.pack_2bit(raw[:, keep]).view(np.uint8).reshape(m, -1)[:, :(n + 3) // 4])
    host = engine(pkg, n, m)
    host.load_genotypes_host(0, sub, pkg.LDP_GENO_REF)
    dev = engine(pkg, n, m)
    dev.set_sample_map(raw_n, keep)
    dev.load_pgen_records(0, f)
    assert_same_rows(host, dev, m)
    assert np.array_equal(host.run(), dev.run())
    # allele_ct > 2 under a map that is not a plain subset of the file's samples (a sample twice: the chrX layouts): refused
    twice = engine(pkg, n + 1, m)
    twice.set_sample_map(raw_n, np.concatenate([keep, keep[:1]]).astype(np.uint32))
    with pytest.raises(pkg.LdpError) as ei:
        twice.load_pgen_records(0, f, allele_cts=np.full(m, 3))
    assert ei.value.code == pkg.LDP_ERR_UNSUPPORTED
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,raw_n,seed,max_alt,keep_frac", [(120, 130, 21, 3, 0.7), (90, 400, 22, 6, 0.5), (60, 5000, 23, 4, 0.9), (70, 64, 24, 2, 0.3)])
def test_multiallelic_collapse_over_a_subset_of_the_files_samples(gpu_pkg, tmp_path, m, raw_n, seed, max_alt, keep_frac):
    """Founders among non-founders: the engine's samples are a subset of the file's (ldp_set_sample_map without repeats), and the allele
    counts that choose a multiallelic variant's major allele run over THEM (the allele-frequency pass counts founders,
    plink2_filter.cc:2113-2153) -- pgen_aux1_kernel with a sample mask, against the numpy collapse of the reader's allele pairs over
    the same samples; the major allele differs from the whole file's for some variants by construction (a small subset)."""
    pkg = gpu_pkg
    alt_ct, lo, hi = make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, raw_n, seed, max_alt=max_alt, missing=0.03 if raw_n < 1000 else 0.002)
    cp = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "mv.pgen"))
    keep = np.flatnonzero(np.random.default_rng(seed).random(raw_n) < keep_frac).astype(np.uint32)
    n = len(keep)
    rows = f.read()
    host = engine(pkg, n, m)
    want_maj = np.full(m, 0xffffffff, dtype=np.uint32)
    whole_maj = np.full(m, 0xffffffff, dtype=np.uint32)
    for v in range(m):
        glo, ghi = f.read_alleles(v, int(alt_ct[v])) if alt_ct[v] > 1 else (None, None)
        if alt_ct[v] > 1:
            c, maj, mf = collapse(glo[keep], ghi[keep], int(alt_ct[v]))
            want_maj[v] = maj
            whole_maj[v] = collapse(glo, ghi, int(alt_ct[v]))[1]
            packed = np.ascontiguousarray(T.pack_2bit(c[None, :]).view(np.uint8).reshape(1, -1)[:, :(n + 3) // 4])
            host.load_genotypes_host(v, packed, pkg.LDP_GENO_INVERSE)
            host.set_maj_freqs(v, np.array([mf]))
        else:
            codes = ((rows[v][:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(-1)[:raw_n][keep].astype(np.uint8)
            packed = np.ascontiguousarray(T.pack_2bit(codes[None, :]).view(np.uint8).reshape(1, -1)[:, :(n + 3) // 4])
            host.load_genotypes_host(v, packed, pkg.LDP_GENO_REF)
    dev = engine(pkg, n, m)
    dev.set_sample_map(raw_n, keep)
    got_maj = dev.load_pgen_records(0, f, allele_cts=alt_ct + 1)
    assert np.array_equal(got_maj, want_maj)
    if keep_frac <= 0.5:
        assert (want_maj != whole_maj).any()          # (the subset's major allele is not always the file's)
    assert_same_rows(host, dev, m)
    assert np.array_equal(host.run(), dev.run())
    # in pieces, and after the device copies were released (the mask is rebuilt from the host's map)
    dev2 = engine(pkg, n, m)
    dev2.set_sample_map(raw_n, keep)
    for a, b in [(0, 7), (7, m // 2), (m // 2, m)]:
        dev2.load_pgen_records(a, f, a, b - a, allele_cts=(alt_ct + 1)[a:b])
    assert_same_rows(host, dev2, m)
    f.close()


def test_malformed_records_are_refused(gpu_pkg, tmp_path):
    pkg = gpu_pkg
    src = open(os.path.join(GOLD, "varwidth_small.pgen"), "rb").read()
    f = pkg.PgenFile(os.path.join(GOLD, "varwidth_small.pgen"))
    m, n = f.variant_ct, f.sample_ct
    recs, _ = f.record_index()
    ptr, nbytes = f.file_bytes()
    data = np.frombuffer(src, dtype=np.uint8).copy()
    rng = np.random.default_rng(5)
    refused = 0
    for trial in range(40):
        bad = data.copy()
        q = int(rng.integers(0, m))
        r = recs[q]
        if (int(r.vrtype) & 7) == 0 or r.length < 3:
            continue
        mode = trial % 3
        cut = (pkg.ldp_pgen_rec * m)()
        for k in range(m):
            cut[k].offset, cut[k].length, cut[k].vrtype, cut[k].allele_ct = recs[k].offset, recs[k].length, recs[k].vrtype, 2
        if mode == 0:
            cut[q].length = int(rng.integers(1, r.length))          # truncated record
        elif mode == 1:
            bad[r.offset + int(rng.integers(0, r.length))] ^= 0xff   # a flipped byte
        else:
            bad[r.offset:r.offset + r.length] = rng.integers(0, 256, size=r.length, dtype=np.uint8)
        eng = engine(pkg, n, m)
        maj = np.zeros(m, dtype=np.uint32)
        rc = eng._L.ldp_load_pgen_records(eng._h, 0, m, bad.ctypes.data_as(pkg.ctypes.c_void_p), len(bad), pkg.LDP_MEM_HOST, cut, None, n,
                                          maj.ctypes.data_as(pkg.ctypes.POINTER(pkg.ctypes.c_uint32)))
        # a damaged record is either refused or decodes to SOME row (a flipped genotype value is not detectable); it never crashes,
        # and a refused call leaves the engine usable
        assert rc in (pkg.LDP_OK, pkg.LDP_ERR_INVALID)
        if rc == pkg.LDP_ERR_INVALID:
            refused += 1
            eng.load_pgen_records(0, f)
            eng.run()
    assert refused >= 5
    # records outside the buffer, LD-compressed first record without a base
    eng = engine(pkg, n, m)
    with pytest.raises(pkg.LdpError):
        eng._ck(eng._L.ldp_load_pgen_records(eng._h, 0, m, pkg.ctypes.c_void_p(ptr), 10, pkg.LDP_MEM_HOST, recs, None, n, None))
    ld = [q for q in range(m) if (int(recs[q].vrtype) & 6) == 2]
    if ld:
        one = (pkg.ldp_pgen_rec * 1)()
        one[0].offset, one[0].length, one[0].vrtype, one[0].allele_ct = recs[ld[0]].offset, recs[ld[0]].length, recs[ld[0]].vrtype, 2
        with pytest.raises(pkg.LdpError):
            eng._ck(eng._L.ldp_load_pgen_records(eng._h, ld[0], 1, pkg.ctypes.c_void_p(ptr), nbytes, pkg.LDP_MEM_HOST, one, None, n, None))
    f.close()


def phased_engine(pkg, n, m, window=40, r2=0.3):
    eng = pkg.LdPruneEngine(2 * n, window, 1, False, r2, order=2, device=0)   # haplotypes: two per sample of the file
    chr_idx, bps = positions(m)
    eng.set_variants(chr_idx, bps)
    return eng


def test_phase_track_decoded_on_the_device(gpu_pkg):
    """ldp_load_pgen_records_phased (--indep-pairphase): the hardcall-phase track (auxiliary track 2) of a reference-written,
    fully phased VCF import decoded on the device -- against the host reader's LDP_GENO_PHASED rows (test_pairphase.py pins those to
    the reference) loaded from host memory: haplotype rows bit for bit, records, frequencies, prune set; in one call and in calls
    that cut the LD chains."""
    pkg = gpu_pkg
    f = pkg.PgenFile(os.path.join(GOLD, "phased_small.pgen"))
    m, n = f.variant_ct, f.sample_ct
    rows = f.read_phased()
    host = phased_engine(pkg, n, m)
    host.load_genotypes_host(0, rows, pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
    dev = phased_engine(pkg, n, m)
    dev.load_pgen_records_phased(0, f)
    assert_same_rows(host, dev, m)
    assert np.array_equal(host.run(), dev.run()) and host.run().sum() > 0
    dev2 = phased_engine(pkg, n, m)
    cuts = [0, 1, 5, m // 3, m // 3 + 1, (2 * m) // 3, m]
    for a, b in zip(cuts[:-1], cuts[1:]):
        dev2.load_pgen_records_phased(a, f, a, b - a)
    assert_same_rows(host, dev2, m)
    for e in (host, dev, dev2):
        e.close()
    f.close()


def test_device_phase_decode_reports_the_first_unphased_variant(gpu_pkg):
    """A partially phased file (the reference: "variant #40 is not fully phased", plink2_ld.cc:2045-2049): the fully phased head loads,
    a call that reaches variant 40 returns LDP_ERR_UNPHASED with that variant, whatever the launch size, and the engine stays usable."""
    pkg = gpu_pkg
    f = pkg.PgenFile(os.path.join(GOLD, "phased_partial.pgen"))
    m, n = f.variant_ct, f.sample_ct
    z = np.load(os.path.join(GOLD, "phased_partial.npz"))
    first_bad = int(np.argmax(((z["raw"] == 1) & (z["phasepresent"] == 0)).any(axis=1)))
    assert first_bad == 40
    dev = phased_engine(pkg, n, m)
    dev.load_pgen_records_phased(0, f, 0, first_bad)
    host = phased_engine(pkg, n, m)
    host.load_genotypes_host(0, f.read_phased(0, first_bad), pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
    for v in range(first_bad):
        assert all(np.array_equal(a, b) for a, b in zip(host.planes(v), dev.planes(v)))
    for rows_per_launch in (0, 7):
        dev.set_option("decode_rows", rows_per_launch)
        with pytest.raises(pkg.LdpError) as ei:
            dev.load_pgen_records_phased(0, f)
        assert ei.value.code == pkg.LDP_ERR_UNPHASED and ei.value.variant == first_bad
    dev.set_option("decode_rows", 0)
    dev.load_pgen_records_phased(0, f, 0, first_bad)   # still usable
    dev.close()
    host.close()
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed,miss", [(300, 40, 1, 0.0), (700, 1000, 2, 0.02), (120, 70001, 3, 0.01)])
def test_phase_tracks_of_reference_imported_vcfs(gpu_pkg, tmp_path, m, n, seed, miss):
    """Phased VCFs imported by the reference (--vcf ... --make-pgen: phase tracks with and without explicit phasepresent beside
    every main-track record type), decoded on the device against the host reader; up to 70,001 samples (rows assembled outside LDS)."""
    pkg = gpu_pkg
    raw, pp, pi = T.synth_phased(m, n, seed=seed, missing_rate=miss)
    T.write_vcf(str(tmp_path / "p.vcf"), raw, ["1"] * m, np.arange(m) + 1, phasepresent=(raw == 1), phaseinfo=pi)
    cp = T.run_ref(["--vcf", "p.vcf", "--make-pgen", "--out", "p"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "p.pgen"))
    assert (f.variant_ct, f.sample_ct) == (m, n)
    host = phased_engine(pkg, n, m)
    host.load_genotypes_host(0, f.read_phased(threads=4), pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
    dev = phased_engine(pkg, n, m)
    dev.load_pgen_records_phased(0, f)
    ra, rb = host.variant_recs(), dev.variant_recs()
    for name in ("nm_ct", "sum", "ssq", "flags"):
        assert np.array_equal(ra[name], rb[name]), name
    for v in range(0, m, max(1, m // 40)):
        assert all(np.array_equal(a, b) for a, b in zip(host.planes(v), dev.planes(v))), v
    assert np.array_equal(host.maj_freqs(), dev.maj_freqs())
    assert np.array_equal(host.run(), dev.run())
    host.close()
    dev.close()
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("m,n,seed,max_alt,decode_rows,no_lds", [(260, 50, 1, 3, 0, 0), (400, 333, 2, 6, 7, 0), (150, 1100, 3, 9, 0, 1), (90, 70001, 4, 4, 0, 0)])
def test_phased_multiallelic_records_on_the_device(gpu_pkg, tmp_path, m, n, seed, max_alt, decode_rows, no_lds):
    """--indep-pairphase records with several ALT alleles (PgrGetInv1P -> Get1MP, pgenlib_read.cc:7016, :6962): the collapse on the major allele
    AND the phase bits of the collapsed row in pgen_aux1_kernel -- the phase track counts every het call of the file (ALTx/ALTy ones too), a het
    of the collapsed row takes its bit in the reference's reading (ldtools.pairphase_hap_rows_multiallelic, pinned to the reference's prune lists
    in test_pairphase.py).  Against the rows plink2-hip used to build on the host; some ALTx/ALTy hets between two non-major alleles are left
    unphased in the VCF (explicit phasepresent bits; they collapse to homozygous calls, so the variant still counts as fully phased)."""
    pkg = gpu_pkg
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed, max_alt=max_alt, multi_rate=0.5, missing_rate=0.02 if n < 5000 else 0.001, ld_copy_prob=0.6, redraw=0.1)
    rng = np.random.default_rng(seed + 100)
    unph = np.zeros((m, n), dtype=bool)
    for v in range(m):
        k = int(alt_ct[v]) + 1
        nm = first[v] >= 0
        cnt = [int((first[v][nm] == a).sum() + (second[v][nm] == a).sum()) for a in range(k)]
        maj, _ = T.major_allele_multi(cnt)
        unph[v] = nm & (first[v] != second[v]) & (first[v] != maj) & (second[v] != maj) & (rng.random(n) < 0.5)
    T.write_vcf_haps(str(tmp_path / "p.vcf"), first, second, alt_ct, ["1"] * m, np.arange(m) + 1, unphased=unph)
    cp = T.run_ref(["--vcf", "p.vcf", "--make-pgen", "--out", "p"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "p.pgen"))
    assert (f.variant_ct, f.sample_ct) == (m, n) and f.has_multiallelic
    host = phased_engine(pkg, n, m)
    host.load_genotypes_host(0, f.read_phased(threads=4), pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
    multi = [v for v in range(m) if alt_ct[v] > 1]
    assert unph[multi].any() and len(multi) > m // 4
    for v in multi:
        lo, hi, pp, pi = f.read_alleles_phased(v, int(alt_ct[v]))
        nm = lo != 255
        cnt = [int((lo[nm] == a).sum() + (hi[nm] == a).sum()) for a in range(int(alt_ct[v]) + 1)]
        maj, mf = T.major_allele_multi(cnt)
        a, b, sw = lo.astype(int), hi.astype(int), pi.astype(bool)
        fst, snd = np.where(sw, b, a), np.where(sw, a, b)
        if maj >= 1:
            flip = (a == maj) & (b != maj)
            fst, snd = np.where(flip, snd, fst), np.where(flip, fst, snd)
        assert not (nm & ((a == maj) != (b == maj)) & ~pp.astype(bool)).any()
        hap = np.full(2 * n, 3, dtype=np.uint8)
        hap[1::2] = np.where(nm, np.where(fst != maj, 2, 0), 3)
        hap[0::2] = np.where(nm, np.where(snd != maj, 2, 0), 3)
        row = np.ascontiguousarray(T.pack_2bit(hap[None, :]).view(np.uint8).reshape(1, -1)[:, :(2 * n + 3) // 4])
        host.load_genotypes_host(v, row, pkg.LDP_GENO_INVERSE)
        host.set_maj_freqs(v, [mf])
    dev = phased_engine(pkg, n, m)
    if decode_rows:
        dev.set_option("decode_rows", decode_rows)
    if no_lds:
        dev.set_option("decode_no_lds", 1)
    dev.load_pgen_records_phased(0, f, allele_cts=alt_ct + 1)
    ra, rb = host.variant_recs(), dev.variant_recs()
    for name in ("nm_ct", "sum", "ssq", "flags"):
        assert np.array_equal(ra[name], rb[name]), name
    for v in (multi if n < 5000 else multi[:12]):
        assert all(np.array_equal(x, y) for x, y in zip(host.planes(v), dev.planes(v))), (v, int(alt_ct[v]))
    assert np.array_equal(host.maj_freqs(), dev.maj_freqs())
    assert np.array_equal(host.run(), dev.run())
    host.close()
    dev.close()
    f.close()


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
def test_phased_multiallelic_record_with_an_unphased_het_is_reported(gpu_pkg, tmp_path):
    """... and a het of the COLLAPSED row without phase is the reference's "variant #k is not fully phased" (plink2_ld.cc:2045-2049): the lowest such
    variant comes back, whether its record has one ALT allele (pgen_phase_kernel) or several (pgen_aux1_kernel)."""
    pkg = gpu_pkg
    m, n = 120, 64
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, 9, max_alt=4, multi_rate=0.5, missing_rate=0.0, ld_copy_prob=0.5, redraw=0.2)
    unph = np.zeros((m, n), dtype=bool)
    picked = []
    for want_multi in (True, False):
        for v in range(40 if want_multi else 70, m):
            if (alt_ct[v] > 1) != want_multi:
                continue
            k = int(alt_ct[v]) + 1
            cnt = [int((first[v] == a).sum() + (second[v] == a).sum()) for a in range(k)]
            maj, _ = T.major_allele_multi(cnt)
            hets = np.flatnonzero((first[v] != second[v]) & ((first[v] == maj) != (second[v] == maj)))
            if len(hets):
                unph[v, hets[0]] = True
                picked.append(v)
                break
    assert len(picked) == 2 and picked[0] < picked[1]
    T.write_vcf_haps(str(tmp_path / "p.vcf"), first, second, alt_ct, ["1"] * m, np.arange(m) + 1, unphased=unph)
    cp = T.run_ref(["--vcf", "p.vcf", "--make-pgen", "--out", "p"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    f = pkg.PgenFile(str(tmp_path / "p.pgen"))
    dev = phased_engine(pkg, n, m)
    with pytest.raises(pkg.LdpError) as ei:
        dev.load_pgen_records_phased(0, f, allele_cts=alt_ct + 1)
    assert ei.value.code == pkg.LDP_ERR_UNPHASED and ei.value.variant == picked[0]
    with pytest.raises(pkg.LdpError) as ei:
        dev.load_pgen_records_phased(picked[0] + 1, f, picked[0] + 1, m - picked[0] - 1, allele_cts=(alt_ct + 1)[picked[0] + 1:])
    assert ei.value.code == pkg.LDP_ERR_UNPHASED and ei.value.variant == picked[1]
    dev.load_pgen_records_phased(0, f, 0, picked[0], allele_cts=(alt_ct + 1)[:picked[0]])   # the head is fine
    dev.close()
    f.close()
