"""Sample-mapped rows (ldp_set_sample_map + LDP_GENO_MAPPED): the chrX / chrY / MT rows of plink2_ld.cc:1356-1388 built
on the device, against the same rows built on the host (what plink2-hip did before, pinned against the reference binary
by tests/test_cli.py::test_cli_sex_chromosomes_match_reference) and loaded as LDP_GENO_INVERSE + explicit frequencies."""
import numpy as np
import pytest

import ldtools as T
from test_host_logic import make_positions

pytestmark = pytest.mark.gpu


def host_rows(raw, src, het_missing):
    """numpy restatement of the host builder: (inverse-coded rows, maj_freq)"""
    cols = raw[:, src]                                   # raw REF codes of the engine's columns
    ref_ct = 2 * (cols == 0).sum(1) + (cols == 1).sum(1)
    alt_ct = 2 * (cols == 2).sum(1) + (cols == 1).sum(1)
    tot = ref_ct + alt_ct
    ref_freq = np.full(raw.shape[0], 0.5)
    nz = tot > 0
    ref_freq[nz] = ref_ct[nz].astype(np.float64) * (1.0 / tot[nz].astype(np.float64))
    alt_major = ~(ref_freq >= 0.5)
    mf = np.where(alt_major, np.maximum(1.0 - ref_freq, 0.0), ref_freq)
    out = cols.copy()
    out[(cols == 1) & (het_missing[None, :] != 0)] = 3
    inv = np.array([2, 1, 0, 3], dtype=out.dtype)
    out[alt_major] = inv[out[alt_major]]
    return out, mf, alt_major


@pytest.mark.parametrize("n_raw,males,nonmales,enc", [(90, 31, 29, "ref"), (257, 100, 64, "bed"), (1500, 700, 650, "ref"), (64, 64, 0, "ref"),
                                                       (333, 0, 300, "bed"), (70000, 33000, 36000, "ref")])
def test_mapped_rows_match_host_built_rows(gpu_pkg, n_raw, males, nonmales, enc):
    pkg = gpu_pkg
    rng = np.random.default_rng(n_raw)
    m = 700 if n_raw < 10000 else 96
    raw = T.synth_raw_codes(m, n_raw, seed=n_raw + 1, missing_rate=0.04)
    raw[2] = 1   # all het: the males' calls all become missing
    raw[3] = 3
    perm = rng.permutation(n_raw)
    part1 = np.sort(perm[:males])                         # het -> missing (males on chrX; everyone on chrY / MT)
    part2 = np.sort(perm[males:males + nonmales])         # twice (chrX non-males)
    src = np.concatenate([part1, part2, part2]).astype(np.uint32)
    het = np.concatenate([np.ones(males, np.uint8), np.zeros(2 * nonmales, np.uint8)])
    fct = len(src)
    chr_idx, bps = make_positions(m, 2, 7)
    rows, mf, alt_major = host_rows(raw, src, het)

    a = pkg.LdPruneEngine(fct, 40, 1, False, 0.3, device=0)
    a.set_variants(chr_idx, None)
    a.load_genotypes_host(0, T.pack_2bit(rows), pkg.LDP_GENO_INVERSE)
    a.set_maj_freqs(0, mf)
    want_removed = a.run()
    want_recs = a.variant_recs()

    b = pkg.LdPruneEngine(fct, 40, 1, False, 0.3, device=0)
    b.set_variants(chr_idx, None)
    b.set_sample_map(n_raw, src, het)
    rec = (n_raw + 3) // 4
    if enc == "ref":
        packed = np.ascontiguousarray(T.pack_2bit(raw).view(np.uint8).reshape(m, -1)[:, :rec])
        b.load_genotypes_host(0, packed, pkg.LDP_GENO_REF | pkg.LDP_GENO_MAPPED)
    else:
        lut = np.array([3, 2, 0, 1], dtype=np.uint8)
        packed = np.ascontiguousarray(T.pack_2bit(lut[raw]).view(np.uint8).reshape(m, -1)[:, :rec])
        b.load_genotypes_host(0, packed, pkg.LDP_GENO_BED | pkg.LDP_GENO_MAPPED)
    got_removed = b.run()
    got_recs = b.variant_recs()
    for v in range(m):
        assert np.array_equal(a.planes(v)[0], b.planes(v)[0]), (v, "hom plane")
        assert np.array_equal(a.planes(v)[1], b.planes(v)[1]), (v, "ref2het plane")
        assert (want_recs[v]["nm_ct"], want_recs[v]["sum"], want_recs[v]["ssq"]) == (got_recs[v]["nm_ct"], got_recs[v]["sum"], got_recs[v]["ssq"])
        assert (want_recs[v]["flags"] & 6) == (got_recs[v]["flags"] & 6)
        assert bool(got_recs[v]["flags"] & 1) == bool(alt_major[v])
    assert np.array_equal(b.maj_freqs(), mf)  # exact doubles
    assert np.array_equal(got_removed, want_removed)
    assert 0 < int(want_removed.sum()) < m
    a.close()
    b.close()


def test_mapped_rows_need_a_map(gpu_pkg):
    pkg = gpu_pkg
    e = pkg.LdPruneEngine(10, 5, 1, False, 0.3, device=0)
    e.set_variants(np.zeros(4, dtype=np.uint32), None)
    with pytest.raises(pkg.LdpError):
        e.load_genotypes_host(0, np.zeros((4, 8), dtype=np.uint8), pkg.LDP_GENO_REF | pkg.LDP_GENO_MAPPED)
    with pytest.raises(pkg.LdpError):
        e.set_sample_map(8, np.arange(10, dtype=np.uint32), None)  # sources beyond the file's samples
    e.close()


@pytest.mark.parametrize("n_raw,keep,flag,enc", [(1000, 0.95, 0, "ref"), (1000, 0.9, 1, "bed"), (4099, 0.99, 0, "bed"), (4099, 1.0, 1, "ref"), (50, 0.8, 0, "ref")])
def test_mapped_subset_with_long_runs(gpu_pkg, n_raw, keep, flag, enc):
    """A founder subset with few gaps (or all samples, hets -> missing): the gather's sixteen-neighbours fast path, both
    encodings, window offsets 0..3 within a byte."""
    pkg = gpu_pkg
    rng = np.random.default_rng(n_raw + flag)
    m = 300
    raw = T.synth_raw_codes(m, n_raw, seed=n_raw + 3, missing_rate=0.05)
    raw[1] = 1
    src = np.flatnonzero(rng.random(n_raw) < keep).astype(np.uint32)
    het = np.full(len(src), flag, np.uint8)
    chr_idx, bps = make_positions(m, 2, 9)
    rows, mf, alt_major = host_rows(raw, src, het)
    a = pkg.LdPruneEngine(len(src), 30, 1, False, 0.3, device=0)
    a.set_variants(chr_idx, None)
    a.load_genotypes_host(0, T.pack_2bit(rows), pkg.LDP_GENO_INVERSE)
    a.set_maj_freqs(0, mf)
    b = pkg.LdPruneEngine(len(src), 30, 1, False, 0.3, device=0)
    b.set_variants(chr_idx, None)
    b.set_sample_map(n_raw, src, het)
    rec = (n_raw + 3) // 4
    codes = raw if enc == "ref" else np.array([3, 2, 0, 1], dtype=np.uint8)[raw]
    packed = np.ascontiguousarray(T.pack_2bit(codes).view(np.uint8).reshape(m, -1)[:, :rec])
    b.load_genotypes_host(0, packed, (pkg.LDP_GENO_REF if enc == "ref" else pkg.LDP_GENO_BED) | pkg.LDP_GENO_MAPPED)
    for v in range(m):
        assert np.array_equal(a.planes(v)[0], b.planes(v)[0]), (v, "hom plane")
        assert np.array_equal(a.planes(v)[1], b.planes(v)[1]), (v, "ref2het plane")
    assert np.array_equal(b.maj_freqs(), mf)
    assert np.array_equal(a.run(), b.run())
    a.close()
    b.close()


def test_mapped_inverse_rows(gpu_pkg):
    """LDP_GENO_INVERSE | LDP_GENO_MAPPED: what the reference-side binding loads (PgrGetInv1 rows of all samples)."""
    pkg = gpu_pkg
    n_raw, m = 301, 400
    rng = np.random.default_rng(5)
    inv_all = T.synth_raw_codes(m, n_raw, seed=77, missing_rate=0.05)   # read as counts of the non-major allele
    males = np.sort(rng.choice(n_raw, size=120, replace=False))
    others = np.setdiff1d(np.arange(n_raw), males)[:150]
    src = np.concatenate([males, others, others]).astype(np.uint32)
    het = np.concatenate([np.ones(len(males), np.uint8), np.zeros(2 * len(others), np.uint8)])
    cols = inv_all[:, src].copy()
    cols[(cols == 1) & (het[None, :] != 0)] = 3
    mf = 0.5 + 0.5 * rng.random(m)
    chr_idx, _ = make_positions(m, 2, 3)
    a = pkg.LdPruneEngine(len(src), 25, 1, False, 0.2, device=0)
    a.set_variants(chr_idx, None)
    a.load_genotypes_host(0, T.pack_2bit(cols), pkg.LDP_GENO_INVERSE)
    a.set_maj_freqs(0, mf)
    b = pkg.LdPruneEngine(len(src), 25, 1, False, 0.2, device=0)
    b.set_variants(chr_idx, None)
    b.set_sample_map(n_raw, src, het)
    rec = (n_raw + 3) // 4
    b.load_genotypes_host(0, np.ascontiguousarray(T.pack_2bit(inv_all).view(np.uint8).reshape(m, -1)[:, :rec]), pkg.LDP_GENO_INVERSE | pkg.LDP_GENO_MAPPED)
    b.set_maj_freqs(0, mf)
    for v in range(0, m, 7):
        assert np.array_equal(a.planes(v)[0], b.planes(v)[0]) and np.array_equal(a.planes(v)[1], b.planes(v)[1]), v
    ra, rb = a.run(), b.run()
    assert np.array_equal(ra, rb) and 0 < int(ra.sum()) < m
    a.close()
    b.close()
