"""--indep-pairphase (plink2_ld.cc:1449-2163): the oracle restatement against the reference's recorded prune sets,
the hardcall-phase reader against reference-written .pgen files, and (GPU) the HIP path against the oracle.
Golden files: tests/golden/make_golden_pairphase.py."""
import ctypes
import filecmp
import re
import os
import subprocess

import numpy as np
import pytest

import ldtools as T
import __graft_entry__ as ge

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "pgen")


def _grid(z):
    for k, g in enumerate(z["grid"]):
        win, r2, order = str(g).split("|")
        win = win.split()
        is_bp = win[0].endswith("kb")
        window = int(float(win[0][:-2]) * 1000 * (1 + T.K_SMALL_EPSILON)) if is_bp else int(win[0])
        step = 1 if (is_bp or len(win) < 2) else int(win[1])
        yield k, window, step, is_bp, float(r2), int(order)


def _chr_idx(z):
    return (z["chroms"] - 1).astype(np.uint32)


def test_oracle_pairphase_matches_reference_golden():
    z = np.load(os.path.join(GOLD, "phased_small.npz"))
    rows, mf, unphased, hap_ct = T.oracle_hapsplit(z["raw"], z["phasepresent"], z["phaseinfo"])
    assert not unphased.any()
    for k, window, step, is_bp, r2, order in _grid(z):
        got, _ = T.oracle_indep_pairphase(rows, hap_ct, _chr_idx(z), z["bps"], mf, window, step, is_bp, r2, order)
        assert np.array_equal(got, z["removed_%d" % k]), (k, window, r2, order)


def test_oracle_detects_unphased_hets():
    z = np.load(os.path.join(GOLD, "phased_partial.npz"))
    _, _, unphased, _ = T.oracle_hapsplit(z["raw"], z["phasepresent"], z["phaseinfo"])
    want = ((z["raw"] == 1) & (z["phasepresent"] == 0)).any(axis=1)
    assert np.array_equal(unphased, want)
    assert int(np.argmax(unphased)) == 40 and "variant #40 is not fully phased" in str(z["ref_error"])


def test_haplotype_statistics_are_a_quarter_of_the_genotype_coded_ones():
    """The device carries haplotype h as genotype code 2h (x = 1 - 2h); the reference's nm / sum / dot
    (plink2_ld.cc:1456-1481) then satisfy N*dot_x - S1*S2 = 4*(nm*dot - sum1*sum2), N*ssq - S^2 = 4*sum*(nm-sum)."""
    raw, pp, pi = T.synth_phased(40, 77, seed=3, missing_rate=0.1)
    rows, _, _, hap_ct = T.oracle_hapsplit(raw, pp, pi)
    wc = (hap_ct + 63) // 64
    hap = T.bitmap_to_bool(rows[:, :wc].copy().reshape(-1), wc * 64 * 40).reshape(40, -1)[:, :hap_ct]
    nm = T.bitmap_to_bool(rows[:, wc:].copy().reshape(-1), wc * 64 * 40).reshape(40, -1)[:, :hap_ct]
    codes = np.where(nm, np.where(hap, 2, 0), 3).astype(np.uint8)
    hom, r2h, vaggs = T.oracle_split(T.pack_2bit(codes), hap_ct)
    for a, b in [(0, 1), (3, 9), (10, 39), (20, 21)]:
        g = T.oracle_pair_stats(hom, r2h, vaggs, hap_ct, a, b)
        h = T.oracle_hap_pair_stats(rows, hap_ct, a, b)
        assert g.nm == h.nm
        assert g.nm * g.dot - g.sum1 * g.sum2 == 4 * (h.nm * h.dot - h.sum1 * h.sum2)
        assert g.nm * g.ssq1 - g.sum1 ** 2 == 4 * h.sum1 * (h.nm - h.sum1)
        assert g.nm * g.ssq2 - g.sum2 ** 2 == 4 * h.sum2 * (h.nm - h.sum2)


def _unpack_phased(rows, n):
    cb = (n + 3) // 4
    off = (cb + 3) & ~3
    b = rows[:, :cb]
    out = np.empty((rows.shape[0], cb * 4), dtype=np.uint8)
    for k in range(4):
        out[:, k::4] = (b >> (2 * k)) & 3
    codes = out[:, :n]
    phase = np.unpackbits(rows[:, off:], axis=1, bitorder="little")[:, :n]
    return codes, phase


def test_phase_track_reader_against_reference_written_file():
    pkg = ge.load_package()
    z = np.load(os.path.join(GOLD, "phased_small.npz"))
    m, n = z["raw"].shape
    pg = pkg.PgenFile(os.path.join(GOLD, "phased_small.pgen"))
    assert (pg.variant_ct, pg.sample_ct) == (m, n)
    rows = pg.read_phased()
    codes, phase = _unpack_phased(rows, n)
    assert np.array_equal(codes, z["raw"])
    assert np.array_equal(phase, z["phaseinfo"] & (z["raw"] == 1))
    # ragged ranges, one thread and many
    for first, cnt, threads in [(0, 1, 1), (7, 300, 3), (499, 1, 0), (250, 250, 0)]:
        part = pg.read_phased(first, cnt, threads=threads)
        assert np.array_equal(part, rows[first:first + cnt])
    pg.close()


def test_phase_track_reader_reports_unphased_hets():
    pkg = ge.load_package()
    z = np.load(os.path.join(GOLD, "phased_partial.npz"))
    m, n = z["raw"].shape
    pg = pkg.PgenFile(os.path.join(GOLD, "phased_partial.pgen"))
    rows = pg.read_phased(0, 40)  # the fully phased head
    codes, phase = _unpack_phased(rows, n)
    assert np.array_equal(codes, z["raw"][:40])
    assert np.array_equal(phase, (z["phaseinfo"] & z["phasepresent"])[:40])
    with pytest.raises(pkg.LdpError) as ei:
        pg.read_phased()
    assert ei.value.code == pkg.LDP_ERR_UNPHASED and ei.value.unphased_variant == 40
    # only the samples in the mask matter (the reference checks after founder subsetting)
    unphased = (z["raw"] == 1) & (z["phasepresent"] == 0)
    mask = ~unphased[40:60].any(axis=0)
    if mask.any():
        pg.read_phased(0, 60, sample_mask=mask)
        first_bad = int(np.argmax((unphased & mask[None, :]).any(axis=1))) if (unphased & mask[None, :]).any() else None
        if first_bad is not None:
            with pytest.raises(pkg.LdpError) as ei:
                pg.read_phased(sample_mask=mask)
            assert ei.value.unphased_variant == first_bad
    pg.close()


def _multi_arrays(pkg):
    z = np.load(os.path.join(GOLD, "phased_multi.npz"))
    m, n = z["first"].shape
    pg = pkg.PgenFile(os.path.join(GOLD, "phased_multi.pgen"))
    lo = np.zeros((m, n), dtype=np.uint8)
    hi = np.zeros((m, n), dtype=np.uint8)
    pp = np.zeros((m, n), dtype=np.uint8)
    pi = np.zeros((m, n), dtype=np.uint8)
    for v in range(m):
        lo[v], hi[v], pp[v], pi[v] = pg.read_alleles_phased(v, int(z["alt_ct"][v]))
    return z, pg, lo, hi, pp, pi


def test_multiallelic_phase_reader_against_reference_written_file():
    """allele pairs + phase of every het, multiallelic ones included, from a file the reference imported from a VCF"""
    pkg = ge.load_package()
    z, pg, lo, hi, pp, pi = _multi_arrays(pkg)
    a, b = z["first"].astype(int), z["second"].astype(int)
    assert sum(pg.is_multiallelic(v) for v in range(pg.variant_ct)) > 50
    assert np.array_equal(lo, np.where(a < 0, 255, np.minimum(a, b)))
    assert np.array_equal(hi, np.where(a < 0, 255, np.maximum(a, b)))
    het = (a != b) & (a >= 0)
    assert np.array_equal(pp.astype(bool), het)
    assert np.array_equal(pi.astype(bool), het & (a > b))  # set = the higher allele on the first haplotype
    # the bulk reader hands multiallelic records through with their main-track codes and no phase
    rows = pg.read_phased()
    codes, phase = _unpack_phased(rows, pg.sample_ct)
    for v in range(pg.variant_ct):
        want = np.where(lo[v] == 255, 3, np.minimum(lo[v], 1) + np.minimum(hi[v], 1))
        assert np.array_equal(codes[v], want)
        if pg.is_multiallelic(v):
            assert not phase[v].any()
        else:
            assert np.array_equal(phase[v].astype(bool), het[v] & (a[v] > b[v]))
    pg.close()


def test_multiallelic_phase_reader_partially_phased_file():
    """explicit phasepresent bits over all hets, multiallelic ones included (pgen_spec.tex:541-562)"""
    pkg = ge.load_package()
    z = np.load(os.path.join(GOLD, "phased_multi_partial.npz"))
    pg = pkg.PgenFile(os.path.join(GOLD, "phased_multi_partial.pgen"))
    a, b = z["first"].astype(int), z["second"].astype(int)
    het = (a != b) & (a >= 0)
    seen_explicit = 0
    for v in range(pg.variant_ct):
        lo, hi, pp, pi = pg.read_alleles_phased(v, int(z["alt_ct"][v]))
        assert np.array_equal(lo, np.where(a[v] < 0, 255, np.minimum(a[v], b[v])))
        assert np.array_equal(hi, np.where(a[v] < 0, 255, np.maximum(a[v], b[v])))
        want_pp = het[v] & ~z["unphased"][v]
        assert np.array_equal(pp.astype(bool), want_pp), v
        assert np.array_equal(pi.astype(bool), want_pp & (a[v] > b[v])), v
        seen_explicit += int(z["unphased"][v].any() and want_pp.any())
    assert seen_explicit > 30
    pg.close()


def test_oracle_multiallelic_collapse_pairwise_matches_reference_golden():
    """--indep-pairwise on multiallelic sites: major allele by GetMajIdxMulti, genotype = copies of non-major alleles
    (PgrGetInv1 -> Get1Multiallelic), CPU-only against lists recorded from the reference"""
    pkg = ge.load_package()
    z, pg, lo, hi, pp, pi = _multi_arrays(pkg)
    pg.close()
    m, n = lo.shape
    codes = np.zeros((m, n), dtype=np.uint8)
    mf = np.zeros(m)
    for v in range(m):
        k = int(z["alt_ct"][v]) + 1
        nm = lo[v] != 255
        cnt = [int((lo[v][nm] == a).sum() + (hi[v][nm] == a).sum()) for a in range(k)]
        maj, mf[v] = T.major_allele_multi(cnt)
        codes[v] = np.where(nm, (lo[v] != maj).astype(np.uint8) + (hi[v] != maj).astype(np.uint8), 3)
    for k, window, step, is_bp, r2, order in _grid(z):
        got, _ = T.oracle_indep_pairwise(T.pack_2bit(codes), n, _chr_idx(z), z["bps"], mf, window, step, is_bp, r2, order)
        assert np.array_equal(got, z["removed_wise_%d" % k]), (k, window, r2, order)


def test_oracle_indep_preferred_matches_reference_golden():
    """--indep-preferred: listed variants enter the scan with (major frequency - 1) (plink2_ld.cc:916-918)"""
    z = np.load(os.path.join(GOLD, "preferred.npz"))
    raw = z["raw"]
    inv, mf, _ = T.oracle_prepare(raw)
    differs = False
    for k, window, step, is_bp, r2, order in _grid(z):
        got, _ = T.oracle_indep_pairwise(inv, raw.shape[1], _chr_idx(z), z["bps"], mf - z["preferred"].astype(np.float64), window, step, is_bp, r2, order)
        assert np.array_equal(got, z["removed_%d" % k]), (k, window, r2, order)
        assert not (got & z["preferred"] & ~z["removed_plain_%d" % k]).all()
        differs |= not np.array_equal(z["removed_%d" % k], z["removed_plain_%d" % k])
    assert differs  # (the list changed the outcome, so the test exercises something)


def test_oracle_sex_chromosome_layouts_match_reference_golden():
    """chrX / chrY / MT as the reference's loaders shape them (ldtools.sex_chromosome_rows), through the oracle scan,
    against prune lists recorded from the reference for BOTH --indep-pairwise and --indep-pairphase (CPU-only pin of
    what the GPU tests check end to end)."""
    z = np.load(os.path.join(GOLD, "sexed_phased.npz"))
    raw, pi, chroms, bps = z["raw"], z["phaseinfo"], [str(c) for c in z["chroms"]], z["bps"]
    founders, sexes = z["founders"], z["sexes"]
    n_f = int(founders.sum())
    for k, window, step, is_bp, r2, order in _grid(z):
        for mode in ("wise", "phase"):
            want = z["removed_%s_%d" % (mode, k)]
            got = np.zeros(len(chroms), dtype=bool)
            for name in dict.fromkeys(chroms):
                vs = np.array([i for i, c in enumerate(chroms) if c == name])
                zeros = np.zeros(len(vs), dtype=np.uint32)
                if name in ("X", "Y", "MT"):
                    codes, mf = T.sex_chromosome_rows(raw[vs], founders, sexes, name, phaseinfo=pi[vs] if (mode == "phase" and name == "X") else None)
                    if mode == "phase":
                        res, _ = T.oracle_indep_pairphase(T.haploid_codes_to_hap_rows(codes), codes.shape[1], zeros, bps[vs], mf, window, step, is_bp, r2, order)
                    else:
                        # the virtual-sample row is already major-allele-inverse up to orientation: r^2 does not care
                        res, _ = T.oracle_indep_pairwise(T.pack_2bit(codes), codes.shape[1], zeros, bps[vs], mf, window, step, is_bp, r2, order)
                elif mode == "phase":
                    rows, mf, unphased, hap_ct = T.oracle_hapsplit(raw[vs][:, founders], (raw[vs][:, founders] == 1).astype(np.uint8), pi[vs][:, founders])
                    assert not unphased.any()
                    res, _ = T.oracle_indep_pairphase(rows, hap_ct, zeros, bps[vs], mf, window, step, is_bp, r2, order)
                else:
                    inv, mf, _ = T.oracle_prepare(raw[vs][:, founders])
                    res, _ = T.oracle_indep_pairwise(inv, n_f, zeros, bps[vs], mf, window, step, is_bp, r2, order)
                got[vs] = res
            assert np.array_equal(got, want), (mode, k, np.flatnonzero(got != want)[:10])


def test_portable_bit_deposit_path():
    """the phase decoder uses pext/pdep when the host has BMI2; ldp_pgen_debug_force_portable(1) -- which tests/conftest.py calls when
    LDTEST_PGEN_PORTABLE is set: the library itself reads no environment -- forces the portable loops"""
    import sys
    env = dict(os.environ, LDTEST_PGEN_PORTABLE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "not gpu", "-k", "reader", "-p", "no:cacheprovider"],
                       env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-800:]


def test_oracle_pairphase_multiallelic_collapse_matches_reference_golden():
    """PgrGetInv1P -> Get1MP on multiallelic variants: the collapse on the major allele and the reference's reading of
    phaseinfo (ldtools.pairphase_hap_rows_multiallelic).  The physically right haplotype assignment does NOT give the
    reference's lists -- recorded here so that nobody 'fixes' it."""
    pkg = ge.load_package()
    z, pg, lo, hi, pp, pi = _multi_arrays(pkg)
    pg.close()
    n = lo.shape[1]
    rows, mf, unphased = T.pairphase_hap_rows_multiallelic(lo, hi, pp, pi, z["alt_ct"])
    rows_phys, _, _ = T.pairphase_hap_rows_multiallelic(lo, hi, pp, pi, z["alt_ct"], quirk=False)
    assert not unphased.any()
    physical_differs = False
    for k, window, step, is_bp, r2, order in _grid(z):
        got, _ = T.oracle_indep_pairphase(rows, 2 * n, _chr_idx(z), z["bps"], mf, window, step, is_bp, r2, order)
        assert np.array_equal(got, z["removed_%d" % k]), (k, window, r2, order)
        phys, _ = T.oracle_indep_pairphase(rows_phys, 2 * n, _chr_idx(z), z["bps"], mf, window, step, is_bp, r2, order)
        physical_differs |= not np.array_equal(phys, z["removed_%d" % k])
    assert physical_differs


# ---------------------------------------------------------------------------------------------------------- GPU
def _engine_rows(pkg, raw, pi):
    return pkg.pack_phased_rows(T.pack_2bit(raw).view(np.uint8).reshape(raw.shape[0], -1), pi & (raw == 1), raw.shape[1])


@pytest.mark.gpu
def test_hip_pairphase_matches_reference_golden():
    pkg = ge.load_package()
    z = np.load(os.path.join(GOLD, "phased_small.npz"))
    m, n = z["raw"].shape
    rows = _engine_rows(pkg, z["raw"], z["phaseinfo"])
    for k, window, step, is_bp, r2, order in _grid(z):
        eng = pkg.LdPruneEngine(2 * n, window, step, is_bp, r2, order=order, device=0)
        eng.set_variants(_chr_idx(z), z["bps"])
        eng.load_genotypes_host(0, rows, pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
        got = eng.run()
        eng.close()
        assert np.array_equal(got, z["removed_%d" % k]), (k, window, r2, order)


@pytest.mark.gpu
@pytest.mark.parametrize("n,miss", [(33, 0.0), (100, 0.03), (257, 0.0), (1000, 0.01), (2049, 0.0)])
def test_hip_pairphase_matches_oracle(n, miss):
    pkg = ge.load_package()
    m = 400
    raw, pp, pi = T.synth_phased(m, n, seed=1000 + n, missing_rate=miss)
    raw[5] = 2
    rng = np.random.default_rng(n)
    chr_idx = np.sort(rng.integers(0, 2, size=m)).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(2):
        sel = np.where(chr_idx == c)[0]
        bps[sel] = np.sort(rng.integers(1, 50000, size=len(sel)))
    hrows, mf, unphased, hap_ct = T.oracle_hapsplit(raw, (raw == 1).astype(np.uint8), pi)
    assert not unphased.any()
    rows = _engine_rows(pkg, raw, pi)
    for window, step, is_bp, r2, order in [(60, 7, False, 0.3, 2), (60, 7, False, 0.3, 1), (8000, 1, True, 0.5, 2), (8000, 1, True, 0.1, 1)]:
        want, _ = T.oracle_indep_pairphase(hrows, hap_ct, chr_idx, bps, mf, window, step, is_bp, r2, order)
        eng = pkg.LdPruneEngine(hap_ct, window, step, is_bp, r2, order=order, device=0)
        eng.set_variants(chr_idx, bps)
        # several uneven pieces
        for a, b in [(0, 123), (123, 124), (124, m)]:
            eng.load_genotypes_host(a, rows[a:b], pkg.LDP_GENO_REF | pkg.LDP_GENO_PHASED)
        got = eng.run()
        assert np.array_equal(eng.maj_freqs(), mf)
        eng.close()
        assert np.array_equal(got, want), (n, miss, window, r2, order)
    # the seam's own form: major-allele-inverse codes + phaseinfo re-oriented with them (PgrGetInv1P), caller-supplied
    # frequencies, rows resident in device memory at an odd stride
    import torch
    inv, mf2, altmaj = T.oracle_prepare(raw)
    pi_inv = np.where(altmaj[:, None].astype(bool), pi ^ 1, pi) & (raw == 1)
    rows_inv = pkg.pack_phased_rows(inv.view(np.uint8).reshape(m, -1), pi_inv, n)
    stride = rows_inv.shape[1] + 3
    padded = np.zeros((m, stride), dtype=np.uint8)
    padded[:, :rows_inv.shape[1]] = rows_inv
    dev = torch.from_numpy(padded).cuda()
    torch.cuda.synchronize()
    eng = pkg.LdPruneEngine(hap_ct, 60, 7, False, 0.3, order=2, device=0)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_device(0, m, dev.data_ptr(), stride, pkg.LDP_GENO_INVERSE | pkg.LDP_GENO_PHASED)
    eng.set_maj_freqs(0, mf2)
    got = eng.run()
    eng.close()
    want, _ = T.oracle_indep_pairphase(hrows, hap_ct, chr_idx, bps, mf, 60, 7, False, 0.3, 2)
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------------------- CLI
import filecmp  # noqa: E402
import subprocess  # noqa: E402


def _cli(pkg):
    path = pkg.build_cli()
    assert path and os.path.exists(path)
    return path


def _run(cli, args, cwd):
    return subprocess.run([cli] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)


def _phased_fileset(tmp, m, n, seed, chrom_plan, sexes=None, nonfounders=0, unphased_rate=0.0, missing=0.03):
    """phased VCF -> the reference's import -> <tmp>/p.{pgen,pvar,psam}; .psam rewritten with parents / sex"""
    raw, pp, pi = T.synth_phased(m, n, seed, missing_rate=missing)
    if unphased_rate:
        drop = np.random.default_rng(seed + 1).random(raw.shape) < unphased_rate
        drop[:m // 3] = False
        pp = (pp.astype(bool) & ~drop).astype(np.uint8)
        pi = pi & pp
    chroms, bps = [], []
    for name, cnt in chrom_plan:
        chroms += [name] * cnt
        bps += list((3000000 if name == "X" else 1000) + 173 * np.arange(cnt))  # chrX: clear of PAR1
    assert len(chroms) == m
    T.write_vcf(os.path.join(tmp, "p.vcf"), raw, chroms, np.array(bps), pp, pi)
    lines = ["#IID\tPAT\tMAT\tSEX"]
    for s in range(n):
        nf = (s % 11 == 3) and (s // 11 < nonfounders)
        sx = "NA" if (sexes is None or sexes[s] == 0) else str(sexes[s])
        lines.append("s%d\t%s\t%s\t%s" % (s, "s0" if nf else "0", "s1" if nf else "0", sx))
    open(os.path.join(tmp, "in.psam"), "w").write("\n".join(lines) + "\n")
    T.ref_import_vcf(os.path.join(tmp, "p.vcf"), os.path.join(tmp, "p"), extra=["--allow-extra-chr", "--psam", "in.psam"])
    return raw, pp, pi


def test_cli_pairphase_flag_parses(tmp_path):
    pkg = ge.load_package()
    cli = _cli(pkg)
    z = np.load(os.path.join(GOLD, "phased_small.npz"))
    m, n = z["raw"].shape
    # a fileset around the reference-written phased .pgen
    import shutil
    shutil.copy(os.path.join(GOLD, "phased_small.pgen"), str(tmp_path / "g.pgen"))
    with open(str(tmp_path / "g.pvar"), "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for v in range(m):
            f.write("%d\t%d\tsnp%d\tA\tC\n" % (z["chroms"][v], z["bps"][v], v))
    with open(str(tmp_path / "g.psam"), "w") as f:
        f.write("#IID\tSEX\n" + "".join("s%d\tNA\n" % s for s in range(n)))
    r = _run(cli, ["--pfile", "g", "--indep-pairphase", "20kb", "0.5", "--dry-run"], str(tmp_path))
    assert r.returncode == 0, r.stdout
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("dry-run:")][0]
    assert "founders=%d " % n in line and "window=20000 " in line and "window_is_bp=1" in line
    r = _run(cli, ["--pfile", "g", "--indep-pairphase", "50", "60", "0.5", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "--indep-pairphase window-increment" in r.stdout
    r = _run(cli, ["--pfile", "g", "--indep-pairphase", "50", "5", "0.5", "--indep-pairwise", "50", "5", "0.5", "--dry-run"], str(tmp_path))
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,r2,order,nonfounders", [(["30kb"], "0.5", 2, 0), (["60", "7"], "0.2", 1, 5), (["100", "1"], "0.8", 2, 5)])
def test_cli_pairphase_autosomes_match_reference(tmp_path, wargs, r2, order, nonfounders):
    assert T.have_ref()
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    _phased_fileset(tmp, 700, 131, seed=11 + order, chrom_plan=[("0", 3), ("1", 300), ("2", 250), ("7", 147)], nonfounders=nonfounders)
    common = ["--pfile", "p", "--indep-pairphase"] + wargs + [r2] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = _run(cli, common + ["--out", "hip"], tmp)
    assert got.returncode == 0, got.stdout
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), ext
    assert 0 < len(open(os.path.join(tmp, "hip.prune.out")).read().split()) < 697
    assert "Ignoring 3 chromosome 0 variants" in got.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,order,unknown", [(["40kb"], 2, True), (["70", "9"], 1, False)])
def test_cli_pairphase_sex_chromosomes_match_reference(tmp_path, wargs, order, unknown):
    """chrX: male founders one haplotype each (hets missing), non-males two, split by phase; chrY / MT haploid
    (plink2_ld.cc:2040-2110)."""
    assert T.have_ref()
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    n = 140
    rng = np.random.default_rng(5)
    sexes = rng.choice([1, 2, 0] if unknown else [1, 2], size=n, p=[0.45, 0.45, 0.1] if unknown else [0.5, 0.5])
    founders = np.array([not ((s % 11 == 3) and (s // 11 < 6)) for s in range(n)])
    while T.ref_pairphase_chrx_is_unreliable(sexes, founders):  # a layout where the reference itself is not reproducible
        sexes[np.flatnonzero(founders & (sexes != 1))[0]] = 1
    _phased_fileset(tmp, 900, n, seed=21, chrom_plan=[("1", 180), ("2", 180), ("X", 180), ("Y", 180), ("MT", 180)], sexes=sexes, nonfounders=6)
    common = ["--pfile", "p", "--indep-pairphase"] + wargs + ["0.3"] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = _run(cli, common + ["--out", "hip"], tmp)
    assert got.returncode == 0, got.stdout
    ref_out = open(os.path.join(tmp, "ref.prune.out")).read().split()
    hip_out = open(os.path.join(tmp, "hip.prune.out")).read().split()
    diff = sorted(set(ref_out) ^ set(hip_out), key=lambda x: int(x[3:]))
    assert not diff, "differs on %d variants, e.g. %s" % (len(diff), diff[:10])
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), ext


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,r2,order", [(["30kb"], "0.5", 2), (["60", "7"], "0.2", 1)])
def test_cli_pairphase_multiallelic_matches_reference(tmp_path, wargs, r2, order):
    """multiallelic + phased input (collapse on the major allele with phase, Get1MP) next to biallelic variants"""
    assert T.have_ref()
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    m, n = 600, 131
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed=40 + order, max_alt=4)
    chroms = ["1"] * 250 + ["2"] * 200 + ["9"] * 150
    bps = np.concatenate([1000 + 151 * np.arange(250), 1000 + 151 * np.arange(200), 1000 + 151 * np.arange(150)])
    T.write_vcf_haps(os.path.join(tmp, "p.vcf"), first, second, alt_ct, chroms, bps)
    lines = ["#IID\tPAT\tMAT\tSEX"] + ["s%d\t%s\t%s\tNA" % (s, "s0" if s in (14, 47) else "0", "s1" if s in (14, 47) else "0") for s in range(n)]
    open(os.path.join(tmp, "in.psam"), "w").write("\n".join(lines) + "\n")
    T.ref_import_vcf(os.path.join(tmp, "p.vcf"), os.path.join(tmp, "p"), extra=["--psam", "in.psam"])
    common = ["--pfile", "p", "--indep-pairphase"] + wargs + [r2] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = _run(cli, common + ["--out", "hip"], tmp)
    assert got.returncode == 0, got.stdout
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), ext
    # an unphased het at a multiallelic site is refused like the reference does
    unph = np.zeros(first.shape, dtype=bool)
    v_bad = int(np.flatnonzero(alt_ct > 1)[5])
    s_bad = int(np.flatnonzero((first[v_bad] != second[v_bad]) & (first[v_bad] >= 0))[0])
    unph[v_bad, s_bad] = True
    T.write_vcf_haps(os.path.join(tmp, "u.vcf"), first, second, alt_ct, chroms, bps, unphased=unph)
    T.ref_import_vcf(os.path.join(tmp, "u.vcf"), os.path.join(tmp, "u"))
    refu = T.run_ref(["--pfile", "u", "--indep-pairphase", "50", "5", "0.5", "--out", "refu"], tmp)
    gotu = _run(cli, ["--pfile", "u", "--indep-pairphase", "50", "5", "0.5", "--out", "hipu"], tmp)
    assert refu.returncode == gotu.returncode, (refu.stdout[-300:], gotu.stdout[-300:])
    assert [ln for ln in refu.stdout.splitlines() if ln.startswith("Error")] == [ln for ln in gotu.stdout.splitlines() if ln.startswith("Error")]


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,r2,order", [(["20kb"], "0.4", 2), (["80", "9"], "0.15", 1)])
def test_cli_pairphase_multiallelic_on_the_device_matches_reference(tmp_path, wargs, r2, order):
    """... and with every sample a founder no row is built on the host any more (round 5): main track, multiallelic collapse and phase bits all come
    from ldp_load_pgen_records_phased -- the front-end says so under --timing --, and --debug-host-decode gives the same lists the old way."""
    assert T.have_ref()
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    m, n = 900, 157
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed=60 + order, max_alt=6, multi_rate=0.4)
    chroms = ["1"] * 500 + ["3"] * 400
    bps = np.concatenate([1000 + 97 * np.arange(500), 1000 + 97 * np.arange(400)])
    T.write_vcf_haps(os.path.join(tmp, "p.vcf"), first, second, alt_ct, chroms, bps)
    T.ref_import_vcf(os.path.join(tmp, "p.vcf"), os.path.join(tmp, "p"))
    common = ["--pfile", "p", "--indep-pairphase"] + wargs + [r2] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = _run(cli, common + ["--timing", "--out", "hip"], tmp)
    assert got.returncode == 0, got.stdout
    line = re.search(r"host-built rows: (\d+) multiallelic \((\d+) more have REF as the major allele: main track as loaded; (\d+) collapsed on the device\)", got.stdout)
    assert line and int(line.group(1)) == 0 and int(line.group(3)) == int((alt_ct > 1).sum()), got.stdout[-600:]
    old = _run(cli, common + ["--debug-host-decode", "--timing", "--out", "old"], tmp)
    assert old.returncode == 0 and re.search(r"host-built rows: (\d+) multiallelic", old.stdout).group(1) == str(int((alt_ct > 1).sum()))
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), ext
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "old" + ext), shallow=False), ext


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,r2,order,unknown,nonfounders", [(["25kb"], "0.3", 2, True, False), (["70", "11"], "0.15", 1, False, True), (["40kb"], "0.6", 2, False, False)])
def test_cli_pairphase_multiallelic_on_sex_chromosomes_matches_reference(tmp_path, wargs, r2, order, unknown, nonfounders):
    """Variants with several ALT alleles on chrX / chrY / MT under --indep-pairphase (refused with exit 63 until round 5): the major allele from the
    chromosome's own allele-frequency weights, chrX's non-male founders as two haplotypes split in Get1MP's reading, everybody else one haplotype
    with het calls missing (plink2_ld.cc:2040-2097)."""
    assert T.have_ref()
    from test_clump import multiallelic_clump_fileset
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    m, n = 1000, 143
    names = ["2", "X", "Y", "MT"]
    alt_ct, chroms, _ = multiallelic_clump_fileset(tmp_path, m, n, 90 + order, chrom_of=lambda v: names[(4 * v) // m], max_alt=5, multi_rate=0.4)
    rng = np.random.default_rng(17)
    lines = ["#IID\tPAT\tMAT\tSEX"]
    for s in range(n):
        sx = str(int(rng.integers(1, 3)))
        if unknown and rng.random() < 0.12:
            sx = "NA"
        par = ("s0", "s1") if (nonfounders and s % 9 == 5) else ("0", "0")
        lines.append("s%d\t%s\t%s\t%s" % (s, par[0], par[1], sx))
    sexes_chk = np.array([0 if ln.split("\t")[3] == "NA" else int(ln.split("\t")[3]) for ln in lines[1:]])
    founders_chk = np.array([ln.split("\t")[1] == "0" for ln in lines[1:]])
    assert not T.ref_pairphase_chrx_is_unreliable(sexes_chk, founders_chk), "a sample layout on which the reference itself is not reproducible (ldtools)"
    open(os.path.join(tmp, "d.psam"), "w").write("\n".join(lines) + "\n")
    common = ["--pfile", "d", "--indep-pairphase"] + wargs + [r2] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = _run(cli, common + ["--out", "hip"], tmp)
    assert got.returncode == 0, got.stdout
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), ext
    ids_out = set(open(os.path.join(tmp, "hip.prune.out")).read().split())
    multi_ids = {"snp%d" % v for v in range(m) if alt_ct[v] > 1 and chroms[v] != "2"}
    assert len(ids_out & multi_ids) > 10 and len(multi_ids - ids_out) > 10, "multiallelic variants on the sex chromosomes on both lists"


@pytest.mark.gpu
def test_cli_pairphase_refuses_partially_phased_like_reference(tmp_path):
    assert T.have_ref()
    pkg = ge.load_package()
    cli = _cli(pkg)
    tmp = str(tmp_path)
    _phased_fileset(tmp, 300, 90, seed=31, chrom_plan=[("1", 300)], unphased_rate=0.2)
    common = ["--pfile", "p", "--indep-pairphase", "50", "5", "0.5"]
    ref = T.run_ref(common + ["--out", "ref"], tmp)
    got = _run(cli, common + ["--out", "hip"], tmp)
    ref_err = [ln for ln in ref.stdout.splitlines() if ln.startswith("Error")]
    hip_err = [ln for ln in got.stdout.splitlines() if ln.startswith("Error")]
    assert ref.returncode == got.returncode == 7, (ref.returncode, got.returncode, got.stdout)
    assert ref_err == hip_err and "is not fully phased" in hip_err[0]
    # .bed input: every het is unphased
    T.run_ref(["--pfile", "p", "--make-bed", "--out", "b"], tmp)
    refb = T.run_ref(["--bfile", "b", "--indep-pairphase", "50", "5", "0.5", "--out", "refb"], tmp)
    gotb = _run(cli, ["--bfile", "b", "--indep-pairphase", "50", "5", "0.5", "--out", "hipb"], tmp)
    assert refb.returncode == gotb.returncode == 7
    assert [ln for ln in refb.stdout.splitlines() if ln.startswith("Error")] == [ln for ln in gotb.stdout.splitlines() if ln.startswith("Error")]
