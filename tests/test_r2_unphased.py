"""--r2-unphased matrices (BASELINE config 4 path): the tile kernel's matrix mode against the oracle's
integer statistics (r^2 doubles must be bit-identical: same integers, same IEEE operations as ComputeR2,
plink2_ld.cc:6654-6682), against the golden doubles recorded from the reference, and -- through plink2-hip --
byte-for-byte against the files the reference binary writes."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import ldtools as T
from test_golden import GOLDEN, load

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "plink2")

pytestmark = pytest.mark.gpu


def oracle_r2_lower(raw):
    m, n = raw.shape
    inv, mf, _ = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    out = np.zeros((m, m), dtype=np.float64)
    for j in range(m):
        for i in range(j + 1):
            st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
            cov, v1, v2 = T.oracle_r2(st)
            prod = v1 * v2
            out[j, i] = np.nan if (st.nm == 0 or prod == 0.0) else cov * cov / prod
    return out


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32),
                          np.ascontiguousarray(b).view(np.uint64 if b.dtype == np.float64 else np.uint32))


@pytest.mark.parametrize("m,n,miss", [(70, 90, 0.0), (150, 130, 0.05), (300, 40, 0.2), (45, 1100, 0.01)])
def test_matrix_rows_match_oracle(gpu_pkg, m, n, miss):
    pkg = gpu_pkg
    raw = T.synth_raw_codes(m, n, seed=m + n, missing_rate=miss)
    raw[3] = 0   # monomorphic: undefined r^2 (NaN) against everything, incl. itself
    raw[5] = 3   # all missing
    want = oracle_r2_lower(raw)
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_matrix(m)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    got = eng.r2_unphased_rows()
    il = np.tril_indices(m)
    nan_w, nan_g = np.isnan(want[il]), np.isnan(got[il])
    assert np.array_equal(nan_w, nan_g)
    assert np.array_equal(want[il][~nan_w], got[il][~nan_g])            # exact doubles
    assert (got[np.triu_indices(m, 1)] == 0).all()
    # NaN carries the reference's bit pattern
    assert set(got[il][nan_g].view(np.uint64)) <= {0xfff8000000000000}
    # row chunks + float output
    r0, cnt = m // 3, min(20, m - m // 3)
    part = eng.r2_unphased_rows(r0, cnt, as_float=True)
    ref32 = want[r0:r0 + cnt, :r0 + cnt].astype(np.float32)
    for q in range(cnt):
        a, b = part[q, :r0 + q + 1], ref32[q, :r0 + q + 1]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    eng.close()


def test_matrix_matches_golden_reference_doubles(gpu_pkg):
    pkg = gpu_pkg
    for name in ("kat_missing.npz", "kat_quirk_o2.npz"):
        g = load([p for p in GOLDEN if p.endswith(name)][0])
        eng = pkg.LdPruneEngine(g["n"], 2, 1, False, 0.5, device=0)
        eng.set_variants_matrix(g["m"])
        eng.load_genotypes_host(0, T.pack_2bit(g["raw"]), pkg.LDP_GENO_REF)
        got = eng.r2_unphased_rows()
        il = np.tril_indices(g["m"])
        assert same_bits(got[il], g["r2_square"][il])
        eng.close()


@pytest.mark.parametrize("shape,enc", [("square", "bin"), ("square0", "bin4"), ("triangle", "bin"), ("triangle", "bin4"),
                                       ("yes-really", "bin4")])  # (an encoding without a shape: square, plink2_help.cc:1015)
def test_cli_matrix_files_byte_identical(gpu_pkg, tmp_path, shape, enc):
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 700, 150
    raw = T.synth_raw_codes(m, n, seed=9, missing_rate=0.03)
    raw[10] = 2
    raw[11] = 3
    chroms = ["0"] * 5 + ["1"] * 400 + ["7"] * 295
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, np.arange(m) + 1)
    ref = T.run_ref(["--pfile", "d", "--r2-unphased", shape, enc, "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased", shape, enc, "--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.unphased.vcor2.bin.vars"), str(tmp_path / "hip.unphased.vcor2.bin.vars"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.unphased.vcor2.bin"), str(tmp_path / "hip.unphased.vcor2.bin"), shallow=False)


@pytest.mark.parametrize("bp_radius,var_radius", [(3000, 0x7fffffff), (100000, 6), (900, 3), (0, 0x7fffffff)])
def test_band_rows_match_oracle(gpu_pkg, bp_radius, var_radius):
    """Windowed plan of the --r2-unphased table: the band of r^2 doubles against the oracle, window membership
    against UpdateVcorWindow's rule (bp[B] - bp[A] <= bp_radius, B - A <= var_ct_radius, same chromosome)."""
    pkg = gpu_pkg
    m, n = 260, 150
    raw = T.synth_raw_codes(m, n, seed=77, missing_rate=0.04)
    raw[7] = 0
    rng = np.random.default_rng(3)
    chr_idx = np.repeat(np.arange(4), [100, 1, 80, 79]).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(4):
        sel = np.where(chr_idx == c)[0]
        bps[sel] = np.sort(rng.integers(1, 40000, size=len(sel)))
    want = oracle_r2_lower(raw)
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_vcor(chr_idx, bps, bp_radius, var_radius)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    lo, cand = eng.band()
    for j in range(m):
        i = j
        while i > 0 and chr_idx[i - 1] == chr_idx[j] and int(bps[j]) - int(bps[i - 1]) <= bp_radius and j - (i - 1) <= var_radius:
            i -= 1
        assert lo[j] == i, (j, lo[j], i)
    for r0, cnt in ((0, m), (37, 101), (m - 5, 5)):
        got = eng.r2_unphased_band_rows(r0, cnt)
        k = 0
        for j in range(r0, r0 + cnt):
            for i in range(int(lo[j]), j):
                w, g = want[j, i], got[k]
                assert (np.isnan(w) and np.isnan(g)) or w == g, (i, j, w, g)
                k += 1
        assert k == len(got)
    eng.close()


VCOR_CASES = [
    [],
    ["--ld-window-kb", "5", "--ld-window-r2", "0.05"],
    ["--ld-window", "7", "--ld-window-r2", "0"],
    ["--ld-window-kb", "0.4", "--ld-window", "3", "--ld-window-r2", "0.5"],
]


@pytest.mark.parametrize("extra", VCOR_CASES)
def test_cli_vcor_table_byte_identical(gpu_pkg, tmp_path, extra):
    """--r2-unphased without a matrix shape: the .vcor table (default columns) against the reference's file."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 900, 160
    raw = T.synth_raw_codes(m, n, seed=19, missing_rate=0.02)
    raw[10] = 2
    raw[11] = 3
    chroms = ["0"] * 4 + ["1"] * 500 + ["3"] * 1 + ["7"] * 395
    rng = np.random.default_rng(8)
    pos = np.concatenate([np.arange(4) + 1, np.sort(rng.integers(1, 60000, 500)), [5], np.sort(rng.integers(1, 2000000, 395))])
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, pos)
    ref = T.run_ref(["--pfile", "d", "--r2-unphased"] + extra + ["--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased"] + extra + ["--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    assert os.path.getsize(str(tmp_path / "ref.vcor")) > 100
    assert filecmp.cmp(str(tmp_path / "ref.vcor"), str(tmp_path / "hip.vcor"), shallow=False)


@pytest.mark.parametrize("shape", ["square", "square0", "triangle"])
def test_cli_text_matrix_byte_identical(gpu_pkg, tmp_path, shape):
    """--r2-unphased <shape> without bin/bin4: the tab-delimited text matrix (dtoa_g formatting)."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 230, 140
    raw = T.synth_raw_codes(m, n, seed=29, missing_rate=0.03)
    raw[10] = 2
    raw[11] = 3
    chroms = ["0"] * 3 + ["1"] * 127 + ["9"] * 100
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, np.arange(m) + 1)
    ref = T.run_ref(["--pfile", "d", "--r2-unphased", shape, "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased", shape, "--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.unphased.vcor2.vars"), str(tmp_path / "hip.unphased.vcor2.vars"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.unphased.vcor2"), str(tmp_path / "hip.unphased.vcor2"), shallow=False)


@pytest.mark.parametrize("extra", [[], ["--ld-window-r2", "0.02"], ["--ld-window-r2", "0"], ["--ld-window-r2", "0.7"]])
def test_cli_inter_chr_table_byte_identical(gpu_pkg, tmp_path, extra):
    """--r2-unphased inter-chr (the command BASELINE config 4 names): every pair A < B across and within chromosomes,
    chromosome 0 included, filtered by --ld-window-r2 only (plink2_ld.cc:11082-11116)."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 700, 150
    raw = T.synth_raw_codes(m, n, seed=23, missing_rate=0.02)
    raw[10] = 2
    raw[11] = 3
    chroms = ["0"] * 4 + ["1"] * 300 + ["3"] * 1 + ["7"] * 395
    rng = np.random.default_rng(9)
    pos = np.concatenate([np.arange(4) + 1, np.sort(rng.integers(1, 60000, 300)), [5], np.sort(rng.integers(1, 2000000, 395))])
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, pos)
    ref = T.run_ref(["--pfile", "d", "--r2-unphased", "inter-chr"] + extra + ["--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased", "inter-chr"] + extra + ["--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    assert os.path.getsize(str(tmp_path / "ref.vcor")) > 100
    assert filecmp.cmp(str(tmp_path / "ref.vcor"), str(tmp_path / "hip.vcor"), shallow=False)
    # the window flags do not combine with an all-pairs mode
    bad = subprocess.run([cli, "--pfile", "d", "--r2-unphased", "inter-chr", "--ld-window-kb", "5", "--out", "x", "--dry-run"], cwd=str(tmp_path),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert bad.returncode != 0 and "All-pairs" in bad.stdout


def test_device_side_hit_filter_matches_dense_rows(gpu_pkg):
    """ldp_r2_unphased_hits = the dense rows filtered by |r^2| >= min (NaN never passes), incl. buffer overflow reporting."""
    m, n = 500, 333
    raw = T.synth_raw_codes(m, n, seed=77, missing_rate=0.02)
    raw[7] = 0
    raw[8] = 3
    eng = gpu_pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_matrix(m)
    eng.load_genotypes_host(0, T.pack_2bit(raw), gpu_pkg.LDP_GENO_REF)
    dense = eng.r2_unphased_rows()
    for thr, first, cnt in [(0.3, 0, m), (0.05, 100, 217), (1e-9, 0, 64)]:
        hits, found = eng.r2_unphased_hits(thr, first, cnt, capacity=1 << 18)
        want = [(i, j, dense[j, i]) for j in range(first, first + cnt) for i in range(j) if abs(dense[j, i]) >= thr]
        want.sort()
        assert found == len(want) == len(hits)
        assert [(int(h["first"]), int(h["second"])) for h in hits] == [(a, b) for a, b, _ in want]
        assert np.array_equal(hits["r2"], np.array([w[2] for w in want]))
    hits, found = eng.r2_unphased_hits(0.05, 0, m, capacity=10)
    assert found > 10 and len(hits) == 10
    eng.close()


def test_device_side_hit_filter_on_the_windowed_plan(gpu_pkg):
    """ldp_r2_unphased_hits on a ldp_set_variants_vcor plan = the band's pairs filtered, global variant indices
    (the plan leaves out variants without partners, so local and global indices differ)."""
    m, n = 420, 200
    raw = T.synth_raw_codes(m, n, seed=31, missing_rate=0.01)
    rng = np.random.default_rng(4)
    chr_idx = np.sort(rng.integers(0, 3, size=m)).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(3):
        sel = np.where(chr_idx == c)[0]
        gaps = rng.integers(1, 400, size=len(sel))
        gaps[rng.random(len(sel)) < 0.05] += 50000  # isolated variants: no partner inside the window
        bps[sel] = np.cumsum(gaps)
    eng = gpu_pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_vcor(chr_idx, bps, 3000, 25)
    eng.load_genotypes_host(0, T.pack_2bit(raw), gpu_pkg.LDP_GENO_REF)
    lo, cand = eng.band()
    dense = eng.r2_unphased_band_rows()
    want = []
    off = 0
    for j in range(m):
        for i in range(int(lo[j]), j):
            r2 = dense[off + (i - int(lo[j]))]
            if abs(r2) >= 0.1:
                want.append((i, j, r2))
        off += j - int(lo[j])
    want.sort()
    hits, found = eng.r2_unphased_hits(0.1, 0, m, capacity=1 << 18)
    assert found == len(want) == len(hits) and found > 50
    assert [(int(h["first"]), int(h["second"])) for h in hits] == [(a, b) for a, b, _ in want]
    assert np.array_equal(hits["r2"], np.array([w[2] for w in want]))
    part, found2 = eng.r2_unphased_hits(0.1, 100, 150, capacity=1 << 18)
    assert [(int(h["first"]), int(h["second"])) for h in part] == [(a, b) for a, b, _ in want if 100 <= b < 250]
    eng.close()


@pytest.mark.parametrize("mods,ext", [([], ".vcor"), (["inter-chr"], ".vcor"), (["square"], ".unphased.vcor2"), (["triangle"], ".unphased.vcor2")])
def test_cli_zs_outputs_decompress_to_the_reference_text(gpu_pkg, tmp_path, mods, ext):
    """'zs': <out><ext>.zst whose decompressed text equals the reference's uncompressed file"""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 260, 120
    raw = T.synth_raw_codes(m, n, seed=41, missing_rate=0.02)
    chroms = ["1"] * 150 + ["4"] * 110
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, np.concatenate([np.arange(150), np.arange(110)]) * 300 + 1)
    ref = T.run_ref(["--pfile", "d", "--r2-unphased"] + mods + ["--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased"] + mods + ["zs", "--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    back = subprocess.run([T.REF_BIN, "--zst-decompress", "hip" + ext + ".zst"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
    assert back.returncode == 0
    assert back.stdout == open(str(tmp_path / ("ref" + ext)), "rb").read()


def test_cli_r2_multiallelic_variants_match_reference(gpu_pkg, tmp_path):
    """multiallelic variants in --r2-unphased: major-vs-rest collapse (or REF-vs-rest with 'ref-based'), the table's
    ambiguity guard (plink2_ld.cc:11063-11072) and its 'allow-ambiguous-allele' override"""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 260, 110
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed=3, max_alt=4, multi_rate=0.4)
    T.write_vcf_haps(os.path.join(tmp, "d.vcf"), first, second, alt_ct, ["1"] * 150 + ["6"] * 110, np.concatenate([np.arange(150), np.arange(110)]) * 211 + 1,
                     unphased=np.ones(first.shape, dtype=bool))
    T.ref_import_vcf(os.path.join(tmp, "d.vcf"), os.path.join(tmp, "d"))

    def both(args):
        ref = T.run_ref(["--pfile", "d", "--r2-unphased"] + args + ["--out", "ref"], tmp)
        got = subprocess.run([cli, "--pfile", "d", "--r2-unphased"] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        return ref, got

    ref, got = both([])
    assert ref.returncode == got.returncode == 7
    assert [ln for ln in ref.stdout.splitlines() if ln.startswith("Error")] == [ln for ln in got.stdout.splitlines() if ln.startswith("Error")]
    for args, ext in [(["allow-ambiguous-allele", "--ld-window-r2", "0.05"], ".vcor"), (["allow-ambiguous-allele", "ref-based", "--ld-window-r2", "0.05"], ".vcor"),
                      (["square", "bin"], ".unphased.vcor2.bin"), (["triangle", "bin4", "ref-based"], ".unphased.vcor2.bin"),
                      (["inter-chr", "allow-ambiguous-allele"], ".vcor")]:
        ref, got = both(args)
        assert ref.returncode == 0 and got.returncode == 0, (args, ref.stdout[-300:], got.stdout[-300:])
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), args


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,miss", [(260, 90, 0.0), (333, 70, 0.04)])
def test_matrix_column_blocks_match_the_rows(gpu_pkg, m, n, miss):
    """ldp_r2_unphased_block / _block_hits (the pieces of `--parallel k n` and of a device shard): every block of the lower
    triangle equals the same entries of the full rows; hits of a block = the full hit list restricted to it."""
    pkg = gpu_pkg
    raw = T.synth_raw_codes(m, n, seed=3 * m + n, missing_rate=miss)
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_matrix(m)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    full = eng.r2_unphased_rows()
    hits_all, found_all = eng.r2_unphased_hits(0.05)
    assert found_all == len(hits_all)
    for (r0, rc, c0, cc) in [(0, m, 0, m), (100, 60, 37, 50), (200, m - 200, 0, 33), (64, 64, 64, 64), (5, 20, 100, 40), (130, 70, 96, 64)]:
        blk = eng.r2_unphased_block(r0, rc, c0, cc)
        want = np.zeros_like(blk)
        for j in range(r0, r0 + rc):
            hi = min(j + 1, c0 + cc)
            if hi > c0:
                want[j - r0, :hi - c0] = full[j, c0:hi]
        assert np.array_equal(np.isnan(blk), np.isnan(want))
        assert np.array_equal(blk[~np.isnan(blk)], want[~np.isnan(want)])
        h, found = eng.r2_unphased_block_hits(0.05, r0, rc, c0, cc)
        sel = hits_all[(hits_all["second"] >= r0) & (hits_all["second"] < r0 + rc) & (hits_all["first"] >= c0) & (hits_all["first"] < c0 + cc)]
        assert found == len(sel) and np.array_equal(h, np.sort(sel, order=["first", "second"]))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,miss,mfma", [(260, 90, 0.0, 1), (333, 70, 0.04, 1), (200, 1100, 0.01, 1), (150, 130, 0.1, 0)])
def test_pair_tuples_of_dense_blocks_from_the_pair_kernels(gpu_pkg, m, n, miss, mfma):
    """ldp_pair_stats_block: the six integers of every pair of a block, written by the tile kernels' epilogue instead of the r^2 they would
    form from them -- equal to the one-wave-per-pair reference kernel's (ldp_pair_stats), on complete rows, rows with missing calls,
    monomorphic and all-missing rows, on the matrix pipe and on the popcount kernels."""
    pkg = gpu_pkg
    raw = T.synth_raw_codes(m, n, seed=5 * m + n, missing_rate=miss)
    raw[3] = 0
    raw[5] = 3
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_option("pair_mfma", mfma)
    eng.set_variants_matrix(m)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    for (r0, rc, c0, cc) in [(0, m, 0, m), (100, 45, 37, 50), (64, 64, 64, 64), (130, m - 130, 96, 54)]:
        blk = eng.pair_stats_block(r0, rc, c0, cc)
        first, second = [], []
        for j in range(r0, r0 + rc):
            for i in range(c0, min(j, c0 + cc)):
                first.append(i)
                second.append(j)
        ref = eng.pair_stats(first, second)
        got = blk[np.array(second) - r0, np.array(first) - c0]
        assert np.array_equal(got, ref)
        mask = np.ones(blk.shape, dtype=bool)
        mask[np.array(second) - r0, np.array(first) - c0] = False
        assert not blk[mask].view(np.uint32).any()      # nothing outside the pairs i < j
    eng.close()


def _x_weighted_numpy(ta, tm, fa_i, fa_j, fm_i, fm_j, both_x, unsquared):
    """ComputeXR2's arithmetic (plink2_ld.cc:7160-7185) for one pair in numpy longdouble: every fma of the reference is one rounding of
    an exactly representable product-sum here as long as the 64-bit significand holds it -- the caller only asks for the value where
    double arithmetic cannot be on a rounding boundary (tolerance 2 ulp), bit-exactness is the CLI tests' business (vs the reference)."""
    def counts(t, f1, f2):
        n = int(t["nm"]); g1 = n - int(t["sum1"]); q1 = n - 2 * int(t["sum1"]) + int(t["ssq1"])
        g2 = n - int(t["sum2"]); q2 = n - 2 * int(t["sum2"]) + int(t["ssq2"]); d = n - int(t["sum1"]) - int(t["sum2"]) + int(t["dot"])
        if f1:
            q1 = 4 * n - 4 * g1 + q1; g1 = 2 * n - g1; d = 2 * g2 - d
        if f2:
            q2 = 4 * n - 4 * g2 + q2; g2 = 2 * n - g2; d = 2 * g1 - d
        return n, g1, q1, g2, q2, d
    a = counts(ta, fa_i, fa_j)
    mm = counts(tm, fm_i, fm_j) if tm is not None else (0,) * 6
    if not a[0]:
        return np.nan
    w = 0.5 if both_x else (1.0 - 0.5 * 1.4142135623730951)
    L = np.longdouble
    wn, g1, q1, g2, q2, d = [np.float64(L(x) - L(w) * L(y)) for x, y in zip(a, mm)]
    var1 = np.float64(L(q1) * L(wn) - L(np.float64(g1 * g1)))
    var2 = np.float64(L(q2) * L(wn) - L(np.float64(g2 * g2)))
    if not (var1 > 0 and var2 > 0):
        return np.nan
    cov = np.float64(L(d) * L(wn) - L(np.float64(g1 * g2)))
    r = min(1.0, float(cov * cov / (var1 * var2)))
    if unsquared:
        r = np.sqrt(r) * (-1.0 if cov < 0 else 1.0)
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("unsquared,as_float", [(False, False), (True, False), (False, True)])
def test_chrx_weighted_blocks_on_the_device(gpu_pkg, monkeypatch, unsquared, as_float):
    """ldp_r2_unphased_block_x / _x_hits (ComputeXR2): the all-founder and the male-founder engines' tuples from the pair kernels, combined
    by x_weighted_kernel -- only the pairs with a chrX variant are touched; values against the same arithmetic in numpy; one chunk or
    many; hits = the dense values filtered."""
    pkg = gpu_pkg
    m, n = 300, 160
    rng = np.random.default_rng(11)
    raw = T.synth_raw_codes(m, n, seed=21, missing_rate=0.03, ld_copy_prob=0.7)
    is_x = np.zeros(m, dtype=np.uint8)
    is_x[100:190] = 1
    male = np.flatnonzero(rng.random(n) < 0.45)
    flip_all = (rng.random(m) < 0.3).astype(np.uint8)
    flip_male = (rng.random(m) < 0.3).astype(np.uint8)
    e_all = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    e_all.set_variants_matrix(m)
    e_all.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    e_m = pkg.LdPruneEngine(len(male), 2, 1, False, 0.5, device=0)
    e_m.set_variants_matrix(m)
    e_m.load_genotypes_host(0, T.pack_2bit(raw[:, male]), pkg.LDP_GENO_REF)
    for (r0, rc, c0, cc) in [(0, m, 0, m), (90, 120, 50, 100), (200, 100, 0, 96), (200, 100, 150, 64)]:
        base = e_all.r2_unphased_block(r0, rc, c0, cc, as_float=as_float)
        ta, tm = e_all.pair_stats_block(r0, rc, c0, cc), e_m.pair_stats_block(r0, rc, c0, cc)
        outs = []
        for x_rows in (0, 7):
            e_all.set_option("x_rows", x_rows)
            outs.append(e_all.r2_unphased_block_x(base.copy(), e_m, is_x, r0, c0, flip_all, flip_male, unsquared))
        e_all.set_option("x_rows", 0)
        got = outs[0]
        assert same_bits(outs[0], outs[1])
        touched = np.zeros(got.shape, dtype=bool)
        for j in range(r0, r0 + rc):
            for i in range(c0, min(j, c0 + cc)):
                if is_x[i] or is_x[j]:
                    touched[j - r0, i - c0] = True
                    want = _x_weighted_numpy(ta[j - r0, i - c0], tm[j - r0, i - c0], flip_all[i], flip_all[j], flip_male[i], flip_male[j], bool(is_x[i] and is_x[j]), unsquared)
                    g = float(got[j - r0, i - c0])
                    assert (np.isnan(want) and np.isnan(g)) or abs(g - want) <= (2e-7 if as_float else 5e-16) * max(1.0, abs(want)), (i, j, g, want)
        assert (touched.any() or (r0, c0) == (200, 0)) and same_bits(got[~touched], base[~touched])   # (one block has no chrX row or column)
        if not as_float:
            assert set(got[touched][np.isnan(got[touched])].view(np.uint64)) <= {0xfff8000000000000}
            hits, found = e_all.r2_unphased_block_x_hits(e_m, is_x, 0.05, r0, rc, c0, cc, flip_all, flip_male, unsquared)
            jj, ii = np.nonzero(touched & (np.abs(got) >= 0.05))
            want_hits = sorted((int(i) + c0, int(j) + r0) for j, i in zip(jj, ii))
            assert found == len(want_hits) and [(int(h["first"]), int(h["second"])) for h in hits] == want_hits
            assert same_bits(np.array([h["r2"] for h in hits]), np.array([got[j - r0, i - c0] for i, j in want_hits]))
    # without male founders the weights drop out but the formula (clamp, variance guard) stays
    base = e_all.r2_unphased_block(0, m, 0, m)
    none = e_all.r2_unphased_block_x(base.copy(), None, is_x, 0, 0, None, None, unsquared)
    ta = e_all.pair_stats_block(0, m, 0, m)
    for (i, j) in [(120, 150), (5, 130), (150, 250)]:
        want = _x_weighted_numpy(ta[j, i], None, 0, 0, 0, 0, bool(is_x[i] and is_x[j]), unsquared)
        assert (np.isnan(want) and np.isnan(none[j, i])) or abs(none[j, i] - want) <= 5e-16 * max(1.0, abs(want))
    e_all.close()
    e_m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mods,extra,ext", [(["inter-chr"], ["--ld-window-r2", "0.02"], ".vcor"), (["square", "bin"], [], ".unphased.vcor2.bin"),
                                            (["inter-chr", "ref-based"], ["--ld-window-r2", "0"], ".vcor"),
                                            ([], ["--ld-window-kb", "30", "--ld-window-r2", "0.02"], ".vcor"),       # windowed table, filter on the device
                                            ([], ["--ld-window-kb", "50", "--ld-window-r2", "0"], ".vcor"),          # windowed table, every pair (dense band)
                                            (["ref-based"], ["--ld-window", "40", "--ld-window-r2", "0.1"], ".vcor")])
def test_cli_chrx_device_path_equals_the_pair_lists(gpu_pkg, tmp_path, mods, extra, ext):
    """plink2-hip's chrX values of dense rows, of the inter-chr table and (round 5) of the windowed table come from ldp_r2_unphased_block_x[_hits] --
    the windowed table through two all-pairs engines that hold the chrX run alone --; --debug-x-host takes the same pairs as lists through the
    one-wave-per-pair kernel and the host arithmetic (pinned to the reference in test_cli_r2_with_chrx_matches_reference, which now runs the
    device path): byte-identical files, also when the device path works in many row chunks."""
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    _x_fileset(tmp_path, m=900, n=210, seed=13)
    flag = "--r-unphased" if "ref-based" in mods else "--r2-unphased"
    outs = []
    for tag, hook in (("dev", []), ("chunks", ["--debug-x-rows", "50"]), ("host", ["--debug-x-host"])):
        got = subprocess.run([cli, "--pfile", "sx", flag] + mods + extra + hook + ["--out", tag], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                             timeout=900)
        assert got.returncode == 0, got.stdout
        outs.append(open(os.path.join(tmp, tag + ext), "rb").read())
    assert len(outs[0]) > 10000 and outs[0] == outs[2] and outs[1] == outs[2]


def _concat(paths):
    return b"".join(open(p, "rb").read() for p in paths)


@pytest.mark.gpu
@pytest.mark.parametrize("mods", [["square0", "bin"], ["triangle", "bin4"], ["square", "bin"], ["triangle"], ["square0"], ["square", "bin4"]])
def test_cli_matrix_parallel_pieces_match_reference(gpu_pkg, tmp_path, mods):
    """`--parallel k n` for the matrix shapes (VcorMatrix, plink2_ld.cc:9800-9824): every piece byte-identical to the
    reference's piece, .vars written by piece 1 only, and the pieces concatenate to the undistributed file."""
    cli = gpu_pkg.build_cli()
    m, n = 157, 120
    raw = T.synth_raw_codes(m, n, seed=77, missing_rate=0.03)
    chroms = ["1"] * 80 + ["2"] * (m - 80)
    bps = (1000 + 137 * np.arange(m)).astype(np.uint32)
    T.write_bed(str(tmp_path / "d"), raw, chroms, bps)
    ext = ".unphased.vcor2" + (".bin" if ("bin" in mods or "bin4" in mods) else "")
    for k in (1, 2, 3):
        a = subprocess.run([REF, "--bfile", "d", "--r2-unphased"] + mods + ["--parallel", str(k), "3", "--out", "ref"], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        b = subprocess.run([cli, "--bfile", "d", "--r2-unphased"] + mods + ["--parallel", str(k), "3", "--out", "hip"], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert a.returncode == 0, a.stdout[-800:]
        assert b.returncode == 0, b.stdout[-800:]
        assert open(str(tmp_path / ("ref%s.%d" % (ext, k))), "rb").read() == open(str(tmp_path / ("hip%s.%d" % (ext, k))), "rb").read(), (mods, k)
    assert open(str(tmp_path / ("ref%s.vars" % ext)), "rb").read() == open(str(tmp_path / ("hip%s.vars" % ext)), "rb").read()
    whole = subprocess.run([cli, "--bfile", "d", "--r2-unphased"] + mods + ["--out", "all"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=300)
    assert whole.returncode == 0, whole.stdout[-800:]
    assert _concat([str(tmp_path / ("hip%s.%d" % (ext, k))) for k in (1, 2, 3)]) == open(str(tmp_path / ("all" + ext)), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("mods,extra", [(["inter-chr"], ["--ld-window-r2", "0.1"]), (["inter-chr"], ["--ld-window-r2", "0"]),
                                        ([], ["--ld-window-kb", "5", "--ld-window-r2", "0.05"]), ([], ["--ld-window-kb", "3", "--ld-window-r2", "0"])])
def test_cli_table_parallel_pieces_match_reference(gpu_pkg, tmp_path, mods, extra):
    """`--parallel k n` for the .vcor tables (VcorTable, plink2_ld.cc:11157-11168): shards by first variant, header in piece 1."""
    cli = gpu_pkg.build_cli()
    m, n = 211, 100
    raw = T.synth_raw_codes(m, n, seed=91, missing_rate=0.02)
    chroms = ["1"] * 100 + ["3"] * (m - 100)
    bps = (500 + 97 * np.arange(m)).astype(np.uint32)
    T.write_bed(str(tmp_path / "d"), raw, chroms, bps)
    for k in (1, 2, 3, 4):
        a = subprocess.run([REF, "--bfile", "d", "--r2-unphased"] + mods + extra + ["--parallel", str(k), "4", "--out", "ref"], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        b = subprocess.run([cli, "--bfile", "d", "--r2-unphased"] + mods + extra + ["--parallel", str(k), "4", "--out", "hip"], cwd=str(tmp_path),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert a.returncode == 0, a.stdout[-800:]
        assert b.returncode == 0, b.stdout[-800:]
        assert open(str(tmp_path / ("ref.vcor.%d" % k)), "rb").read() == open(str(tmp_path / ("hip.vcor.%d" % k)), "rb").read(), (mods, extra, k)


LD_SNP_CASES = [
    ["--ld-snp", "snp200"],
    ["--ld-snp", "snp200", "--ld-window-r2", "0", "--ld-window-kb", "3"],
    ["--ld-snps", "snp100-snp140,snp7", "snp650", "--ld-window-r2", "0.05"],
    ["--ld-snps", "snp140-snp100", "--ld-window-kb", "1.5", "--ld-window-r2", "0"],
    ["--ld-snp-list", "rows.txt", "--ld-window-r2", "0.1"],
    ["--ld-snp-list", "rows.txt", "--ld-window-kb", "2", "--ld-window-r2", "0"],
    ["inter-chr", "--ld-snp-list", "rows.txt", "--ld-window-r2", "0.3"],
]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", LD_SNP_CASES)
def test_cli_vcor_row_variants_match_reference(gpu_pkg, tmp_path, extra):
    """--ld-snp / --ld-snps / --ld-snp-list: the row variants lead their lines and see both sides of their windows."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    m, n = 900, 160
    raw = T.synth_raw_codes(m, n, seed=23, missing_rate=0.02)
    raw[12] = 2
    chroms = ["0"] * 4 + ["1"] * 500 + ["3"] * 1 + ["7"] * 395
    rng = np.random.default_rng(9)
    pos = np.concatenate([np.arange(4) + 1, np.sort(rng.integers(1, 60000, 500)), [5], np.sort(rng.integers(1, 2000000, 395))])
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, pos)
    with open(str(tmp_path / "rows.txt"), "w") as f:
        f.write("snp2 snp30\nsnp31 nosuchid snp32\nsnp504\nsnp505 snp899 snp506\n")   # snp2 sits on chromosome 0
    mods = [x for x in extra if not x.startswith("--") and x == "inter-chr"]
    rest = [x for x in extra if x != "inter-chr"]
    ref = T.run_ref(["--pfile", "d", "--r2-unphased"] + mods + rest + ["--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "d", "--r2-unphased"] + mods + rest + ["--out", "hip"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    want = open(str(tmp_path / "ref.vcor")).read()
    have = open(str(tmp_path / "hip.vcor")).read()
    assert want.count("\n") >= 2
    if want != have:
        wl, hl = want.split("\n"), have.split("\n")
        bad = [(a, b) for a, b in zip(wl, hl) if a != b]
        raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))


COLS_CASES = [
    ("pfile", ["cols=chrom,id,ref,alt"], []),
    ("pfile", ["cols=+maj,+nonmaj,+freq"], ["--ld-window-r2", "0.05"]),
    ("bfile", ["cols=-chrom,-pos,+ref,+alt1"], []),                    # a .bed's REF alleles are provisional: the column appears
    ("bfile", ["cols=+ref,-maybeprovref"], []),
    ("pfile", ["cols=pos,provref,ref"], ["--ld-window-kb", "3"]),
    ("pfile", ["cols=id"], ["--ld-window-r2", "0.5"]),
    ("pfile", ["cols=+freq,-id", "inter-chr"], ["--ld-window-r2", "0.3"]),
    ("bfile", ["cols=maj,nonmaj,freq", "zs"], ["--ld-window-r2", "0"]),
    ("mkpgen", ["cols=+ref,+alt,+maj"], []),                            # written by the reference from the .bed: every REF provisional
]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,mods,extra", COLS_CASES)
def test_cli_vcor_column_sets_match_reference(gpu_pkg, tmp_path, fmt, mods, extra):
    """--r2-unphased cols=: the .vcor table with other column sets (VcorTableWriteThread, plink2_ld.cc:10836-10960)."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 500, 120
    raw = T.synth_raw_codes(m, n, seed=23, missing_rate=0.03)
    raw[7] = 0                                   # monomorphic: frequency 0, NaN r^2
    raw[8, :] = 3                                # nothing observed: frequency 0.5 by convention
    chroms = ["1"] * 300 + ["4"] * 200
    rng = np.random.default_rng(4)
    pos = np.concatenate([np.sort(rng.integers(1, 40000, 300)), np.sort(rng.integers(1, 40000, 200))])
    T.write_pgen_fixed(os.path.join(tmp, "d"), raw, chroms, pos)
    T.write_bed(os.path.join(tmp, "d"), raw, chroms, pos)
    src = ["--" + fmt, "d"]
    if fmt == "mkpgen":
        mk = T.run_ref(["--bfile", "d", "--make-pgen", "--out", "e"], tmp)
        assert mk.returncode == 0, mk.stdout
        src = ["--pfile", "e"]
    ref = T.run_ref(src + ["--r2-unphased"] + mods + extra + ["--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli] + src + ["--r2-unphased"] + mods + extra + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    ext = ".vcor.zst" if "zs" in mods else ".vcor"
    if ext.endswith(".zst"):
        a = subprocess.run([T.REF_BIN, "--zst-decompress", os.path.join(tmp, "ref" + ext)], stdout=subprocess.PIPE).stdout
        b = subprocess.run([T.REF_BIN, "--zst-decompress", os.path.join(tmp, "hip" + ext)], stdout=subprocess.PIPE).stdout
        assert len(a) > 100 and a == b
    else:
        want, have = open(os.path.join(tmp, "ref" + ext)).read(), open(os.path.join(tmp, "hip" + ext)).read()
        assert len(want) > 100
        if want != have:
            wl, hl = want.split("\n"), have.split("\n")
            bad = [(a, b) for a, b in zip(wl, hl) if a != b]
            raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))


@pytest.mark.gpu
def test_cli_vcor_column_sets_at_multiallelic_variants(gpu_pkg, tmp_path):
    """Allele columns lift the ambiguity guard: MAJ / NONMAJ / NONMAJ_FREQ of multiallelic variants, REF / ALT with 'ref-based'."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 260, 110
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed=5, max_alt=4, multi_rate=0.4)
    T.write_vcf_haps(os.path.join(tmp, "d.vcf"), first, second, alt_ct, ["1"] * 150 + ["6"] * 110, np.concatenate([np.arange(150), np.arange(110)]) * 211 + 1,
                     unphased=np.ones(first.shape, dtype=bool))
    T.ref_import_vcf(os.path.join(tmp, "d.vcf"), os.path.join(tmp, "d"))
    for mods in (["cols=+maj,+nonmaj,+freq"], ["cols=+ref,+alt", "ref-based"], ["cols=+maj,+alt1"], ["cols=+ref,+alt,+maj,+freq", "ref-based"],
                 ["cols=+nonmaj", "inter-chr"]):
        args = ["--pfile", "d", "--r2-unphased"] + mods + ["--ld-window-r2", "0.05"]
        ref = T.run_ref(args + ["--out", "ref"], tmp)
        got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert ref.returncode == 0 and got.returncode == 0, (mods, ref.stdout[-300:], got.stdout[-300:])
        assert filecmp.cmp(os.path.join(tmp, "ref.vcor"), os.path.join(tmp, "hip.vcor"), shallow=False), mods
    # the guard itself: 'ref-based' wants ref or alt, the default wants maj or nonmaj
    for mods in (["cols=+ref"], ["cols=+maj", "ref-based"]):
        args = ["--pfile", "d", "--r2-unphased"] + mods
        ref = T.run_ref(args + ["--out", "ref"], tmp)
        got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert ref.returncode == got.returncode == 7, mods


R_CASES = [
    # fmt, modifiers, extra flags, output extension
    ("pfile", [], [], ".vcor"),                                                        # default columns: MAJ names the allele r's sign refers to
    ("bfile", ["ref-based"], ["--ld-window-r2", "0.04"], ".vcor"),                    # REF orientation (+ PROVISIONAL_REF? for a .bed)
    ("pfile", ["cols=+nonmaj,+freq,-maj"], ["--ld-window-r2", "0"], ".vcor"),
    ("pfile", ["inter-chr", "cols=id,maj"], ["--ld-window-r2", "0.25"], ".vcor"),
    ("pfile", ["ref-based", "cols=id,alt1", "allow-ambiguous-allele"], ["--ld-window-kb", "4"], ".vcor"),
    ("pfile", ["square", "bin"], [], ".unphased.vcor1.bin"),
    ("bfile", ["triangle", "bin4"], [], ".unphased.vcor1.bin"),
    ("pfile", ["square0", "bin", "ref-based"], [], ".unphased.vcor1.bin"),
    ("pfile", ["triangle"], [], ".unphased.vcor1"),
    ("pfile", ["square", "ref-based"], [], ".unphased.vcor1"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,mods,extra,ext", R_CASES)
def test_cli_r_unphased_matches_reference(gpu_pkg, tmp_path, fmt, mods, extra, ext):
    """--r-unphased: r = +-sqrt(r^2) with the sign of the covariance (plink2_ld.cc:9633-9641, :10640-10647), in major-allele
    or REF orientation; tables, binary and text matrices."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 420, 130
    raw = T.synth_raw_codes(m, n, seed=31, missing_rate=0.03, ld_copy_prob=0.7)
    rng = np.random.default_rng(2)
    for v in rng.choice(m, size=m // 3, replace=False):   # negative correlations: flip a third of the variants
        raw[v] = np.where(raw[v] < 3, 2 - raw[v], 3)
    raw[5] = 0
    chroms = ["2"] * 250 + ["9"] * 170
    pos = np.concatenate([np.sort(rng.integers(1, 30000, 250)), np.sort(rng.integers(1, 30000, 170))])
    T.write_pgen_fixed(os.path.join(tmp, "d"), raw, chroms, pos)
    T.write_bed(os.path.join(tmp, "d"), raw, chroms, pos)
    args = ["--" + fmt, "d", "--r-unphased"] + mods + extra
    ref = T.run_ref(args + ["--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    want, have = open(os.path.join(tmp, "ref" + ext), "rb").read(), open(os.path.join(tmp, "hip" + ext), "rb").read()
    assert len(want) > 100
    if ext == ".vcor":
        assert b"-0." in want, "the case must contain negative correlations"
        if want != have:
            wl, hl = want.split(b"\n"), have.split(b"\n")
            bad = [(a, b) for a, b in zip(wl, hl) if a != b]
            raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))
    assert want == have
    if ext.endswith("vcor1.bin") or ext.endswith("vcor1"):
        assert open(os.path.join(tmp, "ref" + ext + ".vars")).read() == open(os.path.join(tmp, "hip" + ext + ".vars")).read()


@pytest.mark.gpu
def test_cli_r_unphased_guards_and_multiallelic(gpu_pkg, tmp_path):
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 200, 100
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed=8, max_alt=3, multi_rate=0.3)
    T.write_vcf_haps(os.path.join(tmp, "d.vcf"), first, second, alt_ct, ["1"] * m, np.arange(m) * 180 + 1, unphased=np.ones(first.shape, dtype=bool))
    T.ref_import_vcf(os.path.join(tmp, "d.vcf"), os.path.join(tmp, "d"))

    def both(mods):
        args = ["--pfile", "d", "--r-unphased"] + mods
        ref = T.run_ref(args + ["--out", "ref"], tmp)
        got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        return ref, got

    for mods in ([], ["ref-based", "cols=+alt"], ["cols=+nonmaj,-maj"], ["ref-based", "cols=+alt1", "allow-ambiguous-allele"], ["square", "bin"]):
        ref, got = both(mods)
        assert ref.returncode == 0 and got.returncode == 0, (mods, ref.stdout[-300:], got.stdout[-300:])
        ext = ".unphased.vcor1.bin" if "square" in mods else ".vcor"
        assert filecmp.cmp(os.path.join(tmp, "ref" + ext), os.path.join(tmp, "hip" + ext), shallow=False), mods
    for mods in (["cols=-maj"], ["ref-based", "cols=-ref"], ["ref-based", "cols=-ref,+alt1"]):   # nothing names the allele / alt1 alone at multiallelic sites
        ref, got = both(mods)
        assert ref.returncode == got.returncode == 7, (mods, ref.returncode, got.returncode, got.stdout[-300:])


def _rewrite_cm(bim_path, cms):
    rows = [ln.split("\t") for ln in open(bim_path).read().splitlines()]
    assert len(rows) == len(cms) and len(rows[0]) == 6
    with open(bim_path, "w") as f:
        for r, c in zip(rows, cms):
            r[2] = c
            f.write("\t".join(r) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--ld-window-cm", "0.5"], ["--ld-window-cm", "0.05", "--ld-window-r2", "0"], ["--ld-window-cm", "2", "--ld-window-kb", "8"],
                                   ["--ld-window-cm", "1", "--ld-window", "6", "--ld-window-r2", "0.01"], ["--ld-window-cm", "0"],
                                   ["--ld-window-cm", "0.25", "--r-unphased"]])
def test_cli_ld_window_cm_matches_reference(gpu_pkg, tmp_path, extra):
    """--ld-window-cm: the centimorgan window of UpdateVcorWindow (open at the far end, intersected with the kb and count windows)."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 600, 100
    raw = T.synth_raw_codes(m, n, seed=37, missing_rate=0.02, ld_copy_prob=0.7)
    chroms = ["1"] * 350 + ["5"] * 250
    rng = np.random.default_rng(6)
    pos = np.concatenate([np.sort(rng.integers(1, 60000, 350)), np.sort(rng.integers(1, 60000, 250))])
    T.write_bed(os.path.join(tmp, "d"), raw, chroms, pos)
    # quarter-centimorgan grid with repeats: plenty of pairs exactly one radius apart
    steps = rng.choice([0.0, 0.25, 0.25, 0.5, 0.05], size=m)
    cm = np.concatenate([np.cumsum(steps[:350]), np.cumsum(steps[350:])])
    _rewrite_cm(os.path.join(tmp, "d.bim"), ["%g" % c for c in cm])
    flag = ["--r-unphased"] if "--r-unphased" in extra else ["--r2-unphased"]
    extra = [x for x in extra if x != "--r-unphased"]
    args = ["--bfile", "d"] + flag + extra
    ref = T.run_ref(args + ["--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert got.returncode == 0, got.stdout
    want, have = open(os.path.join(tmp, "ref.vcor")).read(), open(os.path.join(tmp, "hip.vcor")).read()
    if want != have:
        wl, hl = want.split("\n"), have.split("\n")
        bad = [(a, b) for a, b in zip(wl, hl) if a != b]
        raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))
    if extra[1] != "0":
        assert len(want) > 200


@pytest.mark.gpu
def test_cli_ld_window_cm_edge_cases(gpu_pkg, tmp_path):
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 120, 60
    raw = T.synth_raw_codes(m, n, seed=3, ld_copy_prob=0.8)
    T.write_bed(os.path.join(tmp, "d"), raw, ["2"] * m, np.arange(m) * 50 + 1)

    def both(args):
        ref = T.run_ref(["--bfile", "d", "--r2-unphased"] + args + ["--out", "ref"], tmp)
        got = subprocess.run([cli, "--bfile", "d", "--r2-unphased"] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        return ref, got

    # a file without centimorgan positions: the flag has nothing to act on
    ref, got = both(["--ld-window-cm", "0.001"])
    assert ref.returncode == got.returncode == 0 and filecmp.cmp(os.path.join(tmp, "ref.vcor"), os.path.join(tmp, "hip.vcor"), shallow=False)
    # decreasing positions
    _rewrite_cm(os.path.join(tmp, "d.bim"), ["%g" % (10 - 0.01 * k) for k in range(m)])
    ref, got = both(["--ld-window-cm", "1"])
    assert ref.returncode == got.returncode == 7 and "nondecreasing CM values" in got.stdout
    ref, got = both(["square", "--ld-window-cm", "1"])
    assert ref.returncode == got.returncode == 8
    ref, got = both(["--ld-window-cm", "-1"])
    assert ref.returncode == got.returncode == 8 and "Invalid --ld-window-cm argument" in got.stdout


def _x_fileset(tmp_path, m=360, n=150, seed=7, unknown_sex=True, nonfounders=5):
    """chr1 + chrX + chr5 (chrX in the middle of the file), males / females / unknown sex, some non-founders."""
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.04, ld_copy_prob=0.7)
    per = m // 3
    chroms = ["1"] * per + ["X"] * per + ["5"] * (m - 2 * per)
    rng = np.random.default_rng(seed)
    bps = np.concatenate([np.sort(rng.integers(1, 50000, per)), np.sort(rng.integers(1, 50000, per)), np.sort(rng.integers(1, 50000, m - 2 * per))]).astype(np.uint32)
    sexes = rng.choice([1, 2, 0] if unknown_sex else [1, 2], size=n, p=[0.5, 0.4, 0.1] if unknown_sex else [0.5, 0.5])
    male = np.flatnonzero(sexes == 1)
    for v in range(per, 2 * per):                 # hemizygous males: hets are rare there, not absent
        het = raw[v, male] == 1
        keep = rng.random(het.sum()) < 0.1
        vals = raw[v, male]
        vals[np.flatnonzero(het)[~keep]] = 2 * rng.integers(0, 2, size=int((~keep).sum()))
        raw[v, male] = vals
    prefix = str(tmp_path / "sx")
    T.write_pgen_fixed(prefix, raw, chroms, bps, sexes=sexes)
    T.write_bed(prefix, raw, chroms, bps)
    fam, psam = [], ["#IID\tPAT\tMAT\tSEX"]
    for s_ in range(n):
        nf = (s_ % 11 == 3) and (s_ // 11 < nonfounders)
        fam.append("s%d s%d %s %s %d -9" % (s_, s_, "s0" if nf else "0", "s1" if nf else "0", sexes[s_]))
        psam.append("s%d\t%s\t%s\t%s" % (s_, "s0" if nf else "0", "s1" if nf else "0", "NA" if sexes[s_] == 0 else str(sexes[s_])))
    open(prefix + ".fam", "w").write("\n".join(fam) + "\n")
    open(prefix + ".psam", "w").write("\n".join(psam) + "\n")
    return prefix


X_CASES = [
    ("pfile", "--r2-unphased", [], [], ".vcor"),
    ("bfile", "--r2-unphased", [], ["--ld-window-r2", "0", "--ld-window-kb", "6"], ".vcor"),
    ("pfile", "--r2-unphased", ["inter-chr"], ["--ld-window-r2", "0.04"], ".vcor"),
    ("pfile", "--r2-unphased", ["inter-chr"], ["--ld-window-r2", "0"], ".vcor"),
    ("bfile", "--r2-unphased", ["cols=+maj,+nonmaj,+freq"], ["--ld-window-r2", "0.1"], ".vcor"),
    ("pfile", "--r2-unphased", ["square", "bin"], [], ".unphased.vcor2.bin"),
    ("pfile", "--r2-unphased", ["triangle", "bin4"], [], ".unphased.vcor2.bin"),
    ("bfile", "--r2-unphased", ["square0"], [], ".unphased.vcor2"),
    ("pfile", "--r-unphased", [], ["--ld-window-r2", "0.02"], ".vcor"),
    ("pfile", "--r-unphased", ["ref-based", "inter-chr"], ["--ld-window-r2", "0.05"], ".vcor"),
    ("pfile", "--r-unphased", ["triangle", "bin", "ref-based"], [], ".unphased.vcor1.bin"),
    ("pfile", "--r2-unphased", [], ["--ld-snps", "snp130-snp150,snp40", "--ld-window-kb", "8"], ".vcor"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,flag,mods,extra,ext", X_CASES)
def test_cli_r2_with_chrx_matches_reference(gpu_pkg, tmp_path, fmt, flag, mods, extra, ext):
    """Pairs with a chrX variant: the male founders weighted down in all six sums (ComputeXR2, plink2_ld.cc:7122-7190) -- by
    1/2 inside chrX, by 1 - sqrt(2)/2 against an autosome -- with the chrX-aware major allele where the orientation matters."""
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    _x_fileset(tmp_path)
    args = ["--" + fmt, "sx", flag] + mods + extra
    ref = T.run_ref(args + ["--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert got.returncode == 0, got.stdout
    want, have = open(os.path.join(tmp, "ref" + ext), "rb").read(), open(os.path.join(tmp, "hip" + ext), "rb").read()
    assert len(want) > 200
    if ext == ".vcor" or ext.endswith("vcor2"):
        if want != have:
            wl, hl = want.split(b"\n"), have.split(b"\n")
            bad = [(a, b) for a, b in zip(wl, hl) if a != b]
            raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))
    elif want != have:
        dt = np.float32 if "bin4" in mods else np.float64
        a, b = np.frombuffer(want, dtype=dt), np.frombuffer(have, dtype=dt)
        ai, bi = a.view(np.uint32 if "bin4" in mods else np.uint64), b.view(np.uint32 if "bin4" in mods else np.uint64)
        bad = np.flatnonzero(ai != bi)
        raise AssertionError("%d of %d elements differ, first at %d: %r (%x) vs %r (%x)" % (len(bad), len(a), bad[0], a[bad[0]], ai[bad[0]], b[bad[0]], bi[bad[0]]))


@pytest.mark.gpu
def test_cli_r2_chrx_without_male_founders_and_refusals(gpu_pkg, tmp_path):
    assert T.have_ref()
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    prefix = _x_fileset(tmp_path, unknown_sex=False)
    # every sample female: the male terms vanish but the chrX formula (clamp at 1, variance guard) stays
    lines = open(prefix + ".fam").read().splitlines()
    for sex_code in ("2", "1"):     # ... and with male founders only chrX is not special at all (Vcor :9946-9951)
        open(prefix + ".fam", "w").write("\n".join(" ".join(ln.split()[:4] + [sex_code, "-9"]) for ln in lines) + "\n")
        for args in (["--bfile", "sx", "--r2-unphased", "--ld-window-r2", "0.05"], ["--bfile", "sx", "--r2-unphased", "inter-chr", "--ld-window-r2", "0.05"]):
            ref = T.run_ref(args + ["--out", "ref"], tmp)
            got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
            assert ref.returncode == 0 and got.returncode == 0, (ref.stdout[-300:], got.stdout[-300:])
            assert filecmp.cmp(os.path.join(tmp, "ref.vcor"), os.path.join(tmp, "hip.vcor"), shallow=False), (sex_code, args)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,flag,mods,extra,ext", [
    ("pfile", "--r2-unphased", [], ["--ld-window-r2", "0.03"], ".vcor"),
    ("bfile", "--r2-unphased", ["ref-based", "cols=+ref,+alt1"], ["--ld-window-kb", "4", "--ld-window-r2", "0"], ".vcor"),
    ("pfile", "--r-unphased", ["ref-based", "inter-chr"], ["--ld-window-r2", "0.03"], ".vcor"),
    ("pfile", "--r2-unphased", ["square", "bin", "ref-based"], [], ".unphased.vcor2.bin"),
])
def test_cli_r2_whole_genome_layout_matches_reference(gpu_pkg, tmp_path, fmt, flag, mods, extra, ext):
    """Autosomes, chrX, chrY and MT in one fileset (only chrX is special in the unphased statistics, Vcor :9611-9632)."""
    assert T.have_ref()
    from test_cli import sexed_fileset
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    sexed_fileset(tmp_path, m=500, n=130)
    args = ["--" + fmt, "sx", flag] + mods + extra
    ref = T.run_ref(args + ["--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli] + args + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert got.returncode == 0, got.stdout
    want, have = open(os.path.join(tmp, "ref" + ext), "rb").read(), open(os.path.join(tmp, "hip" + ext), "rb").read()
    assert len(want) > 200
    if ext == ".vcor" and want != have:
        wl, hl = want.split(b"\n"), have.split(b"\n")
        bad = [(a, b) for a, b in zip(wl, hl) if a != b]
        raise AssertionError("%d vs %d lines, first difference %r" % (len(wl), len(hl), bad[:2]))
    assert want == have
    # what needs the haploid chromosomes' major allele is refused
    for bad_args in (["--r-unphased"], ["--r2-unphased", "cols=+maj"], ["--r2-unphased", "inter-chr"]):
        r = subprocess.run([cli, "--pfile", "sx"] + bad_args + ["--out", "hip2"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 63, (bad_args, r.stdout[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("flag,mods,extra,ext", [
    ("--r2-unphased", ["square", "bin"], [], ".unphased.vcor2.bin"),
    ("--r2-unphased", ["inter-chr", "allow-ambiguous-allele"], ["--ld-window-r2", "0.03"], ".vcor"),
    ("--r2-unphased", ["inter-chr", "cols=+maj,+nonmaj,+freq"], ["--ld-window-r2", "0.1"], ".vcor"),
    ("--r-unphased", ["triangle", "bin"], [], ".unphased.vcor1.bin"),
    ("--r2-unphased", ["inter-chr", "ref-based", "cols=+ref,+alt"], ["--ld-window-r2", "0.05"], ".vcor"),
])
def test_cli_r2_chrx_beside_multiallelic_autosomes_matches_reference(gpu_pkg, tmp_path, flag, mods, extra, ext):
    """All-pairs outputs over chrX AND autosomal variants with several ALT alleles: a pair of a chrX variant with a multiallelic one weighs the male founders'
    tuple of both down (ComputeXR2), so the male founders' engine has to hold the multiallelic variant collapsed on the same major allele as the all-founder engine
    (round 5: it held the main track -- right only where REF is the major allele)."""
    cli = gpu_pkg.build_cli()
    tmp = str(tmp_path)
    m, n = 420, 170
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, 55, max_alt=4, multi_rate=0.6, missing_rate=0.03, ld_copy_prob=0.6, redraw=0.1)
    chroms = ["1"] * 150 + ["X"] * 120 + ["7"] * 150
    for v in range(150, 270):   # chrX stays biallelic (the r^2 outputs refuse multiallelic sites there)
        first[v] = np.where(first[v] > 1, 1, first[v])
        second[v] = np.where(second[v] > 1, 1, second[v])
        alt_ct[v] = 1
    rng = np.random.default_rng(5)
    # (several multiallelic autosomal variants must have a major allele other than REF: those are the ones the fix is about)
    swapped = 0
    for v in list(range(0, 150)) + list(range(270, m)):
        if alt_ct[v] > 1 and rng.random() < 0.5:
            a = int(rng.integers(1, alt_ct[v] + 1))
            f0, s0 = first[v].copy(), second[v].copy()
            first[v] = np.where(f0 == 0, a, np.where(f0 == a, 0, f0))
            second[v] = np.where(s0 == 0, a, np.where(s0 == a, 0, s0))
            swapped += 1
    assert swapped > 30
    bps = np.concatenate([1000 + 211 * np.arange(150), 1000 + 211 * np.arange(120), 1000 + 211 * np.arange(150)])
    T.write_vcf_haps(os.path.join(tmp, "p.vcf"), first, second, alt_ct, ["1"] * m, 1000 + 211 * np.arange(m))
    T.ref_import_vcf(os.path.join(tmp, "p.vcf"), os.path.join(tmp, "p"))
    out, k = [], 0
    for ln in open(os.path.join(tmp, "p.pvar")):
        if not ln.startswith("#"):
            f = ln.split("\t")
            f[0], f[1] = chroms[k], str(int(bps[k]))
            ln = "\t".join(f)
            k += 1
        out.append(ln)
    open(os.path.join(tmp, "p.pvar"), "w").write("".join(out))
    sexes = rng.choice([1, 2, 0], size=n, p=[0.5, 0.4, 0.1])
    psam = ["#IID\tPAT\tMAT\tSEX"] + ["s%d\t%s\t%s\t%s" % (q, "s0" if q % 13 == 4 else "0", "s1" if q % 13 == 4 else "0", "NA" if sexes[q] == 0 else str(sexes[q])) for q in range(n)]
    open(os.path.join(tmp, "p.psam"), "w").write("\n".join(psam) + "\n")
    ref = T.run_ref(["--pfile", "p", flag] + mods + extra + ["--threads", "3", "--out", "ref"], tmp)
    assert ref.returncode == 0, ref.stdout
    got = subprocess.run([cli, "--pfile", "p", flag] + mods + extra + ["--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert got.returncode == 0, got.stdout
    a, b = open(os.path.join(tmp, "ref" + ext), "rb").read(), open(os.path.join(tmp, "hip" + ext), "rb").read()
    assert len(a) > 5000 and a == b
