"""plink2-hip --clump against the reference binary (ClumpReports, plink2_ld.cc:7506-9480): the .clumps report and the
.clumps.missing_id list must be the same bytes.  The report handling (p-value parsing through logarithms, TEST filter,
column search, bins, TOTAL, SP2, the p-value formatter) is exercised on the CPU with a window that pairs nothing; the
r^2 membership runs on the GPU."""
import filecmp
import os

import numpy as np
import pytest

import ldtools as T
from test_cli import cli, run_cli  # noqa: F401  (fixture)

P_FORMATS = ["%.3g", "%.6e", "%.2e", "%.8f", "%.1e", "%g"]


def clump_fileset(tmp_path, m, n, seed, missing_rate=0.02, n_chr=3, spacing=900):
    raw = T.synth_raw_codes(m, n, seed, missing_rate=missing_rate, ld_copy_prob=0.7, redraw=0.04)
    rng = np.random.default_rng(seed + 17)
    per = m // n_chr
    chroms, bps = [], []
    for c in range(n_chr):
        cnt = per if c < n_chr - 1 else m - per * (n_chr - 1)
        pos = np.cumsum(rng.integers(1, spacing, size=cnt)) + 1000  # strictly increasing: no shared positions
        chroms += [str(c + 1)] * cnt
        bps += [int(x) for x in pos]
    prefix = str(tmp_path / "d")
    T.write_bed(prefix, raw, chroms, bps)
    T.write_pgen_fixed(prefix, raw, chroms, bps)
    return prefix, raw, chroms, bps


def write_report(path, m, seed, header=("#CHROM", "POS", "ID", "TEST", "OBS_CT", "P"), sig_rate=0.06, with_test=True, id_name="ID", p_name="P"):
    """A --glm-like report: mostly unremarkable p-values, a few strong ones, odd spellings, repeated and unknown IDs."""
    rng = np.random.default_rng(seed)
    lines = []
    cols = [h for h in header]
    cols = [id_name if h == "ID" else (p_name if h == "P" else h) for h in cols]
    if not with_test:
        cols = [h for h in cols if h != "TEST"]
    lines.append("\t".join(cols))
    odd = ["NA", "nan", "0", "1", "1e-320", "4.9e-324", "INF", ".05", "5E-9", "0.0001", "1e-4", "0.01", "1.00e-02", "+0.05", "0.001",
           "0.05000000000000001", "9.9999949e-5", "9.99999501e-5", "0.99999951", "12345678901234567890e-25", "0.000100000000000000000001"]
    ids = ["snp%d" % v for v in range(m)]
    order = rng.permutation(m)
    for v in order:
        reps = 1 + (rng.random() < 0.05) + (rng.random() < 0.01)
        for _ in range(reps):
            u = rng.random()
            if u < sig_rate:
                p = 10.0 ** (-rng.uniform(4, 40))
            elif u < 3 * sig_rate:
                p = 10.0 ** (-rng.uniform(1, 4.3))
            else:
                p = rng.random()
            ptxt = P_FORMATS[rng.integers(len(P_FORMATS))] % p
            if rng.random() < 0.03:
                ptxt = odd[rng.integers(len(odd))]
            tests = ["ADD"] if with_test else [None]
            if with_test and rng.random() < 0.2:
                tests = ["ADD", "DOMDEV"] if rng.random() < 0.5 else ["GENO_2DF"]
            for t in tests:
                row = {"#CHROM": "1", "POS": "1", "ID": ids[v], "TEST": t, "OBS_CT": "100",
                       "P": ptxt if t == "ADD" or t is None else "%.3g" % rng.random()}
                lines.append("\t".join(row[h] for h in header if (h != "TEST" or with_test)))
    for k in range(25):  # IDs the dataset does not have: the significant ones go to .clumps.missing_id
        p = 10.0 ** (-rng.uniform(0, 9))
        row = {"#CHROM": "1", "POS": "1", "ID": "rs%d_%d" % (rng.integers(1, 300), k % 3), "TEST": "ADD", "OBS_CT": "100", "P": "%.3g" % p}
        lines.append("\t".join(row[h] for h in header if (h != "TEST" or with_test)))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def compare_runs(cli, tmp_path, common):
    ref = T.run_ref(common + ["--threads", "4", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    if not os.path.exists(str(tmp_path / "ref.clumps")):
        assert "No significant --clump results" in ref.stdout
        assert "No significant --clump results" in got.stdout
        assert not os.path.exists(str(tmp_path / "hip.clumps"))
        return ref, got
    want = open(str(tmp_path / "ref.clumps")).read()
    have = open(str(tmp_path / "hip.clumps")).read()
    if want != have:
        wl, hl = want.split("\n"), have.split("\n")
        bad = [(a, b) for a, b in zip(wl, hl) if a != b]
        raise AssertionError("%d/%d lines differ, first: %r" % (len(bad) + abs(len(wl) - len(hl)), len(wl), bad[:2]))
    for ext in (".clumps.missing_id", ".clumps.missing_allele"):
        a, b = str(tmp_path / ("ref" + ext)), str(tmp_path / ("hip" + ext))
        assert os.path.exists(a) == os.path.exists(b)
        if os.path.exists(a):
            assert filecmp.cmp(a, b, shallow=False), ext
    line = [l for l in ref.stdout.split("\n") if "formed from" in l]
    assert line and line[0].strip() in got.stdout
    return ref, got


needs_ref = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/plink2 not built")


@needs_ref
@pytest.mark.parametrize("seed,extra", [
    (1, []),
    (2, ["--clump-p1", "1e-3", "--clump-p2", "0.04"]),
    (3, ["--clump-p1", "0.2", "--clump-p2", "0.5"]),          # thresholds above the last bin boundary
    (4, ["--clump-p1", "1e-9", "--clump-p2", "1e-12"]),       # p2 < p1
    (5, ["--clump-p1", "1e-300"]),                               # nothing significant
])
def test_clump_report_handling_matches_reference(cli, tmp_path, seed, extra):
    """A 1-bp radius over distinct positions pairs nothing, so every index candidate is a clump of its own and no GPU
    is needed: what is compared is the report parsing, ranking, binning and printing."""
    m = 1500
    clump_fileset(tmp_path, m, 40, seed)
    write_report(str(tmp_path / "assoc.txt"), m, seed)
    common = ["--bfile", "d", "--clump", "assoc.txt", "--clump-unphased", "--clump-kb", "0.001"] + extra
    compare_runs(cli, tmp_path, common)


@needs_ref
def test_clump_column_search_and_test_filter(cli, tmp_path):
    m = 600
    clump_fileset(tmp_path, m, 40, 9)
    write_report(str(tmp_path / "a.txt"), m, 21, header=("ID", "TEST", "P", "#CHROM"), id_name="SNP", p_name="PVAL")
    base = ["--bfile", "d", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.01"]
    compare_runs(cli, tmp_path, base + ["--clump", "a.txt", "--clump-p-field", "PVAL"])
    compare_runs(cli, tmp_path, base + ["--clump", "a.txt", "--clump-p-field", "P", "PVAL", "--clump-test", "ADD", "DOMDEV"])
    write_report(str(tmp_path / "b.txt"), m, 22, with_test=False)
    compare_runs(cli, tmp_path, base + ["--clump", "b.txt"])


@needs_ref
@pytest.mark.parametrize("how", ["space", "comma"])
def test_clump_several_reports_without_ld(cli, tmp_path, how):
    """Three reports: the F column, ties between reports (the first wins), a variant's lines from other reports in SP2."""
    m = 900
    clump_fileset(tmp_path, m, 40, 12)
    write_report(str(tmp_path / "a.txt"), m, 31)
    write_report(str(tmp_path / "b.txt"), m, 32, sig_rate=0.1)
    write_report(str(tmp_path / "c.txt"), m, 31)          # the same p-values as a.txt: every best p-value is a tie
    files = ["a.txt", "b.txt", "c.txt"] if how == "space" else ["a.txt,b.txt,c.txt"]
    compare_runs(cli, tmp_path, ["--bfile", "d", "--clump"] + files + ["--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.001", "--clump-p2", "0.05"])
    body = open(str(tmp_path / "hip.clumps")).read().split("\n")
    assert body[0].split("\t")[3] == "F" and any("(" in l for l in body[1:])


@needs_ref
@pytest.mark.parametrize("fmt,extra", [
    ("bfile", ["cols=-chrom,-pos"]),
    ("bfile", ["cols=+ref,+alt1,+alt"]),                       # a .bed's REF alleles are provisional: the column appears
    ("pfile", ["cols=+ref,+alt,+provref"]),
    ("pfile", ["cols=+bounds", "--clump-p1", "1e-3", "--clump-p2", "1e-5"]),   # index variants above p2: '.' bounds
    ("pfile", ["cols=+bounds,+f,+a1"]),
    ("pfile", ["cols=-bins"]),
    ("bfile", ["cols=-total,-bins"]),
    ("pfile", ["cols=sp2"]),
    ("pfile", ["cols=total"]),
    ("bfile", ["cols=chrom,pos"]),
    ("pfile", ["--clump-bins", "0.001,0.01", "0.2"]),
    ("bfile", ["--clump-bins", "1e-5", "--clump-p1", "0.05"]),
    ("pfile", ["--clump-bins", "1e-6,0.0001,0.02,0.3,0.9", "cols=+bounds"]),
])
def test_clump_column_sets_and_bins_match_reference(cli, tmp_path, fmt, extra):
    """cols= and --clump-bins (plink2_ld.cc:7577-7612, :9003-9360): what is kept of a report line -- and with it which variants
    count as observed -- depends on the column set."""
    m = 1200
    clump_fileset(tmp_path, m, 40, 8)
    write_report(str(tmp_path / "assoc.txt"), m, 5)
    mods = [x for x in extra if x.startswith("cols=")]
    rest = [x for x in extra if not x.startswith("cols=")]
    common = ["--" + fmt, "d", "--clump"] + mods + ["assoc.txt", "--clump-unphased", "--clump-kb", "0.001"] + rest
    compare_runs(cli, tmp_path, common)


@needs_ref
def test_clump_column_sets_with_two_reports(cli, tmp_path):
    m = 700
    clump_fileset(tmp_path, m, 40, 2)
    write_report(str(tmp_path / "a.txt"), m, 11)
    write_report(str(tmp_path / "b.txt"), m, 12, sig_rate=0.1)
    for mods in (["cols=-maybef"], ["cols=+f,-sp2"], ["cols=+bounds,-total"]):
        compare_runs(cli, tmp_path, ["--bfile", "d", "--clump"] + mods + ["a.txt", "b.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.001"])
    # index variants above p2: '.' bounds -- except that with several reports and SP2 the reference's scan trips over the report
    # numbers it keeps and counts every member with a kept line (plink2_ld.cc:9271-9279); without SP2 it does not
    for mods in (["cols=+bounds"], ["cols=+bounds,-sp2"]):
        compare_runs(cli, tmp_path, ["--bfile", "d", "--clump"] + mods + ["a.txt", "b.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.05", "--clump-p2", "1e-5"])
        body = open(str(tmp_path / "hip.clumps")).read()
        assert ("\t.\t." in body) == ("-sp2" in mods[0])


@needs_ref
@pytest.mark.parametrize("mode,pname,extra", [
    ([], "LOG10_P", []),
    (["input-only"], "NEG_LOG10_P", ["--clump-log10-p1", "2.5", "--clump-log10-p2", "1"]),
    (["output-only"], "P", ["--clump-p1", "0.003"]),
    ([], "P", ["--clump-log10-p1", "3"]),                       # 'P' is the last name tried for a -log10 column
])
def test_clump_log10_matches_reference(cli, tmp_path, mode, pname, extra):
    """--clump-log10 and --clump-log10-p1/-p2 (plink2.cc:4979-5008, :5211-5232; ClumpReports :7631, :7744-7752, :9214)."""
    import math
    m = 900
    clump_fileset(tmp_path, m, 40, 6)
    write_report(str(tmp_path / "plain.txt"), m, 7)
    rng = np.random.default_rng(3)
    lines = open(str(tmp_path / "plain.txt")).read().splitlines()
    hdr = lines[0].split("\t")
    pc = hdr.index("P")
    as_log = mode != ["output-only"]
    out = ["\t".join(pname if h == "P" else h for h in hdr)]
    for ln in lines[1:]:
        f = ln.split("\t")
        if as_log:
            try:
                v = float(f[pc])
                f[pc] = ("inf" if rng.random() < 0.3 else "INF") if v == 0.0 else ("%.6g" % -math.log10(v)) if (0 < v <= 1) else f[pc]
            except ValueError:
                pass
        out.append("\t".join(f))
    open(str(tmp_path / "assoc.txt"), "w").write("\n".join(out) + "\n")
    common = ["--bfile", "d", "--clump", "assoc.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-log10"] + mode + extra
    compare_runs(cli, tmp_path, common)


def write_ranges(path, chroms, bps, seed, zero_based=False):
    """Regions around some variants: several pieces per name, names that sort differently naturally and bytewise, one name on
    two chromosomes, lines on a chromosome the dataset does not have."""
    rng = np.random.default_rng(seed)
    lines = []
    m = len(bps)
    for g in range(60):
        v = int(rng.integers(0, m))
        name = "gene%d" % (g % 37) if g % 5 else "%dorf%d" % (g % 7, g)
        for _ in range(int(rng.integers(1, 4))):
            a = max(1, int(bps[v]) - int(rng.integers(0, 3000)))
            b = a + int(rng.integers(0, 6000))
            lines.append("%s %d %d %s" % (chroms[v], a - (1 if zero_based else 0), b, name))
    lines.append("22 1 1000000 elsewhere")
    lines.append("%s 1 2 tiny" % chroms[0])
    rng.shuffle(lines)
    open(path, "w").write("\n".join(lines) + "\n")


@needs_ref
@pytest.mark.parametrize("flag,extra", [
    ("--clump-range", []),
    ("--clump-range0", []),
    ("--clump-range", ["--clump-range-border", "2.5"]),
    ("--clump-range", ["--clump-range-border", "40", "--clump-p1", "0.01", "--clump-p2", "1e-6"]),   # index variants above p2: no bounds, no ranges
    ("--clump-range", ["cols=-maybebounds,-sp2,-bins"]),
])
def test_clump_ranges_match_reference(cli, tmp_path, flag, extra):
    """--clump-range[0] / --clump-range-border: the RANGES column and the bounds that come with it (plink2_set.cc:39-330, :495-638;
    plink2_ld.cc:9265-9400)."""
    m = 1000
    prefix, raw, chroms, bps = clump_fileset(tmp_path, m, 40, 4)
    write_report(str(tmp_path / "assoc.txt"), m, 9)
    write_ranges(str(tmp_path / "genes.txt"), chroms, bps, 1, zero_based=(flag == "--clump-range0"))
    mods = [x for x in extra if x.startswith("cols=")]
    rest = [x for x in extra if not x.startswith("cols=")]
    common = ["--bfile", "d", "--clump"] + mods + ["assoc.txt", "--clump-unphased", "--clump-kb", "0.001", flag, "genes.txt"] + rest
    compare_runs(cli, tmp_path, common)
    body = open(str(tmp_path / "hip.clumps")).read().split("\n")
    assert body[0].endswith("RANGES") and any(("," in l.split("\t")[-1]) for l in body[1:-1]), "some clump must meet several regions"


def add_a1_column(src, dst, seed, name="A1"):
    """An allele column behind the report's columns: mostly the dataset's ALT (C) or REF (A), now and then neither."""
    rng = np.random.default_rng(seed)
    lines = open(src).read().splitlines()
    out = [lines[0] + "\t" + name]
    for ln in lines[1:]:
        out.append(ln + "\t" + str(rng.choice(["C", "A", "T", "AC"], p=[0.55, 0.35, 0.07, 0.03])))
    open(dst, "w").write("\n".join(out) + "\n")


@needs_ref
@pytest.mark.parametrize("mods,extra,two", [
    ([], [], False),
    (["cols=+a1,+bounds"], ["--clump-p1", "0.01", "--clump-p2", "1e-4"], False),     # the bounds scan reads the forced-A1 bit too
    (["cols=+a1,+f"], [], True),
    (["cols=+bounds,-total"], ["--clump-p1", "0.05", "--clump-p2", "1e-5"], True),
    (["cols=-sp2,+a1"], [], False),
])
def test_clump_force_a1_matches_reference(cli, tmp_path, mods, extra, two):
    """--clump-force-a1 on a biallelic dataset (ClumpReports :7783-7818, :9186-9196, :9355-9358): lines whose A1 is neither allele
    drop out (the top ones into .clumps.missing_allele), the A1 column and the SP2 suffixes carry the allele."""
    m = 900
    clump_fileset(tmp_path, m, 40, 10)
    write_report(str(tmp_path / "p.txt"), m, 21)
    add_a1_column(str(tmp_path / "p.txt"), str(tmp_path / "a.txt"), 1)
    files = ["a.txt"]
    if two:
        write_report(str(tmp_path / "q.txt"), m, 22, sig_rate=0.1)
        add_a1_column(str(tmp_path / "q.txt"), str(tmp_path / "b.txt"), 2)
        files.append("b.txt")
    common = ["--bfile", "d", "--clump"] + mods + files + ["--clump-unphased", "--clump-kb", "0.001", "--clump-force-a1"] + extra
    compare_runs(cli, tmp_path, common)
    assert os.path.exists(str(tmp_path / "hip.clumps.missing_allele"))


ALT_TEXT = ["C", "G", "T", "AC", "AG", "AT", "CA", "CC", "CG", "CT"]   # (ldtools.write_vcf_haps; REF is "A")


def multiallelic_clump_fileset(tmp_path, m, n, seed, chrom_of=None, max_alt=4, multi_rate=0.35, spacing=700):
    """LD-carrying phased haplotypes with 1..max_alt ALT alleles per site -> VCF -> the reference's own import (variable-width .pgen with the
    multiallelic track).  Returns (alt_ct, chroms, bps)."""
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed, max_alt=max_alt, multi_rate=multi_rate, missing_rate=0.02, ld_copy_prob=0.75, redraw=0.05)
    rng = np.random.default_rng(seed + 5)
    chroms = [chrom_of(v) for v in range(m)] if chrom_of else [str(1 + (3 * v) // m) for v in range(m)]
    bps = [int(x) for x in 1000 + np.cumsum(rng.integers(1, spacing, size=m))]
    # (imported as one chromosome, the .pvar re-labelled afterwards: the reference's VCF import wants sex information for chrX / chrY)
    T.write_vcf_haps(str(tmp_path / "d.vcf"), first, second, alt_ct, ["1"] * m, bps)
    T.ref_import_vcf(str(tmp_path / "d.vcf"), str(tmp_path / "d"))
    out, k = [], 0
    for ln in open(str(tmp_path / "d.pvar")):
        if not ln.startswith("#"):
            f = ln.split("\t")
            f[0] = chroms[k]
            ln = "\t".join(f)
            k += 1
        out.append(ln)
    open(str(tmp_path / "d.pvar"), "w").write("".join(out))
    return alt_ct, chroms, bps


def write_allele_report(path, alt_ct, seed, one_allele_per_variant, sig_rate=0.12, a1_name="A1"):
    """A report with an allele column: (ID, A1) lines over REF and every ALT of the multiallelic sites, now and then an allele the variant
    does not have; one_allele_per_variant: every line of a variant names the same allele (nothing shares a position then)."""
    rng = np.random.default_rng(seed)
    m = len(alt_ct)
    lines = ["#CHROM\tPOS\tID\t%s\tTEST\tOBS_CT\tP" % a1_name]
    fixed = np.random.default_rng(12345).integers(0, 64, size=m)   # (the same choice in every report of a test)
    for v in rng.permutation(m):
        k = int(alt_ct[v]) + 1
        names = ["A"] + ALT_TEXT[:k - 1]
        reps = 1 + int(rng.random() < 0.5) + int(rng.random() < 0.2) + (2 if (k > 2 and not one_allele_per_variant) else 0)
        for _ in range(reps):
            a1 = names[fixed[v] % k] if one_allele_per_variant else names[int(rng.integers(k))]
            if rng.random() < 0.04:
                a1 = ["GT", "N", "CT"][int(rng.integers(3))] if k < 10 else "N"
            u = rng.random()
            p = 10.0 ** (-rng.uniform(4, 30)) if u < sig_rate else (10.0 ** (-rng.uniform(1, 4.3)) if u < 3 * sig_rate else rng.random())
            lines.append("1\t1\tsnp%d\t%s\tADD\t100\t%s" % (v, a1, P_FORMATS[rng.integers(len(P_FORMATS))] % p))
    open(path, "w").write("\n".join(lines) + "\n")


@needs_ref
@pytest.mark.parametrize("mods,extra,two", [
    ([], [], False),
    (["cols=+a1,+bounds,+f"], ["--clump-p1", "0.01", "--clump-p2", "1e-3"], True),
    ([], ["--clump-force-a1"], False),                                   # the stale forced-A1 bit on the multiallelic entries (SP2)
    (["cols=+bounds,-total"], ["--clump-force-a1", "--clump-p1", "0.05", "--clump-p2", "1e-4"], True),
    (["cols=-maybea1"], ["--clump-a1-field", "EFFECT"], False),
])
def test_clump_multiallelic_report_handling_matches_reference(cli, tmp_path, mods, extra, two):
    """A dataset with multiallelic sites: the reference clumps (variant, A1 allele) pairs (ClumpReports, plink2_ld.cc:7776-7819, :9186-9201,
    :9355-9358).  Here every variant is named with one allele only and positions are distinct, so a 1-bp radius pairs nothing and the report
    side -- allele lookup, .clumps.missing_allele, the A1 column, the SP2 suffixes -- is compared without a GPU."""
    m = 700
    alt_ct, _, _ = multiallelic_clump_fileset(tmp_path, m, 60, 31, chrom_of=lambda v: "1")
    a1_name = "EFFECT" if "EFFECT" in extra else "A1"
    write_allele_report(str(tmp_path / "a.txt"), alt_ct, 41, True, a1_name=a1_name)
    files = ["a.txt"]
    if two:
        write_allele_report(str(tmp_path / "b.txt"), alt_ct, 42, True, sig_rate=0.2, a1_name=a1_name)
        files.append("b.txt")
    common = ["--pfile", "d", "--clump"] + mods + files + ["--clump-unphased", "--clump-kb", "0.001"] + extra
    compare_runs(cli, tmp_path, common)
    head = open(str(tmp_path / "hip.clumps")).readline()
    assert ("A1" in head.split("\t")) == ("cols=-maybea1" not in mods)
    assert os.path.exists(str(tmp_path / "hip.clumps.missing_allele"))


@needs_ref
def test_clump_multiallelic_line_without_an_a1_column(cli, tmp_path):
    alt_ct, _, _ = multiallelic_clump_fileset(tmp_path, 120, 50, 32, chrom_of=lambda v: "1")
    write_allele_report(str(tmp_path / "a.txt"), alt_ct, 43, True, a1_name="ALLELE")
    base = ["--pfile", "d", "--clump", "a.txt", "--clump-unphased", "--clump-kb", "0.001"]
    for args in ([], ["--clump-a1-field"]):
        ref = T.run_ref(base + args + ["--out", "ref"], str(tmp_path))
        got = run_cli(cli, base + args + ["--out", "hip"], str(tmp_path))
        assert ref.returncode == got.returncode == 7 and "is multiallelic, but there is no A1 column" in got.stdout, (args, ref.returncode, got.returncode, got.stdout[-300:])


@needs_ref
def test_clump_a1_field_rules(cli, tmp_path):
    m = 300
    clump_fileset(tmp_path, m, 40, 3)
    write_report(str(tmp_path / "p.txt"), m, 4)
    add_a1_column(str(tmp_path / "p.txt"), str(tmp_path / "a.txt"), 5, name="EFFECT")
    base = ["--bfile", "d", "--clump", "a.txt", "--clump-unphased", "--clump-kb", "0.001"]
    compare_runs(cli, tmp_path, base + ["--clump-force-a1", "--clump-a1-field", "ALLELE", "EFFECT"])
    compare_runs(cli, tmp_path, base + ["--clump-a1-field", "EFFECT"])                     # without the force flag a biallelic dataset ignores it
    for args, code in ((["--clump-force-a1"], 7), (["--clump-force-a1", "--clump-a1-field"], 8)):
        ref = T.run_ref(base + args + ["--out", "ref"], str(tmp_path))
        got = run_cli(cli, base + args + ["--out", "hip"], str(tmp_path))
        assert ref.returncode == got.returncode == code, (args, ref.returncode, got.returncode, got.stdout[-300:])


@needs_ref
def test_clump_zs_outputs_decompress_to_the_reference_text(cli, tmp_path):
    import subprocess
    m = 800
    clump_fileset(tmp_path, m, 40, 14)
    write_report(str(tmp_path / "assoc.txt"), m, 15)
    common = ["--bfile", "d", "--clump", "zs", "cols=+bounds", "assoc.txt", "--clump-unphased", "--clump-kb", "0.001"]
    ref = T.run_ref(common + ["--out", "ref"], str(tmp_path))
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert ref.returncode == 0 and got.returncode == 0, (ref.stdout[-300:], got.stdout[-300:])
    for ext in (".clumps.zst", ".clumps.missing_id.zst"):
        a = subprocess.run([T.REF_BIN, "--zst-decompress", str(tmp_path / ("ref" + ext))], stdout=subprocess.PIPE).stdout
        b = subprocess.run([T.REF_BIN, "--zst-decompress", str(tmp_path / ("hip" + ext))], stdout=subprocess.PIPE).stdout
        assert len(a) > 50 and a == b, ext


@needs_ref
@pytest.mark.parametrize("storage", [1, 2, 3])
def test_provisional_ref_column_follows_the_pgen_header(cli, pkg, tmp_path, storage):
    """PROVISIONAL_REF? ('maybeprovref' with a REF column): the .pgen header says none / all / these (control bits 6-7 and the
    flag array behind the header, pgen_spec.tex:196-206) -- through the reader's ldp_pgen_provisional_ref and the column writer."""
    m = 300
    prefix, raw, chroms, bps = clump_fileset(tmp_path, m, 40, 6)
    rng = np.random.default_rng(storage)
    flags = rng.random(m) < 0.3
    data = open(prefix + ".pgen", "rb").read()
    assert data[:3] == bytes([0x6C, 0x1B, 0x02]) and data[11] == 0x40
    body = data[12:]
    bits = np.packbits(flags, bitorder="little").tobytes() if storage == 3 else b""
    open(prefix + ".pgen", "wb").write(data[:11] + bytes([storage << 6]) + bits + body)
    write_report(str(tmp_path / "assoc.txt"), m, 2)
    f = pkg.PgenFile(prefix + ".pgen")
    got_bits = np.zeros((m + 7) // 8, dtype=np.uint8)
    import ctypes
    L = pkg.lib()
    L.ldp_pgen_provisional_ref.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint64]
    assert L.ldp_pgen_provisional_ref(f._h, got_bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(got_bits)) == storage
    if storage == 3:
        assert np.array_equal(np.unpackbits(got_bits, bitorder="little")[:m].astype(bool), flags)
    f.close()
    compare_runs(cli, tmp_path, ["--pfile", "d", "--clump", "cols=+ref", "assoc.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.05"])
    hdr = open(str(tmp_path / "hip.clumps")).readline()
    assert ("PROVISIONAL_REF?" in hdr) == (storage != 1)


@needs_ref
@pytest.mark.parametrize("with_header", [True, False])
def test_provisional_ref_column_from_the_pvar_info(cli, tmp_path, with_header):
    """A .pgen that leaves the provisional-REF flags to the .pvar (control bits 6-7 = 0): INFO/PR marks the variants -- PR alone, first,
    last, or in the middle of the INFO keys (PrInInfo, plink2_pvar.cc:561) -- when the header declares the flag."""
    m = 300
    prefix, raw, chroms, bps = clump_fileset(tmp_path, m, 40, 7)
    data = open(prefix + ".pgen", "rb").read()
    open(prefix + ".pgen", "wb").write(data[:11] + bytes([0x00]) + data[12:])
    rng = np.random.default_rng(9)
    infos = ["PR", "PR;AC=3", "AC=3;PR", "AC=3;PR;DP=9", "AC=3", ".", "APR;DP=2", "DP=2;PRX", "XPR"]
    lines = open(prefix + ".pvar").read().splitlines()
    out = ['##INFO=<ID=PR,Number=0,Type=Flag,Description="Provisional reference allele, may not be based on real reference genome">'] if with_header else []
    out.append(lines[0] + "\tQUAL\tFILTER\tINFO")
    for ln in lines[1:]:
        out.append(ln + "\t.\t.\t" + infos[int(rng.integers(len(infos)))])
    open(prefix + ".pvar", "w").write("\n".join(out) + "\n")
    write_report(str(tmp_path / "assoc.txt"), m, 3)
    for mods in (["cols=+ref"], ["cols=+ref,+provref"]):
        compare_runs(cli, tmp_path, ["--pfile", "d", "--clump"] + mods + ["assoc.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.05"])
    body = open(str(tmp_path / "hip.clumps")).read()
    assert ("\tY\t" in body) == with_header


@needs_ref
@pytest.mark.parametrize("mods", [[], ["just-acgt"]])
def test_snps_only_in_front_of_the_command(cli, tmp_path, mods):
    """--snps-only ['just-acgt'] drops variants while the table loads (LoadPvar, plink2_pvar.cc:1917-1932): seen here through
    --clump, whose report then names IDs the dataset no longer has."""
    m = 400
    prefix, raw, chroms, bps = clump_fileset(tmp_path, m, 40, 12)
    rng = np.random.default_rng(1)
    rows = [ln.split("\t") for ln in open(prefix + ".bim").read().splitlines()]
    for r_ in rows:
        r_[4], r_[5] = [("C", "A"), ("CT", "A"), ("C", "AG"), ("N", "A"), ("c", "t"), (".", "G"), ("<DEL>", "A"), ("*", "T")][int(rng.choice(8, p=[0.6, 0.07, 0.07, 0.06, 0.05, 0.05, 0.05, 0.05]))]
    open(prefix + ".bim", "w").write("\n".join("\t".join(r_) for r_ in rows) + "\n")
    write_report(str(tmp_path / "assoc.txt"), m, 13)
    compare_runs(cli, tmp_path, ["--bfile", "d", "--snps-only"] + mods + ["--clump", "assoc.txt", "--clump-unphased", "--clump-kb", "0.001", "--clump-p1", "0.01"])
    assert os.path.getsize(str(tmp_path / "hip.clumps.missing_id")) > 0


def test_clump_flag_rules(cli, tmp_path):
    clump_fileset(tmp_path, 60, 20, 3)
    write_report(str(tmp_path / "a.txt"), 60, 1)
    r = run_cli(cli, ["--bfile", "d", "--clump", "a.txt"], str(tmp_path))
    assert r.returncode == 63 and "--clump-unphased" in r.stdout
    r = run_cli(cli, ["--bfile", "d", "--clump", "a.txt", "--clump-unphased", "--clump-r2", "1.0"], str(tmp_path))
    assert r.returncode == 8 and "Invalid --clump-r2" in r.stdout
    for args in (["--clump-range-border", "5"], ["--clump-range0", "a.txt", "--clump-range-border", "5"]):
        r = run_cli(cli, ["--bfile", "d", "--clump", "a.txt", "--clump-unphased"] + args, str(tmp_path))
        assert r.returncode == 8 and "must be used with --clump-range" in r.stdout
        if T.have_ref():
            assert T.run_ref(["--bfile", "d", "--clump", "a.txt", "--clump-unphased"] + args + ["--out", "ref"], str(tmp_path)).returncode == 8
    r = run_cli(cli, ["--bfile", "d", "--clump-unphased"], str(tmp_path))
    assert r.returncode == 8
    for args, needle in ((["--clump-bins", "0.01,0.001"], "not in increasing order"), (["--clump-bins", "0.5,1"], "values >= 1"),
                         (["--clump-bins", "0.01x"], "Invalid --clump-bins argument"), (["cols=-bins", "--clump-bins", "0.01"], "has been excluded"),
                         (["cols=+nope"], "Unrecognized ID 'nope' in --clump column set descriptor.")):
        mods = [x for x in args if x.startswith("cols=")]
        rest = [x for x in args if not x.startswith("cols=")]
        r = run_cli(cli, ["--bfile", "d", "--clump"] + mods + ["a.txt", "--clump-unphased"] + rest, str(tmp_path))
        assert r.returncode == 8 and needle in r.stdout.replace("\n", " "), (args, r.returncode, r.stdout[-300:])
        if T.have_ref() and "increasing" not in needle:   # (the reference's order check never updates its running value: it lets these through)
            assert T.run_ref(["--bfile", "d", "--clump"] + mods + ["a.txt", "--clump-unphased"] + rest + ["--out", "ref"], str(tmp_path)).returncode == 8, args
    for args in (["--clump-log10", "sideways"], ["--clump-p1", "0.01", "--clump-log10-p1", "2"], ["--clump-log10-p2", "-1"]):
        r = run_cli(cli, ["--bfile", "d", "--clump", "a.txt", "--clump-unphased"] + args, str(tmp_path))
        assert r.returncode == 8, (args, r.returncode, r.stdout[-200:])
        if T.have_ref():
            assert T.run_ref(["--bfile", "d", "--clump", "a.txt", "--clump-unphased"] + args + ["--out", "ref"], str(tmp_path)).returncode == 8, args
    r = run_cli(cli, ["--bfile", "d", "--clump", "a.txt", "cols=+f", "--clump-unphased"], str(tmp_path))
    assert r.returncode == 8 and "must come before" in r.stdout


@pytest.mark.gpu
def test_clump_several_reports_match_reference(gpu_pkg, cli, tmp_path):
    assert T.have_ref()
    m = 3000
    clump_fileset(tmp_path, m, 210, 3, spacing=400)
    write_report(str(tmp_path / "a.txt"), m, 5)
    write_report(str(tmp_path / "b.txt"), m, 6, sig_rate=0.03)
    compare_runs(cli, tmp_path, ["--pfile", "d", "--clump", "a.txt", "b.txt", "--clump-unphased", "--clump-r2", "0.3", "--clump-kb", "80"])
    body = open(str(tmp_path / "hip.clumps")).read().split("\n")[1:-1]
    assert any(l.split("\t")[-1] != "." for l in body)


CLUMP_CASES = [
    # fmt, m, n, missing, extra
    ("bfile", 3000, 200, 0.0, []),
    ("pfile", 3000, 333, 0.03, ["--clump-r2", "0.2", "--clump-kb", "60"]),
    ("bfile", 5000, 150, 0.01, ["--clump-r2", "0.8", "--clump-p1", "1e-3", "--clump-p2", "0.05", "--clump-kb", "1000"]),
    ("pfile", 2500, 97, 0.05, ["--clump-r2", "0", "--clump-kb", "20"]),
    ("pfile", 4000, 1200, 0.002, ["--clump-r2", "0.35", "--clump-p1", "0.05", "--clump-p2", "0.5"]),
    ("bfile", 3000, 200, 0.01, ["--clump-allow-overlap", "--clump-r2", "0.15", "--clump-p1", "0.01", "--clump-p2", "0.1", "--clump-kb", "100"]),
    ("pfile", 2500, 120, 0.0, ["--clump-allow-overlap", "--clump-r2", "0.05", "--clump-p1", "0.05", "--clump-kb", "30"]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CLUMP_CASES)
def test_clump_matches_reference(gpu_pkg, cli, tmp_path, case):
    fmt, m, n, miss, extra = case
    assert T.have_ref(), "reference binary oracle/_ref/plink2 must travel with the repo snapshot"
    clump_fileset(tmp_path, m, n, m + n, missing_rate=miss, spacing=400)
    write_report(str(tmp_path / "assoc.txt"), m, n)
    common = ["--" + fmt, "d", "--clump", "assoc.txt", "--clump-unphased"] + extra
    ref, got = compare_runs(cli, tmp_path, common)
    body = open(str(tmp_path / "hip.clumps")).read().split("\n")[1:-1]
    assert any(l.split("\t")[-1] != "." for l in body), "the case must form multi-variant clumps"


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,unknown,extra", [
    ("pfile", True, ["--clump-r2", "0.2", "--clump-kb", "30", "--clump-p1", "0.01", "--clump-p2", "0.2"]),
    ("bfile", False, ["--clump-r2", "0.05", "--clump-kb", "100", "--clump-p1", "0.05", "--clump-p2", "0.5", "--clump-allow-overlap"]),
    ("pfile", True, ["--clump-r2", "0.5", "--clump-kb", "15", "--clump-p1", "0.001"]),
])
def test_clump_with_sex_chromosomes_matches_reference(gpu_pkg, cli, tmp_path, fmt, unknown, extra):
    """chrX windows through the male-weighted sums (ComputeXR2 with both variants on chrX), chrY rows with the female founders'
    calls missing, MT as it is (ClumpReports, plink2_ld.cc:8150-8215, :8460-8482, :8823-8835)."""
    assert T.have_ref()
    from test_cli import sexed_fileset
    m = 900
    sexed_fileset(tmp_path, m=m, n=140, seed=11, unknown_sex=unknown)
    write_report(str(tmp_path / "assoc.txt"), m, 3, sig_rate=0.1)
    common = ["--" + fmt, "sx", "--clump", "assoc.txt", "--clump-unphased"] + extra
    compare_runs(cli, tmp_path, common)
    body = open(str(tmp_path / "hip.clumps")).read().split("\n")[1:-1]
    by_chr = {}
    for l in body:
        f = l.split("\t")
        by_chr.setdefault(f[0], []).append(f[-1] != ".")
    assert any(by_chr.get("X", [])) and any(by_chr.get("Y", [])), "multi-variant clumps on chrX and chrY must be formed"


@pytest.mark.gpu
@pytest.mark.parametrize("n,nonfounders,mods,extra", [
    (180, False, [], ["--clump-r2", "0.2", "--clump-kb", "40"]),
    (251, True, ["cols=+a1,+bounds,+f"], ["--clump-r2", "0.1", "--clump-kb", "60", "--clump-p1", "0.01", "--clump-p2", "0.1"]),
    (90, False, [], ["--clump-force-a1", "--clump-r2", "0.3", "--clump-kb", "25"]),
    (333, False, ["cols=+alt,+ref"], ["--clump-allow-overlap", "--clump-r2", "0.05", "--clump-p1", "0.05", "--clump-kb", "30"]),
])
def test_clump_multiallelic_matches_reference(gpu_pkg, cli, tmp_path, n, nonfounders, mods, extra):
    """(variant, A1 allele) pairs as clump members (ClumpHighmemR2, plink2_ld.cc:7282-7470, rows by PgrGetInv1 :8813-8835): every allele of a
    multiallelic site is its own row -- copies of the other alleles -- with its own p-values; alleles of one site pair with each other too."""
    assert T.have_ref()
    m = 1500
    alt_ct, _, _ = multiallelic_clump_fileset(tmp_path, m, n, 100 + n)
    if nonfounders:
        lines = open(str(tmp_path / "d.psam")).read().splitlines()
        out = ["#IID\tPAT\tMAT\tSEX"]
        for k, ln in enumerate(lines[1:]):
            iid = ln.split("\t")[0]
            out.append("%s\t%s\t0\tNA" % (iid, "s0" if (k % 7 == 3) else "0"))
        open(str(tmp_path / "d.psam"), "w").write("\n".join(out) + "\n")
    write_allele_report(str(tmp_path / "a.txt"), alt_ct, n, False)
    files = ["a.txt"]
    if "cols=+a1,+bounds,+f" in mods:
        write_allele_report(str(tmp_path / "b.txt"), alt_ct, n + 1, False, sig_rate=0.05)
        files.append("b.txt")
    compare_runs(cli, tmp_path, ["--pfile", "d", "--clump"] + mods + files + ["--clump-unphased"] + extra)
    body = [l.split("\t") for l in open(str(tmp_path / "hip.clumps")).read().split("\n")[1:-1]]
    assert any(("(" in f[-1]) and (f[-1] != ".") for f in body), "clumps with allele-named members must be formed"


@pytest.mark.gpu
@pytest.mark.parametrize("unknown,extra", [
    (True, ["--clump-r2", "0.15", "--clump-kb", "40", "--clump-p1", "0.01", "--clump-p2", "0.2"]),
    (False, ["--clump-force-a1", "--clump-r2", "0.3", "--clump-kb", "25", "--clump-p1", "0.05"]),
])
def test_clump_multiallelic_on_sex_chromosomes_matches_reference(gpu_pkg, cli, tmp_path, unknown, extra):
    """... and on chrX (male-weighted sums, ComputeXR2), chrY (the female founders' calls missing) and MT."""
    assert T.have_ref()
    m, n = 1200, 150
    names = ["1", "X", "Y", "MT"]
    alt_ct, chroms, _ = multiallelic_clump_fileset(tmp_path, m, n, 77, chrom_of=lambda v: names[(4 * v) // m])
    rng = np.random.default_rng(9)
    out = ["#IID\tSEX"]
    for s in range(n):
        sx = int(rng.integers(1, 3))
        if unknown and rng.random() < 0.1:
            sx = "NA"
        out.append("s%d\t%s" % (s, sx))
    open(str(tmp_path / "d.psam"), "w").write("\n".join(out) + "\n")
    write_allele_report(str(tmp_path / "a.txt"), alt_ct, 5, False, sig_rate=0.15)
    compare_runs(cli, tmp_path, ["--pfile", "d", "--clump", "a.txt", "--clump-unphased"] + extra)
    body = [l.split("\t") for l in open(str(tmp_path / "hip.clumps")).read().split("\n")[1:-1]]
    for c in ("X", "Y"):
        assert any((f[0] == c) and ("(" in f[-1]) for f in body), "allele-named members on chr" + c


@pytest.mark.gpu
def test_clump_chrx_windows_on_the_pair_kernels_equal_the_pair_lists(gpu_pkg, cli, tmp_path):
    """The chrX windows of --clump come from two all-pairs engines over the chrX run (ldp_r2_unphased_block_x_hits: pair kernels + the device-side
    weighting, round 5); --debug-x-host keeps the pair lists and the host arithmetic; --debug-x-rows cuts the device path into many chunks: one file."""
    from test_cli import sexed_fileset
    m = 900
    sexed_fileset(tmp_path, m=m, n=140, seed=12, unknown_sex=True)
    write_report(str(tmp_path / "assoc.txt"), m, 4, sig_rate=0.12)
    common = ["--pfile", "sx", "--clump", "assoc.txt", "--clump-unphased", "--clump-r2", "0.1", "--clump-kb", "60", "--clump-p1", "0.01", "--clump-p2", "0.2"]
    outs = []
    for tag, hook in (("dev", []), ("chunks", ["--debug-x-rows", "40"]), ("host", ["--debug-x-host"])):
        got = run_cli(cli, common + hook + ["--out", tag], str(tmp_path))
        assert got.returncode == 0, got.stdout
        outs.append(open(str(tmp_path / (tag + ".clumps"))).read())
    assert outs[0] == outs[2] and outs[1] == outs[2]
    assert any((l.split("\t")[0] == "X") and (l.split("\t")[-1] != ".") for l in outs[0].split("\n")[1:-1])
