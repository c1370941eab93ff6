"""Variant and sample filters in front of the commands (--chr / --not-chr / --autosome, --extract / --exclude, --keep /
--remove): plink2-hip's output files against the reference binary's on the same command line."""
import filecmp
import os

import numpy as np
import pytest

import ldtools as T
from test_cli import cli, run_cli  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def fileset(tmp_path, m=1200, n=150, seed=3, with_x=True):
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.02, ld_copy_prob=0.6, redraw=0.05)
    names = ["0"] * 5 + ["1"] * 400 + ["2"] * 300 + ["3"] * 200 + ["7"] * 145 + (["X"] * 150 if with_x else ["9"] * 150)
    assert len(names) == m
    rng = np.random.default_rng(seed)
    pos = np.zeros(m, dtype=np.int64)
    start = 0
    for cnt in (5, 400, 300, 200, 145, 150):
        pos[start:start + cnt] = np.sort(rng.choice(np.arange(3000000, 3400000), size=cnt, replace=False))
        start += cnt
    sexes = rng.choice([1, 2], size=n)
    prefix = str(tmp_path / "d")
    T.write_bed(prefix, raw, names, pos)
    T.write_pgen_fixed(prefix, raw, names, pos, sexes=sexes)
    fam = ["f%d s%d %s %s %d -9" % (s // 3, s, "s0" if s % 13 == 5 else "0", "s1" if s % 13 == 5 else "0", sexes[s]) for s in range(n)]
    open(prefix + ".fam", "w").write("\n".join(fam) + "\n")
    psam = ["#IID\tPAT\tMAT\tSEX"] + ["s%d\t%s\t%s\t%d" % (s, "s0" if s % 13 == 5 else "0", "s1" if s % 13 == 5 else "0", sexes[s]) for s in range(n)]
    open(prefix + ".psam", "w").write("\n".join(psam) + "\n")
    with open(str(tmp_path / "vars.txt"), "w") as f:
        f.write("\n".join("snp%d" % v for v in rng.choice(m, size=m // 2, replace=False)) + "\nnot_in_the_file\n")
    with open(str(tmp_path / "vars2.txt"), "w") as f:
        f.write(" ".join("snp%d" % v for v in rng.choice(m, size=m // 6, replace=False)) + "\n")
    pick = rng.choice(n, size=n * 2 // 3, replace=False)
    with open(str(tmp_path / "keep_fam.txt"), "w") as f:          # FID IID, no header
        f.write("".join("f%d s%d\n" % (s // 3, s) for s in pick) + "f999 nobody\nf0 s0\nf0 s0\n")
    with open(str(tmp_path / "keep_iid.txt"), "w") as f:          # IID only, no header (matches FID 0: the .psam has no FID column)
        f.write("".join("s%d\n" % s for s in pick))
    with open(str(tmp_path / "rm_hdr.txt"), "w") as f:            # header line naming the columns
        f.write("#IID\tNOTE\n" + "".join("s%d\tx\n" % s for s in pick[:40]))
    with open(str(tmp_path / "rm_fam_hdr.txt"), "w") as f:
        f.write("# a comment\n#FID\tIID\n" + "".join("f%d\ts%d\n" % (s // 3, s) for s in pick[:40]))
    return prefix


def compare(cli, tmp_path, args, exts, may_be_empty=False):
    ref = T.run_ref(args + ["--threads", "4", "--out", "ref"], str(tmp_path))
    got = run_cli(cli, args + ["--out", "hip"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout[-1500:]
    assert got.returncode == 0, got.stdout[-1500:]
    for e in exts:
        a, b = str(tmp_path / ("ref" + e)), str(tmp_path / ("hip" + e))
        assert (e == ".prune.out") or may_be_empty or (os.path.getsize(a) > 20), e
        assert filecmp.cmp(a, b, shallow=False), (e, " ".join(args))
    return ref, got


PRUNE = ["--indep-pairwise", "60kb", "0.3", "--bad-ld"]
CASES = [
    (["--bfile", "d", "--chr", "1,3"] + PRUNE, None),
    (["--bfile", "d", "--chr", "1-2,", "7"] + PRUNE, None),
    (["--pfile", "d", "--chr", "chr2", "X"] + PRUNE, None),
    (["--bfile", "d", "--not-chr", "2", "X"] + PRUNE, None),
    (["--pfile", "d", "--autosome"] + PRUNE, None),
    (["--bfile", "d", "--extract", "vars.txt"] + PRUNE, "--extract:"),
    (["--pfile", "d", "--exclude", "vars.txt"] + PRUNE, "--exclude:"),
    (["--bfile", "d", "--extract", "vars.txt", "--exclude", "vars2.txt", "--chr", "1-3"] + PRUNE, "--exclude:"),
    (["--bfile", "d", "--keep", "keep_fam.txt"] + PRUNE, "--keep:"),
    (["--pfile", "d", "--keep", "keep_iid.txt"] + PRUNE, "--keep:"),
    (["--pfile", "d", "--remove", "rm_hdr.txt"] + PRUNE, "--remove:"),
    (["--bfile", "d", "--remove", "rm_fam_hdr.txt", "--keep", "keep_fam.txt", "--not-chr", "X"] + PRUNE, "--remove:"),
]


@pytest.mark.parametrize("args,log_key", CASES)
def test_filters_in_front_of_the_prune(gpu_pkg, cli, tmp_path, args, log_key):
    assert T.have_ref()
    fileset(tmp_path)
    ref, got = compare(cli, tmp_path, args, [".prune.in", ".prune.out"])
    if log_key:
        want = [l.strip() for l in ref.stdout.split("\n") if l.startswith(log_key)]
        assert want and want[0] in got.stdout, (want, got.stdout[-600:])


def test_filters_in_front_of_the_r2_table_and_clump(gpu_pkg, cli, tmp_path):
    assert T.have_ref()
    fileset(tmp_path, with_x=False)
    compare(cli, tmp_path, ["--bfile", "d", "--chr", "2-7", "--exclude", "vars2.txt", "--keep", "keep_fam.txt", "--r2-unphased", "--ld-window-kb", "40",
                            "--ld-window-r2", "0.1"], [".vcor"])
    import test_clump as TC
    TC.write_report(str(tmp_path / "assoc.txt"), 1200, 5, sig_rate=0.1)
    compare(cli, tmp_path, ["--pfile", "d", "--extract", "vars.txt", "--remove", "rm_hdr.txt", "--clump", "assoc.txt", "--clump-unphased", "--clump-r2", "0.2",
                            "--clump-kb", "100", "--clump-p1", "0.01"], [".clumps"])


COUNT_CASES = [
    (["--bfile", "d", "--maf", "0.05"] + PRUNE, "removed due to allele frequency"),
    (["--pfile", "d", "--maf"] + PRUNE, "removed due to allele frequency"),
    (["--bfile", "d", "--geno", "0.02"] + PRUNE, "--geno:"),
    (["--pfile", "d", "--geno", "0.017", "--maf", "0.1", "--max-maf", "0.4"] + PRUNE, "removed due to allele frequency"),
    (["--bfile", "d", "--keep", "keep_fam.txt", "--geno", "0.02", "--maf", "0.03", "--chr", "1-3"] + PRUNE, "--geno:"),
    (["--pfile", "v", "--maf", "0.2", "--geno"] + PRUNE, "--geno:"),   # variable-width records: decoded for the counts
]


@pytest.mark.parametrize("args,log_key", COUNT_CASES)
def test_count_based_filters(gpu_pkg, cli, tmp_path, args, log_key):
    """--maf / --max-maf (founders' nonmajor-allele frequency) and --geno (kept samples' missing-call rate): the thresholds with
    the reference's epsilons, on data whose frequencies and missing rates straddle them."""
    assert T.have_ref()
    fileset(tmp_path, with_x=False)
    if "v" in args:
        mk = T.run_ref(["--pfile", "d", "--make-pgen", "--out", "v"], str(tmp_path))
        assert mk.returncode == 0, mk.stdout
    ref, got = compare(cli, tmp_path, args, [".prune.in", ".prune.out"])
    want = [l.strip() for l in ref.stdout.split("\n") if log_key in l]
    assert want and want[0] in got.stdout, (want, got.stdout[-800:])
    n_in = len(open(str(tmp_path / "ref.prune.in")).read().split())
    n_out = len(open(str(tmp_path / "ref.prune.out")).read().split())
    assert 50 < n_in + n_out < 1195   # the filters removed something and left something


def test_count_based_filters_refuse_what_they_do_not_cover(gpu_pkg, cli, tmp_path):
    fileset(tmp_path)
    r = run_cli(cli, ["--bfile", "d", "--maf", "0.05"] + PRUNE, str(tmp_path))
    assert r.returncode == 63 and "chrX" in r.stdout
    r = run_cli(cli, ["--bfile", "d", "--maf", "0.05:minor"] + PRUNE, str(tmp_path))
    assert r.returncode == 63
    r = run_cli(cli, ["--bfile", "d", "--maf", "1.5"] + PRUNE, str(tmp_path))
    assert r.returncode == 8


@pytest.mark.parametrize("extra", [["--max-alleles", "2"], ["--max-alleles", "3", "--geno", "0.08"], ["--max-alleles", "2", "--maf", "0.1"]])
def test_max_alleles_filter(gpu_pkg, cli, tmp_path, extra):
    """--max-alleles drops the multiallelic sites while the variant table loads (LoadPvar); with 2 the count-based filters
    are available on a file that has such sites."""
    from test_pgen_reader import make_multiallelic_vcf
    assert T.have_ref()
    make_multiallelic_vcf(str(tmp_path / "m.vcf"), 700, 160, seed=4, max_alt=4, missing=0.05)
    mk = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert mk.returncode == 0, mk.stdout
    args = ["--pfile", "mv"] + extra + ["--indep-pairwise", "40kb", "0.2"]
    if "3" in extra:
        got = run_cli(cli, args + ["--out", "hip"], str(tmp_path))
        assert got.returncode == 63 and "multiallelic" in got.stdout   # count filters with a triallelic site left: refused
        return
    compare(cli, tmp_path, args, [".prune.in", ".prune.out"])


def test_filters_that_leave_nothing(gpu_pkg, cli, tmp_path):
    """kPglRetDegenerateData (exit 13) with the reference's messages."""
    fileset(tmp_path, with_x=False)
    open(str(tmp_path / "unknown.txt"), "w").write("rs1 rs2\n")
    for args, msg in ((["--bfile", "d", "--extract", "unknown.txt"] + PRUNE, "No variants remaining after main filters"),
                      (["--bfile", "d", "--chr", "21", "--max-alleles", "2"] + PRUNE, "excluded by --chr + --max-alleles."),
                      (["--bfile", "d", "--chr", "21"] + PRUNE, "excluded by --chr.")):
        ref = T.run_ref(args + ["--out", "ref"], str(tmp_path))
        got = run_cli(cli, args + ["--out", "hip"], str(tmp_path))
        assert ref.returncode == got.returncode and ref.returncode in (7, 13), (ref.returncode, got.returncode, got.stdout[-400:])
        assert msg in ref.stdout.replace("\n", " ") and msg in got.stdout, (msg, ref.stdout[-300:], got.stdout[-300:])
    open(str(tmp_path / "nobody.txt"), "w").write("x y\n")
    ref = T.run_ref(["--bfile", "d", "--keep", "nobody.txt"] + PRUNE + ["--out", "ref"], str(tmp_path))
    got = run_cli(cli, ["--bfile", "d", "--keep", "nobody.txt"] + PRUNE + ["--out", "hip"], str(tmp_path))
    assert ref.returncode == got.returncode == 13 and "No samples remaining after main filters" in got.stdout


def test_make_founders_prune_matches_reference(gpu_pkg, cli, tmp_path):
    """Samples whose parents are not in the file become founders: the prune then runs over them too."""
    m, n = 400, 120
    raw = T.synth_raw_codes(m, n, seed=9, missing_rate=0.02)
    T.write_bed(str(tmp_path / "d"), raw, ["1"] * 200 + ["2"] * 200, np.concatenate([np.arange(200), np.arange(200)]) * 150 + 1)
    lines = []
    for s_ in range(n):
        pat, mat = ("0", "0") if s_ % 3 else (("s%d" % ((s_ + 1) % n), "s%d" % ((s_ + 2) % n)) if s_ % 2 else ("gone", "s%d" % ((s_ + 5) % n)))
        lines.append("s%d s%d %s %s 1 -9" % (s_, s_, pat, mat))
    open(str(tmp_path / "d.fam"), "w").write("\n".join(lines) + "\n")
    for mods in ([], ["require-2-missing"]):
        common = ["--bfile", "d", "--make-founders"] + mods + ["--indep-pairwise", "30kb", "0.2"]
        ref = T.run_ref(common + ["--out", "ref"], str(tmp_path))
        got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
        assert ref.returncode == 0 and got.returncode == 0, (ref.stdout[-300:], got.stdout[-300:])
        assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False), mods


@pytest.mark.parametrize("extra", [["--snps-only", "just-acgt"], ["--max-alleles", "1"], ["--snps-only"]])
def test_missing_code_alleles_in_a_bim(gpu_pkg, cli, tmp_path, extra):
    """A .bim whose A1 is '0' (monomorphic PLINK 1 data) or '.': '0' is the reference's default missing-genotype character, so
    `--snps-only just-acgt` keeps such variants (acgtm_table, plink2_pvar.cc:1631) and `--max-alleles` counts the lone ALT as
    no allele at all (allele_ct = 1, plink2_pvar.cc:1940-1948)."""
    assert T.have_ref()
    m, n = 500, 130
    raw = T.synth_raw_codes(m, n, seed=21, missing_rate=0.01, ld_copy_prob=0.6, redraw=0.05)
    raw[::7][raw[::7] == 2] = 0                         # (rows with A1 = 0 carry no A1 allele)
    raw[::7][raw[::7] == 1] = 0
    prefix = str(tmp_path / "d")
    T.write_bed(prefix, raw, ["1"] * 250 + ["2"] * 250, np.concatenate([np.arange(250), np.arange(250)]) * 200 + 1)
    lines = open(prefix + ".bim").read().splitlines()
    for v in range(0, m, 7):
        f = lines[v].split("\t")
        f[4] = "0" if (v % 14) else "."
        lines[v] = "\t".join(f)
    for v in range(3, m, 11):
        f = lines[v].split("\t")
        f[4] = "N"                                      # neither ACGT nor a missing code
        lines[v] = "\t".join(f)
    for v in range(5, m, 13):
        f = lines[v].split("\t")
        f[5] = "CT"                                     # not a SNP
        lines[v] = "\t".join(f)
    open(prefix + ".bim", "w").write("\n".join(lines) + "\n")
    # (--max-alleles 1 leaves the monomorphic rows only: every one of them is pruned and .prune.in is empty, as the reference's)
    compare(cli, tmp_path, ["--bfile", "d"] + extra + ["--indep-pairwise", "30kb", "0.2"], [".prune.in", ".prune.out"], may_be_empty=("1" in extra))
