"""Command-line rules of the r^2 outputs that need no GPU: modifier combinations, column-set descriptors, the exit codes
the reference gives for the same mistakes (plink2.cc:11060-11210, ParseColDescriptor plink2_cmdline.cc:4375)."""
import subprocess

import numpy as np

import ldtools as T


def test_cli_ld_snp_flag_rules(tmp_path):
    import __graft_entry__ as ge
    cli = ge.load_package().build_cli()
    raw = T.synth_raw_codes(60, 30, seed=2)
    T.write_pgen_fixed(str(tmp_path / "d"), raw, ["1"] * 60, np.arange(60) * 10 + 1)
    def run(args):
        return subprocess.run([cli, "--pfile", "d"] + args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    r = run(["--r2-unphased", "square", "--ld-snp", "snp3"])
    assert r.returncode == 8 and "Matrix-only and table-only" in r.stdout
    r = run(["--r2-unphased", "--ld-snp", "snp3", "--ld-snps", "snp4"])
    assert r.returncode == 8 and "cannot be used with" in r.stdout
    r = run(["--indep-pairwise", "50", "5", "0.2", "--ld-snp", "snp3"])
    assert r.returncode == 8
    r = run(["--r2-unphased", "--ld-snps", "snp3-"])
    assert r.returncode == 8 and "Invalid --ld-snps" in r.stdout
    r = run(["--r2-unphased", "--ld-snp", "snp3", "--ld-window", "5"])
    assert r.returncode == 63


def test_cli_cols_flag_rules(tmp_path):
    import __graft_entry__ as ge
    cli = ge.load_package().build_cli()
    raw = T.synth_raw_codes(60, 30, seed=2)
    T.write_pgen_fixed(str(tmp_path / "d"), raw, ["1"] * 60, np.arange(60) * 10 + 1)
    def run(args):
        return subprocess.run([cli, "--pfile", "d"] + args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    r = run(["--r2-unphased", "cols=+nonsense"])
    assert r.returncode == 8 and "Unrecognized ID 'nonsense' in --r2-unphased column set descriptor." in r.stdout
    r = run(["--r2-unphased", "cols=+maj,freq"])
    assert r.returncode == 8 and "either all column set IDs must be" in r.stdout
    r = run(["--r2-unphased", "cols=+d"])
    assert r.returncode == 8 and "does not support computation of D or D'" in r.stdout
    r = run(["--r2-unphased", "square", "cols=+maj"])
    assert r.returncode == 8 and "Matrix-only and table-only" in r.stdout
    r = run(["--r2-unphased", "cols=+maj", "cols=+freq"])
    assert r.returncode == 8 and "Multiple --r2-unphased cols= modifiers." in r.stdout
    if T.have_ref():
        for args in (["--r2-unphased", "cols=+nonsense"], ["--r2-unphased", "cols=+maj,freq"], ["--r2-unphased", "cols=+d"], ["--r2-unphased", "dprime"]):
            ref = T.run_ref(["--pfile", "d"] + args + ["--out", "ref"], str(tmp_path))
            got = run(args)
            assert ref.returncode == got.returncode, (args, ref.returncode, got.returncode)


def test_cli_r_flags_and_windows(tmp_path):
    import __graft_entry__ as ge
    cli = ge.load_package().build_cli()
    raw = T.synth_raw_codes(60, 30, seed=4)
    T.write_bed(str(tmp_path / "d"), raw, ["1"] * 60, np.arange(60) * 10 + 1)

    def run(args):
        return subprocess.run([cli, "--bfile", "d"] + args, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)

    cases = [
        (["--r-unphased", "--r2-unphased"], "mutually"),
        (["--r-unphased", "cols=+dprime"], "does not support computation of D or D'"),
        (["--r-unphased", "square", "--ld-window-cm", "1"], "All-pairs --r-unphased settings"),
        (["--r2-unphased", "--ld-window-cm", "x"], "Invalid --ld-window-cm argument"),
        (["--r2-unphased", "inter-chr", "--ld-window-cm", "2"], "All-pairs --r2-unphased settings"),
        (["--indep-pairwise", "50", "5", "0.2", "--ld-window-cm", "1"], "must be used with"),
    ]
    for args, needle in cases:
        r = run(args)
        assert r.returncode == 8 and needle in r.stdout.replace("\n", " "), (args, r.returncode, r.stdout[-300:])
        if T.have_ref():
            ref = T.run_ref(["--bfile", "d"] + args + ["--out", "ref"], str(tmp_path))
            assert ref.returncode == 8, (args, ref.returncode, ref.stdout[-300:])
    # combinations this binary does not carry say so (63), they are not silently reinterpreted
    r = run(["--r2-unphased", "--ld-snp", "snp3", "--ld-window-cm", "1"])
    assert r.returncode == 63


def test_chrx_weighted_arithmetic_matches_reference(tmp_path):
    """plink2-hip's chrX arithmetic (XWeighted: two integer 6-tuples per pair -> ComputeXR2's doubles, plink2_ld.cc:7122-7190)
    against the reference binary's own `square bin` doubles, without a GPU: the tuples come from the oracle (all founders,
    male founders), the arithmetic from the CLI's --debug-xweighted hook.  'ref-based' on both sides: one orientation, no
    frequency pass to agree on.  Pairs inside chrX (weight 1/2) and chrX x autosome (weight 1 - sqrt(2)/2), r^2 and signed r."""
    import pytest
    if not T.have_ref():
        pytest.skip("oracle/_ref/plink2 not built")
    import __graft_entry__ as ge
    cli = ge.load_package().build_cli()
    m, n = 90, 70
    raw = T.synth_raw_codes(m, n, seed=13, missing_rate=0.05, ld_copy_prob=0.7)
    rng = np.random.default_rng(4)
    sexes = rng.choice([1, 2, 0], size=n, p=[0.5, 0.4, 0.1])
    chroms = ["3"] * 40 + ["X"] * 50
    bps = np.concatenate([np.arange(40), np.arange(50)]) * 100 + 1
    T.write_pgen_fixed(str(tmp_path / "d"), raw, chroms, bps, sexes=sexes)
    male = np.flatnonzero(sexes == 1)
    packed_all, packed_male = T.pack_2bit(raw), T.pack_2bit(np.ascontiguousarray(raw[:, male]))
    hom_a, r2h_a, va = T.oracle_split(packed_all, n)
    hom_m, r2h_m, vm = T.oracle_split(packed_male, len(male))
    pairs = [(i, j) for j in range(40, m) for i in range(j) if (i + 3 * j) % 5 == 0]
    for flag, ext, unsq in (("--r2-unphased", ".unphased.vcor2.bin", 0), ("--r-unphased", ".unphased.vcor1.bin", 1)):
        ref = T.run_ref(["--pfile", "d", flag, "square", "bin", "ref-based", "--out", "ref"], str(tmp_path))
        assert ref.returncode == 0, ref.stdout
        want = np.fromfile(str(tmp_path / ("ref" + ext)), dtype=np.float64).reshape(m, m)
        with open(str(tmp_path / "tuples.txt"), "w") as f:
            for i, j in pairs:
                a = T.oracle_pair_stats(hom_a, r2h_a, va, n, i, j).astuple()
                mm = T.oracle_pair_stats(hom_m, r2h_m, vm, len(male), i, j).astuple()
                f.write("%d %d 0 0 0 0 %s %s\n" % (1 if i >= 40 else 0, unsq, " ".join(map(str, a)), " ".join(map(str, mm))))
        out = subprocess.run([cli, "--debug-xweighted", str(tmp_path / "tuples.txt")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert out.returncode == 0, out.stdout
        got = np.array([int(x, 16) for x in out.stdout.split()], dtype=np.uint64).view(np.float64)
        exp = np.array([want[j, i] for i, j in pairs])
        assert len(got) == len(exp)
        same = (got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))
        assert same.all(), [(pairs[k], got[k], exp[k]) for k in np.flatnonzero(~same)[:5]]
        assert (np.abs(exp[~np.isnan(exp)]) > 0.05).any() and (unsq == 0 or (exp[~np.isnan(exp)] < 0).any())
