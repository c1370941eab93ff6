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
