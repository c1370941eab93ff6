"""The in-process seam exercised from the reference's own LdPrune(): oracle/_ref/plink2_hipld is the
reference binary with tests/integration/indep_pairwise_hip.cc linked in and the IndepPairwise() call inside LdPrune()
(plink2_ld.cc:2700) redirected to it at build time (oracle/Makefile, target ref_hip).  Its output files must be
byte-identical to the stock reference's."""
import os
import subprocess

import numpy as np
import pytest

import ldtools as T
from test_cli import small_fileset

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STOCK = os.path.join(REPO, "oracle", "_ref", "plink2")
PATCHED = os.path.join(REPO, "oracle", "_ref", "plink2_hipld")

CASES = [
    (["--bfile", "d", "--indep-pairwise", "50", "5", "0.2"], {}),
    (["--bfile", "d", "--indep-pairwise", "20kb", "0.5"], {}),
    (["--pfile", "d", "--indep-pairwise", "30", "1", "0.3", "--indep-order", "1"], {}),
    (["--bfile", "d", "--indep-pairwise", "15kb", "0.1"], {"nonfounders": 7}),
    (["--bfile", "d", "--indep-pairwise", "40", "3", "0.4"], {"chr0": 3}),
    (["--bfile", "d", "--indep-pairwise", "25kb", "0.2", "--indep-preferred", "pref.txt"], {}),
]


def run(binary, args, cwd, out):
    return subprocess.run([binary] + args + ["--bad-ld", "--out", out], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)


def both(tmp_path, args, kw, seed):
    prefix, raw, chr_idx, bps = small_fileset(tmp_path, m=400, n=90, seed=seed, **kw)
    if "--indep-preferred" in args:
        with open(str(tmp_path / "pref.txt"), "w") as f:
            f.write("".join("snp%d\n" % i for i in range(0, 400, 7)))
    a = run(STOCK, args, str(tmp_path), "stock")
    b = run(PATCHED, args, str(tmp_path), "hipld")
    assert a.returncode == 0, a.stdout[-1500:]
    assert b.returncode == 0, b.stdout[-1500:]
    for ext in (".prune.in", ".prune.out"):
        assert open(str(tmp_path / ("stock" + ext)), "rb").read() == open(str(tmp_path / ("hipld" + ext)), "rb").read(), ext
    return b.stdout


@pytest.mark.skipif(not (os.path.exists(STOCK) and os.path.exists(PATCHED)), reason="oracle/_ref binaries not built (make -C oracle ref ref_hip)")
def test_patched_reference_runs_its_own_path_without_a_gpu(pkg, tmp_path):
    """Plumbing only (no GPU here): with no HIP device IndepPairwiseHip() hands the job to the reference's IndepPairwise()."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    out = both(tmp_path, CASES[0][0], CASES[0][1], 3)
    assert "--indep-pairwise (HIP" not in out


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
def test_reference_ldprune_through_the_c_abi_matches_stock_reference(gpu_pkg, tmp_path, case):
    assert os.path.exists(STOCK) and os.path.exists(PATCHED), "oracle/_ref/plink2{,_hipld} missing: build() makes them where /root/reference exists"
    args, kw = CASES[case]
    out = both(tmp_path, args, kw, 11 + case)
    assert "--indep-pairwise (HIP" in out  # the engine really ran


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,wargs,unknown,all_male", [("pfile", ["40kb", "0.3"], True, False), ("bfile", ["70", "9", "0.2"], True, False),
                                                         ("pfile", ["100", "1", "0.4", "--indep-order", "1"], False, False), ("bfile", ["30kb", "0.3"], False, True)])
def test_sex_chromosomes_through_the_c_abi(gpu_pkg, tmp_path, fmt, wargs, unknown, all_male):
    """chrX / chrY / MT in the include set: the binding loads PgrGetInv1 rows of all samples and lets one sample map per
    chromosome class do the reference loader's founder / sex subsetting and SetHetMissing (plink2_ld.cc:1357-1388) on the
    device.  Male, female and unknown-sex samples, non-founders; also a file without a single non-male founder (chrX then
    takes the reference's chrY branch)."""
    from test_cli import sexed_fileset
    prefix = sexed_fileset(tmp_path, m=900, n=140, seed=5 + len(wargs), nonfounders=6, unknown_sex=unknown)
    if all_male:
        if fmt == "bfile":
            lines = [l.split() for l in open(prefix + ".fam").read().splitlines()]
            for t in lines:
                t[4] = "1"
            open(prefix + ".fam", "w").write("\n".join(" ".join(t) for t in lines) + "\n")
        else:
            pytest.skip("all-male variant runs on the .fam fileset")
    args = ["--" + fmt, "sx", "--indep-pairwise"] + wargs
    a = run(STOCK, args, str(tmp_path), "stock")
    b = run(PATCHED, args, str(tmp_path), "hipld")
    assert a.returncode == 0, a.stdout[-1500:]
    assert b.returncode == 0, b.stdout[-1500:]
    assert "--indep-pairwise (HIP" in b.stdout
    for ext in (".prune.in", ".prune.out"):
        assert open(str(tmp_path / ("stock" + ext)), "rb").read() == open(str(tmp_path / ("hipld" + ext)), "rb").read(), ext
    removed = open(str(tmp_path / "stock.prune.out")).read().split()
    assert len(removed) > 20


@pytest.mark.gpu
def test_environment_switch_sends_the_job_to_the_reference_path(gpu_pkg, tmp_path):
    m, n = 300, 80
    raw = T.synth_raw_codes(m, n, 5, missing_rate=0.02)
    T.write_bed(str(tmp_path / "d"), raw, ["1"] * m, (1000 + 200 * np.arange(m)).astype(np.uint32))
    env = dict(os.environ, PLINK2_HIP_LDPRUNE="0")
    c = subprocess.run([PATCHED, "--bfile", "d", "--indep-pairwise", "20kb", "0.3", "--out", "off"], cwd=str(tmp_path), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert c.returncode == 0 and "--indep-pairwise (HIP" not in c.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("wargs,nonfounders", [(["30kb", "0.5"], 0), (["60", "7", "0.2", "--indep-order", "1"], 5), (["100", "1", "0.8"], 5)])
def test_reference_pairphase_through_the_c_abi_matches_stock_reference(gpu_pkg, tmp_path, wargs, nonfounders):
    """IndepPairphase() redirected the same way (plink2_ld.cc:2698): PgrGetInv1P rows + phase bits as LDP_GENO_INVERSE |
    LDP_GENO_PHASED on a 2 x founders haplotype engine."""
    from test_pairphase import _phased_fileset
    tmp = str(tmp_path)
    _phased_fileset(tmp, 700, 131, seed=17 + nonfounders, chrom_plan=[("0", 3), ("1", 300), ("2", 250), ("7", 147)], nonfounders=nonfounders)
    args = ["--pfile", "p", "--indep-pairphase"] + wargs
    a = run(STOCK, args, tmp, "stock")
    b = run(PATCHED, args, tmp, "hipld")
    assert a.returncode == 0, a.stdout[-1500:]
    assert b.returncode == 0, b.stdout[-1500:]
    assert "--indep-pairphase (HIP" in b.stdout
    for ext in (".prune.in", ".prune.out"):
        assert open(os.path.join(tmp, "stock" + ext), "rb").read() == open(os.path.join(tmp, "hipld" + ext), "rb").read(), ext
    assert 0 < len(open(os.path.join(tmp, "stock.prune.out")).read().split()) < 697


@pytest.mark.gpu
def test_reference_pairphase_binding_reports_unphased_hets_like_the_reference(gpu_pkg, tmp_path):
    from test_pairphase import _phased_fileset
    tmp = str(tmp_path)
    _phased_fileset(tmp, 300, 90, seed=31, chrom_plan=[("1", 300)], unphased_rate=0.2)
    args = ["--pfile", "p", "--indep-pairphase", "50", "5", "0.5"]
    a = run(STOCK, args, tmp, "stock")
    b = run(PATCHED, args, tmp, "hipld")
    assert a.returncode == b.returncode == 7, (a.returncode, b.returncode, b.stdout[-800:])
    ea = [ln for ln in a.stdout.splitlines() if ln.startswith("Error")]
    eb = [ln for ln in b.stdout.splitlines() if ln.startswith("Error")]
    assert ea == eb and "is not fully phased" in eb[0]
