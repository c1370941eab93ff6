"""plink2-hip, the process-level drop-in: flag parsing and guards on CPU (no GPU needed: --dry-run stops
after planning), byte-identical .prune.in/.prune.out against the reference binary on the GPU box."""
import filecmp
import os
import re
import subprocess

import numpy as np
import pytest

import ldtools as T
from test_host_logic import make_positions


@pytest.fixture(scope="module")
def cli(pkg):
    path = pkg.build_cli()
    assert path and os.path.exists(path)
    return path


def run_cli(cli, args, cwd):
    return subprocess.run([cli] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)


def small_fileset(tmp_path, m=120, n=60, seed=1, nonfounders=0, chr0=0):
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.03)
    chr_idx, bps = make_positions(m, 3, seed + 5)
    chroms = [str(c + 1) for c in chr_idx]
    for k in range(chr0):
        chroms[k] = "0"
    prefix = str(tmp_path / "d")
    T.write_bed(prefix, raw, chroms, bps)
    T.write_pgen_fixed(prefix, raw, chroms, bps)
    if nonfounders:
        lines = open(prefix + ".fam").read().splitlines()
        for s in range(nonfounders):
            t = lines[3 * s + 1].split()
            t[2], t[3] = "s0", "s2"
            lines[3 * s + 1] = " ".join(t)
        open(prefix + ".fam", "w").write("\n".join(lines) + "\n")
        with open(prefix + ".psam", "w") as f:
            f.write("#IID\tPAT\tMAT\tSEX\n")
            for s in range(n):
                nf = (s % 3 == 1) and (s // 3 < nonfounders)
                f.write("s%d\t%s\t%s\t2\n" % (s, "s0" if nf else "0", "s2" if nf else "0"))
    return prefix, raw, chr_idx, bps


def test_dry_run_parses_like_plink(cli, tmp_path):
    prefix, raw, chr_idx, bps = small_fileset(tmp_path)
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "200kb", "0.3", "--dry-run", "--out", "o"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("dry-run:")][0]
    # 200kb -> 200000 bp (plink2.cc:7266); "0.3" -> 3 * 0.1 = 0.30000000000000004 (ScanadvDouble), one ulp above strtod
    assert "window=200000 step=1 window_is_bp=1" in line
    assert ("r2=%s" % float.hex(3 * 0.1)) in line and float.hex(3 * 0.1) != float.hex(0.3)
    cp = run_cli(cli, ["--pfile", "d", "--indep-pairwise", "50", "5", "0.2", "--indep-order", "1", "--dry-run", "--out", "o"], str(tmp_path))
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("dry-run:")][0]
    assert "window=50 step=5 window_is_bp=0" in line and "order=1" in line and ("r2=%s" % float.hex(0.2)) in line
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "0.5", "kb", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
    assert "window=500 step=1 window_is_bp=1" in cp.stdout


@pytest.mark.parametrize("args,needle", [
    (["--indep-pairwise", "50", "60", "0.2"], "window-increment cannot be larger than window size"),
    (["--indep-pairwise", "50kb", "2", "0.2"], "window-increment must be 1"),
    (["--indep-pairwise", "50", "5", "1.0"], "Invalid --indep-pairwise r^2 threshold"),
    (["--indep-pairwise", "50"], "accepts 2-4 arguments"),
    (["--indep-pairwise", "50", "5", "0.2", "--indep-order", "3"], "Invalid --indep-order mode"),
    (["--glm"], "Unrecognized flag"),
])
def test_argument_errors(cli, tmp_path, args, needle):
    small_fileset(tmp_path)
    cp = run_cli(cli, ["--bfile", "d"] + args + ["--dry-run", "--out", "o"], str(tmp_path))
    assert cp.returncode != 0 and needle in cp.stdout, cp.stdout


def test_founder_guard_and_format_errors(cli, tmp_path):
    small_fileset(tmp_path, n=30)
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "50", "5", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
    assert cp.returncode == 13 and "less than 50 samples" in cp.stdout   # kPglRetDegenerateData (plink2.cc:2065-2071), as the reference exits:
    if T.have_ref():
        ref = T.run_ref(["--bfile", "d", "--indep-pairwise", "50", "5", "0.2", "--out", "r"], str(tmp_path))
        assert ref.returncode == cp.returncode == 13 and "less than 50" in ref.stdout
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run", "--out", "o"], str(tmp_path))
    assert cp.returncode == 0
    # wrong .bed size
    with open(str(tmp_path / "d.bed"), "ab") as f:
        f.write(b"\0")
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run", "--out", "o"], str(tmp_path))
    assert cp.returncode != 0 and "Unexpected" in cp.stdout
    # an external-index .pgen without its index, and the extension modes: refused, not mis-read
    for mode, needle in ((0x20, "d.pgen.pgi"), (0x21, "d.pgen.pgi"), (0x12, "storage mode 0x12 is not supported")):
        with open(str(tmp_path / "d.pgen"), "r+b") as f:
            f.seek(2)
            f.write(bytes([mode]))
        cp = run_cli(cli, ["--pfile", "d", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run", "--out", "o"], str(tmp_path))
        assert cp.returncode != 0 and needle in cp.stdout, cp.stdout


def test_cli_refuses_to_compute_without_gpu(cli, pkg, tmp_path):
    if pkg.device_count() > 0:
        pytest.skip("GPU present")
    small_fileset(tmp_path)
    cp = run_cli(cli, ["--bfile", "d", "--indep-pairwise", "50", "5", "0.2", "--out", "o"], str(tmp_path))
    assert cp.returncode == 16 and "no usable HIP device" in cp.stdout
    assert not os.path.exists(str(tmp_path / "o.prune.in"))


CLI_CASES = [
    # fmt, window args, r2, order, nonfounders, chr0, preferred
    ("bfile", ["50", "5"], "0.2", 2, 0, 0, False),
    ("pfile", ["50", "5"], "0.2", 1, 0, 0, False),
    ("bfile", ["20kb"], "0.5", 2, 0, 3, False),
    ("pfile", ["20kb"], "0.3", 2, 7, 0, False),
    ("bfile", ["30", "kb"], "0.1", 1, 5, 2, False),
    ("pfile", ["100", "10"], "0.4", 2, 0, 0, True),
    ("vpfile", ["20kb"], "0.2", 2, 0, 0, False),     # standard variable-width .pgen written by the reference
    ("vpfile", ["60", "3"], "0.5", 1, 9, 4, False),
    ("xpfile", ["25kb"], "0.3", 2, 4, 2, False),      # the same with the index split off into a .pgen.pgi (mode 0x20), named by --pgi
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CLI_CASES)
def test_cli_byte_identical_to_reference(gpu_pkg, cli, tmp_path, case):
    fmt, wargs, r2, order, nonfounders, chr0, preferred = case
    assert T.have_ref(), "reference binary oracle/_ref/plink2 must travel with the repo snapshot"
    prefix, raw, chr_idx, bps = small_fileset(tmp_path, m=900, n=120, seed=len(wargs) + order + nonfounders, nonfounders=nonfounders, chr0=chr0)
    if fmt == "vpfile":
        # let the reference re-encode the fileset into its default variable-width .pgen (mode 0x10)
        mk = T.run_ref(["--pfile", "d", "--make-pgen", "--out", "v"], str(tmp_path))
        assert mk.returncode == 0, mk.stdout
        assert open(str(tmp_path / "v.pgen"), "rb").read(3)[2] == 0x10
        common = ["--pfile", "v", "--indep-pairwise"] + wargs + [r2]
    elif fmt == "xpfile":
        mk = T.run_ref(["--pfile", "d", "--make-pgen", "--out", "v"], str(tmp_path))
        assert mk.returncode == 0, mk.stdout
        T.split_pgen_index(str(tmp_path / "v.pgen"), str(tmp_path / "x.pgen"), str(tmp_path / "x.index"))
        common = ["--pgen", "x.pgen", "--pgi", "x.index", "--pvar", "v.pvar", "--psam", "v.psam", "--indep-pairwise"] + wargs + [r2]
    else:
        common = ["--" + fmt, "d", "--indep-pairwise"] + wargs + [r2]
    if order == 1:
        common += ["--indep-order", "1"]
    if preferred:
        with open(str(tmp_path / "pref.txt"), "w") as f:
            f.write("\n".join("snp%d" % i for i in range(0, 900, 7)) + "\n")
        common += ["--indep-preferred", "pref.txt"]
    ref = T.run_ref(common + ["--threads", "4", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)
    want = re.findall(r"\d+/\d+ variants removed\.", ref.stdout)[-1]
    assert want in got.stdout


@pytest.mark.gpu
def test_cli_config1_toy(gpu_pkg, cli, tmp_path):
    """BASELINE config 1 (1.9/toy.ped -> .bed; 2 samples x 2 variants): rs0 is monomorphic and goes to .prune.out."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy_50_5_0.2.npz"))
    m, n = int(z["m"]), int(z["n"])
    raw = T.unpack_2bit(z["raw_packed"].reshape(m, -1).view(np.uint64), n)
    T.write_bed(str(tmp_path / "toy"), raw, [str(c) for c in z["chroms"]], z["bps"], ids=["rs0", "rs10"])
    cp = run_cli(cli, ["--bfile", "toy", "--indep-pairwise", "50", "5", "0.2", "--out", "o"], str(tmp_path))
    assert cp.returncode != 0 and "less than 50 samples" in cp.stdout
    cp = run_cli(cli, ["--bfile", "toy", "--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--out", "o"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    assert "1/2 variants removed." in cp.stdout
    assert open(str(tmp_path / "o.prune.in")).read() == "rs10\n"
    assert open(str(tmp_path / "o.prune.out")).read() == "rs0\n"


@pytest.mark.gpu
@pytest.mark.parametrize("max_alt,order,wargs", [(2, 2, ["30kb"]), (5, 2, ["80", "7"]), (18, 1, ["30kb"])])
def test_cli_multiallelic_collapse_matches_reference(gpu_pkg, cli, tmp_path, max_alt, order, wargs):
    """BASELINE config 5's mask path: missing calls + multiallelic sites (collapsed major-vs-rest like
    PgrGetInv1 / Get1Multiallelic), imported by the reference from a VCF into a variable-width .pgen."""
    from test_pgen_reader import make_multiallelic_vcf
    assert T.have_ref()
    m, n = 700, 160
    make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed=max_alt, max_alt=max_alt, missing=0.05)
    mk = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert mk.returncode == 0, mk.stdout
    common = ["--pfile", "mv", "--indep-pairwise"] + wargs + ["0.1"]
    if order == 1:
        common += ["--indep-order", "1"]
    ref = T.run_ref(common + ["--threads", "2", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)
    assert 0 < len(open(str(tmp_path / "hip.prune.out")).read().split()) < m


@pytest.mark.gpu
@pytest.mark.parametrize("max_alt,wargs,nonfounder_every", [(3, ["30kb"], 3), (6, ["60", "5"], 2), (2, ["20kb"], 7)])
def test_cli_multiallelic_collapse_over_the_founders_matches_reference(gpu_pkg, cli, tmp_path, max_alt, wargs, nonfounder_every):
    """The same with non-founders in the file: the engines pick the founder columns through a subset sample map and the collapse's
    allele counts run over the founders -- on the device (pgen_aux1_kernel with the map's mask; `--timing` says how many variants took
    which way), byte-identical lists to the reference's."""
    from test_pgen_reader import make_multiallelic_vcf
    assert T.have_ref()
    m, n = 600, 170
    make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed=10 + max_alt, max_alt=max_alt, missing=0.04)
    mk = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert mk.returncode == 0, mk.stdout
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for s_ in range(n):
        nf = (s_ % nonfounder_every == 1) and s_ > 3
        psam.append("s%d\t%s\t%s\t%d" % (s_, "s0" if nf else "0", "s2" if nf else "0", 1 + (s_ % 2)))
    open(str(tmp_path / "mv.psam"), "w").write("\n".join(psam) + "\n")
    common = ["--pfile", "mv", "--indep-pairwise"] + wargs + ["0.1"]
    ref = T.run_ref(common + ["--threads", "2", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--timing", "--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)
    assert 0 < len(open(str(tmp_path / "hip.prune.out")).read().split()) < m
    import re
    line = re.search(r"host-built rows: (\d+) multiallelic \((\d+) more have REF as the major allele: main track as loaded; (\d+) collapsed on the device\)", got.stdout)
    assert line and int(line.group(1)) == 0 and int(line.group(3)) > 50, got.stdout[-600:]


def sexed_fileset(tmp_path, m=900, n=140, seed=5, nonfounders=6, unknown_sex=True):
    """Autosomes + chrX + chrY + MT, males/females/unknown sex, a few non-founders."""
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.04)
    per = m // 5
    chroms = (["1"] * per + ["2"] * per + ["X"] * per + ["Y"] * per + ["MT"] * (m - 4 * per))
    bps = np.concatenate([1000 + 211 * np.arange(per)] * 4 + [1000 + 211 * np.arange(m - 4 * per)]).astype(np.uint32)
    rng = np.random.default_rng(seed)
    sexes = rng.choice([1, 2, 0] if unknown_sex else [1, 2], size=n, p=[0.45, 0.45, 0.1] if unknown_sex else [0.5, 0.5])
    prefix = str(tmp_path / "sx")
    T.write_pgen_fixed(prefix, raw, chroms, bps, sexes=sexes)
    T.write_bed(prefix, raw, chroms, bps)
    fam = []
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for s in range(n):
        nf = (s % 11 == 3) and (s // 11 < nonfounders)
        fam.append("s%d s%d %s %s %d -9" % (s, s, "s0" if nf else "0", "s1" if nf else "0", sexes[s]))
        psam.append("s%d\t%s\t%s\t%s" % (s, "s0" if nf else "0", "s1" if nf else "0", "NA" if sexes[s] == 0 else str(sexes[s])))
    open(prefix + ".fam", "w").write("\n".join(fam) + "\n")
    open(prefix + ".psam", "w").write("\n".join(psam) + "\n")
    return prefix


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,wargs,order,unknown", [("pfile", ["40kb"], 2, True), ("bfile", ["70", "9"], 1, True), ("pfile", ["100", "1"], 2, False)])
def test_cli_sex_chromosomes_match_reference(gpu_pkg, cli, tmp_path, fmt, wargs, order, unknown):
    """chrX (males het->missing, non-males weighted 2x), chrY (non-females, haploid) and MT (haploid):
    plink2_ld.cc:890-901,1066-1082,1356-1388 and the reference's haploid allele-frequency rules."""
    assert T.have_ref()
    sexed_fileset(tmp_path, unknown_sex=unknown)
    common = ["--" + fmt, "sx", "--indep-pairwise"] + wargs + ["0.2"]
    if order == 1:
        common += ["--indep-order", "1"]
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    ref_out = open(str(tmp_path / "ref.prune.out")).read().split()
    hip_out = open(str(tmp_path / "hip.prune.out")).read().split()
    diff = sorted(set(ref_out) ^ set(hip_out), key=lambda x: int(x[3:]))
    assert not diff, "differs on %d variants, e.g. %s" % (len(diff), diff[:10])
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("max_alt,wargs,order,unknown", [(3, ["40kb"], 2, True), (6, ["80", "7"], 1, False)])
def test_cli_multiallelic_variants_on_sex_chromosomes_match_reference(gpu_pkg, cli, tmp_path, max_alt, wargs, order, unknown):
    """Variants with several ALT alleles on chrX / chrY / MT under --indep-pairwise (round 5; refused before): the major allele from the
    chromosome's own allele-frequency rule (plink2_data.cc:2752-2895: chrX non-males' copies count double, chrY / MT diploid-style over
    their founders), PgrGetInv1's collapse on it, then the chromosome's sample layout -- byte-identical lists to the reference's, with
    non-founders and unknown-sex samples in the file."""
    from test_pgen_reader import make_multiallelic_vcf
    assert T.have_ref()
    m, n = 800, 150
    make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed=20 + max_alt, max_alt=max_alt, missing=0.04)
    mk = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert mk.returncode == 0, mk.stdout
    # the same records, re-labelled: a quarter each on chromosome 1, X, Y and MT
    out = []
    k = 0
    for ln in open(str(tmp_path / "mv.pvar")):
        if ln.startswith("#"):
            out.append(ln)
            continue
        f = ln.rstrip("\n").split("\t")
        f[0] = ["1", "X", "Y", "MT"][min(3, (4 * k) // m)]
        out.append("\t".join(f) + "\n")
        k += 1
    open(str(tmp_path / "mv.pvar"), "w").write("".join(out))
    rng = np.random.default_rng(max_alt)
    sexes = rng.choice([1, 2, 0] if unknown else [1, 2], size=n, p=[0.45, 0.45, 0.1] if unknown else [0.5, 0.5])
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for s_ in range(n):
        nf = (s_ % 9 == 4) and s_ > 9
        psam.append("s%d\t%s\t%s\t%s" % (s_, "s0" if nf else "0", "s1" if nf else "0", "NA" if sexes[s_] == 0 else str(sexes[s_])))
    open(str(tmp_path / "mv.psam"), "w").write("\n".join(psam) + "\n")
    common = ["--pfile", "mv", "--indep-pairwise"] + wargs + ["0.1"]
    if order == 1:
        common += ["--indep-order", "1"]
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    ref_out = open(str(tmp_path / "ref.prune.out")).read().split()
    hip_out = open(str(tmp_path / "hip.prune.out")).read().split()
    diff = sorted(set(ref_out) ^ set(hip_out), key=lambda x: int(x[3:]))
    assert not diff, "differs on %d variants, e.g. %s" % (len(diff), diff[:10])
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)
    assert 0 < len(hip_out) < m
    # the r^2 outputs take them too (round 5; exit 63 before): without a column that names the allele both tools ask for one, with
    # 'allow-ambiguous-allele' the windowed table over all four chromosomes is the reference's, byte for byte
    r = run_cli(cli, ["--pfile", "mv", "--r2-unphased", "--out", "no"], str(tmp_path))
    rr = T.run_ref(["--pfile", "mv", "--r2-unphased", "--out", "refno"], str(tmp_path))
    assert r.returncode == rr.returncode == 7 and "allow-ambiguous-allele" in r.stdout
    tab = ["--pfile", "mv", "--r2-unphased", "allow-ambiguous-allele", "--ld-window-kb", "2", "--ld-window-r2", "0.05"]
    rr = T.run_ref(tab + ["--threads", "3", "--out", "reft"], str(tmp_path))
    r = run_cli(cli, tab + ["--out", "hipt"], str(tmp_path))
    assert rr.returncode == 0 and r.returncode == 0, (rr.stdout[-300:], r.stdout[-300:])
    a, b = open(str(tmp_path / "reft.vcor"), "rb").read(), open(str(tmp_path / "hipt.vcor"), "rb").read()
    assert len(a) > 2000 and a == b


def test_vcor_number_formatting_matches_reference(cli, tmp_path):
    """The .vcor writer's 6-significant-digit formatter against 26k (double, text) pairs recorded from the
    reference's own table output (tests/golden/make_golden_vcor.py).  No GPU involved."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pgen", "vcor_format_g6.npz"))
    path = tmp_path / "bits.txt"
    with open(path, "w") as f:
        for b in g["bits"]:
            f.write("%016x\n" % int(b))
    out = subprocess.run([cli, "--debug-format-g6", str(path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    got = out.stdout.split("\n")[:-1]
    want = [str(t) for t in g["texts"]]
    assert len(got) == len(want)
    bad = [(w, x) for w, x in zip(want, got) if w != x]
    assert not bad, bad[:5]


def test_vcor_flag_rules(cli, tmp_path):
    """--ld-window* belong to --r2-unphased tables only (plink2.cc:11175-11191, :12960)."""
    small_fileset(tmp_path)
    r = run_cli(cli, ["--pfile", "d", "--indep-pairwise", "50", "5", "0.2", "--ld-window-kb", "10", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "--ld-window" in r.stdout
    r = run_cli(cli, ["--pfile", "d", "--r2-unphased", "square", "bin", "--ld-window", "5", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "All-pairs" in r.stdout
    r = run_cli(cli, ["--pfile", "d", "--r2-unphased", "triangle", "bin4", "--ld-window-r2", "0.1", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "Matrix-only" in r.stdout
    r = run_cli(cli, ["--pfile", "d", "--r2-unphased", "--ld-window", "1", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "Invalid --ld-window argument" in r.stdout
    # inter-chr is an all-pairs mode: the r^2 filter applies, the window flags do not (plink2.cc:11175-11179)
    r = run_cli(cli, ["--pfile", "d", "--r2-unphased", "inter-chr", "--ld-window-kb", "10", "--dry-run"], str(tmp_path))
    assert r.returncode != 0 and "All-pairs" in r.stdout
    r = run_cli(cli, ["--pfile", "d", "--r2-unphased", "inter-chr", "square", "--dry-run"], str(tmp_path))
    assert r.returncode != 0


def test_zstd_variant_table(cli, tmp_path):
    """<prefix>.pvar.zst / 'vzs' (written by the reference's --make-pgen vzs) parses to the same plan as the plain
    .pvar (libzstd.so.1 is bound by hand: the image has no zstd headers)."""
    if not T.have_ref():
        pytest.skip("reference binary not built")
    small_fileset(tmp_path, m=150, n=60, seed=3)
    r = T.run_ref(["--pfile", "d", "--make-pgen", "vzs", "--out", "z"], str(tmp_path))
    assert r.returncode == 0 and os.path.exists(str(tmp_path / "z.pvar.zst")), r.stdout
    a = run_cli(cli, ["--pfile", "d", "--indep-pairwise", "50", "5", "0.2", "--dry-run"], str(tmp_path))
    b = run_cli(cli, ["--pfile", "z", "vzs", "--indep-pairwise", "50", "5", "0.2", "--dry-run"], str(tmp_path))
    assert a.returncode == 0 and b.returncode == 0, a.stdout + b.stdout
    la = [ln for ln in a.stdout.splitlines() if ln.startswith("dry-run:")]
    lb = [ln for ln in b.stdout.splitlines() if ln.startswith("dry-run:")]
    assert la and la == lb
    # truncated stream is an error, not a short table
    data = open(str(tmp_path / "z.pvar.zst"), "rb").read()
    open(str(tmp_path / "t.pvar.zst"), "wb").write(data[:len(data) // 2])
    for ext in (".pgen", ".psam"):
        os.link(str(tmp_path / ("z" + ext)), str(tmp_path / ("t" + ext)))
    c = run_cli(cli, ["--pfile", "t", "vzs", "--indep-pairwise", "50", "5", "0.2", "--dry-run"], str(tmp_path))
    assert c.returncode != 0 and "zstd" in c.stdout


def test_zstd_output_writer_roundtrip(cli, tmp_path):
    """'zs' outputs go through a hand-bound libzstd stream; the reference's own --zst-decompress must give the text back."""
    if not T.have_ref():
        pytest.skip("reference binary not built")
    rng = np.random.default_rng(1)
    text = "\n".join("%d\t%s" % (k, "ACGT"[k % 4] * int(rng.integers(1, 40))) for k in range(60000)) + "\n"
    open(str(tmp_path / "in.txt"), "w").write(text)
    r = run_cli(cli, ["--debug-zstd", "in.txt", "out.zst"], str(tmp_path))
    assert r.returncode == 0, r.stdout
    assert os.path.getsize(str(tmp_path / "out.zst")) < len(text) // 2
    back = subprocess.run([T.REF_BIN, "--zst-decompress", "out.zst"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120)
    assert back.returncode == 0 and back.stdout.decode() == text


def test_vcor_values_oracle_plus_formatter_reproduce_reference_text(cli, tmp_path):
    """Every r^2 the reference printed in a windowed .vcor table = the oracle's exact double (ComputeR2 arithmetic)
    through plink2-hip's number formatter.  CPU only: pins value + text without the GPU in the loop."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pgen", "vcor_windows.npz"))
    raw = z["raw"]
    n = raw.shape[1]
    inv, mf, _ = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    bits, want = [], []
    for k in range(len(z["settings"])):
        for (i, j), text in zip(z["pairs_%d" % k], z["r2_text_%d" % k]):
            cov, v1, v2 = T.oracle_r2(T.oracle_pair_stats(hom, r2h, vaggs, n, int(i), int(j)))
            bits.append(np.float64(cov * cov / (v1 * v2)).view(np.uint64))
            want.append(str(text))
    path = tmp_path / "bits.txt"
    open(path, "w").write("".join("%016x\n" % int(b) for b in bits))
    out = subprocess.run([cli, "--debug-format-g6", str(path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    got = out.stdout.split("\n")[:-1]
    assert len(got) == len(want) > 30000
    bad = [(w, g) for w, g in zip(want, got) if w != g]
    assert not bad, bad[:5]


def test_inter_chr_filter_and_order_from_oracle_values(cli, tmp_path):
    """--r2-unphased inter-chr with the default filter: the pairs the reference wrote, in its order, are exactly the
    pairs A < B (any chromosomes) whose oracle r^2 is >= 0.2 * (1 - 2^-44), A-major; texts through the formatter."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "pgen", "vcor_windows.npz"))
    raw = z["raw"]
    m, n = raw.shape
    inv, mf, _ = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    thr = 0.2 * (1 - T.K_SMALL_EPSILON)
    pairs, bits = [], []
    for i in range(m):
        for j in range(i + 1, m):
            st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
            cov, v1, v2 = T.oracle_r2(st)
            if st.nm == 0 or v1 * v2 == 0.0:
                continue
            r2 = cov * cov / (v1 * v2)
            if r2 >= thr:
                pairs.append((i, j))
                bits.append(np.float64(r2).view(np.uint64))
    assert pairs == [(int(a), int(b)) for a, b in z["inter_pairs"]]
    path = tmp_path / "bits.txt"
    open(path, "w").write("".join("%016x\n" % int(b) for b in bits))
    out = subprocess.run([cli, "--debug-format-g6", str(path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.stdout.split("\n")[:-1] == [str(t) for t in z["inter_text"]]


@pytest.mark.gpu
@pytest.mark.parametrize("engines,fmt,order", [(2, "bfile", 2), (3, "pfile", 2), (8, "bfile", 1), (8, "vpfile", 2)])
def test_cli_n_engines_match_reference(gpu_pkg, cli, tmp_path, engines, fmt, order):
    """`plink2-hip --gpus N`: the subcontigs LPT-sharded over N engines (ldp_set_shard), every engine loading from the same file
    calls, one host thread per engine, and the shards' removed bits packed, exchanged and stitched (plink2_ld.cc:2686-2694 shard,
    :1418-1426 stitch).  With N devices the exchange is one RCCL all-gather; on a box with fewer, --debug-alias-devices deals
    the engines onto the devices there are and the host carries the segments (RCCL refuses a device twice): every other step is
    the N-device run's.  Byte-identical to the reference and to one engine; N = 8 leaves some engines without a subcontig."""
    assert T.have_ref(), "reference binary oracle/_ref/plink2 must travel with the repo snapshot"
    prefix, raw, chr_idx, bps = small_fileset(tmp_path, m=1500, n=200, seed=21 + engines, nonfounders=(3 if fmt == "pfile" else 0), chr0=4)
    src = ["--" + fmt, "d"]
    if fmt == "vpfile":
        mk = T.run_ref(["--pfile", "d", "--make-pgen", "--out", "v"], str(tmp_path))
        assert mk.returncode == 0, mk.stdout
        src = ["--pfile", "v"]
    common = src + ["--indep-pairwise", "30kb", "0.3"] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "4", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    one = run_cli(cli, common + ["--out", "one"], str(tmp_path))
    assert one.returncode == 0, one.stdout
    alias = ["--debug-alias-devices"] if gpu_pkg.device_count() < engines else []
    many = subprocess.run([cli] + common + alias + ["--gpus", str(engines), "--timing", "--out", "many"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=600)
    assert many.returncode == 0, many.stdout
    assert "(%d GPUs)" % engines in many.stdout and ("%d engines on" % engines) in many.stdout, many.stdout
    # every engine is fed by a thread of its own, all at once (round 6; the reference's main thread fills every worker's slot of a batch,
    # plink2_ld.cc:1292-1417): the --timing lines name each engine's load interval and the overlap
    assert ("%d engines fed concurrently" % engines) in many.stdout, many.stdout
    assert len(re.findall(r"\[timing\] engine \d+ \(device \d+", many.stdout)) == engines, many.stdout
    for ext in (".prune.in", ".prune.out"):
        assert filecmp.cmp(str(tmp_path / ("ref" + ext)), str(tmp_path / ("many" + ext)), shallow=False)
        assert filecmp.cmp(str(tmp_path / ("one" + ext)), str(tmp_path / ("many" + ext)), shallow=False)
    if engines == 3:
        # --debug-serial-feed: one thread feeds the engines in turn, as rounds 2-5 did -- the same files
        ser = subprocess.run([cli] + common + alias + ["--gpus", str(engines), "--timing", "--debug-serial-feed", "--out", "ser"], cwd=str(tmp_path), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=600)
        assert ser.returncode == 0 and "fed concurrently" not in ser.stdout, ser.stdout
        for ext in (".prune.in", ".prune.out"):
            assert filecmp.cmp(str(tmp_path / ("ser" + ext)), str(tmp_path / ("many" + ext)), shallow=False)


def dummy_dosage_fileset(tmp_path, name, n, m, freq, seed):
    """a fileset with dosage tracks written by the reference itself"""
    cp = T.run_ref(["--dummy", str(n), str(m), "dosage-freq=%g" % freq, "--seed", str(seed), "--threads", "2", "--make-pgen", "--out", name], str(tmp_path))
    assert cp.returncode == 0, cp.stdout


def test_cli_dosage_pgen_rules(cli, pkg, tmp_path):
    """A .pgen with dosage tracks: the reference derives allele frequencies (major allele, prune tie-break) from the dosages
    (plink2_data.cc:2421-2443).  --indep-pairwise reproduces that (ldp_pgen_dosage_sums); what would need dosages elsewhere -- the
    r^2 of --r2-unphased / --clump, phased dosages, frequency filters -- is refused instead of being computed from hardcalls."""
    if not T.have_ref():
        pytest.skip("oracle/_ref/plink2 not built")
    dummy_dosage_fileset(tmp_path, "dos", 60, 200, 0.3, 3)
    for args in (["--r2-unphased", "--ld-window-r2", "0.2"], ["--indep-pairphase", "50", "5", "0.2"]):
        out = run_cli(cli, ["--pfile", "dos"] + args + ["--out", "o"], str(tmp_path))
        assert out.returncode == 63 and "dosage" in out.stdout, out.stdout
    assert not os.path.exists(str(tmp_path / "o.prune.in"))
    out = run_cli(cli, ["--pfile", "dos", "--indep-pairwise", "50", "5", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
    assert out.returncode == 0, out.stdout
    # --maf / --max-maf compare the dosage-based frequencies, as the reference's do: same number of variants filtered out, and
    # another number than on the hardcalls of the same file
    flt = ["--maf", "0.3", "--max-maf", "0.47", "--indep-pairwise", "50", "5", "0.2"]
    ref = T.run_ref(["--pfile", "dos"] + flt + ["--out", "rf"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    line = re.findall(r"\d+ variants? removed due to allele frequency threshold", ref.stdout)[-1]
    out = run_cli(cli, ["--pfile", "dos"] + flt + ["--dry-run", "--out", "o"], str(tmp_path))
    assert out.returncode == 0 and line in out.stdout, (line, out.stdout)
    hard = T.run_ref(["--pfile", "dos", "--make-pgen", "erase-dosage", "--out", "hard"], str(tmp_path))
    assert hard.returncode == 0, hard.stdout
    out = run_cli(cli, ["--pfile", "hard"] + flt + ["--dry-run", "--out", "o"], str(tmp_path))
    assert out.returncode == 0 and line not in out.stdout
    if pkg.device_count() == 0:   # the prune itself gets as far as the device
        out = run_cli(cli, ["--pfile", "dos", "--indep-pairwise", "50", "5", "0.2", "--out", "o"], str(tmp_path))
        assert out.returncode == 16 and "no usable HIP device" in out.stdout
    # the reader's sums over the reference's own records (dosage lists, bit arrays) against its --freq
    f = pkg.PgenFile(str(tmp_path / "dos.pgen"))
    assert f.has_dosage()
    cp = T.run_ref(["--pfile", "dos", "--freq", "--out", "fr"], str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    rows = [ln.split() for ln in open(str(tmp_path / "fr.afreq")) if not ln.startswith("#")]
    with_track = 0
    for v in range(f.variant_ct):
        ref_dd, alt_dd = f.dosage_sums(v)
        with_track += f.has_dosage(v)
        x = alt_dd / (ref_dd + alt_dd)
        assert int(rows[v][5]) == (ref_dd + alt_dd) // 16384
        assert abs(float(rows[v][4]) - x) <= (0.6 * 10.0 ** (np.floor(np.log10(x)) - 5) if x else 0.0), (v, rows[v], x)
    assert with_track > 100
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("freq,order,wargs,nonfounders", [(0.3, 2, ["60", "4"], 0), (0.95, 1, ["200kb"], 0), (0.05, 2, ["100", "1"], 9),
                                                          (0.5, 2, ["80", "2", "--maf", "0.2", "--max-maf", "0.48"], 5)])
def test_cli_dosage_frequencies_match_reference(gpu_pkg, cli, tmp_path, freq, order, wargs, nonfounders):
    """--indep-pairwise on files with dosage tracks: r^2 from the hardcalls, the tie-break's allele frequencies from the dosages --
    byte-identical lists.  Random genotypes at a low threshold: a third of the variants go, and which one of a pair goes is decided
    by frequencies that differ between hardcalls and dosages in the second digit."""
    assert T.have_ref(), "reference binary oracle/_ref/plink2 must travel with the repo snapshot"
    dummy_dosage_fileset(tmp_path, "dos", 130, 1500, freq, int(100 * freq) + order)
    if nonfounders:
        # (--dummy writes "#IID SEX PHENO1": give a few samples parents, so that the founders are a subset of the file's samples)
        lines = open(str(tmp_path / "dos.psam")).read().splitlines()
        assert lines[0].split("\t") == ["#IID", "SEX", "PHENO1"]
        out = ["#IID\tPAT\tMAT\tSEX\tPHENO1"]
        for k, ln in enumerate(lines[1:]):
            iid, sex, ph = ln.split("\t")
            out.append("\t".join([iid] + (["per0", "per1"] if (3 <= k < 3 + nonfounders) else ["0", "0"]) + [sex, ph]))
        open(str(tmp_path / "dos.psam"), "w").write("\n".join(out) + "\n")
    k = wargs.index("--maf") if "--maf" in wargs else len(wargs)
    wargs, filters = wargs[:k], wargs[k:]          # (frequency filters: the dosage-based frequencies again)
    common = ["--pfile", "dos", "--indep-pairwise"] + wargs + ["0.03"] + filters + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "4", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, common + ["--out", "hip"], str(tmp_path))
    assert got.returncode == 0, got.stdout
    assert filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "hip.prune.in"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "ref.prune.out"), str(tmp_path / "hip.prune.out"), shallow=False)
    assert re.findall(r"\d+/\d+ variants removed\.", ref.stdout)[-1] in got.stdout
    # ... and the dosages matter: the hardcalls of the same file give another list
    hard = T.run_ref(["--pfile", "dos", "--make-pgen", "erase-dosage", "--out", "hard"], str(tmp_path))
    assert hard.returncode == 0, hard.stdout
    ref2 = T.run_ref(["--pfile", "hard", "--indep-pairwise"] + wargs + ["0.03"] + filters + (["--indep-order", "1"] if order == 1 else []) + ["--threads", "4", "--out", "ref2"], str(tmp_path))
    assert ref2.returncode == 0, ref2.stdout
    if freq >= 0.3:
        assert not filecmp.cmp(str(tmp_path / "ref.prune.in"), str(tmp_path / "ref2.prune.in"), shallow=False)


@pytest.mark.parametrize("dosage", [False, True])
def test_mac_filters_count_like_the_reference(cli, tmp_path, dosage):
    """--mac / --max-mac (plink2.cc:8785-8867, EnforceFreqConstraints plink2_filter.cc:3791): the non-major allele's count over the
    founders -- its dosage sum where records carry dosages -- against thresholds kept in 32768ths of an allele copy, fractional
    arguments rounded as the reference rounds them.  Same number of variants filtered out (the filter runs before a GPU is needed)."""
    if not T.have_ref():
        pytest.skip("oracle/_ref/plink2 not built")
    args = ["--dummy", "90", "400"] + (["dosage-freq=0.4"] if dosage else []) + ["--seed", "21", "--threads", "2", "--make-pgen", "--out", "d"]
    cp = T.run_ref(args, str(tmp_path))
    assert cp.returncode == 0, cp.stdout
    seen = set()
    for flt in (["--mac", "70"], ["--max-mac", "80"], ["--mac", "66.5", "--max-mac", "84.25"], ["--mac", "72", "--maf", "0.42"], ["--mac", "0"]):
        ref = T.run_ref(["--pfile", "d"] + flt + ["--indep-pairwise", "50", "5", "0.2", "--out", "r"], str(tmp_path))
        assert ref.returncode == 0, ref.stdout
        lines = re.findall(r"\d+ variants? removed due to allele frequency threshold", ref.stdout)
        got = run_cli(cli, ["--pfile", "d"] + flt + ["--indep-pairwise", "50", "5", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
        assert got.returncode == 0, got.stdout
        if lines:
            assert lines[-1] in got.stdout, (flt, lines[-1], got.stdout)
            seen.add(lines[-1])
        else:
            assert "removed due to allele frequency" not in got.stdout
    assert len(seen) >= 3
    # nonfounders present: the reference wants to be told whose alleles to count
    lines = open(str(tmp_path / "d.psam")).read().splitlines()
    out = ["#IID\tPAT\tMAT\tSEX\tPHENO1"] + ["\t".join([ln.split("\t")[0]] + (["per0", "per1"] if k in (5, 6) else ["0", "0"]) + ln.split("\t")[1:]) for k, ln in enumerate(lines[1:])]
    open(str(tmp_path / "d.psam"), "w").write("\n".join(out) + "\n")
    ref = T.run_ref(["--pfile", "d", "--mac", "70", "--indep-pairwise", "50", "5", "0.2", "--out", "r"], str(tmp_path))
    got = run_cli(cli, ["--pfile", "d", "--mac", "70", "--indep-pairwise", "50", "5", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
    assert ref.returncode != 0 and got.returncode == 7 and "--ac-founders nor --nonfounders" in ref.stdout.replace("\n", " ") and "--ac-founders nor --nonfounders" in got.stdout.replace("\n", " ")
    ref = T.run_ref(["--pfile", "d", "--mac", "70", "--ac-founders", "--indep-pairwise", "50", "5", "0.2", "--out", "r"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    got = run_cli(cli, ["--pfile", "d", "--mac", "70", "--ac-founders", "--indep-pairwise", "50", "5", "0.2", "--dry-run", "--out", "o"], str(tmp_path))
    assert got.returncode == 0 and re.findall(r"\d+ variants? removed due to allele frequency threshold", ref.stdout)[-1] in got.stdout


def test_make_founders_counts_match_reference(cli, tmp_path):
    """--make-founders ['require-2-missing'] ['first'] (MakeFounders, plink2_filter.cc:4372-4443): which samples become founders,
    judged by both binaries' own log lines (no GPU: --dry-run here, an unrelated cheap command there)."""
    if not T.have_ref():
        pytest.skip("oracle/_ref/plink2 not built")
    m, n = 80, 70
    raw = T.synth_raw_codes(m, n, seed=2)
    T.write_bed(str(tmp_path / "d"), raw, ["1"] * m, np.arange(m) * 100 + 1)
    rng = np.random.default_rng(5)
    lines = []
    for s_ in range(n):
        kind = rng.integers(0, 5)
        pat, mat = "0", "0"
        if kind == 1:
            pat, mat = "s%d" % rng.integers(0, n), "s%d" % rng.integers(0, n)      # both parents in the file
        elif kind == 2:
            pat, mat = "s%d" % rng.integers(0, n), "gone%d" % s_                    # one parent absent
        elif kind == 3:
            pat, mat = "gone%da" % s_, "gone%db" % s_                               # both absent
        elif kind == 4:
            pat, mat = "0", "s%d" % rng.integers(0, n)                              # one parent unknown
        lines.append("s%d s%d %s %s 2 -9" % (s_, s_, pat, mat))
    open(str(tmp_path / "d.fam"), "w").write("\n".join(lines) + "\n")
    open(str(tmp_path / "keep.txt"), "w").write("".join("s%d s%d\n" % (k, k) for k in range(0, n, 2)))
    for mods in ([], ["require-2-missing"], ["first"], ["require-2-missing", "first"]):
        for flt in ([], ["--keep", "keep.txt"]):
            ref = T.run_ref(["--bfile", "d"] + flt + ["--make-founders"] + mods + ["--freq", "--out", "ref"], str(tmp_path))
            got = run_cli(cli, ["--bfile", "d"] + flt + ["--make-founders"] + mods + ["--indep-pairwise", "50", "5", "0.2", "--bad-ld", "--dry-run", "--out", "hip"], str(tmp_path))
            assert ref.returncode == 0 and got.returncode == 0, (mods, flt, ref.stdout[-300:], got.stdout[-300:])
            want = [l.strip() for l in ref.stdout.split("\n") if l.startswith("--make-founders:") or "Skipping --make-founders" in l]
            have = [l.strip() for l in got.stdout.split("\n") if l.startswith("--make-founders:") or "Skipping --make-founders" in l]
            assert want and want == have, (mods, flt, want, have)
