"""Variants with several ALT alleles on chrX / chrY / MT under --indep-pairwise (round 5).  Pinned WITHOUT a GPU: the rule
plink2-hip's host code follows (p2h_tables.cpp: multiallelic_sex_row) -- allele counts weighted as the reference's allele-frequency
pass weighs the founders of each chromosome class (LoadAlleleAndGenoCountsThread, plink2_data.cc:2752-2895: chrX a non-male's allele
copy counts 2 and a male's 1; chrY non-female founders, MT every founder, diploid-style), GetMajIdxMulti on them, PgrGetInv1's collapse
on that allele, then the class's sample layout (males / haploid samples first with "one major + one other" made missing, chrX
non-males twice) -- is restated in numpy, run through the oracle's scan, and must reproduce the prune lists the reference binary
writes for the same fileset.  The GPU side (the same lists from plink2-hip) is tests/test_cli.py."""
import os

import numpy as np
import pytest

import ldtools as T


def build_fileset(tmp_path, m, n, max_alt, seed, unknown):
    from test_pgen_reader import make_multiallelic_vcf
    alt_ct, lo, hi = make_multiallelic_vcf(str(tmp_path / "m.vcf"), m, n, seed=seed, max_alt=max_alt, missing=0.04)
    mk = T.run_ref(["--vcf", "m.vcf", "--make-pgen", "--out", "mv"], str(tmp_path))
    assert mk.returncode == 0, mk.stdout
    out, chroms, bps, k = [], [], [], 0
    for ln in open(str(tmp_path / "mv.pvar")):
        if ln.startswith("#"):
            out.append(ln)
            continue
        f = ln.rstrip("\n").split("\t")
        f[0] = ["1", "X", "Y", "MT"][min(3, (4 * k) // m)]
        chroms.append(f[0])
        bps.append(int(f[1]))
        out.append("\t".join(f) + "\n")
        k += 1
    open(str(tmp_path / "mv.pvar"), "w").write("".join(out))
    rng = np.random.default_rng(seed)
    sexes = rng.choice([1, 2, 0] if unknown else [1, 2], size=n, p=[0.45, 0.45, 0.1] if unknown else [0.5, 0.5])
    founder = np.ones(n, dtype=bool)
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for s in range(n):
        nf = (s % 9 == 4) and s > 9
        founder[s] = not nf
        psam.append("s%d\t%s\t%s\t%s" % (s, "s0" if nf else "0", "s1" if nf else "0", "NA" if sexes[s] == 0 else str(sexes[s])))
    open(str(tmp_path / "mv.psam"), "w").write("\n".join(psam) + "\n")
    return alt_ct, lo, hi, np.array(chroms), np.array(bps, dtype=np.uint32), sexes, founder


def class_rows(lo, hi, alt_ct, part1, part2, haploid):
    """(codes of the class's virtual samples, major-allele frequencies) for the variants given; part1: founders counted once (their
    "one major + one other allele" calls become missing when `haploid`), part2: founders counted -- and laid out -- twice"""
    m = lo.shape[0]
    rows = np.zeros((m, len(part1) + 2 * len(part2)), dtype=np.uint8)
    mf = np.zeros(m)
    for v in range(m):
        k = int(alt_ct[v]) + 1
        cnt = np.zeros(k, dtype=np.int64)
        for smp, w in ((part1, 1), (part2, 2)):
            a, b = lo[v][smp], hi[v][smp]
            ok = a != 255
            cnt += w * (np.bincount(a[ok], minlength=k)[:k] + np.bincount(b[ok], minlength=k)[:k])
        maj, mf[v] = T.major_allele_multi(cnt)

        def codes(smp):
            a, b = lo[v][smp], hi[v][smp]
            return np.where(a == 255, 3, (a != maj).astype(np.uint8) + (b != maj).astype(np.uint8)).astype(np.uint8)
        c1 = codes(part1)
        if haploid:
            c1 = np.where(c1 == 1, 3, c1)
        c2 = codes(part2)
        rows[v] = np.concatenate([c1, c2, c2])
    return rows, mf


@pytest.mark.skipif(not T.have_ref(), reason="reference binary not built")
@pytest.mark.parametrize("max_alt,wargs,order,unknown", [(3, ["40kb"], 2, True), (6, ["80", "7"], 1, False), (2, ["30", "1"], 2, True)])
def test_oracle_with_the_sex_weighted_collapse_reproduces_reference_lists(pkg, tmp_path, max_alt, wargs, order, unknown):
    m, n = 800, 150
    alt_ct, lo, hi, chroms, bps, sexes, founder = build_fileset(tmp_path, m, n, max_alt, 20 + max_alt, unknown)
    common = ["--pfile", "mv", "--indep-pairwise"] + wargs + ["0.1"] + (["--indep-order", "1"] if order == 1 else [])
    ref = T.run_ref(common + ["--threads", "3", "--out", "ref"], str(tmp_path))
    assert ref.returncode == 0, ref.stdout
    want = set(open(str(tmp_path / "ref.prune.out")).read().split())
    is_bp = wargs[0].endswith("kb")
    window = int(float(wargs[0][:-2]) * 1000) if is_bp else int(wargs[0])
    step = 1 if is_bp else int(wargs[1])
    f_idx = np.flatnonzero(founder)
    classes = {"1": (f_idx, np.zeros(0, dtype=np.int64), False),
               "X": (np.flatnonzero(founder & (sexes == 1)), np.flatnonzero(founder & (sexes != 1)), True),
               "Y": (np.flatnonzero(founder & (sexes != 2)), np.zeros(0, dtype=np.int64), True),
               "MT": (f_idx, np.zeros(0, dtype=np.int64), True)}
    got = set()
    for name, (p1, p2, haploid) in classes.items():
        vs = np.flatnonzero(chroms == name)
        rows, mf = class_rows(lo[vs], hi[vs], alt_ct[vs], p1, p2, haploid)
        zeros = np.zeros(len(vs), dtype=np.uint32)
        res, _ = T.oracle_indep_pairwise(T.pack_2bit(rows), rows.shape[1], zeros, bps[vs], mf, window, step, is_bp, 0.1, order)
        got |= {"snp%d" % v for v in vs[np.asarray(res, dtype=bool)]}
    diff = sorted(want ^ got, key=lambda x: int(x[3:]))
    assert not diff, "the restated rule differs from the reference on %d variants, e.g. %s" % (len(diff), diff[:10])
    assert 0 < len(got) < m and (alt_ct[chroms != "1"] > 1).sum() > 100
