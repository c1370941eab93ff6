#!/usr/bin/env python3
"""Randomised end-to-end comparison on the GPU box: plink2-hip against the reference binary (oracle/_ref/plink2) on
random small filesets -- .bed or fixed-width .pgen, chromosome 0 rows, non-founders, missing calls, kb / count windows,
both scan orders -- for --indep-pairwise (.prune.in/.prune.out), --indep-pairphase on phased variable-width .pgen
(autosomes, chrX/chrY/MT with random sexes, non-founders), the --r2-unphased / --r-unphased table (.vcor: windowed incl.
--ld-window-cm, inter-chr, 'ref-based', cols= sets, --ld-snp / --ld-snps / --ld-snp-list row variants, a chrX with random
sexes now and then) and --clump (.clumps: several reports, --clump-allow-overlap, column sets, --clump-bins, -log10
output).  Files must be byte-identical.
    python tests/fuzz_cli.py [--cases 40] [--seed 1]"""
import argparse
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import ldtools as T  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def hip_only(args, idx):
    """plink2-hip's own flags for this case (the reference does not know them): every fourth prune case runs as N engines on the one device
    (--gpus N --debug-alias-devices: subcontig shards, the exchange of removed bits through the host, the stitch) -- the lists must not change."""
    if (("--indep-pairwise" in args) or ("--indep-pairphase" in args)) and (idx % 4 == 1):
        return ["--gpus", str(2 + idx % 3 + 3 * (idx % 8 == 5)), "--debug-alias-devices"]
    return []


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)


def pairphase_multiallelic_case(cli, ref, rng, idx, tmp, execute=True):
    """autosomes with multiallelic sites: phased VCF -> the reference's import -> both tools"""
    n = int(rng.choice([60, 97, 130]))
    m = int(rng.integers(80, 300))
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, int(rng.integers(1, 1 << 30)), max_alt=int(rng.integers(2, 6)),
                                                      multi_rate=float(rng.choice([0.1, 0.4])), missing_rate=float(rng.choice([0.0, 0.03])))
    cut = int(rng.integers(1, m))
    chroms = ["1"] * cut + ["4"] * (m - cut)
    pos = list(np.sort(rng.integers(1, 60000, size=cut))) + list(np.sort(rng.integers(1, 60000, size=m - cut)))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    T.write_vcf_haps(os.path.join(d, "d.vcf"), first, second, alt_ct, chroms, pos)
    if rng.random() < 0.5:
        win = ["%gkb" % float(rng.choice([2, 7.5, 20]))]
    else:
        w = int(rng.integers(2, 120))
        win = [str(w), str(int(rng.integers(1, max(2, w))))]
    args = ["--pfile", "d", "--indep-pairphase"] + win + [str(rng.choice([0.1, 0.2, 0.5, 0.8])), "--indep-order", str(int(rng.integers(1, 3)))]
    if not execute:
        return True, "case %d skipped" % idx
    T.ref_import_vcf(os.path.join(d, "d.vcf"), os.path.join(d, "d"))
    r = run([ref] + args + ["--threads", "2", "--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in (".prune.in", ".prune.out"):
        if not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            return False, "case %d: %s differs: %s (multiallelic, n=%d m=%d)" % (idx, e, " ".join(args), n, m)
    return True, "case %d ok: %s (multiallelic VCF import)" % (idx, " ".join(args))


def sex_multiallelic_case(cli, ref, rng, idx, tmp, execute=True):
    """--indep-pairwise over variants with several ALT alleles on chromosome 1, X, Y and MT (a VCF imported by the reference, its .pvar
    re-labelled), random sexes incl. unknown, non-founders: the sex-weighted allele counts choose the major allele (round 5)"""
    from test_pgen_reader import make_multiallelic_vcf
    n = int(rng.choice([40, 77, 150]))
    m = int(rng.integers(120, 500))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    seed = int(rng.integers(1, 1 << 30))
    max_alt = int(rng.integers(2, 7))
    miss = float(rng.choice([0.0, 0.03, 0.1]))
    names = [str(x) for x in rng.permutation(["1", "X", "Y", "MT"])[:int(rng.integers(2, 5))]]
    order_of = {"1": 0, "X": 1, "Y": 2, "MT": 3}
    names.sort(key=lambda c: order_of[c])
    cuts = np.sort(rng.integers(1, m, size=len(names) - 1))
    sexes = rng.choice([1, 2, 0], size=n, p=[0.45, 0.4, 0.15])
    nonfounder = rng.random(n) < float(rng.choice([0.0, 0.1]))
    nonfounder[:3] = False
    if rng.random() < 0.5:
        win = ["%gkb" % float(rng.choice([5, 20, 60]))]
    else:
        w = int(rng.integers(2, 150))
        win = [str(w), str(int(rng.integers(1, max(2, w))))]
    args = ["--pfile", "d", "--indep-pairwise"] + win + [str(rng.choice([0.05, 0.1, 0.3, 0.6])), "--indep-order", str(int(rng.integers(1, 3)))]
    if (n < 60) and (rng.random() < 0.8):
        args.append("--bad-ld")   # (fewer than 50 founders: without it both tools must refuse, with the same exit code)
    if not execute:
        return True, "case %d skipped" % idx
    make_multiallelic_vcf(os.path.join(d, "d.vcf"), m, n, seed=seed, max_alt=max_alt, missing=miss)
    T.ref_import_vcf(os.path.join(d, "d.vcf"), os.path.join(d, "d"))
    out, k = [], 0
    for ln in open(os.path.join(d, "d.pvar")):
        if ln.startswith("#"):
            out.append(ln)
            continue
        f = ln.rstrip("\n").split("\t")
        f[0] = names[int(np.searchsorted(cuts, k, side="right"))]
        out.append("\t".join(f) + "\n")
        k += 1
    open(os.path.join(d, "d.pvar"), "w").write("".join(out))
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for q in range(n):
        psam.append("s%d\t%s\t%s\t%s" % (q, "s0" if nonfounder[q] else "0", "s1" if nonfounder[q] else "0", "NA" if sexes[q] == 0 else str(sexes[q])))
    open(os.path.join(d, "d.psam"), "w").write("\n".join(psam) + "\n")
    r = run([ref] + args + ["--threads", "2", "--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in (".prune.in", ".prune.out"):
        if not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            return False, "case %d: %s differs: %s (multiallelic on %s, n=%d m=%d seed=%d max_alt=%d)" % (idx, e, " ".join(args), "/".join(names), n, m, seed, max_alt)
    return True, "case %d ok: %s (multiallelic on %s)" % (idx, " ".join(args), "/".join(names))


def clump_multiallelic_case(cli, ref, rng, idx, tmp, execute=True):
    """--clump over (variant, A1 allele) pairs: LD-carrying multiallelic sites on chromosome 1 / X / Y / MT (a VCF imported by the reference),
    random sexes and non-founders, one or two reports with an allele column, --clump-force-a1 / --clump-allow-overlap / column sets"""
    import test_clump as TC
    import pathlib
    n = int(rng.choice([40, 77, 150, 260]))
    m = int(rng.integers(150, 900))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    seed = int(rng.integers(1, 1 << 30))
    max_alt = int(rng.integers(2, 8))
    names = [str(x) for x in rng.permutation(["1", "X", "Y", "MT"])[:int(rng.integers(1, 5))]]
    order_of = {"1": 0, "X": 1, "Y": 2, "MT": 3}
    names.sort(key=lambda c: order_of[c])
    cuts = np.sort(rng.integers(1, m, size=len(names) - 1))
    sexes = rng.choice([1, 2, 0], size=n, p=[0.45, 0.4, 0.15])
    sexes[:2] = [1, 2]
    nonfounder = rng.random(n) < float(rng.choice([0.0, 0.1]))
    nonfounder[:3] = False
    two = rng.random() < 0.4
    args = ["--pfile", "d", "--clump"]
    if rng.random() < 0.5:
        args.append("cols=" + str(rng.choice(["+a1,+bounds", "+f,+alt,+ref", "-sp2,+a1", "+bounds,-total", "-bins"])))
    args += ["a.txt"] + (["b.txt"] if two else [])
    args += ["--clump-unphased", "--clump-r2", str(rng.choice([0, 0.1, 0.3, 0.7])), "--clump-kb", str(rng.choice([1, 10, 60])),
             "--clump-p1", str(rng.choice(["1e-4", "1e-2", "0.3"])), "--clump-p2", str(rng.choice(["1e-2", "0.05", "1e-6"]))]
    if rng.random() < 0.35:
        args.append("--clump-force-a1")
    if rng.random() < 0.3:
        args.append("--clump-allow-overlap")
    if rng.random() < 0.25:
        args += ["--clump-bins", str(rng.choice(["0.001,0.01", "1e-6,1e-3,0.05,0.5", "0.2"]))]
    rep_seed = int(rng.integers(1, 1 << 30))
    multi_rate = float(rng.choice([0.1, 0.4, 0.9]))
    sig_rate = float(rng.choice([0.05, 0.15]))   # (every draw before the early return: --only replays the same stream)
    if not execute:
        return True, "case %d skipped" % idx
    alt_ct, _, _ = TC.multiallelic_clump_fileset(pathlib.Path(d), m, n, seed, chrom_of=lambda v: names[int(np.searchsorted(cuts, v, side="right"))], max_alt=max_alt,
                                                 multi_rate=multi_rate)
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for q in range(n):
        psam.append("s%d\t%s\t%s\t%s" % (q, "s0" if nonfounder[q] else "0", "s1" if nonfounder[q] else "0", "NA" if sexes[q] == 0 else str(sexes[q])))
    open(os.path.join(d, "d.psam"), "w").write("\n".join(psam) + "\n")
    TC.write_allele_report(os.path.join(d, "a.txt"), alt_ct, rep_seed, False, sig_rate=sig_rate)
    if two:
        TC.write_allele_report(os.path.join(d, "b.txt"), alt_ct, rep_seed + 1, False, sig_rate=0.08)
    r = run([ref] + args + ["--threads", "2", "--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if (r.returncode < 0) and ("--clump-force-a1" in args):
        # the reference died of a signal: under --clump-force-a1 an entry of a multiallelic allele carries the forced-A1 bit of the previous biallelic
        # line, and its SP2 printer then reads allele_storage[allele + 1] (plink2_ld.cc:9357) -- past the table's end when that allele is the last one
        return g.returncode == 0, "case %d: the reference crashed (signal %d) on its stale forced-A1 bit, plink2-hip exit %d: %s" % (idx, -r.returncode, g.returncode, " ".join(args))
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in (".clumps", ".clumps.missing_allele"):
        have_ref, have_hip = os.path.exists(os.path.join(d, "ref" + e)), os.path.exists(os.path.join(d, "hip" + e))
        if (not have_ref) and (not have_hip):
            continue
        if (have_ref != have_hip) or not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            return False, "case %d: %s differs: %s (multiallelic --clump on %s, n=%d m=%d seed=%d max_alt=%d)" % (idx, e, " ".join(args), "/".join(names), n, m, seed, max_alt)
    return True, "case %d ok: %s (multiallelic --clump on %s)" % (idx, " ".join(args), "/".join(names))


def sex_multiallelic_pairphase_case(cli, ref, rng, idx, tmp, execute=True):
    """--indep-pairphase over phased multiallelic sites on chromosome 1 / X / Y / MT (the fileset of clump_multiallelic_case), random sexes, non-founders"""
    import test_clump as TC
    import pathlib
    n = int(rng.choice([60, 77, 150, 260]))
    m = int(rng.integers(150, 700))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    seed = int(rng.integers(1, 1 << 30))
    max_alt = int(rng.integers(2, 8))
    names = [str(x) for x in rng.permutation(["1", "X", "Y", "MT"])[:int(rng.integers(1, 5))]]
    order_of = {"1": 0, "X": 1, "Y": 2, "MT": 3}
    names.sort(key=lambda c: order_of[c])
    cuts = np.sort(rng.integers(1, m, size=len(names) - 1))
    sexes = rng.choice([1, 2, 0], size=n, p=[0.45, 0.4, 0.15])
    sexes[:4] = [1, 2, 1, 2]
    nonfounder = rng.random(n) < float(rng.choice([0.0, 0.1]))
    nonfounder[:4] = False
    while ("X" in names) and T.ref_pairphase_chrx_is_unreliable(sexes, ~nonfounder):   # (the reference's chrX loader reads a stale word there: ldtools)
        sexes[np.flatnonzero((~nonfounder) & (sexes != 1))[0]] = 1
    multi_rate = float(rng.choice([0.1, 0.4, 0.9]))
    if rng.random() < 0.5:
        win = ["%gkb" % float(rng.choice([5, 20, 60]))]
    else:
        w = int(rng.integers(2, 150))
        win = [str(w), str(int(rng.integers(1, max(2, w))))]
    args = ["--pfile", "d", "--indep-pairphase"] + win + [str(rng.choice([0.05, 0.1, 0.3, 0.6])), "--indep-order", str(int(rng.integers(1, 3)))]
    if not execute:
        return True, "case %d skipped" % idx
    TC.multiallelic_clump_fileset(pathlib.Path(d), m, n, seed, chrom_of=lambda v: names[int(np.searchsorted(cuts, v, side="right"))], max_alt=max_alt, multi_rate=multi_rate)
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for q in range(n):
        psam.append("s%d\t%s\t%s\t%s" % (q, "s0" if nonfounder[q] else "0", "s1" if nonfounder[q] else "0", "NA" if sexes[q] == 0 else str(sexes[q])))
    open(os.path.join(d, "d.psam"), "w").write("\n".join(psam) + "\n")
    r = run([ref] + args + ["--threads", "2", "--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in (".prune.in", ".prune.out"):
        if not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            return False, "case %d: %s differs: %s (phased multiallelic on %s, n=%d m=%d seed=%d max_alt=%d)" % (idx, e, " ".join(args), "/".join(names), n, m, seed, max_alt)
    return True, "case %d ok: %s (phased multiallelic on %s)" % (idx, " ".join(args), "/".join(names))


def r2_multiallelic_case(cli, ref, rng, idx, tmp, execute=True):
    """the r^2 outputs over autosomal variants with several ALT alleles (often with a major allele other than REF) beside a biallelic chrX: matrices, the
    inter-chr and the windowed table, r and r^2, major-allele and REF orientation, sexes incl. unknown, non-founders"""
    n = int(rng.choice([50, 90, 170]))
    m = int(rng.integers(120, 420))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    seed = int(rng.integers(1, 1 << 30))
    max_alt = int(rng.integers(2, 6))
    with_x = rng.random() < 0.75
    cut1, cut2 = sorted(int(x) for x in rng.integers(20, m - 20, size=2))
    sexes = rng.choice([1, 2, 0], size=n, p=[0.5, 0.4, 0.1])
    sexes[:2] = [1, 2]
    nonfounder = rng.random(n) < float(rng.choice([0.0, 0.1]))
    nonfounder[:4] = False
    multi_rate = float(rng.choice([0.2, 0.6]))
    swap_seed = int(rng.integers(1, 1 << 30))
    kind = int(rng.integers(0, 7))
    thr = str(rng.choice([0, 0.02, 0.2]))
    if kind == 0:
        args = ["--r2-unphased", str(rng.choice(["square", "square0", "triangle"])), str(rng.choice(["bin", "bin4"]))]
    elif kind == 1:
        args = ["--r2-unphased", "inter-chr", "allow-ambiguous-allele", "--ld-window-r2", thr]
    elif kind == 2:
        args = ["--r2-unphased", "inter-chr", "cols=+maj,+nonmaj,+freq", "--ld-window-r2", thr]
    elif kind == 3:
        args = ["--r-unphased", "triangle", "bin"]
    elif kind == 4:
        args = ["--r-unphased", "inter-chr", "cols=+maj,+nonmaj", "--ld-window-r2", thr]
    elif kind == 5:
        args = ["--r2-unphased", "cols=+ref,+alt", "--ld-window-kb", str(rng.choice([5, 20, 80])), "--ld-window-r2", thr] + (["ref-based"] if rng.random() < 0.4 else [])
        if "ref-based" in args:
            args.remove("ref-based")
            args.insert(1, "ref-based")
    else:
        args = ["--r-unphased", "ref-based", "inter-chr", "cols=+ref,+alt", "--ld-window-r2", thr]
    args = ["--pfile", "d"] + args
    ext = ".vcor" if not any(a in args for a in ("bin", "bin4")) else (".unphased.vcor1." if args[2] == "--r-unphased" else ".unphased.vcor2.") + "bin"
    if not execute:
        return True, "case %d skipped" % idx
    first, second, alt_ct = T.synth_multiallelic_haps(m, n, seed, max_alt=max_alt, multi_rate=multi_rate, missing_rate=0.02, ld_copy_prob=0.6, redraw=0.1)
    chroms = ["1"] * cut1 + (["X"] if with_x else ["2"]) * (cut2 - cut1) + ["7"] * (m - cut2)
    srng = np.random.default_rng(swap_seed)
    x_multi = srng.random() < 0.6          # (round 5: multiallelic sites on chrX stay multiallelic -- and, where plink2-hip computes them, on chrY / MT behind the autosomes)
    tail_ok = (kind == 5) or ((kind in (0, 1)) and not with_x)
    if tail_ok and (srng.random() < 0.6):
        t0 = cut2 + (m - cut2) // 2
        for v in range(t0, m):
            chroms[v] = "Y" if v < t0 + (m - t0) // 2 else "MT"
    for v in range(m):
        if (chroms[v] == "X") and not x_multi:
            first[v] = np.where(first[v] > 1, 1, first[v])
            second[v] = np.where(second[v] > 1, 1, second[v])
            alt_ct[v] = 1
        elif alt_ct[v] > 1 and srng.random() < 0.5:   # (an ALT allele takes REF's place: the major allele is then often not REF)
            a = int(srng.integers(1, alt_ct[v] + 1))
            f0, s0 = first[v].copy(), second[v].copy()
            first[v] = np.where(f0 == 0, a, np.where(f0 == a, 0, f0))
            second[v] = np.where(s0 == 0, a, np.where(s0 == a, 0, s0))
    pos = 1000 + np.cumsum(srng.integers(1, 900, size=m))
    T.write_vcf_haps(os.path.join(d, "d.vcf"), first, second, alt_ct, ["1"] * m, pos)
    T.ref_import_vcf(os.path.join(d, "d.vcf"), os.path.join(d, "d"))
    out, k = [], 0
    for ln in open(os.path.join(d, "d.pvar")):
        if not ln.startswith("#"):
            f = ln.split("\t")
            f[0] = chroms[k]
            ln = "\t".join(f)
            k += 1
        out.append(ln)
    open(os.path.join(d, "d.pvar"), "w").write("".join(out))
    psam = ["#IID\tPAT\tMAT\tSEX"]
    for q in range(n):
        psam.append("s%d\t%s\t%s\t%s" % (q, "s0" if nonfounder[q] else "0", "s1" if nonfounder[q] else "0", "NA" if sexes[q] == 0 else str(sexes[q])))
    open(os.path.join(d, "d.psam"), "w").write("\n".join(psam) + "\n")
    r = run([ref] + args + ["--threads", "2", "--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    a, b = os.path.join(d, "ref" + ext), os.path.join(d, "hip" + ext)
    if (not os.path.exists(a)) or (not os.path.exists(b)) or not filecmp.cmp(a, b, shallow=False):
        return False, "case %d: %s differs: %s (multiallelic autosomes%s, n=%d m=%d seed=%d)" % (idx, ext, " ".join(args), " + chrX" if with_x else "", n, m, seed)
    return True, "case %d ok: %s (multiallelic autosomes%s)" % (idx, " ".join(args), " + chrX" if with_x else "")


def pairphase_case(cli, ref, rng, idx, tmp, execute=True):
    if rng.random() < 0.25:
        return pairphase_multiallelic_case(cli, ref, rng, idx, tmp, execute)
    n = int(rng.choice([60, 97, 130, 513]))
    m = int(rng.integers(80, 500))
    raw, pp, pi = T.synth_phased(m, n, int(rng.integers(1, 1 << 30)), missing_rate=float(rng.choice([0.0, 0.02, 0.1])),
                                 redraw=float(rng.choice([0.02, 0.1, 0.3])))
    names = ["1", "2", "5"]
    if rng.random() < 0.5:
        names += ["X", "Y", "MT"]
    cuts = np.sort(rng.integers(0, m, size=len(names) - 1))
    sizes = np.diff(np.concatenate([[0], cuts, [m]]))
    chroms, pos = [], []
    for name, cnt in zip(names, sizes):
        chroms += [name] * int(cnt)
        pos += list((3000000 if name == "X" else 1) + np.sort(rng.integers(1, 60000, size=int(cnt))))
    if rng.random() < 0.3 and sizes[0] > 3:
        chroms[0] = chroms[1] = "0"
    sexes = rng.choice([0, 1, 2], size=n, p=[0.1, 0.45, 0.45])
    parents = [("s0", "s1") if (s > 1 and rng.random() < 0.05) else ("0", "0") for s in range(n)]
    founders = np.array([p == ("0", "0") for p in parents])
    while ("X" in names) and T.ref_pairphase_chrx_is_unreliable(sexes, founders):
        # a sample layout on which the reference reads uninitialised memory (see ldtools): move one founder to the males
        sexes[np.flatnonzero(founders & (sexes != 1))[0]] = 1
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    T.write_pgen_phased(os.path.join(d, "d"), raw, pi, chroms, np.array(pos), sexes=sexes, parents=parents)
    if rng.random() < 0.5:
        win = ["%gkb" % float(rng.choice([0.5, 2, 7.5, 20]))]
    else:
        w = int(rng.integers(2, 120))
        win = [str(w), str(int(rng.integers(1, max(2, w))))]
    args = ["--pfile", "d", "--indep-pairphase"] + win + [str(rng.choice([0.1, 0.2, 0.3, 0.5, 0.8])), "--indep-order", str(int(rng.integers(1, 3)))]
    if not execute:
        return True, "case %d skipped" % idx
    r = run([ref] + args + ["--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode:
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), r.stdout[-300:], g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in (".prune.in", ".prune.out"):
        if not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            a = set(open(os.path.join(d, "ref.prune.out")).read().split())
            b = set(open(os.path.join(d, "hip.prune.out")).read().split())
            where = sorted((int(x[3:]), chroms[int(x[3:])], "ref-only" if x in a else "hip-only") for x in a ^ b)
            return False, "case %d: %s differs: %s (n=%d m=%d chroms=%s) %s\n%s" % (idx, e, " ".join(args), n, m, ",".join(names), where[:12], g.stdout[-500:])
    return True, "case %d ok: %s (%s)" % (idx, " ".join(args), ",".join(names))


def one_case(cli, ref, rng, idx, tmp, execute=True, mode="all"):
    if mode == "pairphase" or rng.random() < 0.3:
        return pairphase_case(cli, ref, rng, idx, tmp, execute)
    n = int(rng.choice([60, 97, 130, 513, 700]))
    m = int(rng.integers(60, 500))
    miss = float(rng.choice([0.0, 0.0, 0.01, 0.08]))
    raw = T.synth_raw_codes(m, n, seed=int(rng.integers(1, 1 << 30)), missing_rate=miss)
    for _ in range(m // 5):
        a = int(rng.integers(1, m))
        keep = rng.random(n) < 0.9
        raw[a] = np.where(keep, raw[max(0, a - int(rng.integers(1, 6)))], raw[a])
    n_zero = int(rng.integers(0, 4))
    names = ["0"] * n_zero
    chr_ct = int(rng.integers(1, 4))
    per = np.sort(rng.integers(0, chr_ct, size=m - n_zero))
    labels = [str(c) for c in rng.choice(np.arange(1, 23), size=chr_ct, replace=False)]
    labels.sort(key=int)
    kind = rng.random()
    if mode == "clump":
        kind = 0.05
    # the r^2 outputs also get a chrX now and then (male founders weighted down, ComputeXR2), with random sexes
    with_x = ((kind >= 0.6) or (kind < 0.12)) and (rng.random() < 0.3)   # (also for --clump)
    sexes = rng.choice([1, 2, 0], size=n, p=[0.45, 0.45, 0.1]) if with_x else np.full(n, 2)
    if with_x:
        labels[-1] = "X"
    names += [labels[c] for c in per]
    pos = np.zeros(m, dtype=np.int64)
    pos[:n_zero] = np.arange(n_zero) + 1
    for c in range(chr_ct):
        sel = np.where(per == c)[0] + n_zero
        pos[sel] = np.sort(rng.integers(1, 80000, size=len(sel)))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    use_bed = bool(rng.random() < 0.5)
    if use_bed:
        T.write_bed(os.path.join(d, "d"), raw, names, pos)
        inp = ["--bfile", "d"]
        if with_x:
            lines = open(os.path.join(d, "d.fam")).read().splitlines()
            open(os.path.join(d, "d.fam"), "w").write("\n".join(" ".join(ln.split()[:4] + [str(sexes[k]), "-9"]) for k, ln in enumerate(lines)) + "\n")
    else:
        T.write_pgen_fixed(os.path.join(d, "d"), raw, names, pos, sexes=sexes if with_x else None)
        inp = ["--pfile", "d"]
    # some samples with parents in the file (non-founders): the engines pick the founder columns themselves
    founders = n
    if rng.random() < 0.4:
        nonf = np.sort(rng.choice(np.arange(2, n), size=int(rng.integers(1, max(2, n // 5))), replace=False))
        founders = n - len(nonf)
        if use_bed:
            lines = open(os.path.join(d, "d.fam")).read().splitlines()
            for sidx in nonf:
                t = lines[sidx].split()
                t[2], t[3] = "s0", "s1"
                lines[sidx] = " ".join(t)
            open(os.path.join(d, "d.fam"), "w").write("\n".join(lines) + "\n")
        else:
            isnf = np.zeros(n, dtype=bool)
            isnf[nonf] = True
            with open(os.path.join(d, "d.psam"), "w") as f:
                f.write("#IID\tPAT\tMAT\tSEX\n")
                for sidx in range(n):
                    f.write("s%d\t%s\t%s\t%s\n" % (sidx, "s0" if isnf[sidx] else "0", "s1" if isnf[sidx] else "0", "NA" if sexes[sidx] == 0 else str(sexes[sidx])))
    # filters in front of the command: a chromosome subset, an ID list to drop, a sample list to keep
    if rng.random() < 0.3:
        how = rng.random()
        if (how < 0.4) and (chr_ct > 1):
            inp = inp + ["--chr", ",".join(labels[:max(1, chr_ct - 1)])]
        elif how < 0.7:
            with open(os.path.join(d, "drop.txt"), "w") as f:
                f.write("\n".join("snp%d" % v for v in rng.choice(m, size=max(1, m // 7), replace=False)) + "\nnone_such\n")
            inp = inp + ["--exclude", "drop.txt"]
        elif founders == n:
            keep = np.sort(rng.choice(n, size=max(55, n * 3 // 4), replace=False))
            with open(os.path.join(d, "keep.txt"), "w") as f:
                f.write("".join(("s%d s%d\n" % (s, s)) if use_bed else ("s%d\n" % s) for s in keep))
            inp = inp + ["--keep", "keep.txt"]
            founders = len(keep)
    if (rng.random() < 0.2) and not with_x:
        inp = inp + [str(rng.choice(["--maf", "--geno"])), str(rng.choice([0.01, 0.02, 0.05, 0.1, 0.3]))]
    if kind < 0.12:
        # --clump on a random report (unknown IDs, repeated lines, odd p-value spellings come from tests/test_clump.py)
        import test_clump as TC
        TC.write_report(os.path.join(d, "assoc.txt"), m, int(rng.integers(1, 1 << 30)), sig_rate=float(rng.choice([0.02, 0.06, 0.2])))
        args = inp + ["--clump", "assoc.txt", "--clump-unphased", "--clump-r2", str(rng.choice([0, 0.1, 0.5, 0.8])), "--clump-kb", str(rng.choice([1, 10, 250])),
                      "--clump-p1", str(rng.choice(["1e-4", "1e-2", "0.3"])), "--clump-p2", str(rng.choice(["1e-2", "0.05", "1e-6"]))]
        if rng.random() < 0.4:
            args.append("--clump-allow-overlap")
        if rng.random() < 0.4:   # column sets decide what is kept of a report line (and which variants count as observed)
            k = args.index("assoc.txt")
            args.insert(k, str(rng.choice(["cols=+bounds", "cols=-bins", "cols=-total,-bins", "cols=sp2", "cols=+ref,+alt1,+f", "cols=chrom,pos,total,bounds"])))
        if (rng.random() < 0.3) and not any(a in ("cols=-bins", "cols=-total,-bins", "cols=sp2", "cols=chrom,pos,total,bounds") for a in args):
            args += ["--clump-bins", str(rng.choice(["0.001,0.01", "1e-6,1e-3,0.05,0.5", "0.2"]))]
        if rng.random() < 0.2:
            args += ["--clump-log10", "output-only"]
        if rng.random() < 0.3:
            TC.write_report(os.path.join(d, "assoc2.txt"), m, int(rng.integers(1, 1 << 30)), sig_rate=0.05)
            k = args.index("assoc.txt")
            args.insert(k + 1, "assoc2.txt")
        outs = [".clumps"]
    elif kind < 0.6:
        if rng.random() < 0.5:
            win = ["%gkb" % float(rng.choice([0.5, 2, 7.5, 20]))]
        else:
            w = int(rng.integers(2, 120))
            win = [str(w), str(int(rng.integers(1, max(2, w))))]
        args = inp + ["--indep-pairwise"] + win + [str(rng.choice([0.1, 0.2, 0.3, 0.5, 0.8])), "--indep-order", str(int(rng.integers(1, 3)))]
        if founders < 50:
            args.append("--bad-ld")
        outs = [".prune.in", ".prune.out"]
    else:
        # r^2 or signed r, orientation, column set
        flag = ["--r-unphased"] if rng.random() < 0.3 else ["--r2-unphased"]
        mods = ["ref-based"] if rng.random() < 0.3 else []
        if rng.random() < 0.4:
            mods.append(str(rng.choice(["cols=+maj,+nonmaj,+freq", "cols=+ref,+alt1,+alt", "cols=-chrom,-pos,+maj,+ref", "cols=id,ref,maj,freq", "cols=+provref,+ref"])))
        if rng.random() < 0.25:
            args = inp + flag + ["inter-chr"] + mods + ["--ld-window-r2", str(rng.choice([0, 0.05, 0.2, 0.6]))]
        else:
            args = inp + flag + mods + ["--ld-window-kb", str(rng.choice([1, 5, 30])), "--ld-window-r2", str(rng.choice([0, 0.05, 0.2, 0.6]))]
            if rng.random() < 0.5:
                args += ["--ld-window", str(int(rng.integers(2, 40)))]
            elif use_bed and (rng.random() < 0.4):
                # centimorgan positions on a coarse grid (pairs exactly one radius apart) and a centimorgan window
                rows = [ln.split("\t") for ln in open(os.path.join(d, "d.bim")).read().splitlines()]
                cm, last = 0.0, None
                for r_ in rows:
                    if r_[0] != last:
                        cm, last = 0.0, r_[0]
                    cm += float(rng.choice([0.0, 0.125, 0.25, 0.5]))
                    r_[2] = "%g" % cm
                open(os.path.join(d, "d.bim"), "w").write("\n".join("\t".join(r_) for r_ in rows) + "\n")
                args += ["--ld-window-cm", str(rng.choice([0.25, 0.5, 1, 3]))]
        if ("--ld-window" not in args) and ("--ld-window-cm" not in args) and (rng.random() < 0.35):
            # row variants: a single ID, ranges, or a list file with an unknown ID in it
            pick = sorted(int(x) for x in rng.choice(np.arange(n_zero, m), size=int(rng.integers(1, 6)), replace=False))
            how = rng.random()
            if how < 0.3:
                args += ["--ld-snp", "snp%d" % pick[0]]
            elif how < 0.65:
                toks = ["snp%d" % v for v in pick[:-2]] + (["snp%d-snp%d" % (pick[-2], pick[-1])] if len(pick) >= 2 else ["snp%d" % pick[0]])
                args += ["--ld-snps", ",".join(toks)]
            else:
                with open(os.path.join(d, "rows.txt"), "w") as f:
                    f.write(" ".join("snp%d" % v for v in pick) + "\nnot_there\n")
                args += ["--ld-snp-list", "rows.txt"]
        outs = [".vcor"]
    if not execute:
        return True, "case %d skipped" % idx
    r = run([ref] + args + ["--out", "ref"], d)
    g = run([cli] + args + hip_only(args, idx) + ["--out", "hip"], d)
    if r.returncode != g.returncode and not (r.returncode != 0 and g.returncode != 0):
        return False, "case %d: exit codes differ (ref %d, hip %d): %s\n%s" % (idx, r.returncode, g.returncode, " ".join(args), g.stdout[-400:])
    if r.returncode != 0:
        return True, "case %d: both refuse (%s)" % (idx, " ".join(args))
    for e in outs:
        have_ref, have_hip = os.path.exists(os.path.join(d, "ref" + e)), os.path.exists(os.path.join(d, "hip" + e))
        if (e == ".clumps") and (not have_ref) and (not have_hip):
            continue  # (nothing significant: both skip the file)
        if (have_ref != have_hip) or not filecmp.cmp(os.path.join(d, "ref" + e), os.path.join(d, "hip" + e), shallow=False):
            return False, "case %d: %s differs: %s (n=%d m=%d miss=%g %s, %d founders)" % (idx, e, " ".join(args), n, m, miss, "bed" if use_bed else "pgen", founders)
    return True, "case %d ok: %s" % (idx, " ".join(args))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", default="all", choices=["all", "pairphase", "clump", "sexmulti", "clumpmulti", "sexmultiphase", "r2multi"])
    ap.add_argument("--only", type=int, default=None, help="replay the random stream but execute only this case")
    ap.add_argument("--keep", default=None, help="directory to keep the case files in (default: a temporary directory)")
    args = ap.parse_args()
    pkg = ge.load_package()
    cli = pkg.build_cli()
    ref = os.path.join(REPO, "oracle", "_ref", "plink2")
    if not os.path.exists(ref):
        sys.exit("oracle/_ref/plink2 is missing")
    rng = np.random.default_rng(args.seed)
    import contextlib
    with (contextlib.nullcontext(args.keep) if args.keep else tempfile.TemporaryDirectory()) as tmp:
        if args.keep:
            os.makedirs(tmp, exist_ok=True)
        for k in range(args.cases):
            if args.mode == "sexmulti":
                ok, desc = sex_multiallelic_case(cli, ref, rng, k, tmp, execute=(args.only is None or k == args.only))
            elif args.mode == "sexmultiphase":
                ok, desc = sex_multiallelic_pairphase_case(cli, ref, rng, k, tmp, execute=(args.only is None or k == args.only))
            elif args.mode == "r2multi":
                ok, desc = r2_multiallelic_case(cli, ref, rng, k, tmp, execute=(args.only is None or k == args.only))
            elif args.mode == "clumpmulti":
                ok, desc = clump_multiallelic_case(cli, ref, rng, k, tmp, execute=(args.only is None or k == args.only))
            else:
                ok, desc = one_case(cli, ref, rng, k, tmp, execute=(args.only is None or k == args.only), mode=args.mode)
            if not ok:
                print("MISMATCH", desc, "(--seed %d)" % args.seed)
                sys.exit(1)
            if k % 10 == 0:
                print(desc, flush=True)
    print("%d cases byte-identical to the reference" % args.cases)


if __name__ == "__main__":
    main()
