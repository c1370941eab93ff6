#!/usr/bin/env python3
"""CPU-only randomised check of the test ORACLE (oracle/ldoracle.c) against the reference binary (oracle/_ref/plink2):
--indep-pairwise on .bed / fixed-width .pgen and --indep-pairphase on phased variable-width .pgen, random shapes,
chrX/chrY/MT sample layouts with random sexes and non-founders (ldtools.sex_chromosome_rows),
windows, thresholds, scan orders, missing rates.  No GPU involved: this pins the checker the GPU parity tests use.
    python tests/fuzz_oracle.py [--cases 100] [--seed 1]"""
import argparse
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
import ldtools as T  # noqa: E402


def sexed_case(rng, idx, tmp):
    """autosomes + chrX + chrY + MT, random sexes and non-founders, both prune commands on one phased fileset"""
    n = int(rng.choice([60, 97, 130]))
    m = int(rng.integers(120, 400))
    raw, pp, pi = T.synth_phased(m, n, int(rng.integers(1, 1 << 30)), missing_rate=float(rng.choice([0.0, 0.03])), redraw=float(rng.choice([0.05, 0.2])))
    names = ["1", "3", "X", "Y", "MT"]
    cuts = np.sort(rng.integers(0, m, size=len(names) - 1))
    sizes = np.diff(np.concatenate([[0], cuts, [m]]))
    chroms, bps = [], []
    for name, cnt in zip(names, sizes):
        chroms += [name] * int(cnt)
        bps += list((3000000 if name == "X" else 1) + np.sort(rng.integers(1, 60000, size=int(cnt))))
    bps = np.array(bps, dtype=np.uint32)
    sexes = rng.choice([0, 1, 2], size=n, p=[0.1, 0.45, 0.45])
    parents = [("s0", "s1") if (s > 1 and rng.random() < 0.06) else ("0", "0") for s in range(n)]
    founders = np.array([p == ("0", "0") for p in parents])
    while T.ref_pairphase_chrx_is_unreliable(sexes, founders):
        sexes[np.flatnonzero(founders & (sexes != 1))[0]] = 1
    if rng.random() < 0.5:
        kb = int(rng.choice([1, 3, 10]))
        wargs, window, step, is_bp = ["%dkb" % kb], int(kb * 1000 * (1 + T.K_SMALL_EPSILON)), 1, True
    else:
        window = int(rng.integers(2, 120))
        step = int(rng.integers(1, max(2, window)))
        wargs, is_bp = [str(window), str(step)], False
    r2 = float(rng.choice([0.1, 0.2, 0.5, 0.8]))
    order = int(rng.integers(1, 3))
    mode = "phase" if rng.random() < 0.5 else "wise"
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    ids = T.write_pgen_phased(os.path.join(d, "d"), raw, pi, chroms, bps, sexes=sexes, parents=parents)
    kept, removed, _ = T.ref_indep_pairwise(os.path.join(d, "d"), wargs, r2, order=order, mode=mode, threads=2)
    want = np.isin(np.array(ids), np.array(removed))
    got = np.zeros(m, dtype=bool)
    n_f = int(founders.sum())
    for name in dict.fromkeys(chroms):
        vs = np.array([i for i, c in enumerate(chroms) if c == name])
        if not len(vs):
            continue
        zeros = np.zeros(len(vs), dtype=np.uint32)
        if name in ("X", "Y", "MT"):
            codes, mf = T.sex_chromosome_rows(raw[vs], founders, sexes, name, phaseinfo=pi[vs] if (mode == "phase" and name == "X") else None)
            if codes.shape[1] < 2:
                return True, "case %d: too few usable founders on chr%s (skipped)" % (idx, name)
            if mode == "phase":
                res, _ = T.oracle_indep_pairphase(T.haploid_codes_to_hap_rows(codes), codes.shape[1], zeros, bps[vs], mf, window, step, is_bp, r2, order)
            else:
                res, _ = T.oracle_indep_pairwise(T.pack_2bit(codes), codes.shape[1], zeros, bps[vs], mf, window, step, is_bp, r2, order)
        elif mode == "phase":
            rows, mf, unphased, hap_ct = T.oracle_hapsplit(raw[vs][:, founders], (raw[vs][:, founders] == 1).astype(np.uint8), pi[vs][:, founders])
            res, _ = T.oracle_indep_pairphase(rows, hap_ct, zeros, bps[vs], mf, window, step, is_bp, r2, order)
        else:
            inv, mf, _ = T.oracle_prepare(raw[vs][:, founders])
            res, _ = T.oracle_indep_pairwise(inv, n_f, zeros, bps[vs], mf, window, step, is_bp, r2, order)
        got[vs] = res
    desc = "case %d: sexed pair%s n=%d m=%d %s r2=%g order=%d removed=%d" % (idx, mode, n, m, " ".join(wargs), r2, order, int(want.sum()))
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got != want)
        desc += " MISMATCH on " + ",".join(sorted({chroms[b] for b in bad}))
    return bool(np.array_equal(got, want)), desc


def one_case(rng, idx, tmp):
    if rng.random() < 0.3:
        return sexed_case(rng, idx, tmp)
    n = int(rng.choice([50, 64, 97, 130, 257, 513]))
    m = int(rng.integers(60, 400))
    miss = float(rng.choice([0.0, 0.0, 0.01, 0.05, 0.2]))
    phased = bool(rng.random() < 0.4)
    n_chr = int(rng.integers(1, 4))
    chr_idx = np.sort(rng.integers(0, n_chr, size=m)).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(n_chr):
        sel = np.where(chr_idx == c)[0]
        bps[sel] = np.sort(rng.integers(1, 60000, size=len(sel)))
    chroms = [str(c + 1) for c in chr_idx]
    if rng.random() < 0.5:
        wargs, window, step, is_bp = ["%dkb" % k for k in [int(rng.choice([1, 3, 10, 40]))]], None, 1, True
        window = int(float(wargs[0][:-2]) * 1000 * (1 + T.K_SMALL_EPSILON))
    else:
        window = int(rng.integers(2, 150))
        step = int(rng.integers(1, max(2, window)))
        wargs, is_bp = [str(window), str(step)], False
    r2 = float(rng.choice([0.1, 0.2, 0.5, 0.8]))  # values ScanadvDouble and strtod agree on
    order = int(rng.integers(1, 3))
    d = os.path.join(tmp, "c%d" % idx)
    os.makedirs(d)
    prefix = os.path.join(d, "d")
    if phased:
        raw, pp, pi = T.synth_phased(m, n, int(rng.integers(1, 1 << 30)), missing_rate=miss, redraw=float(rng.choice([0.02, 0.1, 0.3])))
        ids = T.write_pgen_phased(prefix, raw, pi, chroms, bps)
        rows, mf, unphased, hap_ct = T.oracle_hapsplit(raw, pp, pi)
        got, _ = T.oracle_indep_pairphase(rows, hap_ct, chr_idx, bps, mf, window, step, is_bp, r2, order)
        kept, removed, _ = T.ref_indep_pairwise(prefix, wargs, r2, order=order, mode="phase")
    else:
        raw = T.synth_raw_codes(m, n, int(rng.integers(1, 1 << 30)), missing_rate=miss)
        fmt = "bfile" if rng.random() < 0.5 else "pfile"
        ids = (T.write_bed if fmt == "bfile" else T.write_pgen_fixed)(prefix, raw, chroms, bps)
        inv, mf, _ = T.oracle_prepare(raw)
        got, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, r2, order)
        kept, removed, _ = T.ref_indep_pairwise(prefix, wargs, r2, order=order, fmt=fmt)
    want = np.isin(np.array(ids), np.array(removed))
    desc = "case %d: %s n=%d m=%d miss=%g %s r2=%g order=%d removed=%d" % (idx, "pairphase" if phased else "pairwise", n, m, miss, " ".join(wargs), r2, order, int(want.sum()))
    return bool(np.array_equal(got, want)), desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    if not T.have_ref():
        sys.exit("oracle/_ref/plink2 is missing (make -C oracle ref)")
    rng = np.random.default_rng(args.seed)
    with tempfile.TemporaryDirectory() as tmp:
        for k in range(args.cases):
            ok, desc = one_case(rng, k, tmp)
            if not ok:
                print("MISMATCH", desc, "(--seed %d)" % args.seed)
                sys.exit(1)
            if k % 20 == 0:
                print(desc, flush=True)
    print("%d cases: oracle == reference" % args.cases)


if __name__ == "__main__":
    main()
