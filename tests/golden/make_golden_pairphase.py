#!/usr/bin/env python3
"""Golden vectors for --indep-pairphase, produced by the REFERENCE binary (oracle/_ref/plink2) in this container:
phased VCFs written from seeded numpy haplotypes -> the reference's --vcf import (variable-width .pgen with the
hardcall-phase track) -> its --indep-pairphase prune lists for a grid of windows / thresholds / scan orders.
Kept: the small .pgen files the reference wrote (reader tests), the generating arrays, and the removed sets.
Re-run:  python tests/golden/make_golden_pairphase.py"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ldtools as T  # noqa: E402

GRID = [(["50", "5"], 0.2, 2), (["50", "5"], 0.2, 1), (["50", "5"], 0.5, 2), (["30"], 0.5, 1),
        (["20kb"], 0.2, 2), (["20kb"], 0.5, 1), (["8kb"], 0.8, 2)]


def positions(m, nchr, seed):
    rng = np.random.default_rng(seed)
    per = (m + nchr - 1) // nchr
    chr_idx = np.arange(m) // per
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(nchr):
        idx = np.where(chr_idx == c)[0]
        gaps = rng.integers(1, 600, size=len(idx))
        gaps = np.where(rng.random(len(idx)) < 0.01, gaps + 300000, gaps)
        bps[idx] = 1000 + np.cumsum(gaps)
    return [str(c + 1) for c in chr_idx], bps


def main():
    if not T.have_ref():
        sys.exit("reference binary missing: make -C oracle ref")
    os.makedirs(os.path.join(HERE, "pgen"), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="golden_pp_")
    try:
        # ---- fully phased, some missing calls
        m, n = 500, 150
        raw, pp, pi = T.synth_phased(m, n, seed=20260925, missing_rate=0.02)
        raw[17] = 0  # monomorphic
        pp[17] = 0
        pi[17] = 0
        chroms, bps = positions(m, 3, 11)
        ids = T.write_vcf(os.path.join(tmp, "a.vcf"), raw, chroms, bps, pp, pi)
        T.ref_import_vcf(os.path.join(tmp, "a.vcf"), os.path.join(tmp, "a"))
        out = dict(raw=raw, phasepresent=pp, phaseinfo=pi, chroms=np.array([int(c) for c in chroms], dtype=np.uint32), bps=bps)
        for k, (win, r2, order) in enumerate(GRID):
            kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "a"), win, r2, order=order, mode="phase")
            out["removed_%d" % k] = np.isin(np.array(ids), np.array(removed))
            print("phased_small", win, r2, order, [ln for ln in log.splitlines() if "variants removed" in ln][-1])
        out["grid"] = np.array(["%s|%r|%d" % (" ".join(w), r, o) for w, r, o in GRID])
        shutil.copy(os.path.join(tmp, "a.pgen"), os.path.join(HERE, "pgen", "phased_small.pgen"))
        np.savez_compressed(os.path.join(HERE, "pgen", "phased_small.npz"), **out)
        # ---- partially phased (the reader must report phasepresent exactly; the command must refuse)
        m2, n2 = 120, 67
        raw2, pp2, pi2 = T.synth_phased(m2, n2, seed=7, missing_rate=0.05)
        rng = np.random.default_rng(3)
        drop = rng.random(raw2.shape) < 0.3
        drop[:40] = False     # first 40 variants stay fully phased
        drop[60:70] = True    # ten variants without any phase -> no phase track at all
        pp2 = (pp2.astype(bool) & ~drop).astype(np.uint8)
        pi2 = pi2 & pp2
        chroms2, bps2 = positions(m2, 1, 5)
        T.write_vcf(os.path.join(tmp, "b.vcf"), raw2, chroms2, bps2, pp2, pi2)
        T.ref_import_vcf(os.path.join(tmp, "b.vcf"), os.path.join(tmp, "b"))
        cp = T.run_ref(["--pfile", "b", "--indep-pairphase", "50", "5", "0.5", "--out", "b"], tmp)
        msg = [ln for ln in cp.stdout.splitlines() if ln.startswith("Error")]
        print("partial:", cp.returncode, msg)
        shutil.copy(os.path.join(tmp, "b.pgen"), os.path.join(HERE, "pgen", "phased_partial.pgen"))
        np.savez_compressed(os.path.join(HERE, "pgen", "phased_partial.npz"), raw=raw2, phasepresent=pp2, phaseinfo=pi2,
                            ref_returncode=np.int32(cp.returncode), ref_error=np.array(msg[0] if msg else ""))
        # ---- multiallelic + phased: the collapse with phase (Get1MP, pgenlib_read.cc:6962) incl. its reading of phaseinfo
        m3, n3 = 400, 120
        first, second, alt_ct = T.synth_multiallelic_haps(m3, n3, seed=5)
        chroms3 = ["1"] * 200 + ["2"] * 200
        bps3 = np.concatenate([1000 + 97 * np.arange(200)] * 2).astype(np.uint32)
        ids3 = T.write_vcf_haps(os.path.join(tmp, "c.vcf"), first, second, alt_ct, chroms3, bps3)
        T.ref_import_vcf(os.path.join(tmp, "c.vcf"), os.path.join(tmp, "c"))
        out3 = dict(first=first, second=second, alt_ct=alt_ct.astype(np.uint32), chroms=np.array([int(c) for c in chroms3], dtype=np.uint32), bps=bps3)
        for k, (win, r2, order) in enumerate(GRID):
            kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "c"), win, r2, order=order, mode="phase")
            out3["removed_%d" % k] = np.isin(np.array(ids3), np.array(removed))
            print("phased_multi", win, r2, order, [ln for ln in log.splitlines() if "variants removed" in ln][-1])
            # the unphased command on the same file: major-vs-rest collapse of multiallelic sites (Get1Multiallelic)
            kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "c"), win, r2, order=order)
            out3["removed_wise_%d" % k] = np.isin(np.array(ids3), np.array(removed))
        out3["grid"] = np.array(["%s|%r|%d" % (" ".join(w), r, o) for w, r, o in GRID])
        shutil.copy(os.path.join(tmp, "c.pgen"), os.path.join(HERE, "pgen", "phased_multi.pgen"))
        np.savez_compressed(os.path.join(HERE, "pgen", "phased_multi.npz"), **out3)
        # ---- multiallelic, partially phased: the explicit-phasepresent form of the phase track over multiallelic hets
        m4, n4 = 150, 83
        first4, second4, alt_ct4 = T.synth_multiallelic_haps(m4, n4, seed=9, max_alt=4, multi_rate=0.5)
        unph = (np.random.default_rng(10).random(first4.shape) < 0.35) & (first4 != second4) & (first4 >= 0)
        unph[:20] = False
        T.write_vcf_haps(os.path.join(tmp, "e.vcf"), first4, second4, alt_ct4, ["1"] * m4, 1 + 50 * np.arange(m4), unphased=unph)
        T.ref_import_vcf(os.path.join(tmp, "e.vcf"), os.path.join(tmp, "e"))
        shutil.copy(os.path.join(tmp, "e.pgen"), os.path.join(HERE, "pgen", "phased_multi_partial.pgen"))
        np.savez_compressed(os.path.join(HERE, "pgen", "phased_multi_partial.npz"), first=first4, second=second4, alt_ct=alt_ct4.astype(np.uint32), unphased=unph)
        # ---- chrX / chrY / MT next to autosomes, males / females / unknown sex, a few non-founders: both prune commands
        m5, n5 = 600, 110
        raw5, pp5, pi5 = T.synth_phased(m5, n5, seed=77, missing_rate=0.03)
        plan = [("1", 150), ("2", 100), ("X", 150), ("Y", 80), ("MT", 120)]
        chroms5, bps5 = [], []
        for name, cnt in plan:
            chroms5 += [name] * cnt
            bps5 += list((3000000 if name == "X" else 1000) + 173 * np.arange(cnt))
        rng5 = np.random.default_rng(5)
        sexes5 = rng5.choice([1, 2, 0], size=n5, p=[0.45, 0.45, 0.1])
        parents5 = [("s0", "s1") if (s % 13 == 5) else ("0", "0") for s in range(n5)]
        founders5 = np.array([p == ("0", "0") for p in parents5])
        while T.ref_pairphase_chrx_is_unreliable(sexes5, founders5):
            sexes5[np.flatnonzero(founders5 & (sexes5 != 1))[0]] = 1
        ids5 = T.write_pgen_phased(os.path.join(tmp, "s"), raw5, pi5, chroms5, np.array(bps5), sexes=sexes5, parents=parents5)
        out5 = dict(raw=raw5, phaseinfo=pi5, chroms=np.array(chroms5), bps=np.array(bps5, dtype=np.uint32), sexes=sexes5.astype(np.int8), founders=founders5)
        grid5 = [(["40kb"], 0.3, 2), (["70", "9"], 0.3, 1), (["60", "1"], 0.6, 2)]
        for mode in ("wise", "phase"):
            for k, (win, r2, order) in enumerate(grid5):
                kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "s"), win, r2, order=order, mode=mode, threads=2)
                out5["removed_%s_%d" % (mode, k)] = np.isin(np.array(ids5), np.array(removed))
                print("sexed", mode, win, r2, order, [ln for ln in log.splitlines() if "variants removed" in ln][-1])
        out5["grid"] = np.array(["%s|%r|%d" % (" ".join(w), r, o) for w, r, o in grid5])
        np.savez_compressed(os.path.join(HERE, "pgen", "sexed_phased.npz"), **out5)
        # ---- --indep-preferred (plink2_ld.cc:916-918: listed variants get major frequency - 1, so they win tie-breaks)
        m6, n6 = 500, 140
        raw6 = T.synth_raw_codes(m6, n6, seed=66, missing_rate=0.02)
        chroms6 = ["1"] * 300 + ["2"] * 200
        bps6 = np.concatenate([1000 + 120 * np.arange(300), 1000 + 120 * np.arange(200)]).astype(np.uint32)
        ids6 = T.write_pgen_fixed(os.path.join(tmp, "f"), raw6, chroms6, bps6)
        pref6 = np.random.default_rng(6).random(m6) < 0.3
        open(os.path.join(tmp, "pref.txt"), "w").write("\n".join(i for i, p in zip(ids6, pref6) if p) + "\n")
        out6 = dict(raw=raw6, chroms=np.array([int(c) for c in chroms6], dtype=np.uint32), bps=bps6, preferred=pref6)
        grid6 = [(["50", "5"], 0.2, 2), (["50", "5"], 0.2, 1), (["20kb"], 0.5, 2)]
        for k, (win, r2, order) in enumerate(grid6):
            kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "f"), win, r2, order=order, extra=["--indep-preferred", "pref.txt"])
            out6["removed_%d" % k] = np.isin(np.array(ids6), np.array(removed))
            kept0, removed0, _ = T.ref_indep_pairwise(os.path.join(tmp, "f"), win, r2, order=order)
            out6["removed_plain_%d" % k] = np.isin(np.array(ids6), np.array(removed0))
            print("preferred", win, r2, order, int(out6["removed_%d" % k].sum()), "vs plain", int(out6["removed_plain_%d" % k].sum()))
        out6["grid"] = np.array(["%s|%r|%d" % (" ".join(w), r, o) for w, r, o in grid6])
        np.savez_compressed(os.path.join(HERE, "pgen", "preferred.npz"), **out6)
        # ---- the windowed --r2-unphased table's pair set (UpdateVcorWindow, plink2_ld.cc:10984-11023) with no r^2 filter
        m7, n7 = 400, 90
        raw7 = T.synth_raw_codes(m7, n7, seed=88, missing_rate=0.02)
        raw7[17] = 0   # monomorphic: its pairs are NaN and never written
        rng7 = np.random.default_rng(8)
        chroms7 = ["1"] * 220 + ["5"] * 1 + ["9"] * 179
        bps7 = np.concatenate([np.sort(rng7.integers(1, 40000, 220)), [7], np.sort(rng7.integers(1, 3000000, 179))]).astype(np.uint32)
        ids7 = T.write_pgen_fixed(os.path.join(tmp, "w"), raw7, chroms7, bps7)
        out7 = dict(raw=raw7, chroms=np.array([int(c) for c in chroms7], dtype=np.uint32), bps=bps7)
        settings = [("5", None), ("0.4", "3"), ("1000", "12"), ("30", None)]
        for k, (kb, cnt) in enumerate(settings):
            args = ["--pfile", "w", "--r2-unphased", "--ld-window-kb", kb, "--ld-window-r2", "0", "--out", "w%d" % k] + (["--ld-window", cnt] if cnt else [])
            cp = T.run_ref(args, tmp)
            assert cp.returncode == 0, cp.stdout
            idx = {v: i for i, v in enumerate(ids7)}
            pairs = [(idx[t[2]], idx[t[5]]) for t in (ln.split("\t") for ln in open(os.path.join(tmp, "w%d.vcor" % k)).read().splitlines()[1:])]
            out7["pairs_%d" % k] = np.array(pairs, dtype=np.uint32).reshape(-1, 2)
            out7["r2_text_%d" % k] = np.array([ln.split("\t")[6] for ln in open(os.path.join(tmp, "w%d.vcor" % k)).read().splitlines()[1:]])
            print("vcor window", kb, cnt, len(pairs), "pairs")
        # inter-chr with the default --ld-window-r2 (0.2 * (1 - 2^-44)): all pairs, A-major, chromosome 0 would be kept
        cp = T.run_ref(["--pfile", "w", "--r2-unphased", "inter-chr", "--out", "wi"], tmp)
        assert cp.returncode == 0, cp.stdout
        lines = [ln.split("\t") for ln in open(os.path.join(tmp, "wi.vcor")).read().splitlines()[1:]]
        idx = {v: i for i, v in enumerate(ids7)}
        out7["inter_pairs"] = np.array([(idx[t[2]], idx[t[5]]) for t in lines], dtype=np.uint32).reshape(-1, 2)
        out7["inter_text"] = np.array([t[6] for t in lines])
        print("inter-chr default filter:", len(lines), "pairs")
        out7["settings"] = np.array(["%s|%s" % (kb, cnt or "") for kb, cnt in settings])
        np.savez_compressed(os.path.join(HERE, "pgen", "vcor_windows.npz"), **out7)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
