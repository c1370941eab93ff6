#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE binary
(oracle/_ref/plink2, built from /root/reference by `make -C oracle ref`) in this container.

The reference's own test-suite holds no --indep-pairwise / --r2 vectors, so
these files are the pin: inputs (packed 2-bit REF-based codes, positions) + the reference's
.prune.out set (and, for the KATs, its --r2-unphased doubles).  Re-run:  python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ldtools as T  # noqa: E402


def run_case(name, raw, chroms, bps, window_args, r2, order, extra_checks=None, r2_matrix=False):
    tmp = tempfile.mkdtemp(prefix="golden_")
    try:
        prefix = os.path.join(tmp, name)
        ids = T.write_pgen_fixed(prefix, raw, chroms, bps)
        kept, removed, log = T.ref_indep_pairwise(prefix, window_args, r2, order=order)
        removed_mask = np.isin(np.array(ids), np.array(removed))
        # .bed input must give the same answer
        T.write_bed(prefix + "_bed", raw, chroms, bps)
        kept_b, removed_b, _ = T.ref_indep_pairwise(prefix + "_bed", window_args, r2, order=order, fmt="bfile")
        assert removed_b == removed, "reference disagrees between .pgen and .bed input"
        summary = [ln for ln in log.splitlines() if "variants removed" in ln]
        out = dict(raw_packed=T.pack_2bit(raw).view(np.uint8), m=raw.shape[0], n=raw.shape[1],
                   chroms=np.array([int(c) for c in chroms], dtype=np.uint32), bps=np.asarray(bps, dtype=np.uint32),
                   window_args=np.array([str(a) for a in window_args]), r2=np.float64(r2), order=np.int32(order),
                   removed=removed_mask, ref_summary=np.array(summary[-1] if summary else ""))
        if r2_matrix:
            cp = T.run_ref(["--pfile", name, "--r2-unphased", "square", "bin", "--bad-ld", "--out", name + ".r2"], tmp)
            assert cp.returncode == 0, cp.stdout
            mat = np.fromfile(os.path.join(tmp, name + ".r2.unphased.vcor2.bin"), dtype=np.float64)
            out["r2_square"] = mat.reshape(raw.shape[0], raw.shape[0])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("%-28s %s" % (name, out["ref_summary"]))
    finally:
        shutil.rmtree(tmp)


def kat_quirk():
    # 100 founders, 3 SNPs; hom-minor carriers A = 0-19, B = 0-29, C = 0-39
    raw = np.zeros((3, 100), dtype=np.uint8)
    raw[0, :20] = 2
    raw[1, :30] = 2
    raw[2, :40] = 2
    return raw


def kat_missing():
    raw = kat_quirk()
    raw[0, 40:50] = 1
    raw[1, 45:60] = 1
    raw[2, 60:70] = 1
    raw[0, 90:95] = 3
    raw[1, 85:93] = 3
    return raw


def positions(m, nchr, seed, spacing=300):
    rng = np.random.default_rng(seed)
    per = (m + nchr - 1) // nchr
    chr_idx = np.arange(m) // per
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(nchr):
        idx = np.where(chr_idx == c)[0]
        gaps = rng.integers(1, 2 * spacing, size=len(idx))
        gaps = np.where(rng.random(len(idx)) < 0.01, gaps + 400000, gaps)
        bps[idx] = 1000 + np.cumsum(gaps)
    return [str(c + 1) for c in chr_idx], bps


def toy():
    """config 1: 1.9/toy.ped -> .bed -> codes (2 samples x 2 variants)."""
    tmp = tempfile.mkdtemp(prefix="golden_toy_")
    try:
        for ext in ("ped", "map"):
            shutil.copy("/root/reference/1.9/toy." + ext, os.path.join(tmp, "toy." + ext))
        cp = T.run_ref(["--pedmap", "toy", "--make-bed", "--out", "toyb"], tmp)
        assert cp.returncode == 0, cp.stdout
        bed = np.fromfile(os.path.join(tmp, "toyb.bed"), dtype=np.uint8)[3:]
        bim = [ln.split() for ln in open(os.path.join(tmp, "toyb.bim"))]
        n = len(open(os.path.join(tmp, "toyb.fam")).readlines())
        m = len(bim)
        rec = (n + 3) // 4
        lut = np.array([2, 3, 1, 0], dtype=np.uint8)  # bed code -> pgen code
        raw = np.zeros((m, n), dtype=np.uint8)
        for v in range(m):
            for s in range(n):
                raw[v, s] = lut[(bed[v * rec + s // 4] >> (2 * (s % 4))) & 3]
        chroms = [b[0] for b in bim]
        bps = np.array([int(b[3]) for b in bim], dtype=np.uint32)
        return raw, chroms, bps
    finally:
        shutil.rmtree(tmp)


def main():
    if not T.have_ref():
        sys.exit("reference binary missing: make -C oracle ref")
    raw, chroms, bps = toy()
    run_case("toy_50_5_0.2", raw, chroms, bps, ["50", "5"], 0.2, 2)
    three = (["1"] * 3, np.array([1000, 1001, 1002], dtype=np.uint32))
    run_case("kat_quirk_o2", kat_quirk(), three[0], three[1], ["50", "1"], 0.5, 2, r2_matrix=True)
    run_case("kat_quirk_o1", kat_quirk(), three[0], three[1], ["50", "1"], 0.5, 1)
    run_case("kat_missing", kat_missing(), three[0], three[1], ["50", "1"], 0.5, 2, r2_matrix=True)
    cases = [
        ("rand_count_50_5_o2", 600, 100, 1, ["50", "5"], 0.2, 2, 0.0),
        ("rand_count_50_5_o1", 600, 100, 2, ["50", "5"], 0.2, 1, 0.0),
        ("rand_count_50_miss", 600, 133, 3, ["50"], 0.5, 2, 0.05),
        ("rand_kb_20_o2_miss", 600, 133, 4, ["20kb"], 0.2, 2, 0.05),
        ("rand_kb_20_o1_miss", 600, 133, 5, ["20kb"], 0.5, 1, 0.02),
        ("rand_kb_30_r2_0.1", 1500, 257, 6, ["30kb"], 0.1, 2, 0.01),
        ("rand_count_200_25_o1", 1500, 64, 7, ["200", "25"], 0.5, 1, 0.1),
        ("rand_count_200_25_o2", 1500, 63, 8, ["200", "25"], 0.5, 2, 0.1),
        ("rand_kb_wide_nomiss", 1200, 1025, 9, ["60kb"], 0.2, 2, 0.0),
    ]
    for name, m, n, seed, wa, r2, order, miss in cases:
        raw = T.synth_raw_codes(m, n, seed, missing_rate=miss)
        chroms, bps = positions(m, 3, seed + 1)
        run_case(name, raw, chroms, bps, wa, r2, order)


if __name__ == "__main__":
    main()
