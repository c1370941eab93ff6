#!/usr/bin/env python3
"""Golden vectors for the .vcor number formatting: (IEEE-754 bits of an r^2 double, the text the reference's dtoa_g
writes for it).  The doubles come from the CPU oracle (bit-identical to the reference's ComputeR2 by
tests/golden/*.npz), the text from the reference binary's own `--r2-unphased --ld-window-r2 0` table.
Run from the repo root in a container that has /root/reference built (oracle/_ref/plink2):
    python tests/golden/make_golden_vcor.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ldtools as T  # noqa: E402


def main():
    m, n = 420, 97
    raw = T.synth_raw_codes(m, n, seed=4242, missing_rate=0.03)
    raw[5] = 0
    chroms = ["1"] * 300 + ["2"] * 120
    pos = np.concatenate([np.arange(300) * 37 + 11, np.arange(120) * 53 + 7])
    inv, mf, _ = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    bits, texts = [], []
    with tempfile.TemporaryDirectory() as tmp:
        T.write_pgen_fixed(os.path.join(tmp, "d"), raw, chroms, pos)
        ref = T.run_ref(["--pfile", "d", "--r2-unphased", "--ld-window-kb", "3", "--ld-window-r2", "0", "--out", "ref"], tmp)
        assert ref.returncode == 0, ref.stdout
        for line in open(os.path.join(tmp, "ref.vcor")):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            i, j = int(f[2][3:]), int(f[5][3:])
            st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
            cov, v1, v2 = T.oracle_r2(st)
            r2 = np.float64(cov) * np.float64(cov) / (np.float64(v1) * np.float64(v2))
            bits.append(np.float64(r2).view(np.uint64))
            texts.append(f[6])
    # a few hand-picked values outside what r^2 reaches, text from the same formatter through --freq-less paths is not
    # available, so only r^2-range values are pinned here
    np.savez_compressed(os.path.join(HERE, "pgen", "vcor_format_g6.npz"), bits=np.array(bits, dtype=np.uint64), texts=np.array(texts))
    print(len(bits), "values;", len(set(texts)), "distinct strings; exponent forms:", sum("e" in t for t in texts))


if __name__ == "__main__":
    main()
