#!/usr/bin/env python3
"""Golden vectors for the dosage-track reader (ldp_pgen_dosage_sums) and for --indep-pairwise on a file with dosages, from the
reference binary (oracle/_ref/plink2; run where /root/reference was compiled):
  pgen/dosage_small.pgen   the reference's own `--dummy 60 200 dosage-freq=0.4` (dosage lists and bit arrays, variable-width)
  pgen/dosage_small.npz    its --freq (ALT_FREQS as printed, OBS_CT), its --indep-pairwise 50 5 0.3 / 30kb 0.45 lists (order 2 and 1)
    python tests/golden/make_golden_dosage.py"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ldtools as T  # noqa: E402


def main():
    assert T.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    tmp = tempfile.mkdtemp(prefix="golddos_")
    try:
        n, m = 60, 200
        cp = T.run_ref(["--dummy", str(n), str(m), "dosage-freq=0.4", "--seed", "11", "--threads", "1", "--make-pgen", "--out", "d"], tmp)
        assert cp.returncode == 0, cp.stdout
        cp = T.run_ref(["--pfile", "d", "--freq", "--out", "f"], tmp)
        assert cp.returncode == 0, cp.stdout
        rows = [ln.split() for ln in open(os.path.join(tmp, "f.afreq")) if not ln.startswith("#")]
        out = {"n": n, "m": m, "alt_freq_text": np.array([r[4] for r in rows]), "obs_ct": np.array([int(r[5]) for r in rows], dtype=np.int64)}
        for tag, wargs, r2, order in (("count_o2", ["50", "5"], 0.3, 2), ("kb_o1", ["30kb"], 0.45, 1)):
            kept, removed, log = T.ref_indep_pairwise(os.path.join(tmp, "d"), wargs, r2, order=order, threads=2)
            mask = np.ones(m, dtype=bool)
            mask[[int(x[3:]) for x in kept]] = False
            out["removed_" + tag] = mask
            print(tag, [ln for ln in log.splitlines() if "variants removed" in ln][-1])
        shutil.copy(os.path.join(tmp, "d.pgen"), os.path.join(HERE, "pgen", "dosage_small.pgen"))
        np.savez_compressed(os.path.join(HERE, "pgen", "dosage_small.npz"), **out)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
