#!/usr/bin/env python3
"""tools/e2e_chr22_missing.py -- the chr22-sized end-to-end comparison of bench.py (176,765 variants x 500,000 samples of the bench generator,
`--indep-pairwise 500kb 0.2`, reference plink2 against plink2-hip, every output file compared byte for byte) on a fileset with MISSING CALLS in
every variant: the parity statement for the kernels a fileset with missing calls takes (0.1 %: the SPARSE instantiation of the 8 x 8 tiles;
5 %: the quarter tiles) at the metric's sample count and a whole chromosome's worth of candidate pairs.  Not part of the default bench run: the
reference needs several minutes on rows with missing calls.  One JSON line.

  --missing-rate 0.001    --variants 0 (= the chr22-sized share)    --ref-timeout 1500
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tools"))
import bench_support as support  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--missing-rate", type=float, default=0.001)
    ap.add_argument("--variants", type=int, default=0)
    ap.add_argument("--ref-timeout", type=int, default=1500)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    cfg = dict(bench.CONFIGS["config3"])
    m = args.variants or int(round(cfg["variants"] * bench.CHR22_FRACTION))
    e2e = support.E2EChr22(pkg, torch, cfg, m, bench.SEED, bench.genome_layout, ref_timeout_s=args.ref_timeout, missing_rate=args.missing_rate)
    e2e.start(reference=True, variable_width=False)
    res = e2e.finish()
    res["missing_rate"] = args.missing_rate
    hip = res.get("plink2_hip") or {}
    res["route_line"] = [ln for ln in hip.get("timing_lines", []) if "pair launches by route" in ln]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
