#!/usr/bin/env python3
"""The config-5 density slice of bench.py (500,000 samples x --variants at 290 bp, `500kb 0.2`, 5 % missing calls) stepped with the
four-product form on the tile plan's quarter tiles (pair_mfma_tile4_kernel, the default) and on the parallelogram plan
(`pair_four_tiles` 0): ms per step, the pair kernels' share, the prune sets compared.  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=120000)
    ap.add_argument("--missing-rate", type=float, default=0.05)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--forms", default="tiles,plan")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    cfg = dict(bench.CONFIGS["config3"], variants=args.variants)
    out = {"samples": cfg["samples"], "variants": cfg["variants"], "missing_rate": args.missing_rate, "window_kb": cfg["window_kb"], "r2": cfg["r2"]}
    sets = {}
    for form in args.forms.split(","):
        w = bench.Workload(pkg, torch, cfg, args.missing_rate, 0, 1, 0, {"pair_four_tiles": 1 if form == "tiles" else 0}, None)
        w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctrs = []
        for _ in range(args.steps):
            words, cc = w.step()
            ctrs.append(bench.sum_counters(cc))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        c = ctrs[-1]
        sets[form] = np.array(words, copy=True)
        out[form] = {"ms_per_step": 1e3 * el / args.steps, "pair_kernels_ms": c["ms_pair_kernel"], "missing_call_kernels_ms": c["ms_pair_mfma_general"],
                     "four_tile_launches": c["four_tile_launches"], "wide_tiles": c["wide_tiles"], "pred_true": c["pred_true"],
                     "skipped_frac": c["mfma_skipped_product_stages"] / max(1, c["mfma_product_stages"]), "pairs_counted_exactly": c["sparse_exact_pairs"]}
        w.close()
        torch.cuda.empty_cache()
    forms = list(sets)
    if len(forms) == 2:
        out["prune_sets_identical"] = bool(np.array_equal(sets[forms[0]], sets[forms[1]]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
