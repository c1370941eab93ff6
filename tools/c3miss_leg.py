#!/usr/bin/env python3
"""Config 3's density slice of bench.py (500,000 samples x --variants at 290 bp, `500kb 0.2`) stepped at a list of missing-call
rates, once per option set: which kernel the device-side route picked, ms per step, the pair kernels' share, early termination,
pairs counted exactly, the prune sets of the option sets compared rate by rate.  One JSON line.

  --rates 0,0.001,0.01        missing-call rates (MCAR, every variant)
  --sets name:opt=v,opt=v;..  engine option sets (ldp_debug_set_option), e.g. "tiles:;plan:wide_sparse=0"
"""
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
    ap.add_argument("--config", default="config3", help="bench.CONFIGS entry the slice is cut from (config2: its 1,000,000 variants unless --variants says otherwise)")
    ap.add_argument("--variants", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--rates", default="0,0.001,0.01")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--sets", default="default:")
    ap.add_argument("--maf-min", type=int, default=0, help="floor on the generator's minor-allele frequency in percent (seed bits 56-62; 0 = the bench generator's 1 %%)")
    args = ap.parse_args()
    if args.maf_min:
        bench.SEED = bench.SEED | (args.maf_min << 56)
    import torch
    pkg = ge.load_package()
    cfg = dict(bench.CONFIGS[args.config])
    cfg["variants"] = args.variants or (120000 if args.config != "config2" else cfg["variants"])
    if args.samples:
        cfg["samples"] = args.samples
    sets = []
    for spec in args.sets.split(";"):
        name, _, opts = spec.partition(":")
        sets.append((name, {k: float(v) for k, v in (kv.split("=") for kv in opts.split(",") if kv)}))
    out = {"samples": cfg["samples"], "variants": cfg["variants"], "window_kb": cfg["window_kb"], "r2": cfg["r2"], "maf_min_percent": args.maf_min or 1, "rates": {}}
    for rate in [float(x) for x in args.rates.split(",")]:
        res, words_of = {}, {}
        for name, opts in sets:
            w = bench.Workload(pkg, torch, cfg, rate, 0, 1, 0, opts, None)
            w.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctrs = []
            for _ in range(args.steps):
                words, cc = w.step()
                ctrs.append(bench.sum_counters(cc))
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            c = {k: float(np.mean([q[k] for q in ctrs])) for k in ctrs[-1]}
            r = bench.pair_roofline(c, cfg["samples"], w.local_ct, rate, cfg["variants"], cfg["window_kb"], opts)
            words_of[name] = np.array(words, copy=True)
            res[name] = {"options": opts, "ms_per_step": 1e3 * el / args.steps, "count_pass_ms": c["ms_prepare"], "pair_kernels_ms": c["ms_pair_kernel"],
                         "complete_or_sparse_kernels_ms": c["ms_pair_mfma"], "missing_call_kernels_ms": c["ms_pair_mfma_general"], "kernel": r["kernel"],
                         "routes": r["routes"], "sparse_tile_launches": int(ctrs[-1].get("sparse_tile_launches", 0)), "four_tile_launches": int(c["four_tile_launches"]),
                         "pairs_counted_exactly": int(c["sparse_exact_pairs"]), "candidate_pairs": int(c["candidate_pairs"]), "pred_true": int(c["pred_true"]),
                         "early_termination_skipped_frac": r["mfma"]["early_termination_skipped_frac"], "mfma_executed_tflops": r["mfma"]["executed_tflops"],
                         "variants_removed": int(np.unpackbits(words_of[name].view(np.uint8)).sum())}
            w.close()
            torch.cuda.empty_cache()
        names = list(words_of)
        res["prune_sets_identical"] = bool(all(np.array_equal(words_of[names[0]], words_of[n]) for n in names[1:]))
        out["rates"]["%g" % rate] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
