#!/usr/bin/env python3
"""Print the share / slice records a two-library A / B batch (tools/r06_batch_r.sh) left in gpurun_out/: ab_summary.py <tag> [rounds]"""
import json
import sys

tag = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for rnd in range(1, rounds + 1):
    for which in ("new", "base"):
        try:
            d = json.loads(open("gpurun_out/r06_%s_share_%s_%d.json" % (tag, which, rnd)).read().strip().splitlines()[-1])
            print("share", which, rnd, "ms/step %.2f kernels %.2f frac %.4f removed %s power %s clock %s beyond %.4f" % (
                d["ms_per_step"], d["stage_ms"]["pair_kernels"], d["roofline"]["frac"], d["config"]["variants_removed"], d.get("power_and_clock", {}).get("socket_power_w_median"),
                d.get("power_and_clock", {}).get("shader_clock_mhz_median"), d["roofline"]["mfma"]["computed_beyond_plan_frac"]))
        except Exception as ex:
            print("share", which, rnd, "??", ex)
        try:
            d = json.load(open("gpurun_out/r06_%s_slice_%s_%d.json" % (tag, which, rnd)))
            for rate, v in d["rates"].items():
                for name, x in v.items():
                    if isinstance(x, dict) and "pair_kernels_ms" in x:
                        print("  slice", which, rnd, rate, name, "pair %.2f ms executed %.1f TFLOP/s skipped %.3f removed %s" % (
                            x["pair_kernels_ms"], x["mfma_executed_tflops"], x["early_termination_skipped_frac"], x["variants_removed"]))
        except Exception as ex:
            print("  slice", which, rnd, "??", ex)
