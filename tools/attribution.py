#!/usr/bin/env python3
"""tools/attribution.py -- where the time of pair_mfma_wide_kernel goes (run on the GPU box; MEASUREMENT build of the library).

    LDP_LIB_MEASURE=1 python tools/attribution.py [--variants 120000] [--steps 40] [--ablations 0,32,1,7,8,9,15,16,25] [--share]

For each ablation of the kernel (csrc/ldp_pair_wide.hip, -DLDP_MEASURE only; bits 0-3 give wrong results by construction) and for
early termination on / off: kernel ms per step (HIP events of the engine), the shader clock INSIDE the kernel (clock64 / wall_clock64
summed over every wave), the socket power and clock rocm-smi reports while the steps run, and -- ablation 32 -- the cycles the waves
spent in each phase.  One JSON object per line on stdout; profiles/r05_experiments.md is written from them.
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import time

os.environ["LDP_LIB_MEASURE"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=120000)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--ablations", default="0,32,1,7,8,9,15,16,25")
    ap.add_argument("--modes", default="exhaustive,early")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--probe", action="store_true", help="instead: tools/_bin/energy_probe, one configuration at a time for a few seconds each, with the same power / clock sampling")
    args = ap.parse_args()

    import torch
    import bench
    import __graft_entry__ as ge
    pkg = ge.load_package()
    pkg.build_library(measure=True)   # lib/libldprune_hip_measure.so (-DLDP_MEASURE): a no-op when it is newer than the sources
    L = pkg.lib()
    L.ldp_measure_wide_counters.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    torch.cuda.set_device(0)
    smi = bench.support.SmiSampler()   # (tools/bench_support.py since round 6)

    def smi_summary(tag):
        q = smi.summary(tag)
        return {"smi_samples": int(re.search(r"(\d+) samples", q["source"]).group(1)), "smi_power_w_median": q["socket_power_w_median"], "smi_power_w_max": q["socket_power_w_max"],
                "smi_sclk_mhz_median": q["shader_clock_mhz_median"], "smi_sclk_mhz_min": q["shader_clock_mhz_min"], "smi_power_cap_w": q["socket_power_cap_w"]}
    if args.probe:
        # the instruction alone, by operand data (0 zero, 1 random nibbles, 2 genotypes +-2 coded, 3 genotypes as allele counts) and VALU beside it
        exe = os.path.join(REPO, "tools", "_bin", "energy_probe")
        for data, k in ((3, 0), (3, 3), (3, 6), (2, 0), (0, 0), (1, 0)):
            tag = "probe/%d/%d" % (data, k)
            smi.window = tag
            out = subprocess.run([exe, "long", str(data), str(k), "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120).stdout
            smi.window = None
            try:
                rec = json.loads(out.strip().splitlines()[-1])
            except Exception:
                rec = {"probe_output": out[-300:]}
            rec.update(smi_summary(tag))
            print(json.dumps(rec), flush=True)
        smi.stop()
        return
    cfg = dict(bench.CONFIGS["config3"], variants=args.variants)
    extra = {}
    for kv in args.option:
        k, v = kv.split("=", 1)
        extra[k] = float(v)
    for mode in args.modes.split(","):
        opts = dict(extra)
        if mode == "exhaustive":
            opts["early_exit"] = 0
        wl = bench.Workload(pkg, torch, cfg, 0.0, 0, 1, 0, opts)
        for abl in [int(a) for a in args.ablations.split(",")]:
            os.environ["LDP_DEBUG_WIDE_ABLATE"] = str(abl)
            wl.step()
            wl.step()
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 16)()
            assert L.ldp_measure_wide_counters(buf, 1) == 0
            tag = "%s/%d" % (mode, abl)
            smi.window = tag
            t0 = time.perf_counter()
            ks = []
            for _ in range(args.steps):
                _, ctrs = wl.step()
                ks.append(bench.sum_counters(ctrs))
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            smi.window = None
            assert L.ldp_measure_wide_counters(buf, 1) == 0
            m = [int(x) for x in buf]
            c = ks[-1]
            kms = sum(k["ms_pair_mfma"] for k in ks) / len(ks)
            executed = max(c["mfma_product_stages"] - c["mfma_skipped_product_stages"], 0) + c["mfma_extra_product_stages"]
            rec = {"mode": mode, "ablate": abl, "variants": args.variants, "steps": args.steps, "kernel_ms": kms, "step_ms": 1e3 * wall / args.steps,
                   "mfma_instructions": executed, "pflops": executed * 131072.0 / (kms * 1e-3) / 1e15 if kms > 0 else None,
                   "waves": m[7], "in_kernel_mhz": (100.0 * m[6] / m[8]) if m[8] else None,
                   "wave_cycles_mean": (m[6] / m[7]) if m[7] else None, "options": opts, "pred_true": c["pred_true"]}
            if abl & 32 and m[7]:
                tot = float(m[6])
                rec["phases_frac_of_wave_cycles"] = {"wait_vmcnt": m[0] / tot, "barrier": m[1] / tot, "stage_live": m[2] / tot, "stage_dead": m[3] / tot,
                                                     "checkpoint": m[4] / tot, "epilogue": m[5] / tot,
                                                     "other": 1.0 - sum(m[:6]) / tot}
                rec["cycles_per_live_stage_visit"] = (m[2] / m[9]) if m[9] else None
                rec["cycles_per_dead_stage_visit"] = (m[3] / m[10]) if m[10] else None
                rec["wait_plus_barrier_cycles_per_stage_visit"] = ((m[0] + m[1]) / (m[9] + m[10])) if (m[9] + m[10]) else None
                rec["stage_visits_live_dead"] = [m[9], m[10]]
                if m[11] or m[12] or m[13]:   # the barrier-free kernel: polls at the top of a stage, deferred DMA issues, s_waitcnt vmcnt of the confirmations
                    rec["async_frac_of_wave_cycles"] = {"top_poll_and_issue": m[11] / tot, "deferred_issue_poll": m[12] / tot, "confirm_vmcnt": m[13] / tot}
            rec.update(smi_summary(tag))
            print(json.dumps(rec), flush=True)
        os.environ.pop("LDP_DEBUG_WIDE_ABLATE", None)
        wl.close()
    smi.stop()


if __name__ == "__main__":
    main()
