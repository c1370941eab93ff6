#!/usr/bin/env python3
"""`plink2-hip --gpus N` feeding its N engines concurrently (one feeding thread per engine, bound next to its device, copy threads and pinned
ring of its own; the reference's main thread fills EVERY worker's slot of a batch, plink2_ld.cc:1292-1417) against the one-after-the-other
feed of rounds 2-5 (`--debug-serial-feed`) and against one engine, on a fixed-width fileset of the metric's sample count written by the device
generator.  On a one-GPU box the N engines share the device (`--debug-alias-devices`): the engines' load INTERVALS overlap -- that is what
this shows there --, their rows still cross one PCIe link.  Prints the [timing] lines of every run and one JSON line."""
import argparse
import json
import os
import shutil
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import bench  # noqa: E402
import bench_support as support  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=88000)
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--gpus", type=int, default=4)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    cfg = dict(bench.CONFIGS["config3"], samples=args.samples)
    where = support._scratch_dir(args.variants * ((args.samples + 3) // 4) * 1.1 + 1e9)
    tmp = tempfile.mkdtemp(prefix="ldp_nengine_", dir=where)
    out = {"samples": args.samples, "variants": args.variants, "engines": args.gpus, "devices": torch.cuda.device_count()}
    try:
        chr_idx, bps = bench.genome_layout(args.variants, 1, cfg["spacing"])
        nbytes = support.write_fixed_width_fileset(pkg, torch, tmp, args.samples, args.variants, bench.SEED, chr_idx, bps)
        out["pgen_bytes"] = nbytes
        alias = ["--debug-alias-devices"] if torch.cuda.device_count() < args.gpus else []
        kb = "%gkb" % cfg["window_kb"]
        runs = (("one_engine", 1, []), ("concurrent_feed", args.gpus, alias), ("serial_feed", args.gpus, alias + ["--debug-serial-feed"]))
        for name, gpus, extra in runs:
            r = support.run_plink2_hip(tmp, "g", kb, cfg["r2"], name, gpus=gpus, extra=extra)
            print("==== %s (--gpus %d %s): wall %s s, rc %s" % (name, gpus, " ".join(extra), r["wall_s"], r["rc"]))
            for ln in r["timing_lines"]:
                print("  " + ln)
            out[name] = {"wall_s": r["wall_s"], "wall_s_runs": r["wall_s_runs"], "rc": r["rc"], "file_to_hbm_s": (r["phases"] or {}).get("file_to_hbm_s"),
                         "file_to_hbm_gbs": (nbytes / r["phases"]["file_to_hbm_s"] / 1e9) if (r["phases"] and r["phases"]["file_to_hbm_s"]) else None}
        same = lambda a, b: all(open(os.path.join(tmp, a + e), "rb").read() == open(os.path.join(tmp, b + e), "rb").read() for e in (".prune.in", ".prune.out"))
        out["files_identical"] = bool(same("one_engine", "concurrent_feed") and same("one_engine", "serial_feed"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
