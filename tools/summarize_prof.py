#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs tools/profile.sh collected into the small files kept under profiles/:
  <tag>_kernel_stats.csv   per-kernel calls / total / average / min / max duration (from kernel_trace)
  <tag>_pmc.json           per-kernel mean counter values (one dispatch = one launch)
  <tag>_pmc_traffic.json   HBM bytes per step of the pair kernels, read by bench.py as roofline.traffic
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced streaming read, so the read side is doubled."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    # ---- kernel trace
    rows = []
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    dur = collections.defaultdict(list)
    for r in rows:
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in dur.values()) or 1
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.4f,%.4f,%.4f,%.4f,%.2f\n' % (k, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6,
                                                       100.0 * sum(v) / total))
    # ---- counters
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in pmc.items() if "ldp::" in k}
    n_disp = {k: max(len(v) for v in cs.values()) for k, cs in pmc.items() if "ldp::" in k}  # PMC passes run ONE bench step
    json.dump(out, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    # the workload: from the bench line the traced run printed (trace.log), so that the file can never be labelled with another
    # shape than the one that ran; the kernel sources it ran on: git blob hashes of this snapshot (bench.py replays the traffic only
    # while they equal the tree's)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    line = None
    for ln in open(os.path.join(src, "trace.log"), errors="ignore"):
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    cfgj = (line or {}).get("config", {})
    samples = int(cfgj.get("samples", os.environ.get("LDP_PROF_SAMPLES", "0")))
    variants = int(cfgj.get("variants_rank0", os.environ.get("LDP_PROF_VARIANTS", "0")))
    window_kb = float(cfgj.get("window_kb", os.environ.get("LDP_PROF_WINDOW_KB", "0")))
    m = None
    if line:
        import re
        m = re.search(r"missing rate ([0-9.e+-]+)", cfgj.get("workload", ""))
    missing_rate = float(m.group(1).rstrip(",")) if m else float(os.environ.get("LDP_PROF_MISSING", "0"))
    # every pair kernel that did work in the step (a launch can carry the wide-band tiles AND the parallelogram workgroups)
    per_kernel, total_b, total_read, total_write = {}, 0.0, 0.0, 0.0
    for k, cs in out.items():
        if (("pair_mfma" in k) or ("pair_tiles_kernel" in k)) and "FETCH_SIZE" in cs:
            read_b = cs["FETCH_SIZE"] * 1024 * 2 * n_disp[k]
            write_b = cs.get("WRITE_SIZE", 0.0) * 1024 * n_disp[k]
            short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ldp::", "")
            per_kernel[short] = {"launches_per_step": n_disp[k], "hbm_read_bytes_per_step": read_b, "hbm_write_bytes_per_step": write_b,
                                 "l2_hit_rate": (cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])) if cs.get("TCC_HIT_sum") is not None and (cs.get("TCC_HIT_sum", 0) + cs.get("TCC_MISS_sum", 0)) > 0 else None}
            total_b += read_b + write_b
            total_read += read_b
            total_write += write_b
    if per_kernel:
        main_kernel = max(per_kernel, key=lambda k: per_kernel[k]["hbm_read_bytes_per_step"])
        rows_bytes = ((samples + 511) // 512) * 128
        json.dump({"kernel": main_kernel, "samples": samples, "variants": variants, "window_kb": window_kb, "missing_rate": missing_rate,
                   "hbm_bytes_per_step": total_b, "hbm_read_bytes_per_step": total_read, "hbm_write_bytes_per_step": total_write,
                   "compulsory_bytes_per_step": float(variants) * rows_bytes, "traffic_over_compulsory": total_b / (float(variants) * rows_bytes),
                   "pair_kernels": per_kernel,
                   "note": "sum over the pair kernels of the step of FETCH_SIZE x 1024 x 2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE x 1024",
                   "tag": tag, "sources": bench.source_hashes(),
                   "bench_line_of_the_traced_run": {k: (line or {}).get(k) for k in ("value", "ms_per_step", "steps")}}, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)
    print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read())


if __name__ == "__main__":
    main()
