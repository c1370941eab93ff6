#!/usr/bin/env python3
"""tools/e2e_load_sweep.py -- the file -> HBM leg of plink2-hip on the chr22-sized fileset of bench.py (176,765 x 500,000, 22 GB fixed-width .pgen under /dev/shm),
swept over the copy pool's geometry and the H2D mode (MEASUREMENT build of the library: the shipped one reads no environment).  One line per setting on stdout."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def main():
    import torch
    import bench
    import __graft_entry__ as ge
    pkg = ge.load_package()
    pkg.build_library(measure=True)
    cfg = bench.CONFIGS["config3"]
    variants = int(sys.argv[1]) if len(sys.argv) > 1 else 176765
    import bench_support as support
    e2e = support.E2EChr22(pkg, torch, cfg, variants, bench.SEED, bench.genome_layout).start(reference=False, variable_width=False)
    libdir = tempfile.mkdtemp(prefix="ldp_measure_lib_")
    shutil.copy(pkg.MEASURE_LIB_PATH, os.path.join(libdir, "libldprune_hip.so"))
    cli = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
    kb = "%gkb" % cfg["window_kb"]
    settings = [{}]
    if os.environ.get("LDP_SWEEP") == "stage":   # round 6: the slot size of the pinned ring, alone and with more copy threads / larger tasks
        for mb in (16, 32, 64, 128):
            settings.append({"LDP_DEBUG_STAGE_MB": str(mb)})
        settings += [{"LDP_DEBUG_STAGE_MB": "64", "LDP_DEBUG_COPY_TASK_KB": "1024"}, {"LDP_DEBUG_STAGE_MB": "64", "LDP_DEBUG_COPY_THREADS": "64"},
                     {"LDP_DEBUG_STAGE_MB": "64", "LDP_DEBUG_COPY_THREADS": "64", "LDP_DEBUG_COPY_TASK_KB": "1024"}, {"LDP_DEBUG_STAGE_MB": "64", "LDP_DEBUG_H2D_MODE": "2"}, {}]
        for st in settings:
            st["LDP_DEBUG_LOAD_TIMING"] = "1"
    for th in (() if os.environ.get("LDP_SWEEP") == "stage" else (16, 48, 64, 96)):
        settings.append({"LDP_DEBUG_COPY_THREADS": str(th)})
    if os.environ.get("LDP_SWEEP") != "stage":
        for kbs in (128, 1024, 4096):
            settings.append({"LDP_DEBUG_COPY_TASK_KB": str(kbs)})
        settings += [{"LDP_DEBUG_COPY_THREADS": "64", "LDP_DEBUG_COPY_TASK_KB": "128"}, {"LDP_DEBUG_H2D_MODE": "0"}, {"LDP_DEBUG_H2D_MODE": "2"}, {"LDP_DEBUG_COPY_POOL": "0"}, {}]
    try:
        for st in settings:
            env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""), **st)
            best = None
            for _ in range(2):
                cc = subprocess.run([cli, "--pfile", "g", "--indep-pairwise", kb, repr(cfg["r2"]), "--timing", "--out", "hip"], cwd=e2e.tmp, env=env,
                                    stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
                ph = re.search(r"genotype load[^|]*?([0-9.]+) s \|", cc.stdout)
                tot = re.search(r"\[timing\] total ([0-9.]+) s", cc.stdout)
                if cc.returncode == 0 and ph:
                    lt = [ln for ln in cc.stdout.splitlines() if "load timing" in ln or "waited" in ln][:3]
                    v = (float(ph.group(1)), float(tot.group(1)) if tot else None, lt)
                    best = v if (best is None or v[0] < best[0]) else best
            print(json.dumps({"setting": st, "file_to_hbm_s": best[0] if best else None, "gbs": (e2e.file_bytes / best[0] / 1e9) if best else None, "main_total_s": best[1] if best else None,
                              "load_timing": best[2] if best else None}), flush=True)
    finally:
        subprocess.call(["rm", "-rf", e2e.tmp, libdir])


if __name__ == "__main__":
    main()
