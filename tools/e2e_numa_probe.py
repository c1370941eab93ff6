#!/usr/bin/env python3
"""tools/e2e_numa_probe.py -- does it matter on which socket plink2-hip's threads (copy pool, pinned staging: first touch) run?  The chr22-sized fileset of bench.py,
the shipped binary as it is (it binds itself to the device's NUMA node: p2h_util.cpp, bind_near_device), with --debug-no-bind, and with --debug-no-bind under taskset on
the other node."""
import json, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch, bench
    import __graft_entry__ as ge
    pkg = ge.load_package()
    cfg = bench.CONFIGS["config3"]
    e2e = bench.E2EChr22(pkg, torch, cfg, 176765).start()
    if e2e.ref_proc:
        e2e.ref_proc.kill(); e2e.ref_proc.communicate()
    node = None
    for d in sorted(os.listdir("/sys/class/drm")):
        try:
            if open("/sys/class/drm/%s/device/vendor" % d).read().strip() == "0x1002":
                node = int(open("/sys/class/drm/%s/device/numa_node" % d).read())
                break
        except Exception:
            continue
    cpus = {k: open("/sys/devices/system/node/node%d/cpulist" % k).read().strip() for k in (0, 1)}
    cli = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
    kb = "%gkb" % cfg["window_kb"]
    try:
        for tag, prefix, extra in (("default (bound to the device's node)", [], []), ("--debug-no-bind", [], ["--debug-no-bind"]), ("taskset other node + --debug-no-bind", ["taskset", "-c", cpus[1 - node]], ["--debug-no-bind"]),
                                   ("default, again", [], []), ("--debug-no-bind, again", [], ["--debug-no-bind"])):
            runs = []
            for _ in range(3):
                import time
                t0 = time.perf_counter()
                cc = subprocess.run(prefix + [cli, "--pfile", "g", "--indep-pairwise", kb, repr(cfg["r2"]), "--timing", "--out", "hip"] + extra, cwd=e2e.tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
                wall = time.perf_counter() - t0
                ph = re.search(r"genotype load[^|]*?([0-9.]+) s \|", cc.stdout); tot = re.search(r"\[timing\] total ([0-9.]+) s", cc.stdout)
                runs.append((float(ph.group(1)), float(tot.group(1)), round(wall, 3)) if (cc.returncode == 0 and ph and tot) else None)
            print(json.dumps({"binding": tag, "gpu_numa_node": node, "runs_load_s_total_s_wall_s": runs, "best_gbs": e2e.file_bytes / min(r[0] for r in runs if r) / 1e9}), flush=True)
    finally:
        subprocess.call(["rm", "-rf", e2e.tmp])


if __name__ == "__main__":
    main()
