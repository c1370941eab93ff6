#!/usr/bin/env python3
"""tools/load_numa_probe.py -- does the file -> HBM leg of plink2-hip depend on WHICH socket's memory holds the file's pages?  The fileset (default 60,000
variants x 500,000 samples, 7.5 GB fixed-width .pgen under /dev/shm) is written by this process bound to each NUMA node's CPUs in turn (tmpfs pages
land on the writer's node), then `plink2-hip --timing` loads it three times.  One JSON line per node: the device's node, the writer's node, the load
times.  Optional extra arguments are passed to plink2-hip (e.g. --debug-copy-anywhere)."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


def node_cpus(node):
    try:
        txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    except OSError:
        return None
    cpus = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def main():
    import torch
    import bench
    import bench_support as support
    import __graft_entry__ as ge
    pkg = ge.load_package()
    variants = 60000
    extra = [a for a in sys.argv[1:] if a.startswith("--")]
    cfg = bench.CONFIGS["config3"]
    n = cfg["samples"]
    chr_idx, bps = bench.genome_layout(variants, 1, cfg["spacing"])
    cli = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
    everything = os.sched_getaffinity(0)
    dev_node = None
    try:
        bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
    except Exception:
        bus = None
    for node in (0, 1, 2, 3):
        cpus = node_cpus(node)
        if not cpus or not (cpus & everything):
            continue
        os.sched_setaffinity(0, cpus & everything)
        tmp = tempfile.mkdtemp(prefix="ldp_numa_probe_", dir="/dev/shm")
        try:
            nbytes = support.write_fixed_width_fileset(pkg, torch, tmp, n, variants, bench.SEED, chr_idx, bps)
            os.sched_setaffinity(0, everything)
            loads, lines = [], []
            for _ in range(3):
                cc = subprocess.run([cli, "--pfile", "g", "--indep-pairwise", "%gkb" % cfg["window_kb"], repr(cfg["r2"]), "--timing", "--out", "hip"] + extra, cwd=tmp,
                                    stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
                ph = re.search(r"genotype load[^|]*?([0-9.]+) s \|", cc.stdout)
                loads.append(float(ph.group(1)) if (cc.returncode == 0 and ph) else None)
                lines = [ln for ln in cc.stdout.splitlines() if ("NUMA" in ln or "numa" in ln)][:3]
            print(json.dumps({"file_written_on_node": node, "pgen_gb": nbytes / 1e9, "file_to_hbm_s": loads, "gbs": [nbytes / t / 1e9 if t else None for t in loads], "numa_lines": lines,
                              "extra": extra}), flush=True)
        finally:
            os.sched_setaffinity(0, everything)
            subprocess.call(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
