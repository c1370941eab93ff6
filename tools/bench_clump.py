#!/usr/bin/env python3
"""--clump end to end: reference plink2 (all host cores) against plink2-hip on the same files, same command line.
Genotypes come from the benchmark generator; the association report names every variant, with a sprinkling of
strong p-values.  Writes one JSON object per shape to stdout / --out.  The .clumps files must be identical."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def write_report(path, m, seed, sig_rate):
    rng = np.random.default_rng(seed)
    p = rng.random(m)
    sig = rng.random(m) < sig_rate
    p[sig] = 10.0 ** (-rng.uniform(4, 30, size=int(sig.sum())))
    mid = (~sig) & (rng.random(m) < 0.05)
    p[mid] = 10.0 ** (-rng.uniform(1.3, 4, size=int(mid.sum())))
    with open(path, "w") as f:
        f.write("#CHROM\tPOS\tID\tTEST\tP\n")
        f.write("".join("1\t1\tsnp%d\tADD\t%.6g\n" % (i, p[i]) for i in range(m)))
    return int((p <= 1e-4).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--variants", type=int, default=1000000)
    ap.add_argument("--spacing", type=int, default=1400)
    ap.add_argument("--sig-rate", type=float, default=0.002)
    ap.add_argument("--kb", type=float, default=250.0)
    ap.add_argument("--r2", type=float, default=0.5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    m, n = args.variants, args.samples
    chr_idx, bps = bench.genome_layout(m, 1, args.spacing)
    stride = (n + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(bench.SEED, 0, m, n, 0.001, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    del buf
    tmp = tempfile.mkdtemp(prefix="clumpbench_")
    res = {"samples": n, "variants": m, "spacing_bp": args.spacing, "clump_kb": args.kb, "clump_r2": args.r2, **bench.host_description()}
    try:
        bench.write_plink1_fileset(os.path.join(tmp, "d"), host, n, chr_idx, bps)
        res["index_candidates"] = write_report(os.path.join(tmp, "assoc.txt"), m, 7, args.sig_rate)
        common = ["--bfile", "d", "--clump", "assoc.txt", "--clump-unphased", "--clump-kb", repr(args.kb), "--clump-r2", repr(args.r2)]
        ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
        cli_bin = pkg.build_cli()
        t0 = time.perf_counter()
        cp = subprocess.run([ref_bin] + common + ["--threads", str(os.cpu_count() or 1), "--out", "ref"], cwd=tmp, stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True, timeout=3000)
        res["reference_wall_s"] = time.perf_counter() - t0
        res["reference_rc"] = cp.returncode
        walls = []
        for _ in range(2):
            t0 = time.perf_counter()
            cc = subprocess.run([cli_bin] + common + ["--timing", "--out", "hip"], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                timeout=3000)
            walls.append(time.perf_counter() - t0)
        res["plink2_hip_wall_s"] = min(walls)
        res["plink2_hip_rc"] = cc.returncode
        res["plink2_hip_timing"] = [l for l in cc.stdout.split("\n") if l.startswith("[timing]")]
        res["summary"] = [l.strip() for l in cp.stdout.split("\n") if "formed from" in l]
        same = (cp.returncode == 0 and cc.returncode == 0 and
                open(os.path.join(tmp, "ref.clumps")).read() == open(os.path.join(tmp, "hip.clumps")).read())
        res["clumps_identical"] = bool(same)
        res["speedup"] = res["reference_wall_s"] / res["plink2_hip_wall_s"]
        if not same:
            res["ref_tail"] = cp.stdout[-400:]
            res["hip_tail"] = cc.stdout[-400:]
    finally:
        subprocess.call(["rm", "-rf", tmp])
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
