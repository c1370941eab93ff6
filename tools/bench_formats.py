#!/usr/bin/env python3
"""plink2-hip end to end on the same data as .bed and as the reference's default variable-width .pgen (storage mode 0x10,
written by `plink2 --make-pgen`): how much the host-side record decode costs.  One JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def timed(cmd, cwd, reps=2):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        cp = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
        dt = time.perf_counter() - t0
        if cp.returncode != 0:
            return None, cp.stdout[-500:]
        if best is None or dt < best:
            best, out = dt, cp.stdout
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=200000)
    ap.add_argument("--missing-rate", type=float, default=0.001)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    m, n = args.variants, args.samples
    chr_idx, bps = bench.genome_layout(m, 1, 2875)
    stride = (n + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(bench.SEED, 0, m, n, args.missing_rate, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    del buf
    tmp = tempfile.mkdtemp(prefix="fmtbench_")
    res = {"samples": n, "variants": m, "missing_rate": args.missing_rate, **bench.host_description()}
    try:
        bench.write_plink1_fileset(os.path.join(tmp, "d"), host, n, chr_idx, bps)
        ref_bin = os.path.join(REPO, "oracle", "_ref", "plink2")
        cli = pkg.build_cli()
        t, out = timed([ref_bin, "--bfile", "d", "--make-pgen", "--out", "v"], tmp, reps=1)
        res["make_pgen_s"] = t
        res["pgen_bytes"] = os.path.getsize(os.path.join(tmp, "v.pgen"))
        res["bed_bytes"] = os.path.getsize(os.path.join(tmp, "d.bed"))
        res["pgen_mode"] = open(os.path.join(tmp, "v.pgen"), "rb").read(3)[2]
        job = ["--indep-pairwise", "200kb", "0.5"]
        for name, inp in (("bed", ["--bfile", "d"]), ("pgen_variable_width", ["--pfile", "v"])):
            t, out = timed([cli] + inp + job + ["--timing", "--out", "hip_" + name], tmp)
            res["plink2_hip_%s_s" % name] = t
            res["plink2_hip_%s_timing" % name] = [l for l in (out or "").split("\n") if l.startswith("[timing]")]
            t, out = timed([ref_bin] + inp + job + ["--threads", str(os.cpu_count() or 1), "--out", "ref_" + name], tmp, reps=1)
            res["reference_%s_s" % name] = t
        same = all(open(os.path.join(tmp, "hip_%s.prune.in" % a)).read() == open(os.path.join(tmp, "ref_bed.prune.in")).read()
                   for a in ("bed", "pgen_variable_width"))
        res["prune_in_identical"] = bool(same)
    finally:
        subprocess.call(["rm", "-rf", tmp])
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
