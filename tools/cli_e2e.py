#!/usr/bin/env python3
"""End-to-end wall-clock of plink2-hip vs reference plink2 on the same .bed fileset (GPU box).
    python tools/cli_e2e.py --variants 1000000 --samples 50000"""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=1000000)
    ap.add_argument("--window-kb", type=float, default=200.0)
    ap.add_argument("--r2", type=float, default=0.5)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--pgen", action="store_true", help="convert the .bed with the reference's --make-pgen first (variable-width .pgen) and time both tools on that")
    ap.add_argument("--vcor", action="store_true", help="time the --r2-unphased table (--ld-window-kb = --window-kb, --ld-window-r2 = --r2) instead")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n, m = args.samples, args.variants
    chr_idx, bps = bench.genome_layout(m, 1, 2875)
    stride = (n + 3) // 4
    tmp = tempfile.mkdtemp(prefix="ldcli_")
    lut = np.zeros(256, dtype=np.uint8)
    conv = [3, 2, 0, 1]
    for b in range(256):
        lut[b] = sum(conv[(b >> (2 * k)) & 3] << (2 * k) for k in range(4))
    with open(os.path.join(tmp, "s.bed"), "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        step = 100000
        for v0 in range(0, m, step):
            cnt = min(step, m - v0)
            buf = torch.empty((cnt, stride), dtype=torch.uint8, device="cuda")
            pkg.synth_genotypes_device(bench.SEED, v0, cnt, n, 0.0, buf.data_ptr(), stride)
            torch.cuda.synchronize()
            f.write(lut[buf.cpu().numpy()].tobytes())
    with open(os.path.join(tmp, "s.bim"), "w") as f:
        f.write("".join("%d\tsnp%d\t0\t%d\tC\tA\n" % (chr_idx[i] + 1, i, bps[i]) for i in range(m)))
    with open(os.path.join(tmp, "s.fam"), "w") as f:
        f.write("".join("s%d s%d 0 0 2 -9\n" % (s, s) for s in range(n)))
    del buf
    torch.cuda.empty_cache()
    common = ["--bfile", "s", "--indep-pairwise", "%gkb" % args.window_kb, repr(args.r2)]
    outs = (".prune.in", ".prune.out")
    if args.vcor:
        common = ["--bfile", "s", "--r2-unphased", "--ld-window-kb", "%g" % args.window_kb, "--ld-window-r2", repr(args.r2)]
        outs = (".vcor",)
    if args.pgen:
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "oracle", "_ref", "plink2"), "--bfile", "s", "--make-pgen", "--out", "s", "--threads", str(os.cpu_count())], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print("reference --make-pgen rc", cp.returncode, "wall %.1f s, .pgen %.2f GB" % (time.perf_counter() - t0, os.path.getsize(os.path.join(tmp, "s.pgen")) / 1e9))
        common = ["--pfile", "s"] + common[2:]
    for rep in range(2):
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")] + common + ["--timing", "--out", "hip"], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_hip = time.perf_counter() - t0
        print("plink2-hip rc", cp.returncode, "wall %.3f s" % t_hip)
        print("\n".join(ln for ln in cp.stdout.splitlines() if "timing" in ln or "removed" in ln or "written" in ln or "Error" in ln or "timeline" in ln or "recs copy" in ln))
    if not args.no_ref:
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "oracle", "_ref", "plink2")] + common + ["--threads", str(os.cpu_count()), "--out", "ref"], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_ref = time.perf_counter() - t0
        print("reference rc", cp.returncode, "wall %.3f s" % t_ref, "speedup %.1fx" % (t_ref / t_hip))
        same = all(open(os.path.join(tmp, "hip" + e)).read() == open(os.path.join(tmp, "ref" + e)).read() for e in outs)
        print("files identical:", same)
    subprocess.call(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
