#!/usr/bin/env python3
"""End-to-end wall-clock of plink2-hip vs reference plink2 on the same .bed fileset (GPU box).
    python tests/cli_e2e.py --variants 1000000 --samples 50000"""
import argparse, os, subprocess, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def phased(args, pkg, torch, tmp, chr_idx, bps):
    """2n synthetic pseudo-samples -> haplotypes h = (code >= 1), sample s = haplotypes (2s, 2s+1); written as a
    standard variable-width .pgen whose records are raw main track + hardcall-phase track (every het phased)."""
    n, m = args.samples, args.variants
    hstride = (2 * n + 3) // 4
    rec = (n + 3) // 4
    dev = "cuda"
    shifts = torch.tensor([0, 2, 4, 6], dtype=torch.uint8, device=dev)
    w4 = torch.tensor([1, 4, 16, 64], dtype=torch.uint8, device=dev)
    w8 = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)
    blocks = (m + 65535) // 65536
    header_len = 12 + 8 * blocks + 4 * m
    lens = np.zeros(m, dtype=np.uint32)
    vrt = np.zeros(m, dtype=np.uint8)
    t0 = time.perf_counter()
    with open(os.path.join(tmp, "s.body"), "wb") as f:
        step = max(1, int(4e8 // (2 * n)))
        for v0 in range(0, m, step):
            cnt = min(step, m - v0)
            buf = torch.empty((cnt, hstride), dtype=torch.uint8, device=dev)
            pkg.synth_genotypes_device(bench.SEED, v0, cnt, 2 * n, 0.0, buf.data_ptr(), hstride)
            torch.cuda.synchronize()
            hap = (((buf.unsqueeze(-1) >> shifts) & 3) >= 1).reshape(cnt, -1)[:, :2 * n]
            ha, hb = hap[:, 0::2], hap[:, 1::2]
            geno = (ha.to(torch.uint8) + hb.to(torch.uint8))
            pad = (-n) % 4
            if pad:
                geno = torch.nn.functional.pad(geno, (0, pad))
            packed = (geno.reshape(cnt, -1, 4) * w4).sum(dim=-1, dtype=torch.uint8)
            het = ha ^ hb
            hct = het.sum(dim=1)
            pos = torch.cumsum(het.to(torch.int32), dim=1)  # 1-based slot among the hets = bit index in aux track 2
            width = ((int(hct.max().item()) + 1 + 7) // 8) * 8
            bits = torch.zeros((cnt, width), dtype=torch.uint8, device=dev)
            rows_i = torch.arange(cnt, device=dev).unsqueeze(1).expand_as(pos)
            sel = het & ha  # ALT on the first haplotype: phaseinfo set ("1|0")
            bits[rows_i[sel], pos[sel].long()] = 1
            aux = (bits.reshape(cnt, -1, 8) * w8).sum(dim=-1, dtype=torch.uint8)
            auxlen = torch.where(hct > 0, 1 + hct // 8, torch.zeros_like(hct))
            mat = torch.cat([packed, aux], dim=1)
            col = torch.arange(mat.shape[1], device=dev).unsqueeze(0)
            keep = col < (rec + auxlen).unsqueeze(1)
            f.write(mat[keep].cpu().numpy().tobytes())
            lens[v0:v0 + cnt] = (rec + auxlen).cpu().numpy()
            vrt[v0:v0 + cnt] = np.where(hct.cpu().numpy() > 0, 0x10, 0)
            del buf, hap, ha, hb, geno, packed, het, pos, bits, aux, mat, keep, rows_i, sel
    with open(os.path.join(tmp, "s.pgen"), "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x10]) + np.uint32(m).tobytes() + np.uint32(n).tobytes() + bytes([0x40 | 6]))
        offs = header_len + np.concatenate([[0], np.cumsum(lens.astype(np.uint64))])
        for b in range(blocks):
            f.write(np.uint64(offs[b * 65536]).tobytes())
        l3 = np.zeros((m, 3), dtype=np.uint8)
        for k in range(3):
            l3[:, k] = (lens >> (8 * k)) & 255
        for b in range(blocks):
            lo, hi = b * 65536, min(m, (b + 1) * 65536)
            f.write(vrt[lo:hi].tobytes())
            f.write(l3[lo:hi].tobytes())
        with open(os.path.join(tmp, "s.body"), "rb") as g:
            while True:
                chunk = g.read(1 << 28)
                if not chunk:
                    break
                f.write(chunk)
    os.remove(os.path.join(tmp, "s.body"))
    with open(os.path.join(tmp, "s.pvar"), "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join("%d\t%d\tsnp%d\tA\tC\n" % (chr_idx[i] + 1, bps[i], i) for i in range(m)))
    with open(os.path.join(tmp, "s.psam"), "w") as f:
        f.write("#IID\tSEX\n" + "".join("s%d\tNA\n" % s for s in range(n)))
    torch.cuda.empty_cache()
    print("phased .pgen written: %.2f GB in %.1f s" % (os.path.getsize(os.path.join(tmp, "s.pgen")) / 1e9, time.perf_counter() - t0))
    common = ["--pfile", "s", "--indep-pairphase", "%gkb" % args.window_kb, repr(args.r2)]
    t_hip = None
    for rep in range(2):
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")] + common + ["--timing", "--out", "hip"], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_hip = time.perf_counter() - t0
        print("plink2-hip rc", cp.returncode, "wall %.3f s" % t_hip)
        print("\n".join(ln for ln in cp.stdout.splitlines() if "timing" in ln or "removed" in ln or "Error" in ln))
    if not args.no_ref:
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "oracle", "_ref", "plink2")] + common + ["--threads", str(os.cpu_count()), "--out", "ref"], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t_ref = time.perf_counter() - t0
        print("reference rc", cp.returncode, "wall %.3f s" % t_ref, "speedup %.1fx" % (t_ref / t_hip))
        print("\n".join(ln for ln in cp.stdout.splitlines() if "removed" in ln or "Error" in ln))
        same = all(open(os.path.join(tmp, "hip" + e)).read() == open(os.path.join(tmp, "ref" + e)).read() for e in (".prune.in", ".prune.out"))
        print("files identical:", same)
    subprocess.call(["rm", "-rf", tmp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--variants", type=int, default=1000000)
    ap.add_argument("--window-kb", type=float, default=200.0)
    ap.add_argument("--r2", type=float, default=0.5)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--h2d-modes", default="", help="comma-separated LDP_DEBUG_H2D_MODE values to time plink2-hip with (they only reach a plink2-hip linked against the measurement build, -DLDP_MEASURE)")
    ap.add_argument("--env-sets", default="", help="semicolon-separated sets of NAME=VALUE,NAME=VALUE to time plink2-hip with on the same fileset (measurement)")
    ap.add_argument("--reps", type=int, default=3, help="plink2-hip runs per setting; the reported speed-up uses the BEST wall of each tool (HIP start-up "
                    "varies by 0.1-0.3 s from run to run on one box), all walls are printed")
    ap.add_argument("--flag-sets", default="", help="semicolon-separated sets of extra plink2-hip flags to time on the same fileset, e.g. ';--debug-load-map' (the first is the "
                    "one the speed-up line uses)")
    ap.add_argument("--pgen", action="store_true", help="convert the .bed with the reference's --make-pgen first (variable-width .pgen) and time both tools on that")
    ap.add_argument("--phased", action="store_true", help="--indep-pairphase on a phased variable-width .pgen (haplotypes = the synthetic generator's pseudo-samples, paired up)")
    ap.add_argument("--inter-chr", action="store_true", help="time --r2-unphased inter-chr --ld-window-r2 <--r2> (all pairs; keep --variants small)")
    ap.add_argument("--vcor", action="store_true", help="time the --r2-unphased table (--ld-window-kb = --window-kb, --ld-window-r2 = --r2) instead")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n, m = args.samples, args.variants
    chr_idx, bps = bench.genome_layout(m, 1, 2875)
    stride = (n + 3) // 4
    tmp = tempfile.mkdtemp(prefix="ldcli_")
    if args.phased:
        return phased(args, pkg, torch, tmp, chr_idx, bps)
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
    if args.inter_chr:
        common = ["--bfile", "s", "--r2-unphased", "inter-chr", "--ld-window-r2", repr(args.r2)]
        outs = (".vcor",)
    if args.pgen:
        t0 = time.perf_counter()
        cp = subprocess.run([os.path.join(REPO, "oracle", "_ref", "plink2"), "--bfile", "s", "--make-pgen", "--out", "s", "--threads", str(os.cpu_count())], cwd=tmp,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print("reference --make-pgen rc", cp.returncode, "wall %.1f s, .pgen %.2f GB" % (time.perf_counter() - t0, os.path.getsize(os.path.join(tmp, "s.pgen")) / 1e9))
        common = ["--pfile", "s"] + common[2:]
    settings = [("LDP_DEBUG_H2D_MODE=" + mode) for mode in args.h2d_modes.split(",")] if args.h2d_modes else [""]
    if args.env_sets:
        settings = args.env_sets.split(";")
    if args.flag_sets:
        settings = ["FLAGS " + fs for fs in args.flag_sets.split(";")]
    t_hip = None
    for setting in settings:
        env = dict(os.environ)
        extra = []
        if setting.startswith("FLAGS "):
            extra = setting[6:].split()
        else:
            for kv in filter(None, setting.split(",")):
                name, _, value = kv.partition("=")
                env[name] = value
        if setting:
            print("--- " + setting)
        walls = []
        for rep in range(max(1, args.reps)):
            t0 = time.perf_counter()
            cp = subprocess.run([os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")] + common + extra + ["--timing", "--out", "hip"], cwd=tmp,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
            walls.append(time.perf_counter() - t0)
            print("plink2-hip rc", cp.returncode, "wall %.3f s" % walls[-1])
            print("\n".join(ln for ln in cp.stdout.splitlines() if "timing" in ln or "removed" in ln or "written" in ln or "Error" in ln or "timeline" in ln or "recs copy" in ln))
        print("plink2-hip walls:", " ".join("%.3f" % w for w in walls), "best %.3f median %.3f" % (min(walls), sorted(walls)[len(walls) // 2]))
        t_hip = min(walls) if t_hip is None else t_hip   # (the speed-up line is the first setting's)
    if not args.no_ref:
        ref_walls = []
        for rep in range(2):
            t0 = time.perf_counter()
            cp = subprocess.run([os.path.join(REPO, "oracle", "_ref", "plink2")] + common + ["--threads", str(os.cpu_count()), "--out", "ref"], cwd=tmp,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            ref_walls.append(time.perf_counter() - t0)
        t_ref = min(ref_walls)
        print("reference rc", cp.returncode, "walls", " ".join("%.3f" % w for w in ref_walls), "best %.3f s" % t_ref, "speedup (best / best) %.1fx" % (t_ref / t_hip))
        same = all(open(os.path.join(tmp, "hip" + e)).read() == open(os.path.join(tmp, "ref" + e)).read() for e in outs)
        print("files identical:", same)
    subprocess.call(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
