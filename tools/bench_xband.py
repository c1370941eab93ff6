#!/usr/bin/env python3
"""tools/bench_xband.py -- chrX windows of the windowed r^2 table and of --clump: the pair kernels + device-side weighting over the chrX run's own all-pairs engines
(round 5, the default) against the pair lists with the host arithmetic (--debug-x-host).  One JSON line per command."""
import json, os, re, subprocess, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch, bench
    import __graft_entry__ as ge
    pkg = ge.load_package()
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    stride = (n + 3) // 4
    tmp = tempfile.mkdtemp(prefix="xband_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        dev = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
        pkg.synth_genotypes_device(bench.SEED, 0, m, n, 0.01, dev.data_ptr(), stride)
        torch.cuda.synchronize()
        with open(os.path.join(tmp, "x.pgen"), "wb") as f:
            f.write(bytes([0x6C, 0x1B, 0x02]) + np.uint32(m).tobytes() + np.uint32(n).tobytes() + bytes([0x40]))
            f.write(memoryview(dev.cpu().numpy()))
        del dev
        rng = np.random.default_rng(1)
        with open(os.path.join(tmp, "x.pvar"), "w") as f:
            f.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join("X\t%d\tsnp%d\tA\tC\n" % (1000 + 100 * i, i) for i in range(m)))
        with open(os.path.join(tmp, "x.psam"), "w") as f:
            f.write("#IID\tSEX\n" + "".join("s%d\t%d\n" % (q, 1 + (q % 2)) for q in range(n)))
        with open(os.path.join(tmp, "assoc.txt"), "w") as f:
            f.write("#CHROM\tPOS\tID\tTEST\tOBS_CT\tP\n" + "".join("X\t1\tsnp%d\tADD\t100\t%.3g\n" % (i, 10.0 ** (-rng.uniform(0, 12)) if rng.random() < 0.3 else rng.random()) for i in range(m)))
        cli = os.path.join(REPO, "plink-ng_amd", "bin", "plink2-hip")
        cmds = {"windowed table, 50 kb (500 variants), r2 >= 0.2": ["--pfile", "x", "--r2-unphased", "--ld-window-kb", "50", "--ld-window-r2", "0.2"],
                "--clump, 50 kb, r2 0.2": ["--pfile", "x", "--clump", "assoc.txt", "--clump-unphased", "--clump-kb", "50", "--clump-r2", "0.2", "--clump-p1", "1e-3", "--clump-p2", "0.05"]}
        for name, args in cmds.items():
            res = {"command": name, "variants_on_chrX": m, "samples": n}
            outs = {}
            for tag, hook in (("pair_kernels", []), ("pair_lists_host_arithmetic", ["--debug-x-host"])):
                walls = []
                for _ in range(2):
                    t0 = time.perf_counter()
                    cc = subprocess.run([cli] + args + hook + ["--out", tag], cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
                    walls.append(time.perf_counter() - t0)
                    assert cc.returncode == 0, cc.stdout[-500:]
                res[tag + "_wall_s"] = min(walls)
                ext = ".vcor" if "--r2-unphased" in args else ".clumps"
                outs[tag] = open(os.path.join(tmp, tag + ext), "rb").read()
            res["files_identical"] = outs["pair_kernels"] == outs["pair_lists_host_arithmetic"]
            res["output_bytes"] = len(outs["pair_kernels"])
            print(json.dumps(res), flush=True)
    finally:
        subprocess.call(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
