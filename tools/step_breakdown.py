import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
n, m = 50000, 1000000
chr_idx, bps = bench.genome_layout(m, 1, 2875)
eng = pkg.LdPruneEngine(n, pkg.kb_window(200), 1, True, 0.5, device=0)
eng.set_variants(chr_idx, bps)
subs = eng.subcontigs()
stride = (n + 3) // 4
geno = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
pkg.synth_genotypes_device(bench.SEED, 0, m, n, 0.0, geno.data_ptr(), stride)
torch.cuda.synchronize()
for it in range(4):
    t0 = time.perf_counter()
    for ln, first in subs:
        eng.load_genotypes_device(first, ln, geno.data_ptr() + first * stride, stride, pkg.LDP_GENO_REF)
    t1 = time.perf_counter()
    bm = eng.run_bitmap()
    t2 = time.perf_counter()
    c = eng.counters()
    print("load calls %.2f ms | run %.2f ms (engine total %.2f: prepare %.2f pair %.2f replay %.2f) | step %.2f" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, c["ms_run_total"], c["ms_prepare"], c["ms_pair_kernel"], c["ms_replay"], (t2 - t0) * 1e3))
# single load call for the whole table
for it in range(3):
    t0 = time.perf_counter()
    eng.load_genotypes_device(0, m, geno.data_ptr(), stride, pkg.LDP_GENO_REF)
    t1 = time.perf_counter()
    bm = eng.run_bitmap()
    t2 = time.perf_counter()
    c = eng.counters()
    print("ONE load call %.2f ms | run %.2f ms (engine total %.2f: prepare %.2f pair %.2f replay %.2f) | step %.2f" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, c["ms_run_total"], c["ms_prepare"], c["ms_pair_kernel"], c["ms_replay"], (t2 - t0) * 1e3))
