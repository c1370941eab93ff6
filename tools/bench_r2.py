#!/usr/bin/env python3
"""Config-4 style measurement: --r2-unphased all-pairs r^2 over a block of variants (no window), reported as
variant-pairs/s of the tile kernel in matrix mode (lower triangle of a V x V block, float output).
    python tools/bench_r2.py --samples 500000 --variants 32768"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--variants", type=int, default=32768)
    ap.add_argument("--missing-rate", type=float, default=0.0)
    ap.add_argument("--rows-per-call", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n, m = args.samples, args.variants
    stride = (n + 3) // 4
    geno = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(20260925 + 4, 0, m, n, args.missing_rate, geno.data_ptr(), stride)
    torch.cuda.synchronize()
    eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    eng.set_variants_matrix(m)
    eng.load_genotypes_device(0, m, geno.data_ptr(), stride, pkg.LDP_GENO_REF)
    eng.r2_unphased_rows(0, min(64, m), as_float=True)  # warm-up
    t0 = time.perf_counter()
    pairs = 0
    kms = 0.0
    nan_ct = 0
    for r0 in range(0, m, args.rows_per_call):
        rows = min(args.rows_per_call, m - r0)
        out = eng.r2_unphased_rows(r0, rows, as_float=True)
        c = eng.counters()
        pairs += c["candidate_pairs"]
        kms += c["ms_pair_kernel"]
        nan_ct += int((out != out).sum())
    wall = time.perf_counter() - t0
    print(json.dumps({"metric": "variant-pairs/s (--r2-unphased matrix rows, bin4)", "samples": n, "variants": m, "pairs": pairs,
                      "wall_s": wall, "pairs_per_s_wall": pairs / wall, "kernel_ms": kms, "pairs_per_s_kernel": pairs / (kms * 1e-3),
                      "algorithmic_GBps_kernel": pairs * (n / 2.0) / (kms * 1e-3) / 1e9, "nan_entries": nan_ct,
                      "missing_rate": args.missing_rate}))
    eng.close()


if __name__ == "__main__":
    main()
