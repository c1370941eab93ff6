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


def chrx(pkg, torch, args):
    """chrX pairs of an inter-chr table (ComputeXR2): all-founder and male-founder engines, hits at |r^2| >= 0.2."""
    import numpy as np
    n, m, k = args.samples, args.variants, args.chrx
    stride = (n + 3) // 4
    geno = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(20260925 + 4, 0, m, n, args.missing_rate, geno.data_ptr(), stride)
    torch.cuda.synchronize()
    e_all = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=0)
    e_all.set_variants_matrix(m)
    e_all.load_genotypes_device(0, m, geno.data_ptr(), stride, pkg.LDP_GENO_REF)
    n_male = int(0.45 * n) & ~3          # (the first samples: a byte-aligned prefix of every row is the male founders' row)
    e_m = pkg.LdPruneEngine(n_male, 2, 1, False, 0.5, device=0)
    e_m.set_variants_matrix(m)
    e_m.load_genotypes_device(0, m, geno.data_ptr(), stride, pkg.LDP_GENO_REF)
    is_x = np.zeros(m, dtype=np.uint8)
    is_x[m - k:] = 1
    pairs = k * (m - k) + k * (k - 1) // 2
    e_all.r2_unphased_block_x_hits(e_m, is_x, 0.2, m - 64, 64, 0, m)   # warm-up
    t0 = time.perf_counter()
    hits, found = e_all.r2_unphased_block_x_hits(e_m, is_x, 0.2, 0, m, 0, m, capacity=1 << 22)
    wall = time.perf_counter() - t0
    # the same pairs as lists through the one-wave-per-pair kernel (what the band writers and round 3 did), on a sample of the rows
    rows = np.arange(m - k, m, max(1, k // 32))
    first = np.concatenate([np.arange(0, j, dtype=np.uint32) for j in rows])
    second = np.concatenate([np.full(j, j, dtype=np.uint32) for j in rows])
    t0 = time.perf_counter()
    e_all.pair_stats(first, second)
    e_m.pair_stats(first, second)
    wall_lists = time.perf_counter() - t0
    print(json.dumps({"metric": "chrX variant-pairs/s (--r2-unphased inter-chr, ComputeXR2)", "samples": n, "male_founders": n_male, "variants": m, "chrx_variants": k,
                      "chrx_pairs": pairs, "hits_at_0.2": int(found), "wall_s": wall, "pairs_per_s": pairs / wall,
                      "pair_lists": {"pairs": int(len(first)), "wall_s": wall_lists, "pairs_per_s": len(first) / wall_lists,
                                     "what": "ldp_pair_stats on both engines (one wave per pair), host arithmetic not included"},
                      "missing_rate": args.missing_rate}))
    e_all.close()
    e_m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=500000)
    ap.add_argument("--variants", type=int, default=32768)
    ap.add_argument("--missing-rate", type=float, default=0.0)
    ap.add_argument("--rows-per-call", type=int, default=4096)
    ap.add_argument("--chrx", type=int, default=0, help="the last K variants are on chrX, 45 %% of the founders male: time their pairs' weighted r^2 "
                    "(ldp_r2_unphased_block_x_hits: two engines' tuples from the pair kernels) against the pair lists of ldp_pair_stats")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as ge
    pkg = ge.load_package()
    n, m = args.samples, args.variants
    if args.chrx:
        return chrx(pkg, torch, args)
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
