#!/usr/bin/env python3
"""Randomised differential test on the GPU box: the HIP path (early termination on, through the C ABI) against the
CPU oracle over random shapes -- sample counts around the 512-sample chunk boundaries, count/kb windows with steps,
both scan orders, thresholds from 0.02 to 0.95, missing rates from 0 to 20 %, LD blocks, monomorphic and all-missing
rows, several chromosomes.  Prints the first mismatching case (seed) and exits non-zero.
    python tests/fuzz_parity.py [--cases 150] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import ldtools as T  # noqa: E402
import __graft_entry__ as ge  # noqa: E402


def one_case(pkg, rng, idx, wide_missing=False, wide_async=False, wide_sparse=False):
    n = int(rng.choice([33, 64, 100, 511, 512, 513, 1000, 1536, 2047, 2049, 3000, 5000, 9000]))
    if wide_sparse:
        n = int(rng.choice([1536, 2049, 5000, 9000, 20000, 40000]))   # --wide-sparse: enough 512-sample stages for the checkpoints to fire
    m = int(rng.integers(40, 700 if n <= 3000 else 350))
    miss = float(rng.choice([0.0, 0.0, 0.0, 0.001, 0.003, 0.01, 0.05, 0.2]))
    if wide_sparse:
        miss = float(rng.choice([0.0001, 0.0003, 0.001, 0.002, 0.004, 0.005]))   # ... a FEW missing calls in every row: the tile kernel's SPARSE instantiation
    if wide_async:
        miss = 0.0                                    # --wide-async: complete data through the tile plan on the barrier-free kernel
    if wide_missing:
        miss = float(rng.choice([0.01, 0.05, 0.2]))   # --wide-missing: every case on the missing-call kernels ...
    raw = T.synth_raw_codes(m, n, seed=int(rng.integers(1, 1 << 30)), missing_rate=miss)
    # sprinkle structure: copies with noise (LD), monomorphic rows, an all-missing row, rare variants
    for _ in range(m // 6):
        a = int(rng.integers(1, m))
        src = max(0, a - int(rng.integers(1, 12)))
        keep = rng.random(n) < rng.choice([0.5, 0.8, 0.95, 1.0])
        raw[a] = np.where(keep, raw[src], raw[a])
    if m > 20:
        raw[int(rng.integers(0, m))] = int(rng.integers(0, 3))
        raw[int(rng.integers(0, m))] = 3
        v = int(rng.integers(0, m))
        raw[v] = 0
        raw[v, rng.choice(n, size=max(1, n // 200), replace=False)] = 1
    n_chr = int(rng.integers(1, 5))
    chr_idx = np.sort(rng.integers(0, n_chr, size=m)).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(n_chr):
        sel = np.where(chr_idx == c)[0]
        bps[sel] = np.sort(rng.integers(1, 60000, size=len(sel)))
    is_bp = bool(rng.random() < 0.5)
    if is_bp:
        window, step = int(rng.integers(200, 30000)), 1
    else:
        window = int(rng.integers(2, 260))
        step = int(rng.integers(1, max(2, window)))
    r2 = float(rng.choice([0.02, 0.1, 0.2, 0.35, 0.5, 0.7, 0.9, 0.95]))
    order = int(rng.integers(1, 3))
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, r2, order)
    eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
    wide = int(rng.choice([-1, -1, 0, 1, 3]))  # (drawn for every case, so that the sequence of cases stays the same)
    if wide_missing or wide_sparse:
        wide = int(idx % 3)                        # ... over the tile plan (quarter tiles of the four-product form unless switched off below)
    if wide_async:
        wide = int(idx % 3)
        eng.set_option("wide_async", 1)
    if wide >= 0:
        eng.set_option("wide_min_reach", wide)  # send narrower bands through the 8 x 8 tile plan of the wide-band kernel too
    if idx % 3 == 2:
        eng.set_option("pair_four", 0)   # the six-product form of the missing-call kernel (prune launches default to four)
    if idx % 4 == 3:
        eng.set_option("pair_four_tiles", 0)  # ... on the parallelogram plan in wide bands too (default: quarter tiles of the tile plan)
    if (idx % 5 == 4) and not wide_sparse:
        eng.set_option("pair_sparse", 0)  # rows with a few missing calls through the missing-call kernel too
    if wide_sparse and (idx % 7 == 6):
        eng.set_option("early_exit", 0)   # ... exhaustively now and then
    eng.set_variants(chr_idx, bps)
    packed = T.pack_2bit(raw)
    if rng.random() < 0.5:
        eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
    else:  # several calls, uneven pieces
        cut = sorted(set([0, m] + [int(x) for x in rng.integers(1, m, size=3)]))
        for a, b in zip(cut[:-1], cut[1:]):
            eng.load_genotypes_host(a, packed[a:b], pkg.LDP_GENO_REF)
    got = eng.run()
    ctr = eng.counters()
    eng.close()
    ok = np.array_equal(got, want)
    desc = "case %d: n=%d m=%d miss=%g %s window=%d step=%d r2=%g order=%d chr=%d wide_min_reach=%d tiles=%d four_tile_launches=%d sparse_tile_launches=%d recounted=%d removed=%d skipped=%.2f" % (
        idx, n, m, miss, "bp" if is_bp else "count", window, step, r2, order, n_chr, wide, ctr["wide_tiles"], ctr["four_tile_launches"], ctr["sparse_tile_launches"],
        ctr["sparse_exact_pairs"], int(want.sum()),
        (ctr["mfma_skipped_product_stages"] / ctr["mfma_product_stages"]) if ctr["mfma_product_stages"] else 0.0)
    return ok, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wide-missing", action="store_true", help="every case has missing calls and takes the wide-band tile plan (pair_mfma_tile4_kernel)")
    ap.add_argument("--wide-sparse", action="store_true", help="every case has a FEW missing calls (0.01-0.5 %) and takes the tile plan: pair_mfma_wide_kernel's SPARSE instantiation")
    ap.add_argument("--wide-async", action="store_true", help="every case is complete data on the tile plan, run by pair_mfma_wide_async_kernel (engine option wide_async)")
    args = ap.parse_args()
    pkg = ge.load_package()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    skipped_any = sparse_tiles = 0
    for k in range(args.cases):
        ok, desc = one_case(pkg, rng, k, args.wide_missing, args.wide_async, args.wide_sparse)
        if "skipped=0.00" not in desc:
            skipped_any += 1
        if "sparse_tile_launches=0" not in desc:
            sparse_tiles += 1
        if not ok:
            print("MISMATCH", desc, "(--seed %d)" % args.seed)
            sys.exit(1)
        if k % 25 == 0:
            print(desc, flush=True)
    print("%d cases identical to the oracle (%d with early termination firing, %d on the tile kernel's SPARSE instantiation), %.1f s" % (args.cases, skipped_any, sparse_tiles,
                                                                                                                                     time.time() - t0))


if __name__ == "__main__":
    main()
