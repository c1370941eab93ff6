"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Integer work is compared bit-exactly: bit-planes, per-variant aggregates, the 6-tuple of every
candidate pair out of the tile kernel, and the final prune set."""
import ctypes
import os

import numpy as np
import pytest

import ldtools as T
from test_host_logic import make_positions

pytestmark = pytest.mark.gpu


def oracle_recs(raw, n):
    inv, mf, altmaj = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    return inv, mf, altmaj, hom, r2h, vaggs


@pytest.mark.parametrize("n", [2, 31, 32, 33, 63, 64, 65, 100, 1023, 1024, 1025, 2049, 5000])
@pytest.mark.parametrize("encoding", ["ref", "bed", "inverse"])
def test_prepare_planes_and_aggregates(gpu_pkg, n, encoding):
    pkg = gpu_pkg
    m = 40
    raw = T.synth_raw_codes(m, n, seed=n, missing_rate=0.07)
    raw[3] = 0          # monomorphic hom-REF
    raw[4] = 2          # monomorphic hom-ALT
    raw[5] = 3          # all missing
    raw[6] = 1          # all het
    if n >= 4:
        raw[7] = 0
        raw[7, : n // 2] = 2  # ref_ct == alt_ct: tie goes to REF (plink2_common.h:559-567)
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    eng = pkg.LdPruneEngine(n, 10, 1, False, 0.2, device=0)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    if encoding == "ref":
        eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    elif encoding == "bed":
        lut = np.array([3, 2, 0, 1], dtype=np.uint8)
        rec = (n + 3) // 4
        eng.load_genotypes_host(0, np.ascontiguousarray(T.pack_2bit(lut[raw]).view(np.uint8).reshape(m, -1)[:, :rec]), pkg.LDP_GENO_BED)
    else:
        eng.load_genotypes_host(0, inv, pkg.LDP_GENO_INVERSE)
        eng.set_maj_freqs(0, mf)
    recs = eng.variant_recs()
    lib = T.oracle()
    for v in range(m):
        gh, gr = eng.planes(v)
        w32 = (n + 31) // 32
        assert np.array_equal(gh, hom[v].view(np.uint32)[:w32]), (v, "hom plane")
        assert np.array_equal(gr, r2h[v].view(np.uint32)[:w32]), (v, "ref2het plane")
        assert (recs[v]["nm_ct"], recs[v]["sum"], recs[v]["ssq"]) == (vaggs[v].nm_ct, vaggs[v].sum, vaggs[v].ssq)
        assert bool(recs[v]["flags"] & 2) == bool(lib.ldo_is_monomorphic(ctypes.byref(vaggs[v])))
        assert bool(recs[v]["flags"] & 4) == (vaggs[v].nm_ct != n)
        if encoding != "inverse":
            assert bool(recs[v]["flags"] & 1) == bool(altmaj[v])
    assert np.array_equal(eng.maj_freqs(), mf)  # exact doubles
    eng.close()


@pytest.mark.parametrize("n", [40000, 200001, 500000, 1100003])
def test_prepare_large_sample_counts(gpu_pkg, n):
    """Every register-budget variant of prepare_kernel (rows of 1k .. 34k plane dwords), incl. 500k samples
    (BASELINE configs 3-5) and the two-pass fallback above 1M samples."""
    pkg = gpu_pkg
    m = 6
    raw = T.synth_raw_codes(m, n, seed=n % 1000, missing_rate=0.03)
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    eng = pkg.LdPruneEngine(n, 10, 1, False, 0.2, device=0)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    recs = eng.variant_recs()
    w32 = (n + 31) // 32
    for v in range(m):
        gh, gr = eng.planes(v)
        assert np.array_equal(gh, hom[v].view(np.uint32)[:w32])
        assert np.array_equal(gr, r2h[v].view(np.uint32)[:w32])
        assert (recs[v]["nm_ct"], recs[v]["sum"], recs[v]["ssq"]) == (vaggs[v].nm_ct, vaggs[v].sum, vaggs[v].ssq)
    assert np.array_equal(eng.maj_freqs(), mf)
    st = eng.pair_stats([0, 1, 2], [3, 4, 5])
    for k, (i, j) in enumerate([(0, 3), (1, 4), (2, 5)]):
        assert tuple(int(x) for x in st[k]) == T.oracle_pair_stats(hom, r2h, vaggs, n, i, j).astuple()
    removed, stats = eng.run_with_stats()  # 15 candidate pairs through the tile kernel at this row length
    k = 0
    for j in range(m):
        for i in range(j):
            assert tuple(int(x) for x in stats[k]) == T.oracle_pair_stats(hom, r2h, vaggs, n, i, j).astuple()
            k += 1
    eng.close()


@pytest.mark.parametrize("n,miss", [(65, 0.0), (100, 0.1), (1025, 0.02), (4097, 0.3), (50000, 0.05)])
def test_pair_stats_reference_kernel(gpu_pkg, n, miss):
    pkg = gpu_pkg
    m = 24
    raw = T.synth_raw_codes(m, n, seed=7 * n + 1, missing_rate=miss)
    raw[0] = np.where(raw[0] == 3, 0, raw[0])  # one complete variant among incomplete ones
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    eng = pkg.LdPruneEngine(n, 10, 1, False, 0.2, device=0)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    first, second = np.triu_indices(m, 1)
    got = eng.pair_stats(first, second)
    for k in range(len(first)):
        st = T.oracle_pair_stats(hom, r2h, vaggs, n, int(first[k]), int(second[k]))
        assert tuple(int(x) for x in got[k]) == st.astuple(), (first[k], second[k])
    eng.close()


def check_run(pkg, raw, chr_idx, bps, window, step, is_bp, r2, order, shard_world=1, device_input=False):
    m, n = raw.shape
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, r2, order)
    union = np.zeros(m, dtype=bool)
    for rank in range(shard_world):
        eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
        eng.set_variants(chr_idx, bps)
        if shard_world > 1:
            eng.set_shard(rank, shard_world)
        packed = T.pack_2bit(raw)
        if device_input:
            import torch
            t = torch.from_numpy(packed.view(np.int64)).to("cuda:0")
            torch.cuda.synchronize()
            eng.load_genotypes_device(0, m, t.data_ptr(), packed.strides[0], pkg.LDP_GENO_REF)
        else:
            eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
        if shard_world == 1:
            removed, stats = eng.run_with_stats()
            lo, cand = eng.band()
            assert len(stats) == cand
            k = 0
            for j in range(m):
                for i in range(int(lo[j]), j):
                    st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
                    assert tuple(int(x) for x in stats[k]) == st.astuple(), ("pair", i, j)
                    k += 1
            ctr = eng.counters()
            assert ctr["candidate_pairs"] == cand
            assert ctr["computed_pairs"] >= cand
            # the production path (early termination of hopeless tiles enabled) must agree
            assert np.array_equal(eng.run(), removed)
            assert eng.counters()["pred_true"] == ctr["pred_true"]
        else:
            removed = eng.run()
        union |= removed
        eng.close()
    assert np.array_equal(union, want), "removed sets differ: %d vs %d" % (union.sum(), want.sum())


RUN_CASES = [
    # m, n, seed, window, step, is_bp, r2, order, missing
    (300, 70, 1, 50, 5, False, 0.2, 2, 0.0),       # fast path only, one block per J-tile
    (300, 70, 2, 50, 5, False, 0.2, 1, 0.0),
    (300, 65, 3, 30, 1, False, 0.5, 2, 0.05),      # general path
    (300, 1100, 4, 15000, 1, True, 0.2, 2, 0.0),   # two k-chunks
    (300, 1100, 5, 15000, 1, True, 0.5, 1, 0.03),
    (420, 33, 6, 200, 40, False, 0.1, 2, 0.0),     # window > 128 distances: several blocks per J-tile
    (420, 33, 7, 300, 1, False, 0.3, 2, 0.1),
    (200, 40, 8, 2, 1, False, 0.3, 2, 0.0),        # minimal window
    (333, 2100, 9, 60, 60, False, 0.2, 2, 0.0),    # step == window, 3 k-chunks
]


@pytest.mark.parametrize("case", RUN_CASES)
def test_tile_kernel_and_prune_set(gpu_pkg, case):
    m, n, seed, window, step, is_bp, r2, order, miss = case
    raw = T.synth_raw_codes(m, n, seed, missing_rate=miss)
    chr_idx, bps = make_positions(m, 3, seed + 100)
    check_run(gpu_pkg, raw, chr_idx, bps, window, step, is_bp, r2, order)


def test_mixed_missingness_tiles(gpu_pkg):
    """Only a stretch of variants has missing calls: neighbouring tiles take different kernels."""
    m, n = 400, 130
    raw = T.synth_raw_codes(m, n, seed=21, missing_rate=0.0)
    rng = np.random.default_rng(5)
    raw[150:190] = np.where(rng.random((40, n)) < 0.1, 3, raw[150:190])
    chr_idx, bps = make_positions(m, 2, 77)
    check_run(gpu_pkg, raw, chr_idx, bps, 40, 3, False, 0.2, 2)


def test_partial_reload_keeps_the_route_of_the_resident_rows(gpu_pkg):
    """Rows with missing calls are loaded in bulk, then a few complete rows are loaded AGAIN one at a time (what plink2-hip does
    with multiallelic and MT rows).  The kernel choice must still know about the missing calls of the rows that stayed resident:
    a route derived from the last load alone would send the launch to the complete-data kernel, which takes nm = founder_ct."""
    pkg = gpu_pkg
    m, n = 300, 900
    raw = T.synth_raw_codes(m, n, seed=77, missing_rate=0.04, ld_copy_prob=0.6, redraw=0.05)
    redo = [5, 130, 131, 299]
    for v in redo:
        raw[v] = np.where(raw[v] == 3, 0, raw[v])      # the re-loaded rows are complete
    chr_idx, bps = make_positions(m, 2, 91)
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 40, 1, False, 0.2, 2)
    eng = pkg.LdPruneEngine(n, 40, 1, False, 0.2, order=2, device=0)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    for v in redo:                                       # host-built rows in the caller's own orientation, like the CLI's
        eng.load_genotypes_host(v, np.ascontiguousarray(inv[v:v + 1]), pkg.LDP_GENO_INVERSE)
        eng.set_maj_freqs(v, mf[v:v + 1])
    got = eng.run()
    c = eng.counters()
    assert c["route_complete_launches"] == 0 and c["route_general_launches"] + c["route_sparse_launches"] > 0
    assert np.array_equal(got, want)
    removed, stats = eng.run_with_stats()
    lo, _ = eng.band()
    k = 0
    for j in range(m):
        for i in range(int(lo[j]), j):
            assert tuple(int(x) for x in stats[k]) == T.oracle_pair_stats(hom, r2h, vaggs, n, i, j).astuple(), (i, j)
            k += 1
    eng.close()
    # the same through the r^2 rows of --r2-unphased (all-pairs plan)
    eng = pkg.LdPruneEngine(n, 40, 1, False, 0.2, order=2, device=0)
    eng.set_variants_matrix(m)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    for v in redo:
        eng.load_genotypes_host(v, np.ascontiguousarray(inv[v:v + 1]), pkg.LDP_GENO_INVERSE)
    r2 = eng.r2_unphased_rows()
    eng.close()
    for (i, j) in ((4, 5), (5, 130), (129, 131), (7, 299), (130, 131)):
        st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
        nm, s1, q1, s2, q2, dot = st.astuple()
        cov = float(dot * nm - s1 * s2)
        vp = float(q1 * nm - s1 * s1) * float(q2 * nm - s2 * s2)
        assert r2[j, i] == (cov * cov) / vp, (i, j)


@pytest.mark.parametrize("n,miss", [(3, 0.0), (64, 0.0), (255, 0.02), (256, 0.0), (257, 0.05), (1000, 0.0), (1021, 0.01), (4099, 0.0), (50000, 0.001)])
def test_rows_written_into_the_engines_image_are_counted_in_place(gpu_pkg, n, miss):
    """ldp_map_rows: a device-side producer writes REF-coded rows straight into the resident image (its trailing bits and the
    padding up to the stage boundary left as garbage) and ldp_load_genotypes() counts them where they are.  Records, planes,
    every candidate pair's integers and the prune set equal those of the same rows loaded from host memory, and the oracle's."""
    import torch
    pkg = gpu_pkg
    m = 150
    raw = T.synth_raw_codes(m, n, seed=n + 7, missing_rate=miss, ld_copy_prob=0.6, redraw=0.1)
    chr_idx, bps = make_positions(m, 2, 60 + n % 7)
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 30, 1, False, 0.3, 2)
    packed = np.ascontiguousarray(T.pack_2bit(raw).view(np.uint8).reshape(m, -1)[:, :(n + 3) // 4])
    if n % 4:
        packed[:, -1] |= np.uint8((0xff << (2 * (n % 4))) & 0xaa)   # garbage in the trailing bit pairs of the last byte
    eng = pkg.LdPruneEngine(n, 30, 1, False, 0.3, order=2, device=0)
    eng.set_variants(chr_idx, bps)
    subs = eng.subcontigs()
    assert sum(ln for ln, _ in subs) == m
    for ln, first in subs:
        ptr, stride = eng.map_rows(first, ln)
        assert stride % 64 == 0 and stride >= (n + 3) // 4
        img = torch.full((ln, stride), 0x5a, dtype=torch.uint8, device="cuda")          # (0x5a: not a valid padding)
        img[:, :packed.shape[1]] = torch.from_numpy(packed[first:first + ln]).cuda()
        torch.cuda.synchronize()
        assert pkg.hip_memcpy_dtod(ptr, img.data_ptr(), ln * stride) == 0
        torch.cuda.synchronize()
        eng.load_genotypes_device(first, ln, ptr, stride, pkg.LDP_GENO_REF)
    recs = eng.variant_recs()
    w32 = (n + 31) // 32
    for v in range(m):
        gh, gr = eng.planes(v)
        assert np.array_equal(gh, hom[v].view(np.uint32)[:w32]) and np.array_equal(gr, r2h[v].view(np.uint32)[:w32]), v
        assert (recs[v]["nm_ct"], recs[v]["sum"], recs[v]["ssq"]) == (vaggs[v].nm_ct, vaggs[v].sum, vaggs[v].ssq)
        assert bool(recs[v]["flags"] & 1) == bool(altmaj[v])
    assert np.array_equal(eng.maj_freqs(), mf)
    got = eng.run()
    assert np.array_equal(got, want)
    removed, stats = eng.run_with_stats()
    lo, _ = eng.band()
    k = 0
    for j in range(m):
        for i in range(int(lo[j]), j):
            assert tuple(int(x) for x in stats[k]) == T.oracle_pair_stats(hom, r2h, vaggs, n, i, j).astuple(), (i, j)
            k += 1
    # a second pass over the same rows (what a benchmark step does) gives the same answer: the padding fix is idempotent
    for ln, first in subs:
        ptr, stride = eng.map_rows(first, ln)
        eng.load_genotypes_device(first, ln, ptr, stride, pkg.LDP_GENO_REF)
    assert np.array_equal(eng.run(), want)
    # a pointer into the image that is not the mapped row itself is refused, not read while it is written
    ptr, stride = eng.map_rows(subs[0][1], subs[0][0])
    with pytest.raises(pkg.LdpError):
        eng.load_genotypes_device(subs[0][1] + 1, 1, ptr, stride, pkg.LDP_GENO_REF)
    eng.close()


@pytest.mark.parametrize("n,miss", [(300, 0.0), (2100, 0.03)])
def test_popcount_kernels_on_bit_planes_agree_with_the_matrix_pipe(gpu_pkg, n, miss):
    """The two resident formats (2-bit code image + matrix pipe, bit-planes + popcount kernels) give the same integers."""
    pkg = gpu_pkg
    m = 260
    raw = T.synth_raw_codes(m, n, seed=n, missing_rate=miss, ld_copy_prob=0.6, redraw=0.1)
    chr_idx, bps = make_positions(m, 2, 9)
    out = []
    for mfma in (1, 0):
        eng = pkg.LdPruneEngine(n, 40, 1, False, 0.3, order=2, device=0)
        eng.set_option("pair_mfma", mfma)
        eng.set_variants(chr_idx, bps)
        eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
        removed, stats = eng.run_with_stats()
        c = eng.counters()
        assert (c["mfma_block_products"] > 0) == bool(mfma)
        planes = [eng.planes(v) for v in (0, 7, m - 1)]
        out.append((removed, stats, planes, eng.run()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][3], out[1][3])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


WIDE_CASES = [
    # m, n, seed, window, step, is_bp, r2, order, min_reach
    (900, 300, 1, 500, 1, False, 0.2, 2, 12),       # 16 row-blocks of reach: the tile plan by itself
    (1300, 3000, 2, 400, 7, False, 0.5, 2, 12),     # count window with a step, two subcontigs
    (1000, 1100, 3, 150000, 1, True, 0.3, 1, 6),    # kb windows with gaps: subcontigs of every length, some wide by this measure
    (700, 70, 4, 50, 5, False, 0.2, 2, 0),          # a narrow band forced through the tiles (diagonal tiles only)
    (520, 513, 5, 519, 1, False, 0.1, 2, 0),        # one window spans everything: three J tiles, the last one ragged
    (257, 40, 6, 256, 1, False, 0.4, 2, 0),         # a J tile of one variant
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_wide_band_tiles_match_oracle(gpu_pkg, case):
    """pair_mfma_wide_kernel (8 x 8 block tiles, eight waves: the plan of wide bands such as config 3's): every candidate
    pair's integers and the prune set against the oracle, and early termination -- whole waves stopping at checkpoints --
    invisible in the result."""
    pkg = gpu_pkg
    m, n, seed, window, step, is_bp, r2, order, min_reach = case
    raw = T.synth_raw_codes(m, n, seed, missing_rate=0.0, ld_copy_prob=0.6, redraw=0.08)
    chr_idx, bps = make_positions(m, 2, seed + 40) if is_bp else (np.repeat(np.arange(2, dtype=np.uint32), (m + 1) // 2)[:m], None)
    inv, mf, altmaj, hom, r2h, vaggs = oracle_recs(raw, n)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps if bps is not None else np.arange(m, dtype=np.uint32), mf, window, step, is_bp, r2, order)
    res = {}
    # the forms of the tile kernel: a workgroup barrier per 512-sample stage ("wide_async" 0) with the diagonal tiles in 2 x 3 rectangles
    # ("wide_diag_kernel" 1, the default) or in the 2 x 4 rectangles of every other tile (0), and the barrier-free one whose waves exchange
    # counters through LDS (256-sample stages, pair_mfma_wide_async_kernel; 2 x 4 rectangles everywhere)
    diag_form = {}
    for ee, wa, dk in ((1, 0, 1), (0, 0, 1), (1, 0, 0), (0, 0, 0), (1, 1, 0), (0, 1, 0)):
        eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
        eng.set_option("wide_min_reach", min_reach)
        eng.set_option("early_exit", ee)
        eng.set_option("wide_async", wa)
        eng.set_option("wide_diag_kernel", dk)
        eng.set_variants(chr_idx, bps)
        eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
        got = eng.run()
        c = eng.counters()
        assert c["wide_tiles"] > 0 and c["route_complete_launches"] > 0
        assert np.array_equal(got, want), (ee, wa, dk, int(got.sum()), int(want.sum()))
        if dk:
            diag_form[ee] = c
            eng.close()
            continue
        if wa:
            assert (c["pred_true"], c["mfma_skipped_product_stages"], c["mfma_extra_product_stages"]) == \
                   (res[ee]["pred_true"], res[ee]["mfma_skipped_product_stages"], res[ee]["mfma_extra_product_stages"]), (ee, "the two forms retire the same work")
        else:
            res[ee] = c
            # the diagonal tiles' own rectangles decide the same pairs (a retired product's pairs are all false either way)
            assert diag_form[ee]["pred_true"] == c["pred_true"] and diag_form[ee]["mfma_block_products"] == c["mfma_block_products"]
        if ee == 0:
            removed, stats = eng.run_with_stats()
            lo, _ = eng.band()
            k = 0
            for j in range(m):
                for i in range(int(lo[j]), j):
                    assert tuple(int(x) for x in stats[k]) == T.oracle_pair_stats(hom, r2h, vaggs, n, i, j).astuple(), (i, j)
                    k += 1
            assert np.array_equal(removed, want)
        eng.close()
    assert res[0]["mfma_skipped_product_stages"] == 0 and res[1]["pred_true"] == res[0]["pred_true"]
    # the same rows through the parallelogram plan alone
    eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
    eng.set_option("wide_min_reach", 1e9)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    assert np.array_equal(eng.run(), want) and eng.counters()["wide_tiles"] == 0
    eng.close()


def test_wide_band_with_missing_calls_keeps_the_tile_plan(gpu_pkg):
    """Wide-band subcontigs keep their 8 x 8 tile plan whatever the rows miss: a few missing calls -> the tile kernel's SPARSE
    instantiation (exact dot product, interval epilogue; ldp_counters.sparse_tile_launches says it ran), many -> the four-product form on
    quarter tiles.  Option "wide_sparse" 0 restores rounds 2-5 (the interval epilogue over the parallelogram plan): same prune set."""
    pkg = gpu_pkg
    m, n = 800, 900
    for miss, route, tiles in ((0.001, "route_sparse_launches", "sparse_tile_launches"), (0.05, "route_general_launches", "four_tile_launches")):
        raw = T.synth_raw_codes(m, n, 17, missing_rate=miss, ld_copy_prob=0.6, redraw=0.08)
        chr_idx = np.zeros(m, dtype=np.uint32)
        bps = np.arange(m, dtype=np.uint32)
        inv, mf, _ = T.oracle_prepare(raw)
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 450, 1, False, 0.3, 2)
        for ws in (1, 0):
            eng = pkg.LdPruneEngine(n, 450, 1, False, 0.3, order=2, device=0)
            eng.set_option("wide_sparse", ws)
            eng.set_variants(chr_idx, None)
            eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
            got = eng.run()
            c = eng.counters()
            eng.close()
            assert c["wide_tiles"] > 0 and c[route] > 0 and c["route_complete_launches"] == 0
            assert (c[tiles] > 0) == bool(ws or tiles == "four_tile_launches")
            assert np.array_equal(got, want)


SPARSE_WIDE_CASES = [
    # m, n, window, step, is_bp, r2, order, min_reach, miss, adversarial rows (rare variants with missing partners, rows that miss 7 %)
    (1500, 20000, 600, 1, False, 0.2, 2, 12, 0.001, True),     # config 3's threshold; 19 row-blocks of reach, 40 stages
    (1500, 30000, 600, 1, False, 0.2, 2, 12, 0.0005, False),   # ... and without the rows that keep every wave alive: checkpoints retire products
    (1100, 6000, 1000, 1, False, 0.5, 2, 12, 0.003, True),     # the window is most of the chromosome: diagonal and far tiles, ragged last J tile
    (1300, 9000, 400, 7, False, 0.5, 1, 12, 0.0003, False),    # count window with a step, --indep-order 1
    (1000, 30000, 150000, 1, True, 0.3, 2, 6, 0.004, True),    # kb windows with gaps, close to the route's limit (0.5 % on average)
    (700, 3000, 50, 5, False, 0.2, 2, 0, 0.002, True),         # a narrow band forced through the tiles (diagonal tiles only)
    (520, 513, 519, 1, False, 0.1, 2, 0, 0.001, False),        # one window spans everything; two stages, no checkpoint
]


@pytest.mark.parametrize("case", SPARSE_WIDE_CASES)
def test_wide_band_tiles_with_a_few_missing_calls(gpu_pkg, case):
    """pair_mfma_wide_kernel's SPARSE instantiation (DESIGN 4.1d on the tile plan; the reference's per-pair dispatch, plink2_ld.cc:699-723):
    same prune set and the same number of true predicates with early termination on and off, as the interval epilogue over the
    parallelogram plan ("wide_sparse" 0), as the six-product kernel, and as the oracle -- with pairs that only become correlated in the
    last 40 % of the samples, rare variants whose partners' missing calls sit on their carriers, complete rows and rows that miss 7 %."""
    pkg = gpu_pkg
    m, n, window, step, is_bp, r2, order, min_reach, miss, adversarial = case
    rng = np.random.default_rng(n + m)
    raw = T.synth_raw_codes(m, n, seed=n % 83 + 7, missing_rate=miss, ld_copy_prob=0.6, redraw=0.08)
    cut = int(0.6 * n)
    for v in range(40, m, 11):             # late LD with a variant 37 rows back (another row-block)
        raw[v, :cut] = rng.permutation(raw[v, :cut])
        raw[v, cut:] = raw[v - 37, cut:]
    tail = np.arange(int(0.75 * n), n)
    for v in (range(5, m - 40, 53) if adversarial else ()):   # a rare variant whose carriers come late; copies of it nearby and 33 rows on, missing on some of the carriers
        raw[v] = 0
        carriers = rng.choice(tail, size=min(30, len(tail)), replace=False)
        raw[v, carriers] = 1
        raw[v + 1] = raw[v]
        raw[v + 1, carriers[:4]] = 3
        raw[v + 33] = raw[v]
        raw[v + 33, carriers[:12]] = 3
    raw[rng.choice(m, size=30, replace=False)] = np.where(raw[3] == 3, 0, raw[3])[None, :]   # complete rows, in LD with each other
    for v in (rng.choice(m, size=6, replace=False) if adversarial else ()):
        raw[v, rng.random(n) < 0.07] = 3                                                    # a handful of rows far beyond the mean
    chr_idx, bps = make_positions(m, 2, 47) if is_bp else (np.repeat(np.arange(2, dtype=np.uint32), (m + 1) // 2)[:m], None)
    packed = T.pack_2bit(raw)

    def run(options):
        eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
        eng.set_option("wide_min_reach", min_reach)
        for name, value in options.items():
            eng.set_option(name, value)
        eng.set_variants(chr_idx, bps)
        eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
        got = eng.run()
        c = eng.counters()
        eng.close()
        return got, c

    on, c1 = run({})
    off, c0 = run({"early_exit": 0})
    plan, cp = run({"wide_sparse": 0})
    six, c6 = run({"pair_sparse": 0, "pair_four": 0, "early_exit": 0})
    for c in (c1, c0):
        assert c["wide_tiles"] > 0 and c["sparse_tile_launches"] > 0 and c["route_sparse_launches"] > 0
        assert c["route_general_launches"] == 0 and c["route_complete_launches"] == 0
    assert cp["sparse_tile_launches"] == 0 and cp["route_sparse_launches"] > 0 and c6["route_general_launches"] > 0 and c6["sparse_tile_launches"] == 0
    assert c0["mfma_skipped_product_stages"] == 0
    assert np.array_equal(off, six) and c0["pred_true"] == c6["pred_true"] > 0
    assert np.array_equal(plan, six) and cp["pred_true"] == c6["pred_true"]
    assert np.array_equal(on, six) and c1["pred_true"] == c0["pred_true"]   # (a retired product's pairs are all false)
    if (n >= 9000) and not adversarial:
        assert c1["mfma_skipped_product_stages"] > 0
    assert c1["sparse_exact_pairs"] < 0.2 * c1["candidate_pairs"]
    if n <= 9000:
        inv, mf, _ = T.oracle_prepare(raw)
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps if bps is not None else np.arange(m, dtype=np.uint32), mf, window, step, is_bp, r2, order)
        assert np.array_equal(on, want)


_RCCL_ONE_RANK = r"""
import sys
import numpy as np
sys.path.insert(0, {repo!r})
sys.path.insert(0, {tests!r})
import __graft_entry__ as ge
import ldtools as T
from test_host_logic import make_positions
pkg = ge.load_package()
m, n = 900, 200
raw = T.synth_raw_codes(m, n, seed=5, missing_rate=0.01)
chr_idx, bps = make_positions(m, 5, 3)
eng = pkg.LdPruneEngine(n, 20000, 1, True, 0.2, device=0)
eng.set_variants(chr_idx, bps)
eng.set_shard(0, 1)
eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
words = eng.run_bitmap()
comm = pkg.comm_init_all([0])[0]
try:
    out = eng.allgather_removed(comm, words)
finally:
    pkg.comm_destroy(comm)
nw = (m + 63) // 64
assert np.array_equal(out[:nw], words[:nw]) and int(np.unpackbits(words[:nw].view(np.uint8)).sum()) > 50
# a communicator whose size does not match the shard is refused (and aborted: a rank that cannot enter the collective must not leave
# its peers waiting)
eng2 = pkg.LdPruneEngine(n, 20000, 1, True, 0.2, device=0)
eng2.set_variants(chr_idx, bps)
eng2.set_shard(1, 2)
comm = pkg.comm_init_all([0])[0]
refused = False
try:
    eng2.allgather_removed(comm, words)
except pkg.LdpError:
    refused = True
finally:
    pkg.comm_destroy(comm)   # (a no-op on the aborted communicator)
assert refused
# ... and RCCL still works afterwards in the same process
comm = pkg.comm_init_all([0])[0]
out = eng.allgather_removed(comm, words)
pkg.comm_destroy(comm)
assert np.array_equal(out[:nw], words[:nw])
eng.close()
eng2.close()
print("rccl one rank: ok")
"""


def test_rccl_allgather_from_the_c_abi_with_one_rank(gpu_pkg):
    """ldp_allgather_removed: the one exchange step of a multi-GPU prune driven from a C/C++ host -- segments in shard order,
    ONE ncclAllGather on device buffers, stitched into global variant order.  One rank here (the box has one GPU): RCCL is
    bound and initialised, the collective runs, and the result is the rank's own bitmap; a mismatched communicator is refused and
    aborted.  (The two-rank stitch is covered on the CPU: tests/test_distributed_gloo.py.)  In a process of its own: an aborted
    communicator's helper threads have no business inside the test runner."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cp = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK.format(repo=os.path.dirname(here), tests=here)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                        timeout=600, env=env)
    assert cp.returncode == 0 and "rccl one rank: ok" in cp.stdout, (cp.stdout[-500:], cp.stderr[-1500:])


def test_device_pointer_input(gpu_pkg):
    m, n = 300, 257
    raw = T.synth_raw_codes(m, n, seed=31, missing_rate=0.02)
    chr_idx, bps = make_positions(m, 2, 78)
    check_run(gpu_pkg, raw, chr_idx, bps, 20000, 1, True, 0.2, 2, device_input=True)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_union_equals_unsharded(gpu_pkg, world):
    m, n = 900, 96
    raw = T.synth_raw_codes(m, n, seed=41, missing_rate=0.01)
    chr_idx, bps = make_positions(m, 7, 79)
    check_run(gpu_pkg, raw, chr_idx, bps, 10000, 1, True, 0.2, 2, shard_world=world)


def test_degenerate_inputs(gpu_pkg):
    pkg = gpu_pkg
    # every chromosome has one variant: no subcontigs, nothing loaded, nothing removed (plink2_ld.cc:2182)
    eng = pkg.LdPruneEngine(50, 1000, 1, True, 0.2, device=0)
    eng.set_variants(np.arange(5, dtype=np.uint32), np.full(5, 100, dtype=np.uint32))
    assert eng.subcontigs() == []
    assert not eng.run().any()
    eng.close()
    # empty variant table
    eng = pkg.LdPruneEngine(50, 1000, 1, True, 0.2, device=0)
    eng.set_variants(np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32))
    assert len(eng.run()) == 0
    eng.close()
    # a monomorphic variant alone in a length-1 subcontig is never loaded and stays in (SURVEY 8(c) edge (i))
    raw = np.zeros((3, 60), dtype=np.uint8)
    raw[1, :10] = 1
    raw[2, :10] = 2
    eng = pkg.LdPruneEngine(60, 1000, 1, True, 0.2, device=0)
    eng.set_variants(np.zeros(3, dtype=np.uint32), np.array([1000, 5000000, 5000100], dtype=np.uint32))
    assert eng.subcontigs() == [(2, 1)]
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    removed = eng.run()
    assert not removed[0]
    eng.close()


def test_preferred_variants(gpu_pkg):
    """--indep-preferred subtracts 1.0 from the major frequency (plink2_ld.cc:916-918): with every pair in
    LD, a preferred variant always wins its tie-breaks."""
    pkg = gpu_pkg
    m, n = 12, 200
    base = T.synth_raw_codes(1, n, seed=3)[0]
    raw = np.tile(base, (m, 1))
    eng = pkg.LdPruneEngine(n, 50, 1, False, 0.5, device=0)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    pref = np.zeros(m, dtype=bool)
    pref[4] = True
    eng.set_preferred(pref)
    removed = eng.run()
    assert not removed[4] and removed.sum() == m - 1
    eng.close()


def test_config2_sample_count(gpu_pkg):
    """BASELINE config 2's sample count (N = 50,000, 200 kb window at ~2.9 kb spacing), a few thousand
    variants: prune set against the oracle."""
    pkg = gpu_pkg
    m, n = 1500, 50000
    raw = T.synth_raw_codes(m, n, seed=2, missing_rate=0.0)
    chr_idx = (np.arange(m) // 750).astype(np.uint32)
    bps = (10000 + 2875 * (np.arange(m) % 750)).astype(np.uint32)
    inv, mf, _ = T.oracle_prepare(raw)
    want, evals = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 200000, 1, True, 0.5, 2)
    eng = pkg.LdPruneEngine(n, 200000, 1, True, 0.5, device=0)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, T.pack_2bit(raw), pkg.LDP_GENO_REF)
    got = eng.run()
    ctr = eng.counters()
    eng.close()
    assert np.array_equal(got, want)
    assert ctr["candidate_pairs"] >= evals


def _run_early_exit(pkg, packed, n, chr_idx, bps, window, step, is_bp, r2, order, enabled):
    eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order, device=0)
    eng.set_option("early_exit", 1 if enabled else 0)  # (a switch of this engine, not of the process)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
    removed = eng.run()
    ctr = eng.counters()
    eng.close()
    return removed, ctr


@pytest.mark.parametrize("n,r2,order,miss", [(2100, 0.5, 2, 0.0), (5000, 0.2, 2, 0.0), (5000, 0.9, 1, 0.0), (20000, 0.5, 2, 0.0), (20000, 0.05, 2, 0.0),
                                             (2100, 0.5, 2, 0.002), (5000, 0.5, 1, 0.05), (20000, 0.5, 2, 0.001), (20000, 0.2, 2, 0.01),
                                             (20000, 0.7, 2, 0.2)])
def test_early_termination_is_invisible(gpu_pkg, n, r2, order, miss):
    """Waves that stop at a checkpoint (ldp_device.h) may only ever skip pairs whose predicate is false: the
    prune set and the number of true predicates are those of the exhaustive run and of the oracle.  With missing
    calls the tiles take the general kernel and its interval bound."""
    m = 700
    raw = T.synth_raw_codes(m, n, seed=n % 97, missing_rate=miss)
    chr_idx, bps = make_positions(m, 2, 5)
    packed = T.pack_2bit(raw)
    off, c0 = _run_early_exit(gpu_pkg, packed, n, chr_idx, bps, 150, 1, False, r2, order, False)
    on, c1 = _run_early_exit(gpu_pkg, packed, n, chr_idx, bps, 150, 1, False, r2, order, True)
    assert c0["early_exit_unit_chunks"] == 0 and c0["mfma_skipped_product_stages"] == 0
    assert np.array_equal(on, off)
    assert c1["pred_true"] == c0["pred_true"]
    assert c1["tile_unit_chunks"] == c0["tile_unit_chunks"] > 0
    if r2 >= 0.5 and miss <= 0.01:
        # unrelated pairs are provably hopeless early on (whichever kernel family owned the tiles; the matrix-pipe kernels
        # for rows with missing calls -- six products, or the interval epilogue -- have no early termination)
        assert c1["early_exit_unit_chunks"] + c1["mfma_skipped_product_stages"] > 0 or c1["route_general_launches"] + c1["route_sparse_launches"] > 0
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 150, 1, False, r2, order)
    assert np.array_equal(on, want)


def _run_sparse(pkg, packed, n, chr_idx, bps, r2, options):
    eng = pkg.LdPruneEngine(n, 150, 1, False, r2, order=2, device=0)
    for name, value in options.items():
        eng.set_option(name, value)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
    removed = eng.run()
    ctr = eng.counters()
    eng.close()
    return removed, ctr


@pytest.mark.parametrize("n,miss,r2,redraw,frac", [
    (20000, 0.001, 0.5, 0.05, None),    # the default limit (0.5 % of the calls): nearly every pair is settled by its intervals
    (20000, 0.0003, 0.5, 0.29, None),   # planted r^2 ~ 0.504: a crowd of pairs next to the threshold
    (50000, 0.003, 0.2, 0.55, None),
    (9000, 0.001, 0.8, 0.1, None),
    (6000, 0.02, 0.5, 0.29, 0.08),    # a limit far beyond the useful one: wide intervals, most pairs resolved exactly
    (3000, 0.05, 0.1, 0.68, 0.2),
])
def test_rows_with_a_few_missing_calls(gpu_pkg, n, miss, r2, redraw, frac):
    """launches whose rows miss only a few calls stay with the complete-data matrix kernel; per-variant
    counts confine the pairwise-complete statistics to intervals, interval arithmetic settles the predicate of the
    clear pairs and the rest are counted exactly.  Same prune set as the six-product kernel and as the oracle."""
    m = 700
    raw = T.synth_raw_codes(m, n, seed=n % 89 + 3, missing_rate=miss, ld_copy_prob=0.7, redraw=redraw)
    chr_idx, bps = make_positions(m, 2, 5)
    packed = T.pack_2bit(raw)
    opts = {} if frac is None else {"sparse_frac": frac}
    got, c1 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, opts)
    six, c0 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0, "pair_four": 0})
    # which kernel owned the launches is read from the route words the device wrote, not from timings
    assert c0["sparse_exact_pairs"] == 0 and c0["route_general_launches"] > 0 and c0["route_sparse_launches"] == 0 and c0["route_complete_launches"] == 0
    # ... and the four-product form of that kernel (two sums of squares from per-variant intervals): the same set,
    # nearly every pair settled without a recount
    four, c4 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0})
    assert c4["route_general_launches"] > 0 and c4["route_sparse_launches"] == 0
    assert np.array_equal(four, six) and c4["pred_true"] == c0["pred_true"]
    assert c4["sparse_exact_pairs"] < (0.05 if frac is None else 0.5) * c4["candidate_pairs"]
    assert c1["route_sparse_launches"] > 0 and c1["route_general_launches"] == 0 and c1["route_complete_launches"] == 0, "the launches must have taken the interval path"
    assert np.array_equal(got, six)
    assert c1["pred_true"] == c0["pred_true"] > 0
    if frac is None:
        assert c1["sparse_exact_pairs"] < 0.2 * c1["candidate_pairs"]
    else:
        assert c1["sparse_exact_pairs"] > 0.03 * c1["candidate_pairs"]
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 150, 1, False, r2, 2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,miss,r2,redraw", [
    (20000, 0.01, 0.5, 0.05),
    (20000, 0.05, 0.2, 0.55),     # config 5's rate and threshold, with planted pairs around it
    (6000, 0.2, 0.5, 0.29),       # wide intervals next to a crowd of pairs at r^2 ~ 0.5: many recounts, same set
    (3001, 0.6, 0.1, 0.3),        # most calls missing
    (50000, 0.02, 0.8, 0.1),
])
def test_four_product_form_of_the_missing_call_kernel(gpu_pkg, n, miss, r2, redraw):
    """prune launches over rows with missing calls multiply four products per pair (dot, nm, the two sums); the sums of
    squares come from per-variant intervals -- ssq1 = hom_i - hm, |S_i - sum1| <= hm <= #(j missing, i called), same parity -- and the
    predicate, monotone in both, is decided where the ends agree; the rest is recounted exactly.  Same prune set and the same
    number of true predicates as the six-product form and the oracle, with and without early termination, also when some rows are
    complete and some are mostly missing."""
    m = 700
    raw = T.synth_raw_codes(m, n, seed=n % 97 + 11, missing_rate=miss, ld_copy_prob=0.7, redraw=redraw)
    rng = np.random.default_rng(n)
    raw[rng.choice(m, size=40, replace=False)] = np.where(raw[0] == 3, 0, raw[0])[None, :]   # some complete rows (copies: in LD with each other)
    for v in rng.choice(m, size=20, replace=False):
        raw[v, rng.random(n) < 0.7] = 3                                                        # and some that are mostly missing
    chr_idx, bps = make_positions(m, 2, 5)
    packed = T.pack_2bit(raw)
    six, c6 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0, "pair_four": 0})
    four, c4 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0})
    four_x, c4x = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0, "early_exit": 0})
    # ... and with the operands of engines beyond 1,800,000 founders (x and n instead of allele counts and missing flags)
    four_xn, c4n = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, r2, {"pair_sparse": 0, "pair_gu": 0})
    assert c6["route_general_launches"] > 0 and c4["route_general_launches"] > 0 and c6["sparse_exact_pairs"] == 0
    assert np.array_equal(four, six) and np.array_equal(four_x, six) and np.array_equal(four_xn, six)
    assert c4n["pred_true"] == c4["pred_true"] and c4n["mfma_skipped_product_stages"] == c4["mfma_skipped_product_stages"]
    assert c4x["pred_true"] == c6["pred_true"] > 0 and c4["pred_true"] <= c4x["pred_true"]   # (a retired product's pairs are false)
    assert c4x["mfma_skipped_product_stages"] == 0
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 150, 1, False, r2, 2)
    assert np.array_equal(four, want)


def _run_wide(pkg, packed, n, chr_idx, window, r2, options):
    eng = pkg.LdPruneEngine(n, window, 1, False, r2, order=2, device=0)
    for name, value in options.items():
        eng.set_option(name, value)
    eng.set_variants(chr_idx, None)
    eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
    removed = eng.run()
    ctr = eng.counters()
    eng.close()
    return removed, ctr


@pytest.mark.parametrize("n,m,window,miss,r2", [
    (20000, 1500, 600, 0.05, 0.2),     # config 5's rate and threshold; 19 row-blocks of reach
    (6000, 1100, 1000, 0.02, 0.5),     # the window is most of the chromosome: diagonal and far tiles, ragged last J tile
    (900, 1300, 420, 0.3, 0.1),        # four stages, no checkpoint reached; wide intervals, many recounts
    (50000, 900, 450, 0.01, 0.7),
])
def test_four_product_form_on_quarter_tiles_of_wide_bands(gpu_pkg, n, m, window, miss, r2):
    """subcontigs that have the 8 x 8 tile plan take the four-product form in QUARTER tiles (pair_mfma_tile4_kernel: eight
    waves, 4 J x 4 V row-blocks per stage, two products per wave) instead of the parallelogram plan.  Same prune set and the same
    number of true predicates as that plan, the six-product form and the oracle, with and without early termination -- with pairs
    that only become correlated in the last 40 % of the samples, and rows that are complete or mostly missing among the others."""
    pkg = gpu_pkg
    rng = np.random.default_rng(n + m)
    raw = T.synth_raw_codes(m, n, seed=n % 83 + 3, missing_rate=miss, ld_copy_prob=0.6, redraw=0.1)
    cut = int(0.6 * n)
    for v in range(40, m, 11):             # late LD with a variant 37 rows back (another row-block, often another quarter tile)
        raw[v, :cut] = rng.permutation(raw[v, :cut])
        raw[v, cut:] = raw[v - 37, cut:]
    raw[rng.choice(m, size=30, replace=False)] = np.where(raw[3] == 3, 0, raw[3])[None, :]
    for v in rng.choice(m, size=15, replace=False):
        raw[v, rng.random(n) < 0.7] = 3
    chr_idx = (np.arange(m) >= m - 200).astype(np.uint32)   # a second, short chromosome (narrow plan) behind the wide one
    packed = T.pack_2bit(raw)
    tiles, ct = _run_wide(pkg, packed, n, chr_idx, window, r2, {"pair_sparse": 0})
    tiles_x, ctx = _run_wide(pkg, packed, n, chr_idx, window, r2, {"pair_sparse": 0, "early_exit": 0})
    plan, cp = _run_wide(pkg, packed, n, chr_idx, window, r2, {"pair_sparse": 0, "pair_four_tiles": 0})
    tiles_xn, ctn = _run_wide(pkg, packed, n, chr_idx, window, r2, {"pair_sparse": 0, "pair_gu": 0})
    assert np.array_equal(tiles_xn, tiles) and ctn["pred_true"] == ct["pred_true"] and ctn["four_tile_launches"] > 0
    assert ctn["mfma_skipped_product_stages"] == ct["mfma_skipped_product_stages"]
    six, c6 = _run_wide(pkg, packed, n, chr_idx, window, r2, {"pair_sparse": 0, "pair_four": 0, "early_exit": 0})
    assert ct["wide_tiles"] > 0 and ct["four_tile_launches"] > 0 and ctx["four_tile_launches"] > 0
    assert cp["four_tile_launches"] == 0 and c6["four_tile_launches"] == 0 and c6["route_general_launches"] > 0
    assert np.array_equal(tiles, six) and np.array_equal(tiles_x, six) and np.array_equal(plan, six)
    assert ctx["pred_true"] == c6["pred_true"] > 0 and ct["pred_true"] <= ctx["pred_true"]
    assert ctx["mfma_skipped_product_stages"] == 0
    if n >= 6000:
        assert ct["mfma_skipped_product_stages"] > 0
    if n <= 6000:
        inv, mf, _ = T.oracle_prepare(raw)
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, np.arange(m, dtype=np.uint32), mf, window, 1, False, r2, 2)
        assert np.array_equal(tiles, want)


def test_route_follows_the_mean_not_the_worst_row(gpu_pkg):
    """A handful of rows with many missing calls among rows with few: the launch still takes the interval path (its
    pairs with those rows are counted exactly); when such rows are common it goes to the six-product kernel."""
    n, m = 20000, 700
    rng = np.random.default_rng(5)
    chr_idx, bps = make_positions(m, 2, 5)
    for n_bad, sparse in ((6, True), (80, False)):
        raw = T.synth_raw_codes(m, n, seed=41, missing_rate=0.0005, ld_copy_prob=0.7, redraw=0.1)
        for v in rng.choice(m, size=n_bad, replace=False):
            raw[v, rng.random(n) < 0.07] = 3
        packed = T.pack_2bit(raw)
        got, c1 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, 0.5, {})
        six, c0 = _run_sparse(gpu_pkg, packed, n, chr_idx, bps, 0.5, {"pair_sparse": 0, "pair_four": 0})
        assert np.array_equal(got, six) and c1["pred_true"] == c0["pred_true"]
        assert c0["route_general_launches"] > 0 and c0["route_sparse_launches"] == 0
        if sparse:
            assert c1["route_sparse_launches"] > 0 and c1["route_general_launches"] == 0 and c1["sparse_exact_pairs"] > 0
        else:
            assert c1["route_general_launches"] > 0 and c1["route_sparse_launches"] == 0
            assert c1["sparse_exact_pairs"] < 0.01 * c1["candidate_pairs"]   # (the four-product form recounts what its intervals leave open)
        inv, mf, _ = T.oracle_prepare(raw)
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 150, 1, False, 0.5, 2)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("order", [2, 1])
def test_predicate_rows_as_csr_dense_and_overflow_agree(gpu_pkg, order):
    """The predicate rows go back to the host as their non-zero words (ldp_pred_csr.hip: compacted on the device, written into pinned host memory,
    replayed entry by entry -- plink2_ld.cc:1093-1097 writes a removed bit where it is decided); `pred_csr` 0 copies the dense rows as rounds 1-5
    did; a CSR buffer that is too small (`csr_capacity`) makes the run fall back to the dense rows.  Same prune set, both replay orders, several
    launch groups, a second run on the same engine."""
    pkg = gpu_pkg
    m, n = 2600, 700
    raw = T.synth_raw_codes(m, n, seed=77, missing_rate=0.0, ld_copy_prob=0.7, redraw=0.1)
    chr_idx, bps = make_positions(m, 3, 11)
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 30000, 1, True, 0.3, order)
    assert 50 < want.sum() < m - 50
    packed = T.pack_2bit(raw)
    for options in ({}, {"pred_csr": 0}, {"csr_capacity": 16}):
        eng = pkg.LdPruneEngine(n, 30000, 1, True, 0.3, order=order, device=0)
        for name, value in options.items():
            eng.set_option(name, value)
        eng.set_variants(chr_idx, bps)
        eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
        got = eng.run()
        again = eng.run()
        c = eng.counters()
        eng.close()
        assert np.array_equal(got, want), options
        assert np.array_equal(again, want), options
        assert c["pred_true"] > 16


def test_early_termination_late_correlation(gpu_pkg):
    """Adversarial layout: pairs that look unrelated over the first 45 % of the samples and are identical over
    the rest.  A bound that extrapolated from the visited samples would drop them; the remainder bound keeps them."""
    n, m = 6000, 260
    rng = np.random.default_rng(11)
    raw = T.synth_raw_codes(m, n, seed=12, missing_rate=0.0)
    cut = int(0.45 * n)
    for v in range(1, m, 2):
        raw[v, cut:] = raw[v - 1, cut:]
        raw[v, :cut] = rng.permutation(raw[v, :cut])
    chr_idx = np.zeros(m, dtype=np.uint32)
    packed = T.pack_2bit(raw)
    inv, mf, _ = T.oracle_prepare(raw)
    for r2 in (0.2, 0.3):
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, np.arange(m, dtype=np.uint32), mf, 100, 1, False, r2, 2)
        on, c1 = _run_early_exit(gpu_pkg, packed, n, chr_idx, None, 100, 1, False, r2, 2, True)
        off, c0 = _run_early_exit(gpu_pkg, packed, n, chr_idx, None, 100, 1, False, r2, 2, False)
        assert want.sum() >= m // 4
        assert np.array_equal(off, want)
        assert np.array_equal(on, want)
        assert c1["pred_true"] == c0["pred_true"]


def test_early_termination_missing_calls_adversarial(gpu_pkg):
    """Missing calls placed where they hurt an interval bound most: rare variants whose carriers sit in the last
    third of the samples, partners whose missing calls cover exactly those carriers (so the pairwise-complete
    variance collapses), plus pairs that only become correlated late."""
    n, m = 6000, 240
    rng = np.random.default_rng(5)
    raw = T.synth_raw_codes(m, n, seed=33, missing_rate=0.0)
    tail = np.arange(int(0.7 * n), n)
    for v in range(0, m, 6):
        raw[v] = 0
        carriers = rng.choice(tail, size=25, replace=False)
        raw[v, carriers] = 1                         # rare variant, carriers late
        raw[v + 1, carriers[:20]] = 3                # partner missing on most of them
        raw[v + 2] = raw[v]                          # perfect copy of the rare variant ...
        raw[v + 2, carriers[:5]] = 3                 # ... with a few of the carriers missing
        raw[v + 3, :int(0.5 * n)] = rng.permutation(raw[v + 3, :int(0.5 * n)])
        raw[v + 3, int(0.5 * n):] = raw[v + 4, int(0.5 * n):]
        raw[v + 3, rng.choice(n, size=60, replace=False)] = 3
    chr_idx = np.zeros(m, dtype=np.uint32)
    packed = T.pack_2bit(raw)
    inv, mf, _ = T.oracle_prepare(raw)
    for r2 in (0.2, 0.5):
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, np.arange(m, dtype=np.uint32), mf, 60, 1, False, r2, 2)
        on, c1 = _run_early_exit(gpu_pkg, packed, n, chr_idx, None, 60, 1, False, r2, 2, True)
        off, c0 = _run_early_exit(gpu_pkg, packed, n, chr_idx, None, 60, 1, False, r2, 2, False)
        assert np.array_equal(off, want)
        assert np.array_equal(on, want)
        assert c1["pred_true"] == c0["pred_true"]


@pytest.mark.parametrize("n,r2,miss", [(20000, 0.5, 0.002), (20000, 0.5, 0.0045), (9000, 0.2, 0.001), (50000, 0.7, 0.003)])
def test_early_termination_with_a_few_missing_calls(gpu_pkg, n, r2, miss):
    """Narrow bands (config 2's shape: ~70 variants per window) whose rows miss a few calls take the single-form kernel with the
    interval epilogue (4.1d).  Same prune set and true-predicate count with early termination on and off, as the six-product
    kernel (which does stop at checkpoints) and as the oracle -- with pairs that only become correlated in the last 40 % of the
    samples, and with the missing calls of some rows sitting on their partners' rare carriers."""
    pkg = gpu_pkg
    m = 1500
    rng = np.random.default_rng(n + int(1000 * r2))
    raw = T.synth_raw_codes(m, n, seed=n % 89 + 5, missing_rate=miss, ld_copy_prob=0.6, redraw=0.08)
    cut = int(0.6 * n)
    for v in range(10, m, 9):          # late LD: unrelated over the first 60 %, a copy of the neighbour afterwards
        raw[v, :cut] = rng.permutation(raw[v, :cut])
        raw[v, cut:] = raw[v - 1, cut:]
    tail = np.arange(int(0.75 * n), n)
    for v in range(5, m - 2, 37):      # a rare variant whose carriers come late, a partner missing on a handful of them
        raw[v] = 0
        carriers = rng.choice(tail, size=30, replace=False)
        raw[v, carriers] = 1
        raw[v + 1] = raw[v]
        raw[v + 1, carriers[:4]] = 3
    chr_idx = (np.arange(m) // 750).astype(np.uint32)
    bps = (10000 + 2875 * (np.arange(m) % 750)).astype(np.uint32)
    packed = T.pack_2bit(raw)
    on, c1 = _run_early_exit(pkg, packed, n, chr_idx, bps, 200000, 1, True, r2, 2, True)
    off, c0 = _run_early_exit(pkg, packed, n, chr_idx, bps, 200000, 1, True, r2, 2, False)
    assert c1["route_sparse_launches"] > 0 and c1["route_general_launches"] == 0 and c1["route_complete_launches"] == 0
    assert c0["mfma_skipped_product_stages"] == 0
    assert np.array_equal(on, off) and c1["pred_true"] == c0["pred_true"] > 0
    six, c6 = _run_sparse_bp(pkg, packed, n, chr_idx, bps, r2)
    assert np.array_equal(on, six) and c6["route_general_launches"] > 0
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 200000, 1, True, r2, 2)
    assert np.array_equal(on, want)


def _run_sparse_bp(pkg, packed, n, chr_idx, bps, r2):
    eng = pkg.LdPruneEngine(n, 200000, 1, True, r2, order=2, device=0)
    eng.set_option("pair_sparse", 0)
    eng.set_option("pair_four", 0)
    eng.set_variants(chr_idx, bps)
    eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
    removed = eng.run()
    ctr = eng.counters()
    eng.close()
    return removed, ctr


@pytest.mark.parametrize("miss", [0.0, 0.003])
def test_early_termination_wide_window(gpu_pkg, miss):
    """A 400-variant window: five blocks per J-tile, the far ones (their own second-variant rows, d0 >= 32) stop at the
    first checkpoint and leave the k-loop as a whole; near ones shrink their tile."""
    n, m = 3000, 1300
    raw = T.synth_raw_codes(m, n, seed=71, missing_rate=miss)
    chr_idx = np.zeros(m, dtype=np.uint32)
    chr_idx[900:] = 1
    bps = np.arange(m, dtype=np.uint32)
    packed = T.pack_2bit(raw)
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, 400, 7, False, 0.5, 2)
    on, c1 = _run_early_exit(gpu_pkg, packed, n, chr_idx, bps, 400, 7, False, 0.5, 2, True)
    off, c0 = _run_early_exit(gpu_pkg, packed, n, chr_idx, bps, 400, 7, False, 0.5, 2, False)
    assert np.array_equal(off, want)
    assert np.array_equal(on, want)
    assert c1["pred_true"] == c0["pred_true"]
    if c1["mfma_product_stages"] and miss == 0.0:   # complete data runs on the matrix-pipe kernel (block products x stages)
        assert c1["mfma_skipped_product_stages"] > 0.3 * c1["mfma_product_stages"]
    elif c1["route_general_launches"] + c1["route_sparse_launches"] == 0:   # (the popcount kernels: a run with the matrix pipe switched off)
        assert c1["early_exit_unit_chunks"] > 0.3 * c1["tile_unit_chunks"]


def test_randomised_differential(gpu_pkg):
    """tests/fuzz_parity.py: 80 random shapes (chunk-boundary sample counts, count/kb windows, steps, both orders,
    thresholds 0.02-0.95, 0-20 % missing, LD blocks, degenerate rows) against the oracle, early termination on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(20260925)
    for k in range(80):
        ok, desc = fz.one_case(gpu_pkg, rng, k)
        assert ok, desc


@pytest.mark.parametrize("beyond,missing", [(0, 0.0), (1, 0.0), (257, 0.01)])
def test_sample_counts_around_the_matrix_pipe_limit(gpu_pkg, beyond, missing):
    """f32 accumulators stay integer-exact up to kMfMaxFounders samples (ldp_matrix_pipe_max_founders()); beyond that the popcount
    kernels take the pairs.  Either side of the limit the tile kernels' six-tuples must equal the one-wave-per-pair reference kernel's
    (ldp_pair_stats: an independent, trivially simple kernel), and the counters must say which family ran."""
    import torch
    pkg = gpu_pkg
    limit = pkg.matrix_pipe_max_founders()
    n = limit + beyond
    m = 40
    stride = (n + 3) // 4
    buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
    pkg.synth_genotypes_device(77, 0, m, n, missing, buf.data_ptr(), stride)
    torch.cuda.synchronize()
    eng = pkg.LdPruneEngine(n, 12, 1, False, 0.2, device=0)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    eng.load_genotypes_device(0, m, buf.data_ptr(), stride, pkg.LDP_GENO_REF)
    removed, stats = eng.run_with_stats()
    lo, cand = eng.band()
    first, second = [], []
    for j in range(m):
        for i in range(int(lo[j]), j):
            first.append(i)
            second.append(j)
    assert len(first) == cand == len(stats)
    ref = eng.pair_stats(first, second)
    assert np.array_equal(stats, ref)
    assert int(stats["nm"].max()) <= n and int(stats["nm"].min()) > 0.9 * n
    ctr = eng.counters()
    if n <= limit:
        assert ctr["mfma_block_products"] > 0
    else:
        assert ctr["mfma_block_products"] == 0
    eng.close()
    del buf


def test_four_product_operands_at_their_founder_limit(gpu_pkg):
    """The allele-count / missing-flag operands of the four-product form are exact while 9 N < 2^24 (kMfGuMaxFounders = 1,800,000,
    ldp_mfma_device.h); engines with more founders multiply x and n.  At the limit and just beyond it: the same prune set and the same
    number of true predicates as the six-product form, from rows with 3 % missing calls and planted LD."""
    import torch
    pkg = gpu_pkg
    m = 96
    for n in (1800000, 1800001):
        stride = (n + 3) // 4
        buf = torch.empty((m, stride), dtype=torch.uint8, device="cuda")
        pkg.synth_genotypes_device(91, 0, m, n, 0.03, buf.data_ptr(), stride)
        torch.cuda.synchronize()
        outs = []
        for opts in ({"pair_sparse": 0}, {"pair_sparse": 0, "pair_gu": 0}, {"pair_sparse": 0, "pair_four": 0}):
            eng = pkg.LdPruneEngine(n, 40, 1, False, 0.2, order=2, device=0)
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.set_variants(np.zeros(m, dtype=np.uint32), None)
            eng.load_genotypes_device(0, m, buf.data_ptr(), stride, pkg.LDP_GENO_REF)
            removed = eng.run()
            ctr = eng.counters()
            outs.append((removed, ctr["pred_true"], ctr["route_general_launches"]))
            eng.close()
        assert outs[0][2] > 0 and outs[0][1] > 0 and removed.sum() > 0
        assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[1][0], outs[2][0])
        assert outs[0][1] <= outs[2][1] and outs[0][1] == outs[1][1]     # (early termination retires products whose pairs are false; both operand sets alike)
        del buf


@pytest.mark.parametrize("n,offset", [(1000, 3), (1001, 0), (50000, 12)])
def test_rows_read_from_a_file_descriptor(gpu_pkg, tmp_path, n, offset):
    """ldp_load_genotypes_fd: fixed-width rows pread() straight from a file into the pinned ring -- the same records, maj_freq and
    prune set as the same rows handed over in host memory (row lengths that are and are not a multiple of four bytes; a header in
    front of row 0, as a .bed / fixed-width .pgen has)."""
    pkg = gpu_pkg
    m = 700
    raw = T.synth_raw_codes(m, n, seed=n + offset, missing_rate=0.01)
    chr_idx, bps = make_positions(m, 3, 11)
    packed = T.pack_2bit(raw)
    rb = (n + 3) // 4                                   # rows as a .bed / fixed-width .pgen stores them: ceil(n / 4) bytes, no padding
    rows = np.ascontiguousarray(packed.view(np.uint8).reshape(m, -1)[:, :rb])
    path = str(tmp_path / "rows.bin")
    with open(path, "wb") as f:
        f.write(b"\x6c" * offset)
        f.write(rows.tobytes())
    res = []
    for via_fd in (False, True):
        eng = pkg.LdPruneEngine(n, 20000, 1, True, 0.3, device=0)
        eng.set_variants(chr_idx, bps)
        if via_fd:
            fd = os.open(path, os.O_RDONLY)
            try:
                eng.load_genotypes_fd(0, 300, fd, offset, rb, pkg.LDP_GENO_REF)   # two calls: the second starts mid-file
                eng.load_genotypes_fd(300, m - 300, fd, offset + 300 * rb, rb, pkg.LDP_GENO_REF)
            finally:
                os.close(fd)
        else:
            eng.load_genotypes_host(0, packed, pkg.LDP_GENO_REF)
        res.append((eng.variant_recs().copy(), eng.maj_freqs().copy(), eng.run().copy()))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][2].sum() > 10
    # a file that ends early is an error, not a short read silently taken for rows
    eng = pkg.LdPruneEngine(n, 20000, 1, True, 0.3, device=0)
    eng.set_variants(chr_idx, bps)
    fd = os.open(path, os.O_RDONLY)
    try:
        with pytest.raises(pkg.LdpError):
            eng.load_genotypes_fd(0, m, fd, offset + 64 * rb, rb, pkg.LDP_GENO_REF)
    finally:
        os.close(fd)
    eng.close()
