"""CPU-only tests of the host logic behind the C ABI: subcontig split, window/band planning, LPT
sharding and the greedy replay.  The pair predicate bits are produced here by the ORACLE and pushed
through ldp_debug_replay_pairs(), so no GPU is needed; the result must equal the oracle's own
end-to-end --indep-pairwise."""
import ctypes
import os

import numpy as np
import pytest

import ldtools as T


def make_positions(m, nchr, seed, spacing=300, big_gap_prob=0.01, big_gap=400000):
    rng = np.random.default_rng(seed)
    per = (m + nchr - 1) // nchr
    chr_idx = (np.arange(m) // per).astype(np.uint32)
    bps = np.zeros(m, dtype=np.uint32)
    for c in range(nchr):
        idx = np.where(chr_idx == c)[0]
        gaps = rng.integers(0, 2 * spacing, size=len(idx))  # 0 allowed: duplicate positions
        gaps = np.where(rng.random(len(idx)) < big_gap_prob, gaps + big_gap, gaps)
        bps[idx] = 1000 + np.cumsum(gaps)
    return chr_idx, bps


def oracle_band_predicates(inv, n, lo, thresh_r2):
    """All candidate pairs (lo[j] <= i < j) whose oracle predicate is true."""
    lib = T.oracle()
    hom, r2h, vaggs = T.oracle_split(inv, n)
    thr = lib.ldo_prune_thresh(thresh_r2)
    first, second = [], []
    for j in range(len(lo)):
        for i in range(int(lo[j]), j):
            st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
            if lib.ldo_exceeds(ctypes.byref(st), thr):
                first.append(i)
                second.append(j)
    return np.array(first, dtype=np.uint32), np.array(second, dtype=np.uint32), vaggs


def recs_from_vaggs(pkg, vaggs, n, m):
    lib = T.oracle()
    recs = np.zeros(m, dtype=pkg.VARIANT_REC_DTYPE)
    for v in range(m):
        recs[v]["nm_ct"] = vaggs[v].nm_ct
        recs[v]["sum"] = vaggs[v].sum
        recs[v]["ssq"] = vaggs[v].ssq
        mono = lib.ldo_is_monomorphic(ctypes.byref(vaggs[v]))
        recs[v]["flags"] = (2 if mono else 0) | (4 if vaggs[v].nm_ct != n else 0)
    return recs


CASES = [
    # m, n, seed, window, step, is_bp, r2, order, missing
    (400, 70, 1, 50, 5, False, 0.2, 2, 0.0),
    (400, 70, 2, 50, 5, False, 0.2, 1, 0.0),
    (400, 65, 3, 30, 1, False, 0.5, 2, 0.05),
    (400, 65, 4, 15000, 1, True, 0.2, 2, 0.05),
    (400, 65, 5, 15000, 1, True, 0.5, 1, 0.03),
    (500, 33, 6, 40, 40, False, 0.1, 2, 0.1),
    (500, 33, 7, 2, 1, False, 0.3, 2, 0.0),
    (500, 40, 8, 25, 7, False, 0.3, 1, 0.1),
]


@pytest.mark.parametrize("case", CASES)
def test_replay_matches_oracle(pkg, case):
    m, n, seed, window, step, is_bp, r2, order, miss = case
    raw = T.synth_raw_codes(m, n, seed, missing_rate=miss)
    chr_idx, bps = make_positions(m, 3, seed + 100)
    inv, mf, _ = T.oracle_prepare(raw)
    want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, r2, order)

    eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order)
    eng.set_variants(chr_idx, bps)
    lo, cand = eng.band()
    assert cand == int(np.sum(np.arange(m) - lo))
    first, second, vaggs = oracle_band_predicates(inv, n, lo, r2)
    eng.debug_set_variant_recs(recs_from_vaggs(pkg, vaggs, n, m))
    eng.set_maj_freqs(0, mf)
    got = eng.debug_replay_pairs(first, second)
    assert np.array_equal(got, want), "removed sets differ: %d vs %d" % (got.sum(), want.sum())
    # the same in instalments (the streaming replay of a run resumes every subcontig at window-batch boundaries)
    for steps in (2, 7, 1000):
        eng.set_option("replay_steps", steps)
        again = eng.debug_replay_pairs(first, second)
        assert np.array_equal(again, want), steps
    eng.set_option("replay_steps", 0)
    eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("is_bp,window", [(True, 5000), (True, 100000), (False, 7), (False, 1000)])
def test_subcontig_split_matches_oracle(pkg, seed, is_bp, window):
    m = 3000
    chr_idx, bps = make_positions(m, 5, seed, spacing=200, big_gap_prob=0.02, big_gap=20000)
    # a chromosome with a single variant, and one of length 2
    chr_idx = np.concatenate([chr_idx, [5, 6, 6]]).astype(np.uint32)
    bps = np.concatenate([bps, [10, 10, 2000000]]).astype(np.uint32)
    want, wmax = T.oracle_subcontig_split(chr_idx, bps if is_bp else None, window)
    eng = pkg.LdPruneEngine(100, window, 1, is_bp, 0.2)
    eng.set_variants(chr_idx, bps)
    got = eng.subcontigs()
    assert got == want
    assert eng.counters()["window_max"] == wmax
    eng.close()


def test_band_is_the_window_rule_in_kb_mode(pkg):
    """kb mode: candidates are exactly the in-subcontig pairs with bp[j]-bp[i] <= W."""
    m, W = 2000, 7000
    chr_idx, bps = make_positions(m, 2, 9, spacing=150, big_gap_prob=0.01, big_gap=30000)
    eng = pkg.LdPruneEngine(100, W, 1, True, 0.2)
    eng.set_variants(chr_idx, bps)
    lo, _ = eng.band()
    subs = eng.subcontigs()
    eng.close()
    in_sub = np.full(m, -1)
    for k, (ln, first) in enumerate(subs):
        in_sub[first:first + ln] = k
    for j in range(m):
        if in_sub[j] < 0:
            assert lo[j] == j
            continue
        first = subs[in_sub[j]][1]
        cand = [i for i in range(first, j) if int(bps[j]) - int(bps[i]) <= W]
        want_lo = cand[0] if cand else j
        # the reference's window iterator never reaches back further than the rule, and never less
        assert lo[j] == want_lo, (j, lo[j], want_lo)


def test_count_mode_band_follows_window_iterator(pkg):
    """`50 5`: variant 50 is compared with 5..49 but never with 4."""
    m = 200
    eng = pkg.LdPruneEngine(100, 50, 5, False, 0.2)
    eng.set_variants(np.zeros(m, dtype=np.uint32), None)
    lo, _ = eng.band()
    eng.close()
    assert list(lo[:50]) == [0] * 50
    assert lo[50] == 5 and lo[54] == 5 and lo[55] == 10


def test_lpt_shard_partition(pkg):
    m = 5000
    chr_idx, bps = make_positions(m, 11, 5, spacing=100, big_gap_prob=0.005, big_gap=50000)
    engs = []
    owners = None
    for r in range(4):
        e = pkg.LdPruneEngine(64, 3000, 1, True, 0.2)
        e.set_variants(chr_idx, bps)
        o = e.set_shard(r, 4)
        if owners is None:
            owners = o
        assert np.array_equal(o, owners)  # every rank computes the same assignment
        engs.append(e)
    subs = engs[0].subcontigs()
    loads = np.zeros(4, dtype=np.int64)
    for (ln, _), o in zip(subs, owners):
        loads[o] += ln
    assert loads.max() - loads.min() <= max(ln for ln, _ in subs)
    assert sum(e.counters()["owned_subcontig_ct"] for e in engs) == len(subs)
    for e in engs:
        e.close()


def test_parameter_validation(pkg):
    with pytest.raises(pkg.LdpError):
        pkg.LdPruneEngine(1, 50, 5, False, 0.2)          # < 2 founders (plink2_ld.cc:2537)
    with pytest.raises(pkg.LdpError):
        pkg.LdPruneEngine(100, 50, 51, False, 0.2)       # step > window (plink2.cc:7285)
    with pytest.raises(pkg.LdpError):
        pkg.LdPruneEngine(100, 50000, 2, True, 0.2)      # kb window needs step 1 (plink2.cc:7290)
    with pytest.raises(pkg.LdpError):
        pkg.LdPruneEngine(100, 50, 5, False, 1.0)        # r2 must be < 1 (plink2.cc:7303)
    with pytest.raises(pkg.LdpError) as ei:
        pkg.LdPruneEngine(1 << 30, 50, 5, False, 0.2)    # plink2_ld.cc:1122
    assert ei.value.code == pkg.LDP_ERR_UNSUPPORTED
    e = pkg.LdPruneEngine(100, 5000, 1, True, 0.2)
    with pytest.raises(pkg.LdpError):
        e.set_variants(np.array([0, 0, 0]), np.array([5, 3, 9]))  # unsorted positions
    e.close()


def test_no_cpu_fallback_without_gpu(pkg):
    """The product path must fail loudly when no HIP device is usable."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present; the no-device path cannot be exercised here")
    e = pkg.LdPruneEngine(64, 50, 5, False, 0.2)
    e.set_variants(np.zeros(10, dtype=np.uint32), None)
    with pytest.raises(pkg.LdpError) as ei:
        e.load_genotypes_host(0, np.zeros((10, 16), dtype=np.uint8))
    assert ei.value.code == pkg.LDP_ERR_GPU
    with pytest.raises(pkg.LdpError) as ei:
        e.run()
    assert ei.value.code == pkg.LDP_ERR_GPU
    e.close()


def test_vcor_window_plan_matches_reference_pair_set(pkg):
    """The windowed --r2-unphased plan (ldp_set_variants_vcor; UpdateVcorWindow, plink2_ld.cc:10984-11023) against the
    pair sets of .vcor tables the reference wrote with --ld-window-r2 0: planned band = written pairs + the pairs whose
    r^2 is undefined (NaN never passes the filter).  Host logic only: no GPU."""
    import ldtools as T
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pgen", "vcor_windows.npz"))
    raw = z["raw"]
    m, n = raw.shape
    chr_idx = np.unique(z["chroms"], return_inverse=True)[1].astype(np.uint32)
    inv, mf, _ = T.oracle_prepare(raw)
    hom, r2h, vaggs = T.oracle_split(inv, n)
    for k, s in enumerate(z["settings"]):
        kb, cnt = str(s).split("|")
        bp_radius = int(float(kb) * 1000 * (1 + T.K_SMALL_EPSILON))
        var_radius = (int(cnt) - 1) if cnt else 0x7fffffff
        eng = pkg.LdPruneEngine(n, 2, 1, False, 0.5, device=-1)
        eng.set_variants_vcor(chr_idx, z["bps"], bp_radius, var_radius)
        lo, cand = eng.band()
        eng.close()
        planned = {(i, j) for j in range(m) for i in range(int(lo[j]), j)}
        assert len(planned) == cand
        written = {(int(a), int(b)) for a, b in z["pairs_%d" % k]}
        assert written <= planned, (k, sorted(written - planned)[:5])
        for i, j in planned - written:
            st = T.oracle_pair_stats(hom, r2h, vaggs, n, i, j)
            cov, v1, v2 = T.oracle_r2(st)
            assert (st.nm == 0) or (v1 * v2 == 0.0), (k, i, j)  # ComputeR2: undefined -> NaN -> not written


def _mfma_plan_coverage(pkg, m, nchr, seed, window, step, is_bp, spacing, rank=0, world=1):
    """Every candidate pair must be owned by exactly one live 32 x 32 block product of the matrix-pipe plan, row-block
    slots must point at the blocks the products read, and a workgroup stages at most 16 distinct row-blocks."""
    chr_idx, bps = make_positions(m, nchr, seed, spacing=spacing)
    eng = pkg.LdPruneEngine(100, window, step, is_bp, 0.2, device=-1)
    eng.set_variants(chr_idx, bps)
    if world > 1:
        eng.set_shard(rank, world)
    wgs, lo = eng.debug_mfma_plan()
    eng.close()
    n_local = len(lo)
    span = np.arange(n_local, dtype=np.int64) - lo.astype(np.int64)
    pair_off = np.concatenate([[0], np.cumsum(np.maximum(span, 0))])
    count = np.zeros(int(pair_off[-1]), dtype=np.int32)
    prev_hi = 0
    for wg in wgs:
        n_rb, j_lo, j_hi = int(wg[0]) & 0x3fffffff, int(wg[1]), int(wg[2])
        all_diag = bool((int(wg[0]) >> 30) & 1)
        rb = wg[3:19].astype(np.int64)
        assert 1 <= n_rb <= 16
        assert np.all(np.diff(rb[:n_rb]) > 0)
        assert np.all(rb[:n_rb] < n_local)
        assert j_lo < j_hi <= n_local
        for w in range(4):
            base = 19 + 11 * w
            jv, vv, jend, mask = int(np.int32(wg[base])), int(np.int32(wg[base + 1])), int(wg[base + 2]), int(wg[base + 3])
            slot = wg[base + 4:base + 11].astype(np.int64)
            if jv < 0:
                assert mask == 0 or True
                continue
            assert mask != 0 and j_lo <= jv < jend <= j_hi
            diag = (vv + 96 == jv)
            assert diag or not all_diag   # (the single-form kernel's workgroups hold diagonal wave items only)
            if diag:  # the kernel reads V3 / V4 through the J0 / J1 slots there
                assert (not mask & 0x48) or rb[slot[0]] == jv
                assert (not mask & 0x80) or rb[slot[1]] == jv + 32
            for p in range(8):
                if not (mask >> p) & 1:
                    continue
                q, k = p >> 2, (p & 3) + (p >> 2)
                jfirst, vfirst = jv + 32 * q, vv + 32 * k
                assert rb[slot[q]] == jfirst
                assert vfirst >= 0
                assert rb[slot[2 + k]] == vfirst and slot[2 + k] < n_rb  # (on the diagonal V3 / V4 are J0 / J1 again)
                for j in range(jfirst, min(jfirst + 32, jend)):
                    a, b = max(vfirst, int(lo[j])), min(vfirst + 32, j)
                    if a < b:
                        count[pair_off[j] + a - lo[j]:pair_off[j] + b - lo[j]] += 1
        prev_hi = max(prev_hi, j_hi)
    assert np.all(count == 1), "pairs covered %d..%d times" % (count.min() if len(count) else 1, count.max() if len(count) else 1)
    return len(wgs), len(count)


def _wide_plan_coverage(pkg, m, nchr, seed, window, step, is_bp, spacing, min_reach, rank=0, world=1, big_gap_prob=0.01):
    """The 8 x 8 tile plan of the wide-band kernel (ldp_pair_wide.hip): over the subcontigs that take it every candidate pair is
    owned by exactly one live product of exactly one tile, tiles are aligned to the subcontig start in both directions, and the
    parallelogram workgroups of those subcontigs are marked as standing by."""
    chr_idx, bps = make_positions(m, nchr, seed, spacing=spacing, big_gap_prob=big_gap_prob)
    eng = pkg.LdPruneEngine(100, window, step, is_bp, 0.2, device=-1)
    eng.set_option("wide_min_reach", min_reach)
    eng.set_variants(chr_idx, bps)
    if world > 1:
        eng.set_shard(rank, world)
    wgs, lo = eng.debug_mfma_plan()
    tiles = eng.debug_wide_plan()
    eng.close()
    n_local = len(lo)
    span = np.arange(n_local, dtype=np.int64) - lo.astype(np.int64)
    pair_off = np.concatenate([[0], np.cumsum(np.maximum(span, 0))])
    count = np.zeros(int(pair_off[-1]), dtype=np.int32)
    wide_j = np.zeros(n_local, dtype=bool)   # second variants whose subcontig has the wide plan
    for wg in wgs:
        if int(wg[0]) >> 31:
            wide_j[int(wg[1]):int(wg[2])] = True
    prev = (-1, -1)
    for jv, vv, jend, mlo, mhi in tiles.astype(np.int64):
        mask = int(mlo) | (int(mhi) << 32)
        assert mask and 0 <= vv <= jv < jend <= n_local and (jv - vv) % 256 == 0
        assert (jv, vv) > prev   # J tile by J tile, V tiles ascending
        prev = (jv, vv)
        for a in range(8):
            for b in range(8):
                if not (mask >> (8 * a + b)) & 1:
                    continue
                jfirst, vfirst = jv + 32 * a, vv + 32 * b
                assert vfirst <= jfirst
                for j in range(jfirst, min(jfirst + 32, jend)):
                    assert wide_j[j]
                    x, y = max(vfirst, int(lo[j])), min(vfirst + 32, j)
                    if x < y:
                        count[pair_off[j] + x - lo[j]:pair_off[j] + y - lo[j]] += 1
    for j in range(n_local):
        if span[j] > 0:
            want = 1 if wide_j[j] else 0
            assert np.all(count[pair_off[j]:pair_off[j + 1]] == want), (j, want)
    return len(tiles), int(wide_j.sum())


@pytest.mark.parametrize("m,nchr,seed,window,step,is_bp,spacing,min_reach,gap_prob,expect", [
    (5000, 2, 2, 200000, 1, True, 300, 12, 0.0, "all"),    # ~670 per window = 21 row-blocks: both subcontigs wide
    (9000, 3, 11, 520000, 1, True, 300, 12, 0.0, "all"),   # config 3's density: ~1,730 per window = 54 row-blocks
    (5000, 2, 2, 200000, 1, True, 300, 12, 0.01, None),    # the same density cut into subcontigs of every length by gaps
    (3000, 3, 1, 20000, 1, True, 300, 12, 0.01, "none"),   # ~67 per window: nothing is wide
    (3000, 3, 1, 20000, 1, True, 300, 0, 0.01, "all"),     # ... unless forced: narrow bands through the tile plan
    (2000, 7, 3, 50, 5, False, 300, 1, 0.01, None),        # count windows with a step
    (1500, 40, 4, 15000, 1, True, 200, 2, 0.01, None),     # many short subcontigs, some wide by this measure, some not
    (700, 1, 5, 1000000, 1, True, 100, 12, 0.0, "all"),    # one window spans everything
    (257, 1, 6, 100000, 1, True, 100, 0, 0.0, "all"), (256, 1, 7, 100000, 1, True, 100, 0, 0.0, "all"), (33, 1, 8, 100000, 1, True, 100, 0, 0.0, "all"),
])
def test_wide_plan_covers_every_candidate_pair_once(pkg, m, nchr, seed, window, step, is_bp, spacing, min_reach, gap_prob, expect):
    n_tiles, n_wide = _wide_plan_coverage(pkg, m, nchr, seed, window, step, is_bp, spacing, min_reach, big_gap_prob=gap_prob)
    if expect == "none":
        assert n_tiles == 0 and n_wide == 0
    elif expect == "all":
        assert n_tiles > 0 and (n_wide == m or (gap_prob > 0 and n_wide > 0.95 * m))  # (variants alone in a subcontig are in no plan)
    # the parallelogram plan still covers everything (it owns the launches whose rows have missing calls)
    _mfma_plan_coverage(pkg, m, nchr, seed, window, step, is_bp, spacing)


def test_wide_plan_covers_a_shard(pkg):
    for rank in range(2):
        n_tiles, _ = _wide_plan_coverage(pkg, 6000, 6, 9, 300000, 1, True, 300, 12, rank=rank, world=2, big_gap_prob=0.0)
        assert n_tiles > 0


@pytest.mark.parametrize("m,nchr,seed,window,step,is_bp,spacing", [
    (3000, 3, 1, 20000, 1, True, 300),      # ~67 variants per window: narrow band, diagonal parallelograms only
    (5000, 2, 2, 200000, 1, True, 300),     # ~670 per window: wide band, several parallelograms per J pair
    (2000, 7, 3, 50, 5, False, 300),        # count windows with a step
    (1500, 40, 4, 15000, 1, True, 200),     # many short subcontigs
    (700, 1, 5, 1000000, 1, True, 100),     # one window spans everything
    (33, 1, 6, 1000, 1, True, 100), (64, 1, 7, 100000, 1, True, 100), (65, 2, 8, 100000, 1, True, 100),
])
def test_mfma_plan_covers_every_candidate_pair_once(pkg, m, nchr, seed, window, step, is_bp, spacing):
    n_wg, n_pairs = _mfma_plan_coverage(pkg, m, nchr, seed, window, step, is_bp, spacing)
    assert n_pairs == 0 or n_wg > 0


def test_mfma_plan_covers_a_shard(pkg):
    for rank in range(2):
        _mfma_plan_coverage(pkg, 4000, 6, 9, 30000, 1, True, 300, rank=rank, world=2)


@pytest.mark.parametrize("world", [1, 2, 3, 8, 40])
def test_segment_pack_and_stitch_of_the_c_abi(pkg, world):
    """The exchange of a multi-device prune in its host steps (ldp_pack_removed_segment / ldp_stitch_removed_segments): each rank
    packs the removed bits of its own subcontigs in shard order, the padded segments -- whatever carried them -- stitch back to
    the global bitmap on every rank (plink2_ld.cc:1418-1426).  Against plink_ng_amd.dist's numpy form of the same layout, which is
    what the gloo / RCCL all_gather of bench.py moves; more ranks than subcontigs leaves empty shards."""
    import importlib
    distmod = importlib.import_module("plink_ng_amd.dist")
    m = 5000
    rng = np.random.default_rng(world)
    chr_idx, bps = make_positions(m, 22, 7 + world)
    engines, owners = [], None
    for r in range(world):
        eng = pkg.LdPruneEngine(50, 20000, 1, True, 0.5)
        eng.set_variants(chr_idx, bps)
        owners = eng.set_shard(r, world)
        engines.append(eng)
    subs = engines[0].subcontigs()
    covered = np.zeros(m, dtype=bool)   # (a variant alone in its stretch belongs to no subcontig: never pruned, never exchanged)
    for ln, f0 in subs:
        covered[f0:f0 + ln] = True
    truth = (rng.random(m) < 0.4) & covered
    words = engines[0].segment_words()
    assert words == max(1, distmod.segment_words(subs, owners, world))
    segs = np.zeros((world, words), dtype=np.uint64)
    for r, eng in enumerate(engines):
        assert eng.segment_words() == words
        owned = np.zeros(m, dtype=bool)
        for (ln, f0), o in zip(subs, owners):
            if o == r:
                owned[f0:f0 + ln] = True
        mine = truth & owned
        bm = np.zeros((m + 63) // 64 + 1, dtype=np.uint64)
        pb = np.packbits(mine, bitorder="little")
        bm.view(np.uint8)[:len(pb)] = pb
        segs[r] = eng.pack_removed_segment(bm)
        assert np.array_equal(segs[r].view(np.int64), distmod.pack_local_bits(mine, subs, owners, r, words))
    for eng in engines:
        full = eng.stitch_removed_segments(segs)
        assert np.array_equal(distmod.bitmap_to_mask(full, m), truth)
        eng.close()


def test_operand_coding_identities_of_the_matrix_pipe_kernels():
    """The integer identities the FP4 operand codings rest on (plink-ng_amd/csrc/ldp_mfma_device.h), restated in numpy on random
    rows -- the kernels themselves are pinned on the GPU, this pins the algebra where no GPU is needed.
    Complete rows: G = sum g_i g_j over the samples VISITED (padding coded 11 = 3) gives dot = G - g_bias + S_i + S_j with
    g_bias = N + 9 (visited - N).  Rows with missing calls: the accumulators of g' (the code, 3 at a missing call) and u (missing flag)
    give dot, nm, sum1, sum2 over the pairwise-complete samples through x_from_gu."""
    rng = np.random.default_rng(17)
    for n, stage in [(1000, 256), (513, 512), (77, 256), (4096, 512)]:
        visited = ((n + stage - 1) // stage) * stage
        # ---- complete rows
        gi = rng.integers(0, 3, size=n)
        gj = np.where(rng.random(n) < 0.6, gi, rng.integers(0, 3, size=n))
        pad = np.full(visited - n, 3)
        G = int(np.dot(np.concatenate([gi, pad]), np.concatenate([gj, pad])))
        xi, xj = 1 - gi, 1 - gj
        g_bias = n + 9 * (visited - n)
        assert G - g_bias + int(xi.sum()) + int(xj.sum()) == int(np.dot(xi, xj))
        # ---- rows with missing calls (code 3): g' = code, u = (code == 3)
        ci = np.where(rng.random(n) < 0.07, 3, gi)
        cj = np.where(rng.random(n) < 0.05, 3, gj)
        gpi, gpj = np.concatenate([ci, pad]), np.concatenate([cj, pad])
        ui, uj = (gpi == 3).astype(np.int64), (gpj == 3).astype(np.int64)
        P1, P4, P3, P2 = int(np.dot(gpi, gpj)), int(np.dot(ui, uj)), int(np.dot(ui, gpj)), int(np.dot(gpi, uj))
        n_pad = visited - n
        # each row's own missing count U and allele sum Z over its calls among the n real samples
        Ui, Uj = int((ci == 3).sum()), int((cj == 3).sum())
        Zi, Zj = int(ci[ci != 3].sum()), int(cj[cj != 3].sum())
        uu = P4 - n_pad
        zu, uz = P2 - 3 * P4, P3 - 3 * P4
        zz = P1 - 3 * P2 - 3 * P3 + 9 * P4
        nm = n - Ui - Uj + uu
        S1, S2 = Zi - zu, Zj - uz
        sum1, sum2, dot = nm - S1, nm - S2, nm - S1 - S2 + zz
        both = (ci != 3) & (cj != 3)
        x1, x2 = (1 - ci)[both], (1 - cj)[both]
        assert (nm, sum1, sum2, dot) == (int(both.sum()), int(x1.sum()), int(x2.sum()), int(np.dot(x1, x2)))
        # exactness in f32 accumulators: the largest accumulator stays below 2^24 at the engine limits
    assert 4 * 4000000 + 9 * 511 < 2 ** 24 and 9 * (1800000 + 511) < 2 ** 24
