"""world_size-2 test of the sharded path on CPU (gloo): each rank plans the whole variant table, owns the
subcontigs LPT gives it, produces the removed bits of its shard (predicates come from the oracle and go
through ldp_debug_replay_pairs, since there is no GPU here) and the ranks exchange their segments with the
same all_gather bench.py uses over RCCL.  Result must equal the unsharded oracle prune set on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

import ldtools as T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    sys.path.insert(0, os.path.join(repo, "tests"))
    import __graft_entry__ as ge
    import importlib
    from test_host_logic import make_positions, oracle_band_predicates, recs_from_vaggs
    pkg = ge.load_package()
    distmod = importlib.import_module("plink_ng_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, n, seed, window, step, is_bp, r2, order, miss = case
        raw = T.synth_raw_codes(m, n, seed, missing_rate=miss)
        chr_idx, bps = make_positions(m, 6, seed + 100)
        inv, mf, _ = T.oracle_prepare(raw)
        want, _ = T.oracle_indep_pairwise(inv, n, chr_idx, bps, mf, window, step, is_bp, r2, order)
        eng = pkg.LdPruneEngine(n, window, step, is_bp, r2, order=order)
        eng.set_variants(chr_idx, bps)
        lo, _ = eng.band()
        first, second, vaggs = oracle_band_predicates(inv, n, lo, r2)
        subs = eng.subcontigs()
        owner = eng.set_shard(rank, world)
        eng.debug_set_variant_recs(recs_from_vaggs(pkg, vaggs, n, m))
        eng.set_maj_freqs(0, mf)
        mine = eng.debug_replay_pairs(first, second)
        # only owned variants may be reported
        owned = np.zeros(m, dtype=bool)
        for (ln, f0), o in zip(subs, owner):
            if o == rank:
                owned[f0:f0 + ln] = True
        assert not (mine & ~owned).any()
        full = distmod.allgather_removed(mine, subs, owner, rank, world, m, device="cpu")
        # the bitmap form bench.py uses
        bm = np.zeros((m + 63) // 64 + 1, dtype=np.uint64)
        pb = np.packbits(mine, bitorder="little")
        bm.view(np.uint8)[:len(pb)] = pb
        full2 = distmod.bitmap_to_mask(distmod.allgather_bitmaps(bm, world, device="cpu").numpy(), m)
        ret[rank] = bool(np.array_equal(full, want)) and bool(np.array_equal(full2, want)) and (int(owned.sum()) > 0)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [
    (500, 60, 11, 12000, 1, True, 0.2, 2, 0.02),
    (500, 60, 12, 40, 7, False, 0.3, 1, 0.0),
])
def test_two_rank_shard_and_allgather(pkg, case):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, case, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {0: True, 1: True}


def test_segment_pack_roundtrip(pkg):
    import importlib
    distmod = importlib.import_module("plink_ng_amd.dist")
    rng = np.random.default_rng(1)
    subs = [(100, 0), (37, 100), (260, 150), (5, 500)]
    owner = np.array([0, 1, 2, 1])
    m = 505
    want = rng.random(m) < 0.4
    inside = np.zeros(m, dtype=bool)
    for ln, first in subs:
        inside[first:first + ln] = True
    want &= inside  # variants outside every subcontig are never removed
    words = distmod.segment_words(subs, owner, 3)
    gathered = []
    for r in range(3):
        mine = np.zeros(m, dtype=bool)
        for first, ln, _ in distmod.owned_segments(subs, owner, r):
            mine[first:first + ln] = want[first:first + ln]
        gathered.append(distmod.pack_local_bits(mine, subs, owner, r, words))
    assert np.array_equal(distmod.unpack_all(gathered, subs, owner, m), want)
