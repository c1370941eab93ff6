"""Multi-GPU plumbing: one process per GPU, subcontigs LPT-sharded over ranks (no data-path collective),
and ONE exchange at the end -- an all_gather of every rank's removed-bit segment (its owned variants in
local order, padded to the longest segment) over torch.distributed ("nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The segments are tiny (M/8 bytes in total), so the exchange is
latency-bound; a padded all_gather is the allgatherv the design calls for."""
import numpy as np


def owned_segments(subcontigs, owner, rank):
    """[(first_variant, length, local_offset)] of the subcontigs `rank` owns, in file order."""
    seg, off = [], 0
    for (ln, first), o in zip(subcontigs, owner):
        if o == rank:
            seg.append((first, ln, off))
            off += ln
    return seg


def segment_words(subcontigs, owner, world):
    longest = 0
    for r in range(world):
        longest = max(longest, sum(ln for (ln, _), o in zip(subcontigs, owner) if o == r))
    return (longest + 63) // 64


def pack_local_bits(removed, subcontigs, owner, rank, words):
    seg = owned_segments(subcontigs, owner, rank)
    bits = np.concatenate([removed[first:first + ln] for first, ln, _ in seg]) if seg else np.zeros(0, dtype=bool)
    out = np.zeros(words * 8, dtype=np.uint8)
    pb = np.packbits(bits, bitorder="little")
    out[:len(pb)] = pb
    return out.view(np.int64)


def unpack_all(gathered, subcontigs, owner, variant_ct):
    """gathered[r] = int64 words of rank r's segment -> global removed mask."""
    full = np.zeros(variant_ct, dtype=bool)
    for r, words in enumerate(gathered):
        bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")
        pos = 0
        for (ln, first), o in zip(subcontigs, owner):
            if o == r:
                full[first:first + ln] = bits[pos:pos + ln]
                pos += ln
    return full


def allgather_removed(removed, subcontigs, owner, rank, world, variant_ct, device="cpu"):
    """removed: this rank's bool mask over ALL variants (only owned bits set).  Returns the global mask."""
    import torch
    import torch.distributed as dist
    words = segment_words(subcontigs, owner, world)
    mine = torch.from_numpy(pack_local_bits(removed, subcontigs, owner, rank, words).copy()).to(device)
    out = [torch.zeros(words, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(out, mine)
    return unpack_all([t.cpu().numpy() for t in out], subcontigs, owner, variant_ct)


def allgather_bitmaps(bitmap_u64, world, device="cpu"):
    """The exchange bench.py times: every rank contributes its global-index removed bitmap (uint64 words, only the
    bits of its own subcontigs set); one all_gather, then an OR over the `world` disjoint pieces on the device.
    Returns the combined bitmap as a torch int64 tensor on `device` (no host round trip)."""
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(bitmap_u64.view(np.int64)).to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    acc = out[0]
    for t in out[1:]:
        acc = torch.bitwise_or(acc, t)
    return acc


def bitmap_to_mask(words, variant_ct):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:variant_ct].astype(bool)
