"""Multi-GPU plumbing: the front-end shards by frame pair (independent units).

One process per GPU.  Rank r owns a contiguous slice of the batch; nothing is exchanged on the
data path.  The only collective is the final gather of fixed-size per-pair result records
(F, winner/count/score/n, inlier matches) — all_gather_into_tensor over RCCL on GPUs (each peer
sends its slice once over its own xGMI link), gloo on CPU in the tests.
"""
import numpy as np

REC_HEAD = 9 + 4     # F (9 f32 words) + best (4 i32 words)


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` (first n % world ranks get one more)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def record_words(K):
    """int32 words per pair record: F (9) + best (4) + one word per match slot."""
    return REC_HEAD + K


def pack_records(F, best, matches):
    """(P,9) f32, (P,4) i32, (P,K,2) i32 -> (P, 13 + K) i32 words.  F and best are bit-preserving; a match
    (query index, train index) becomes one word q | t << 16 — keypoint indices are below VSLAM_MAX_KP = 16384 —
    which halves what the gather moves over xGMI."""
    import torch
    P, K = matches.shape[0], matches.shape[1]
    assert K <= 32767
    m = matches.reshape(P, K, 2)
    packed = m[:, :, 0] | (m[:, :, 1] << 16)
    return torch.cat([F.contiguous().view(torch.int32), best, packed], dim=1).contiguous()


def unpack_records(rec, K):
    import torch
    F = rec[:, :9].contiguous().view(torch.float32)
    best = rec[:, 9:13]
    packed = rec[:, 13:13 + K]
    matches = torch.stack([packed & 0xFFFF, (packed >> 16) & 0xFFFF], dim=2)
    return F, best, matches


def slice_sizes(n_items, world):
    """Rows each rank owns under shard_range."""
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_records(rec, world, n_items=None, out=None):
    """all_gather of the per-rank record blocks -> (n_items, words), rank order = global pair order.

    all_gather_into_tensor needs equal blocks.  With n_items % world == 0 (every BASELINE.json configuration) the
    slices are equal and the blocks go out as they are; otherwise each rank pads its block to the largest slice
    (shard_range gives the first n_items % world ranks one row more) and the padding rows are dropped after the
    gather.  n_items defaults to world * rows, i.e. the even case."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return rec
    rows, words = rec.shape
    if n_items is None:
        n_items = world * rows
    sizes = slice_sizes(n_items, world)
    if rows != sizes[dist.get_rank()]:
        raise ValueError(f"rank {dist.get_rank()} holds {rows} records, shard_range({n_items}, rank, {world}) says {sizes[dist.get_rank()]}")
    cap = max(sizes)
    if min(sizes) == cap:
        if out is None:
            out = torch.empty((n_items, words), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec)
        return out
    padded = rec if rows == cap else torch.cat([rec, rec.new_zeros((cap - rows, words))])
    full = torch.empty((world * cap, words), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(full, padded.contiguous())
    return torch.cat([full[r * cap:r * cap + sizes[r]] for r in range(world)])


def pair_seeds(base_seed, lo, hi):
    """Per-pair RANSAC seeds: base ^ global pair index, so a pair's result does not
    depend on which rank or batch position processed it."""
    return (np.arange(lo, hi, dtype=np.uint32) ^ np.uint32(base_seed & 0xFFFFFFFF)).astype(np.uint32)
