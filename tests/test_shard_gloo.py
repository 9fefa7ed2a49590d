"""N > 1 path on CPU: world_size-2 gloo run of the sharding + record gather the bench uses."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vslam_amd import shard  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pair_seeds_are_position_independent():
    a = shard.pair_seeds(0x5EED0002, 0, 256)
    b = np.concatenate([shard.pair_seeds(0x5EED0002, *shard.shard_range(256, r, 8)) for r in range(8)])
    assert np.array_equal(a, b)


def _worker(rank, world, port, total_pairs, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(total_pairs, rank, world)
    P = hi - lo
    g = torch.Generator().manual_seed(1234)          # same stream on every rank: the "global" result
    F_all = torch.randn((total_pairs, 9), generator=g)
    best_all = torch.randint(-1, 2000, (total_pairs, 4), generator=g, dtype=torch.int32)
    m_all = torch.randint(0, 2000, (total_pairs, K, 2), generator=g, dtype=torch.int32)
    rec = shard.pack_records(F_all[lo:hi], best_all[lo:hi], m_all[lo:hi])
    assert rec.shape == (P, shard.record_words(K))
    out = shard.gather_records(rec, world, n_items=total_pairs)
    F, best, m = shard.unpack_records(out, K)
    ok = torch.equal(F.view(torch.int32), F_all.view(torch.int32)) and torch.equal(best, best_all) and torch.equal(m, m_all)
    # max-over-ranks timing reduction as bench.py does it
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == float(world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world,total_pairs", [(2, 64), (2, 65), (3, 64)])   # even slices, and uneven ones (padded gather)
def test_gather_reassembles_global_result(world, total_pairs):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total_pairs, 50, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {r: True for r in range(world)}
