"""CPU tests of the N>1 path (world_size 2, gloo): frame shards are disjoint and cover the batch, the timing
reduction is the max over ranks, the pixel count is the sum.  No collective exists on the data path itself
(independent frames), so this is all there is to get right for `bench.py --gpus N`."""
import os
import sys

import pytest

from fast_slic_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_shard_partitions_exactly():
    for n in (0, 1, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            got = [batch.frame_shard(n, r, world) for r in range(world)]
            flat = [f for s in got for f in s]
            assert flat == list(range(n)), (n, world)
    assert [len(batch.frame_shard(64, r, 8)) for r in range(8)] == [8] * 8     # BASELINE.json configs[3]
    assert batch.weak_shard(8, 3) == list(range(24, 32))
    with pytest.raises(ValueError):
        batch.frame_shard(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from fast_slic_amd import batch as b
    dist = b.init_process_group("gloo", rank, world)
    mine = b.frame_shard(13, rank, world)
    dist.barrier()
    slow = b.max_over_ranks(1.0 + rank, dist)            # rank 1 is the slow one
    total = b.sum_over_ranks(len(mine) * 100.0, dist)
    objs = [None] * world
    dist.all_gather_object(objs, mine)                   # test-only gather, to check the cover
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, slow, total, objs))


def test_two_rank_gloo_reduction_and_cover():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, slow, total, objs in res:
        assert slow == 2.0                                # max over ranks
        assert total == 1300.0                            # sum over ranks
        assert [f for s in objs for f in s] == list(range(13))
        assert set(objs[0]).isdisjoint(objs[1])
