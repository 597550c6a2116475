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


def test_bench_launches_its_own_ranks_bare():
    """`python bench.py --gpus 2` as the driver types it (no torch.distributed.run around it): the script spawns the two
    ranks, they rendezvous on 127.0.0.1 over gloo, the timed region is bracketed by barriers, rank 0 prints ONE line with
    the max-over-ranks time.  --dry-run replaces the GPU work by sleeps (rank 1 is the slow one), nothing else."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["units_all_ranks"] == 40.0
    assert out["ms_per_step"] >= 2.0                      # rank 1 sleeps 2 ms per step: the slowest rank decides
    assert len(out["per_rank_ms"]) == 2 and out["per_rank_ms"][1] > out["per_rank_ms"][0]     # every rank's own clock is in the line


def test_bench_launches_four_ranks_bare_and_reports_each_rank():
    """World size 4 (the driver's N = 1, 2, 4, 8 ladder), bare launch: per-rank times in rank order, the slowest decides."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "10", "--warmup", "1", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 4 and out["units_all_ranks"] == 40.0 and out["ms_per_step"] >= 4.0
    pr = out["per_rank_ms"]
    assert len(pr) == 4 and pr == sorted(pr) and pr[3] >= 40.0


def test_a_dying_rank_takes_the_launch_down():
    """spawn_ranks: one rank exiting non-zero terminates the others instead of leaving them in the barrier until the rendezvous
    times out; the launcher returns a non-zero code."""
    import time
    t0 = time.time()
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(120)"
    rc = batch.spawn_ranks(3, ["-c", code])
    assert rc != 0 and time.time() - t0 < 60


def test_rank_binds_to_its_own_gpu():
    """bind_rank_to_its_gpu: the rank's GPU becomes the only visible device (one HIP context per rank, not one per GPU of the node)."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); from fast_slic_amd import batch\n"
            "os.environ.pop('HIP_VISIBLE_DEVICES', None); d = batch.bind_rank_to_its_gpu(5); print(d, os.environ['HIP_VISIBLE_DEVICES'])\n"
            "os.environ['HIP_VISIBLE_DEVICES'] = '4,5,6,7'; d = batch.bind_rank_to_its_gpu(2); print(d, os.environ['HIP_VISIBLE_DEVICES'])\n"
            "os.environ['HIP_VISIBLE_DEVICES'] = '3'; d = batch.bind_rank_to_its_gpu(0); print(d, os.environ['HIP_VISIBLE_DEVICES'])\n"
            # bench.py --share-device: every rank takes the first visible GPU (the host side of an N-rank launch on a one-GPU box)
            "os.environ.pop('HIP_VISIBLE_DEVICES', None); d = batch.bind_rank_to_its_gpu(5, share_device=True); print(d, os.environ['HIP_VISIBLE_DEVICES'])\n"
            "os.environ['HIP_VISIBLE_DEVICES'] = '4,5,6,7'; d = batch.bind_rank_to_its_gpu(2, share_device=True); print(d, os.environ['HIP_VISIBLE_DEVICES'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout.decode().split() == ["0", "5", "0", "6", "0", "3", "0", "0", "0", "4"]


def test_torchrun_environment_is_respected():
    import json
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(batch.free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode().strip().splitlines()[-1])["n_gpus"] == 1
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--dry-run"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0                               # --gpus must agree with the launcher's WORLD_SIZE
