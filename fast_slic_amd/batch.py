"""Host-side sharding of an independent-frame batch over the GPUs of one node (BASELINE.json configs[3]).

Frames of a batch have no data dependency on each other (every frame owns its Cluster[K] and its label
map), so the multi-GPU path is: one process per GPU, each process runs its own shard
through `Engine.iterate_batch`, and NO data-path collective exists.  `torch.distributed` carries only the
barrier around the timed region and the max-over-ranks of the elapsed time (RCCL on the GPU box, gloo in
the CPU tests).  Nothing here touches a kernel; tests/test_batch_sharding.py runs it at world_size 2.
"""
import os


def frame_shard(n_frames, rank, world):
    """Contiguous block partition (frame b -> rank b // ceil(n/world)): the ids of this rank's frames.

    64 frames on 8 ranks -> 8 per rank.  Ragged tails go to the last non-empty ranks; ranks past the end
    get an empty shard (they still take part in the barrier)."""
    if n_frames < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    per = -(-n_frames // world) if n_frames else 0
    lo = min(rank * per, n_frames)
    hi = min(lo + per, n_frames)
    return list(range(lo, hi))


def weak_shard(frames_per_gpu, rank):
    """Weak-scaling shard used by bench.py: per-GPU work is fixed, rank r owns seeds r*fpg .. r*fpg+fpg-1."""
    return [rank * frames_per_gpu + i for i in range(frames_per_gpu)]


def env_rank_world():
    """(rank, local_rank, world) from the torch.distributed.run environment; (0, 0, 1) when launched bare."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend, rank, world):
    """torch.distributed rendezvous over 127.0.0.1 (the container hostname may not resolve)."""
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def max_over_ranks(value, dist, device="cpu"):
    """Timing reduction of the bench contract: the slowest rank's elapsed time."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device="cpu"):
    """Units processed by the whole job (value = sum of the ranks' pixels / max of the ranks' time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
