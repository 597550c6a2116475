"""Host-side sharding of an independent-frame batch over the GPUs of one node (BASELINE.json configs[3]).

Frames of a batch have no data dependency on each other (every frame owns its Cluster[K] and its label
map), so the multi-GPU path is: one process per GPU, each process runs its own shard
through `Engine.iterate_batch`, and NO data-path collective exists.  `torch.distributed` carries only the
barrier around the timed region and the max-over-ranks of the elapsed time, over gloo everywhere (RCCL is never
initialised: `north_star` says "RCCL unused").  Nothing here touches a kernel; tests/test_batch_sharding.py runs it at world_size 2.
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


def gather_over_ranks(value, dist, world):
    """Every rank's value, in rank order (the per-rank elapsed times of the bench line: a sub-linear scaling point is only
    attributable when the line says which rank was slow)."""
    if dist is None:
        return [float(value)]
    import torch
    mine = torch.tensor([float(value)], dtype=torch.float64)
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def bind_rank_to_its_gpu(local_rank, share_device=False):
    """Before HIP is initialised in this process: make the rank's GPU the ONLY visible device (it then is device 0), so that a
    rank creates one context instead of one per GPU of the node.  An existing HIP_VISIBLE_DEVICES list is narrowed to its
    local_rank-th entry.  Returns the device index to use afterwards (always 0).
    share_device: every rank takes the FIRST visible GPU (bench.py --share-device: the host side of an N-rank launch -- N submit
    loops, N x slots slot threads, N rendezvous members -- exercised on a box that has one GPU)."""
    if share_device:
        local_rank = 0
    cur = os.environ.get("HIP_VISIBLE_DEVICES")
    if cur:
        ids = [x for x in cur.split(",") if x != ""]
        if len(ids) > 1 and local_rank < len(ids):
            os.environ["HIP_VISIBLE_DEVICES"] = ids[local_rank]
        elif len(ids) > 1:
            return local_rank              # cannot narrow: keep the node's numbering
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
    return 0


# ---- launching one process per GPU ----------------------------------------------------------------------------------
def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(world, argv, extra_env=None):
    """Run `argv` (a python command line) as `world` processes, rank r with RANK = LOCAL_RANK = r, WORLD_SIZE = world and a
    rendezvous on 127.0.0.1 -- what `python -m torch.distributed.run --nproc-per-node world` would set up.  Rank 0 inherits
    stdout (it prints the result line); the call returns the largest exit code."""
    import subprocess
    import sys
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies takes the others with it (they would sit in the barrier until the rendezvous times out)
    import time
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, code if code > 0 else 1)
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
    return rc


def gpu_numa_cpus(local_rank, device_index=None):
    """CPUs of the NUMA node the rank's GPU hangs off (device_index: the GPU's index in THIS process when the visible devices
    have been narrowed to the rank's own, bind_rank_to_its_gpu) (sysfs: the PCI device's numa_node and that node's cpulist), or
    None when the topology cannot be read (containers without sysfs, CPU-only test runs)."""
    bdf = None
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank if device_index is None else device_index)
        if hasattr(pr, "pci_bus_id"):
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    except Exception:       # noqa: BLE001 - no GPU / no torch: no pinning
        bdf = None
    try:
        if bdf is None:
            import glob
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/numa_node"))
            path = cards[local_rank] if local_rank < len(cards) else None
        else:
            path = "/sys/bus/pci/devices/%s/numa_node" % str(bdf).lower()
        if not path or not os.path.exists(path):
            return None
        node = int(open(path).read().strip())
        if node < 0:
            return None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        return cpus or None
    except (OSError, ValueError, IndexError):
        return None


def pin_to_gpu_numa_node(local_rank, world, device_index=None):
    """Keep this rank's host threads (slot threads, staging, write-back) on the NUMA node of its GPU; when the node is
    unknown, on an equal share of the visible CPUs.  Returns the CPU set chosen (for the bench line)."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    cpus = gpu_numa_cpus(local_rank, device_index)
    if cpus:
        cpus = [c for c in cpus if c in set(avail)]
    if not cpus and world > 1:
        per = max(1, len(avail) // world)
        cpus = avail[local_rank * per:(local_rank + 1) * per]
    if not cpus:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus
