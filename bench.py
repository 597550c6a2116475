#!/usr/bin/env python
"""bench.py -- megapixels/sec of the SLIC hot path on MI355X (BASELINE.json's metric).

One *step* = one pass of fast_slic.Slic.iterate() (max_iter=10, defaults) over one batch of synthetic
frames per GPU: 8 frames of 1280x720 RGB, K=1600 (BASELINE.json configs[1]; 8 frames/GPU is the sharding of
configs[3], so --gpus 8 is exactly the 64-frame batch).  Frames and label maps are resident in HBM when the
timed region starts (torch tensors; the C ABI gets raw device pointers).  The 8 frames of a step form ONE launch
group (every kernel covers all 8 frames); a step goes to whichever of the engine's slots (stream + host thread) is
free, through the asynchronous submit / done / wait entry points, so several groups are in flight and the host side
of one (cluster upload and write-back, the heap-select on a top-K tie) overlaps the kernels of the others; groups are
collected in completion order.  Every step starts cold from the grid seeding and is complete (labels in HBM,
clusters written back on the host) before the timed region ends.
One process per GPU, frames are independent so there is no data-path collective; torch.distributed (RCCL)
carries only the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.  Extra keys: "roofline" (dominant kernel = k_assign_blk, the 11 assign launches of a
group, HIP-event timed on their own stream inside the engine on the groups of slot 0 during the timed region; its
"alone_on_the_gpu" sub-object repeats the measurement with nothing else in flight), "cpu_baseline" (the unmodified
reference, SlicAvx2 + OpenMP, on this host's cores; the plain-C port if oracle/_ref is absent), "roofline_4k" (the
north-star kernel: full-assign at 3840x2160, K=1600, eight, four and one frame per launch, measured after the timed
region, not part of `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ASSIGN_BYTES_PER_PX = 6.0      # 4 B LAB read + 2 B label write per visited pixel

WORKLOADS = {
    "720p": dict(H=720, W=1280, K=1600, name="1280x720 RGB, K=1600, 10 iters (configs[1]), 8 frames/GPU"),
    "4k": dict(H=2160, W=3840, K=1600, name="3840x2160 RGB, K=1600, 10 iters, 2 frames/GPU"),
    "4k6000": dict(H=2160, W=3840, K=6000, name="3840x2160 RGB, K=6000, 10 iters (configs[2]), 2 frames/GPU"),
}


def pmc_traffic(key):
    """HBM bytes per launch of the assign kernels from the committed rocprofv3 --pmc passes of this workload
    (profiles/r01_pmc_assign_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE); None when the file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_assign_traffic.json")) as f:
            return json.load(f)[key]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(H, W, K, budget_s=12.0):
    """Reference timed on this host: iterate(img, 10) only, fresh clusters per frame (the reference
    re-allocates its context per call anyway), frames of the same synthetic stream as the GPU leg.
    The OpenMP thread count is chosen by a short sweep (the reference's per-iteration `omp critical`
    merges collapse when every hardware thread of a big host is used); `cores` is the count used."""
    from fast_slic_amd.synth import variant
    from oracle import ref, oracle as orc
    ncores = os.cpu_count() or 1
    frames = [variant("A", H, W, seed=s) for s in range(4)]
    if ref.available() and ref.supports_arch("x64/avx2"):
        kind = "reference"
        what = "fast_slic.avx2.SlicAvx2 + OpenMP (oracle/_ref, unmodified reference)"
        init = lambda img: ref.initialize_clusters(img, K)
        cls = [init(f) for f in frames]
        best_t, cores = None, 1
        for nt in [1, 2, 4, 8, 16, 32, 64, 128, 256]:
            if nt > ncores:
                break
            ref.slic_iterate(frames[0], cls[0], num_threads=nt, arch="x64/avx2")   # warm-up (thread pool)
            t = time.perf_counter()
            ref.slic_iterate(frames[1], cls[1], num_threads=nt, arch="x64/avx2")
            t = time.perf_counter() - t
            if best_t is None or t < best_t:
                best_t, cores = t, nt
            elif t > 3.0 * best_t:
                break
        run = lambda img, cl: ref.slic_iterate(img, cl, num_threads=cores, arch="x64/avx2")
    else:
        kind, cores = "port", 1
        what = "plain-C scalar port (oracle/slic_oracle.c)"
        init = lambda img: orc.initialize_clusters(img, K)
        cls = [init(f) for f in frames]
        run = lambda img, cl: orc.slic_iterate(img, cl)
    run(frames[0], cls[0])   # warm-up
    t0 = time.perf_counter()
    n = 0
    best = 1e9
    while True:
        i = n % len(frames)
        t = time.perf_counter()
        run(frames[i], cls[i])
        best = min(best, time.perf_counter() - t)
        n += 1
        if time.perf_counter() - t0 > budget_s and n >= 4:
            break
    elapsed = time.perf_counter() - t0
    return dict(value=round(H * W * n / elapsed / 1e6, 2), unit="megapixels/sec", cores=cores, kind=kind,
                host_threads_available=ncores, best_frame_ms=round(best * 1e3, 3),
                sample="%d frames of %dx%d K=%d, iterate(img,10) only, %s, %d OpenMP threads (best of a 1..%d sweep), %.1f s wall"
                       % (n, W, H, K, what, cores, ncores, elapsed))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--workload", default="720p", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0)
    ap.add_argument("--slots", type=int, default=6, help="streams per GPU; the frames of a step are cut into one group per slot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true")
    args = ap.parse_args()

    import torch
    from fast_slic_amd import batch
    rank, local_rank, world = batch.env_rank_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = batch.init_process_group("nccl", rank, world)     # "nccl" is RCCL on ROCm; barrier + timing reduction only

    from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE
    from fast_slic_amd import _binding as B
    from fast_slic_amd.synth import variant

    wl = WORKLOADS[args.workload]
    H, W, K = wl["H"], wl["W"], wl["K"]
    fpg = args.frames_per_gpu or (8 if args.workload == "720p" else 2)
    nslots = max(1, args.slots)
    eng = Engine(local_rank, nslots)
    params = make_params(10, 10.0, 0.25, 3)

    # this rank's shard of the synthetic frame stream: frames are independent, rank r owns seeds r*fpg .. r*fpg+fpg-1
    seeds = batch.weak_shard(fpg, rank)
    frames = [variant("A", H, W, seed=s) for s in seeds]
    d_rgb = [torch.from_numpy(f).cuda() for f in frames]
    cl0 = np.zeros((fpg, K), CLUSTER_DTYPE)
    for i, f in enumerate(frames):
        B.load_library().fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, cl0[i].ctypes.data)
    # per slot: its own output label maps and cluster blocks (a group stays in flight until its slot is reused)
    d_lab = [[torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in frames] for _ in range(nslots)]
    cls = [np.zeros((fpg, K), CLUSTER_DTYPE) for _ in range(nslots)]
    # raw byte views: copying the structured array field by field costs 200 us per step, a byte copy 10 us
    cl0_raw = cl0.view(np.uint8).reshape(-1)
    cls_raw = [c.view(np.uint8).reshape(-1) for c in cls]
    a_rgb = eng.pointer_array([t.data_ptr() for t in d_rgb])
    a_lab = [eng.pointer_array([t.data_ptr() for t in d_lab[s]]) for s in range(nslots)]
    a_cls = [eng.pointer_array([cls[s][i].ctypes.data for i in range(fpg)]) for s in range(nslots)]
    torch.cuda.synchronize()
    fa_ms, tot_ms, grp, loop_ms, loop_px = [], [], [], [], []
    state = {"n": 0, "pending": [False] * nslots}

    def collect(s):
        eng.wait_group(s)
        state["pending"][s] = False
        g = eng.last_group_frames(s)
        t, f = eng.last_device_times(s)         # HIP events on the slot's own stream, per group launch
        tot_ms.append(t / g)
        if f > 0.0:                             # per-phase events exist only on directly enqueued (timed) groups
            fa_ms.append(f)
        grp.append(g)
        lm, lp, nl = eng.last_assign_loop(s)    # the 10 subsampled assign+update launches, each event-bracketed
        if nl:                                  # (only the steps submitted with per-launch timing on)
            loop_ms.append(lm)
            loop_px.append(lp)

    def free_slot():
        # a slot that is idle, else the first one whose group has completed: groups are collected in completion order
        # (a group with top-K ties takes two more host/GPU round trips than one without)
        for s in range(nslots):
            if not state["pending"][s]:
                return s
        while True:
            for s in range(nslots):
                if eng.group_done(s):
                    collect(s)
                    return s

    prof = {"free_slot": 0.0, "copy": 0.0, "submit": 0.0} if os.environ.get("BENCH_PROFILE") else None

    def step():
        t_a = time.perf_counter()
        s = free_slot()
        t_b = time.perf_counter()
        state["n"] += 1
        np.copyto(cls_raw[s], cl0_raw)          # every step is a cold iterate() from the grid seeding
        eng.set_launch_timing(s == 0)           # per-launch HIP events on the groups of slot 0 (they cost stream time); a
                                                # fixed slot so that every slot replays ONE captured launch sequence
        t_c = time.perf_counter()
        eng.submit_group(s, a_rgb, a_cls[s], a_lab[s], fpg, H, W, K, params)
        state["pending"][s] = True
        if prof is not None:
            prof["free_slot"] += t_b - t_a; prof["copy"] += t_c - t_b; prof["submit"] += time.perf_counter() - t_c

    def drain():
        for s in range(nslots):
            if state["pending"][s]:
                collect(s)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not a step: every slot allocates and carves its arena on first use (hipMalloc + table upload), so touch
    # each slot once before the warm-up steps; otherwise a small --warmup would leave allocations in the timed region
    for _ in range(nslots):
        step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    del fa_ms[:], tot_ms[:], grp[:], loop_ms[:], loop_px[:]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                     # every step of the timed region is complete here
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = batch.max_over_ranks(elapsed, dist, "cuda")
    if prof is not None:
        sys.stderr.write("[bench profile] per step over the whole run (us): " + ", ".join("%s %.1f" % (k, v * 1e6 / state["n"]) for k, v in prof.items()) + "\n")

    if rank == 0:
        total_px = float(H) * W * fpg * world * args.steps
        fa_avg_ms = float(np.mean(fa_ms))
        frames_per_launch = float(np.mean(grp))
        fa_bytes = ASSIGN_BYTES_PER_PX * H * W * frames_per_launch
        lp_bytes = ASSIGN_BYTES_PER_PX * float(np.mean(loop_px))
        lp_ms = float(np.mean(loop_ms))
        gbs = lambda b, ms: b / (ms * 1e-3) / 1e9
        achieved = gbs(fa_bytes + lp_bytes, fa_avg_ms + lp_ms)
        pmc = pmc_traffic("720p_x8") if (args.workload == "720p" and frames_per_launch == 8) else None
        traffic = (pmc["full_assign"]["bytes"] + 10.0 * pmc["fused_assign_update"]["bytes"]) if pmc else None
        out = {
            "metric": "megapixels/sec @10 iters (1280x720 & 4K, K=1600); % HBM roofline",
            "value": round(total_px / elapsed / 1e6, 2),
            "unit": "megapixels/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/u16/i32 (integer SLIC)", "data": "synthetic",
            "config": {"workload": wl["name"], "frames_per_gpu": fpg, "global_frames": fpg * world,
                       "slots_per_gpu": eng.n_slots, "inputs": "device-resident uint8 RGB, device int16 labels",
                       "parallelism": "frames sharded across GPUs, no collective"},
            "device_ms_per_frame": round(float(np.mean(tot_ms)), 4),
            # the dominant kernel is the assign kernel k_assign_blk: per group 10 subsampled launches (fused with the
            # centroid update) + the full-assign launch; achieved = sum of algorithmic bytes / sum of durations
            "roofline": {"kernel": "k_assign_blk, all 11 launches of a group (%dx%d x %g frames per launch)" % (W, H, frames_per_launch),
                         "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "concurrent_streams": eng.n_slots,   # launch durations below are wall time on a GPU shared by this many groups
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, profiles/r01_pmc_assign_traffic.json" if pmc else None,
                         "bytes_per_group": fa_bytes + lp_bytes, "ms_per_group": round(fa_avg_ms + lp_ms, 5),
                         "full_assign": {"bytes_per_launch": fa_bytes, "avg_launch_ms": round(fa_avg_ms, 5),
                                         "achieved": round(gbs(fa_bytes, fa_avg_ms), 2),
                                         "frac": round(gbs(fa_bytes, fa_avg_ms) / HBM_PEAK_GBS, 4)},
                         "fused_assign_update": {"bytes_per_launch": lp_bytes / 10.0, "avg_launch_ms": round(lp_ms / 10.0, 5),
                                                 "achieved": round(gbs(lp_bytes, lp_ms), 2),
                                                 "frac": round(gbs(lp_bytes, lp_ms) / HBM_PEAK_GBS, 4)}},
        }
        # the same launches with the GPU to themselves: groups of slot 0 one after the other, nothing else in flight
        del fa_ms[:], tot_ms[:], grp[:], loop_ms[:], loop_px[:]
        for _ in range(6):
            np.copyto(cls_raw[0], cl0_raw)
            eng.set_launch_timing(True)
            eng.submit_group(0, a_rgb, a_cls[0], a_lab[0], fpg, H, W, K, params)
            state["pending"][0] = True
            collect(0)
        s_fa, s_lp = float(np.mean(fa_ms[2:])), float(np.mean(loop_ms[2:]))
        s_ach = gbs(fa_bytes + lp_bytes, s_fa + s_lp)
        out["roofline"]["alone_on_the_gpu"] = {
            "achieved": round(s_ach, 2), "frac": round(s_ach / HBM_PEAK_GBS, 4),
            "full_assign": {"avg_launch_ms": round(s_fa, 5), "achieved": round(gbs(fa_bytes, s_fa), 2), "frac": round(gbs(fa_bytes, s_fa) / HBM_PEAK_GBS, 4)},
            "fused_assign_update": {"avg_launch_ms": round(s_lp / 10.0, 5), "achieved": round(gbs(lp_bytes, s_lp), 2), "frac": round(gbs(lp_bytes, s_lp) / HBM_PEAK_GBS, 4)},
            "device_ms_per_frame": round(float(np.mean(tot_ms[2:])), 4)}
        if not args.no_4k and args.workload == "720p":
            # the north-star kernel: full-assign at 3840x2160, K=1600, same engine, after the timed region
            H4, W4, K4 = 2160, 3840, 1600
            G4 = 8                                  # frames per launch of the batched operating point (= the headline config's group size)
            imgs = [variant("A", H4, W4, seed=g) for g in range(G4)]
            d4 = [torch.from_numpy(im).cuda() for im in imgs]
            l4 = [torch.empty((H4, W4), dtype=torch.int16, device="cuda") for _ in imgs]
            c4 = np.zeros((G4, K4), CLUSTER_DTYPE)
            for g, im in enumerate(imgs):
                B.load_library().fslic_hip_initialize_clusters(H4, W4, K4, im.ctypes.data, c4[g].ctypes.data)
            c4w = c4.copy()
            p_rgb = eng.pointer_array([t.data_ptr() for t in d4])
            p_lab = eng.pointer_array([t.data_ptr() for t in l4])
            p_cls = eng.pointer_array([c4w[g].ctypes.data for g in range(G4)])
            torch.cuda.synchronize()

            def run4(nf):
                f4, t4, l4ms, l4px = [], [], [], []
                for i in range(8):
                    np.copyto(c4w.view(np.uint8).reshape(-1), c4.view(np.uint8).reshape(-1))
                    eng.set_launch_timing(True)
                    eng.submit_group(0, p_rgb, p_cls, p_lab, nf, H4, W4, K4, params)
                    eng.wait_group(0)
                    if i >= 2:
                        t, f = eng.last_device_times(0)
                        t4.append(t)
                        f4.append(f)
                        lm, lp, _ = eng.last_assign_loop(0)
                        l4ms.append(lm)
                        l4px.append(lp)
                fa = float(np.mean(f4))
                full = ASSIGN_BYTES_PER_PX * H4 * W4 * nf / (fa * 1e-3) / 1e9
                fused = ASSIGN_BYTES_PER_PX * float(np.mean(l4px)) / (float(np.mean(l4ms)) * 1e-3) / 1e9
                return {"frames_per_launch": nf, "achieved": round(full, 2), "frac": round(full / HBM_PEAK_GBS, 4),
                        "avg_launch_ms": round(fa, 5),
                        "fused_assign_update": {"avg_launch_ms": round(float(np.mean(l4ms)) / 10.0, 5),
                                                "achieved": round(fused, 2), "frac": round(fused / HBM_PEAK_GBS, 4)},
                        "device_ms_per_frame": round(float(np.mean(t4)) / nf, 4),
                        "megapixels_per_sec": round(H4 * W4 * nf / (float(np.mean(t4)) * 1e-3) / 1e6, 1)}

            one, mid, four = run4(1), run4(4), run4(G4)
            pmc4 = pmc_traffic("4k_k1600")
            out["roofline_4k"] = {"kernel": "k_assign_blk<16,false,1> (full_assign launch, 3840x2160, K=1600), %d frames per launch" % G4,
                                  "bound": "hbm", "achieved": four["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": four["frac"],
                                  "traffic": pmc4["full_assign"]["bytes"] * G4 if pmc4 else None,
                                  "avg_launch_ms": four["avg_launch_ms"], "fused_assign_update": four["fused_assign_update"],
                                  "device_ms_per_frame": four["device_ms_per_frame"], "megapixels_per_sec": four["megapixels_per_sec"],
                                  "four_frame_launch": mid, "single_frame_launch": one}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, W, K)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
