#!/usr/bin/env python
"""bench.py -- megapixels/sec of the SLIC hot path on MI355X (BASELINE.json's metric).

One *step* = one pass of fast_slic.Slic.iterate() (max_iter=10, defaults) over one batch of synthetic frames per GPU:
8 frames of 1280x720 RGB, K=1600 (BASELINE.json configs[1]; 8 frames/GPU is the sharding of configs[3], so --gpus 8 is
exactly the 64-frame batch).  Frames and label maps are resident in HBM when the timed region starts (torch tensors; the C
ABI gets raw device pointers).  The 8 frames of a step form ONE launch group (every kernel covers all 8 frames); a step is
handed to the engine's submit / drain pipeline (fslic_hip_pipeline_submit), which gives it to whichever slot (stream + host
thread) is free, so several groups are in flight and the host side of one (staging, cluster write-back) overlaps the kernels
of the others.  Every step starts cold from the grid seeding, owns its label maps and cluster blocks, and is complete
(labels in HBM, clusters written back on the host) before the timed region ends.

One process per GPU: `python bench.py --gpus N` spawns the N ranks itself (or runs as one rank of
`python -m torch.distributed.run`); frames are independent, so there is no data-path collective -- torch.distributed (gloo)
carries only the barrier around the timed region and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.  Beyond the contract's keys:
  parity        the label maps and Cluster blocks of the LAST timed step, compared with the oracle after the timed region
  roofline      the dominant kernel of the benched configuration (the fused assign+update launch, 10 per group), HIP-event
                timed on its own stream with the GPU to itself; sub-objects: the same launches inside the timed region
                (several groups share the GPU), the full-assign launch, the sum over a group's 11 assign launches
  roofline_4k   the north-star kernel (full assign at 3840x2160 K=1600) and the sum over the 11 assign launches at 4K
  c3_k6000, lsc_c5   BASELINE configs[2] and [4]: megapixels/sec and their dominant kernel
  end_to_end    host numpy in -> int16 numpy out through fast_slic_amd.Slic(...).iterate()
  cpu_baseline  the unmodified reference (SlicAvx2 + OpenMP; plain-C port if oracle/_ref is absent) on this host's cores
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ASSIGN_BYTES_PER_PX = 6.0      # 4 B LAB read + 2 B label write per visited pixel
LSC_ASSIGN_BYTES_PER_PX = 6.0  # 4 B LAB word + 2 B label: the features are rebuilt from the LAB word (SURVEY.md 8d's recompute design; the materialised one moved 46 B)
PROFILE_TAG = "r06"

WORKLOADS = {
    "720p": dict(H=720, W=1280, K=1600, fpg=8, name="1280x720 RGB, K=1600, 10 iters (configs[1]), 8 frames/GPU"),
    "4k": dict(H=2160, W=3840, K=1600, fpg=8, name="3840x2160 RGB, K=1600, 10 iters, 8 frames/GPU"),
    "4k6000": dict(H=2160, W=3840, K=6000, fpg=4, name="3840x2160 RGB, K=6000, 10 iters (configs[2]), 4 frames/GPU"),
}


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


def measured_copy_bandwidth(eng):
    """Empirical HBM bandwidth of this box (SURVEY.md 8d asks for fractions against the measured peak as well as the spec): the
    library's plain streaming copy kernel (16 bytes per lane and trip, csrc/graph.hip k_copy16 -- the form the MI355X guide quotes
    6.29 TB/s for) over 1 GiB (read 1 GiB + write 1 GiB, far beyond the 256 MB infinity cache), best of 5, HIP events."""
    try:
        return round(eng.copy_bandwidth(1 << 30, 5), 1)
    except (AttributeError, RuntimeError):      # an A/B library of an earlier round (FSLIC_LIB) has no such entry point
        return None


PATH_BYTES_PER_PX = 37.0       # 7 N (LAB) + 10 x 6 N / 3 (fused assign + update) + 6 N (full assign) + 4 N (connectivity lower bound)


def path_roofline(mp_per_s, what):
    """Fraction of the HBM roofline of the WHOLE path at a measured rate (megapixels/s of one GPU): 37 N algorithmic bytes per frame."""
    ach = mp_per_s * 1e6 * PATH_BYTES_PER_PX / 1e9
    return {"what": what, "bytes_per_px": PATH_BYTES_PER_PX, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "ceiling_megapixels_per_sec": round(HBM_PEAK_GBS * 1e9 / PATH_BYTES_PER_PX / 1e6, 0)}


def pmc_traffic(key):
    """HBM bytes per launch of the assign kernels from the committed rocprofv3 --pmc passes of this workload
    (profiles/<round>_pmc_assign_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE); None when absent."""
    for tag in (PROFILE_TAG, "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_pmc_assign_traffic.json" % tag)) as f:
                d = json.load(f)
                return d[key], "profiles/%s_pmc_assign_traffic.json (counters of build %s)" % (tag, d.get("build_commit") or "unrecorded")
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def rocprof_avg_us(csv_name, *needles):
    """Average duration (us) of the first kernel whose name contains every one of `needles` in a committed rocprofv3 summary
    (profiles/<round>_kernel_stats_*.csv, written by scripts/gpu_profiles.sh from a --kernel-trace run); None when absent."""
    for tag in (PROFILE_TAG, "r05", "r04"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s" % (tag, csv_name))) as f:
                for line in f:
                    if all(n in line for n in needles):
                        return float(line.rsplit('",', 1)[1].split(",")[2]), tag
        except (OSError, ValueError, IndexError):
            continue
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(H, W, K, budget_s=8.0):
    """Reference timed on this host: iterate(img, 10) only, fresh clusters per frame (the reference re-allocates its
    context per call anyway), frames of the same synthetic stream as the GPU leg.  The OpenMP thread count is chosen by a
    short sweep (the reference's per-iteration `omp critical` merges collapse when every hardware thread of a big host is
    used); `cores` is the count used.  Also: one thread, the scalar arch, and the reference's own stage split."""
    from fast_slic_amd.synth import variant
    from oracle import ref, oracle as orc
    ncores = os.cpu_count() or 1
    frames = [variant("A", H, W, seed=s) for s in range(4)]
    mp = H * W / 1e6
    if not (ref.available() and ref.supports_arch("x64/avx2")):
        cls = [orc.initialize_clusters(f, K) for f in frames]
        orc.slic_iterate(frames[0], cls[0])
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s or n < 2:
            orc.slic_iterate(frames[n % 4], cls[n % 4]); n += 1
        el = time.perf_counter() - t0
        return dict(value=round(mp * n / el, 2), unit="megapixels/sec", cores=1, kind="port", host_threads_available=ncores,
                    sample="%d frames of %dx%d K=%d, plain-C scalar port (oracle/slic_oracle.c), %.1f s" % (n, W, H, K, el))
    cls = [ref.initialize_clusters(f, K) for f in frames]

    def timed(nt, arch, reps, budget):
        ref.slic_iterate(frames[0], cls[0], num_threads=nt, arch=arch)          # warm-up (thread pool)
        t0, n, best = time.perf_counter(), 0, 1e9
        while n < reps or (time.perf_counter() - t0 < budget):
            t = time.perf_counter()
            ref.slic_iterate(frames[n % 4], cls[n % 4], num_threads=nt, arch=arch)
            best = min(best, time.perf_counter() - t)
            n += 1
        return n, time.perf_counter() - t0, best

    best_t, cores = None, 1
    for nt in [1, 2, 4, 8, 16, 32, 64, 128, 256]:
        if nt > ncores:
            break
        _, _, t = timed(nt, "x64/avx2", 1, 0.0)
        if best_t is None or t < best_t:
            best_t, cores = t, nt
        elif t > 3.0 * best_t:
            break
    n, el, best = timed(cores, "x64/avx2", 4, budget_s)
    out = dict(value=round(mp * n / el, 2), unit="megapixels/sec", cores=cores, kind="reference",
               host_threads_available=ncores, best_frame_ms=round(best * 1e3, 3),
               sample="%d frames of %dx%d K=%d, iterate(img,10) only, fast_slic.avx2.SlicAvx2 + OpenMP (oracle/_ref, unmodified "
                      "reference), %d OpenMP threads (best of a 1..%d sweep), %.1f s wall" % (n, W, H, K, cores, ncores, el))
    try:
        rep = json.loads(ref.last_timing_report())                              # the reference's fstimer report (src/timer.cpp:4-18)
        split = {}                                                              # of the last frame, top-level stages summed by name
        for ch in rep.get("children", []):
            split[ch["name"]] = split.get(ch["name"], 0) + int(ch["duration"])
        out["stage_split_us"] = dict(split, iterate=int(rep.get("duration", 0)))
    except Exception:                                                          # noqa: BLE001 - informational
        pass
    n1, el1, _ = timed(1, "x64/avx2", 2, 1.0)
    out["threads_1"] = dict(value=round(mp * n1 / el1, 2), unit="megapixels/sec", cores=1, arch="x64/avx2")
    ns, els, _ = timed(cores, "standard", 2, 1.0)
    out["scalar_slic"] = dict(value=round(mp * ns / els, 2), unit="megapixels/sec", cores=cores, arch="standard (fast_slic.Slic)")
    return out


# ---------------------------------------------------------------------------------------------------------------------
class Workload(object):
    """Frames of one configuration resident in HBM, plus per-submission label maps and cluster blocks."""

    def __init__(self, eng, H, W, K, seeds, variant_id=0):
        import torch
        from fast_slic_amd import CLUSTER_DTYPE, make_params
        from fast_slic_amd import _binding as B
        from fast_slic_amd.synth import variant
        self.eng, self.H, self.W, self.K, self.n = eng, H, W, K, len(seeds)
        self.params = make_params(10, 10.0, 0.25, 3, variant=variant_id)
        self.frames = [variant("A", H, W, seed=s) for s in seeds]
        self.d_rgb = [torch.from_numpy(f).cuda() for f in self.frames]
        self.cl0 = np.zeros((self.n, K), CLUSTER_DTYPE)
        for i, f in enumerate(self.frames):
            B.load_library().fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, self.cl0[i].ctypes.data)
        self.a_rgb = eng.pointer_array([t.data_ptr() for t in self.d_rgb])
        self.torch = torch
        self.sets = []
        torch.cuda.synchronize()

    def buffers(self, count):
        """`count` more submissions' worth of output label maps (HBM) and cluster blocks (host, seeded)."""
        t = self.torch
        for _ in range(count):
            d_lab = [t.empty((self.H, self.W), dtype=t.int16, device="cuda") for _ in range(self.n)]
            cls = self.cl0.copy()
            self.sets.append((d_lab, cls, self.eng.pointer_array([x.data_ptr() for x in d_lab]),
                              self.eng.pointer_array([cls[i].ctypes.data for i in range(self.n)])))
        t.cuda.synchronize()

    def submit(self, j, n=None):
        _, _, a_lab, a_cls = self.sets[j]
        self.eng.pipeline_submit(self.a_rgb, a_cls, a_lab, n or self.n, self.H, self.W, self.K, self.params)

    def submit_to_slot(self, slot, j, n=None, merged=1):
        """Submission j (or, merged > 1, submissions j .. j+merged-1 as ONE group, the form dynamic batching produces) on a
        slot of the caller's choice (setup: every slot must have seen every form of group three times)."""
        if merged == 1:
            _, _, a_lab, a_cls = self.sets[j]
            self.eng.submit_group(slot, self.a_rgb, a_cls, a_lab, n or self.n, self.H, self.W, self.K, self.params)
            return
        rgb, lab, cls = [], [], []
        for q in range(merged):
            d_lab, c, _, _ = self.sets[j + q]
            rgb += [t.data_ptr() for t in self.d_rgb]
            lab += [t.data_ptr() for t in d_lab]
            cls += [c[i].ctypes.data for i in range(self.n)]
        arrs = (self.eng.pointer_array(rgb), self.eng.pointer_array(cls), self.eng.pointer_array(lab))
        self._keep = getattr(self, "_keep", []) + [arrs]          # the engine copies the pointer arrays at submit; kept anyway
        self.eng.submit_group(slot, arrs[0], arrs[1], arrs[2], self.n * merged, self.H, self.W, self.K, self.params)

    def prepare_slots(self, base, merged_max):
        """Setup, not a step: every slot sees every form of group the pipeline can produce (1 .. merged_max submissions per
        group) three times -- the first carves its arena, the second records and instantiates its hipGraph, the third
        replays it -- addressed slot by slot.  Uses buffer sets base .. base + sets_for_setup(merged_max) - 1."""
        ns = self.eng.n_slots
        j = base
        for merged in range(1, merged_max + 1):
            for rnd in range(3):
                for sl in range(ns):
                    self.submit_to_slot(sl, j, merged=merged)
                    j += merged
                for sl in range(ns):
                    self.eng.wait_group(sl)
        return j - base

    def sets_for_setup(self, merged_max):
        return 3 * self.eng.n_slots * sum(range(1, merged_max + 1))

    def timed_groups(self, reps, n=None, bytes_per_px=ASSIGN_BYTES_PER_PX, merged=1):
        """`reps` groups one after the other with the GPU to themselves and every subsampled assign launch and the
        full-assign launch bracketed by HIP events on the slot's stream: averages over the groups after the first two.
        merged > 1: groups of `merged` submissions, the form dynamic batching produces (merged * n frames per launch)."""
        eng, n = self.eng, n or self.n
        if len(self.sets) < merged:
            self.buffers(merged - len(self.sets))
        d_lab, cls, a_lab, a_cls = self.sets[0]
        rec = []
        eng.set_launch_timing(True)
        try:
            for i in range(reps + 2):
                for q in range(merged):
                    c = self.sets[q][1]
                    np.copyto(c.view(np.uint8).reshape(-1), self.cl0.view(np.uint8).reshape(-1))
                if merged == 1:
                    eng.submit_group(0, self.a_rgb, a_cls, a_lab, n, self.H, self.W, self.K, self.params)
                else:
                    self.submit_to_slot(0, 0, merged=merged)
                eng.wait_group(0)
                if i >= 2:
                    tot, fa = eng.last_device_times(0)
                    lm, lp, nl = eng.last_assign_loop(0)
                    rec.append((tot, fa, lm, lp, nl))
        finally:
            eng.set_launch_timing(False)
        tot, fa, lm, lp, nl = [float(np.mean([r[q] for r in rec])) for q in range(5)]
        n = n * merged
        fa_bytes = bytes_per_px * self.H * self.W * n
        lp_bytes = bytes_per_px * lp
        per_launch = lm / max(nl, 1.0)
        return {"frames_per_launch": n,
                "fused_assign_update": {"bytes_per_launch": lp_bytes / max(nl, 1.0), "avg_launch_ms": round(per_launch, 5),
                                        "launches_per_group": int(nl), "achieved": round(gbs(lp_bytes, lm), 2),
                                        "frac": round(gbs(lp_bytes, lm) / HBM_PEAK_GBS, 4)},
                "full_assign": {"bytes_per_launch": fa_bytes, "avg_launch_ms": round(fa, 5), "achieved": round(gbs(fa_bytes, fa), 2),
                                "frac": round(gbs(fa_bytes, fa) / HBM_PEAK_GBS, 4)},
                "all_assign_launches": {"bytes_per_group": fa_bytes + lp_bytes, "ms_per_group": round(fa + lm, 5),
                                        "achieved": round(gbs(fa_bytes + lp_bytes, fa + lm), 2),
                                        "frac": round(gbs(fa_bytes + lp_bytes, fa + lm) / HBM_PEAK_GBS, 4)},
                "device_ms_per_frame": round(tot / n, 4),
                "megapixels_per_sec_one_group_at_a_time": round(self.H * self.W * n / (tot * 1e-3) / 1e6, 1)}

    merged_max = 1          # submissions a launch group may gather (dynamic batching of the pipeline): set by main()

    def throughput(self, steps, warm=None):
        """Pipelined megapixels/sec over `steps` submissions (fresh buffers per submission), after prepare_slots()."""
        merged_max = self.merged_max
        warm = self.sets_for_setup(merged_max)
        base = len(self.sets)
        self.buffers(steps + warm)
        self.prepare_slots(base, merged_max)
        self.torch.cuda.synchronize()
        gc.collect()
        gc.disable()                              # (see the timed region of main())
        t0 = time.perf_counter()
        for j in range(steps):
            self.submit(base + warm + j)
        tot = self.eng.pipeline_drain()
        self.torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        self.last_timed = base + warm + steps - 1          # buffer set of the last submission of the timed region (parity gates)
        return round(self.H * self.W * self.n * steps / el / 1e6, 1), tot


def parity_of(wl, j):
    """The label maps and Cluster blocks of submission j against the oracle (bit-exact; SURVEY.md 8d 'parity gate')."""
    from oracle import oracle as orc
    d_lab, cls, _, _ = wl.sets[j]
    bad = []
    for i, f in enumerate(wl.frames):
        labels, cl = orc.slic_iterate(f, orc.initialize_clusters(f, wl.K))
        got = d_lab[i].cpu().numpy().view(np.uint16)
        if not np.array_equal(got, labels):
            bad.append("frame %d: %d label pixels differ" % (i, int((got != labels).sum())))
        if cls[i].tobytes() != cl.tobytes():
            bad.append("frame %d: Cluster bytes differ" % i)
    return bad


def lsc_parity_of(wl, j, j_other, measured=None):
    """LSC (f32 path, tolerance-based parity): the label maps of submission j against the committed fixture of
    the unmodified reference (tests/golden/lsc_cases.npz, frame seed 0 of this workload: agreement >= 0.99 as in
    tests/test_gpu_lsc.py) and against submission j_other of the same inputs (run-to-run identity: the sums are fixed point)."""
    bad = []
    measured = {} if measured is None else measured
    try:
        fx = np.load(os.path.join(ROOT, "tests", "golden", "lsc_cases.npz"), allow_pickle=False)
        want = fx["A_%dx%d_k%d/labels" % (wl.H, wl.W, wl.K)]
        want_cl = fx["A_%dx%d_k%d/clusters" % (wl.H, wl.W, wl.K)]
    except (OSError, KeyError):
        want = want_cl = None
    a = [t.cpu().numpy().view(np.uint16) for t in wl.sets[j][0]]
    b = [t.cpu().numpy().view(np.uint16) for t in wl.sets[j_other][0]]

    def best_overlap(p, q):          # fraction of pixels whose segment in p lies in the segment of q it overlaps most (labels are renumbered by the connectivity pass)
        p = p.astype(np.int64).ravel(); q = q.astype(np.int64).ravel()
        nq = int(q.max()) + 1
        u, c = np.unique(p * nq + q, return_counts=True)
        best = np.zeros(int(p.max()) + 1, np.int64)
        np.maximum.at(best, u // nq, c)
        return float(best.sum()) / p.size
    if want is not None:
        agree = min(best_overlap(a[0], want), best_overlap(want, a[0]))
        cl = wl.sets[j][1][0]
        measured["best_overlap_both_ways_min"] = round(agree, 5)
        measured["mean_centre_shift_px"] = round(float(np.mean(np.hypot(cl["y"] - want_cl["y"], cl["x"] - want_cl["x"]))), 4)
        measured["max_centre_shift_px"] = round(float(np.max(np.hypot(cl["y"] - want_cl["y"], cl["x"] - want_cl["x"]))), 3)
        measured["against"] = "tests/golden/lsc_cases.npz (unmodified fast_slic.LSC, 1 thread), frame 0 of the timed region's last submission"
        if agree < 0.99:
            bad.append("frame 0: best overlap with the reference fixture %.4f (< 0.99)" % agree)
        if measured["mean_centre_shift_px"] > 0.1:
            bad.append("frame 0: mean centre shift %.4f px (> 0.1, the floor of tests/test_gpu_lsc.py)" % measured["mean_centre_shift_px"])
    else:
        bad.append("fixture A_%dx%d_k%d missing" % (wl.H, wl.W, wl.K))
    for i in range(wl.n):
        if not np.array_equal(a[i], b[i]):
            bad.append("frame %d differs between two submissions of the same input" % i)
        if wl.sets[j][1][i].tobytes() != wl.sets[j_other][1][i].tobytes():
            bad.append("frame %d: Cluster bytes differ between two submissions of the same input" % i)
    return bad


def single_frame_latency(eng, wl, reps=24):
    """Device time of ONE frame per call (the reference's own calling pattern, cfast_slic.pyx:150-260): groups of one frame on
    slot 0, GPU to itself, replayed from the slot's hipGraph; HIP events around the replay (median of `reps` after 4 warm calls)."""
    if not wl.sets:
        wl.buffers(1)
    d_lab, cls, a_lab, a_cls = wl.sets[0]
    ms, modes = [], set()
    for i in range(reps + 4):
        np.copyto(cls.view(np.uint8).reshape(-1), wl.cl0.view(np.uint8).reshape(-1))
        eng.submit_group(0, wl.a_rgb, a_cls, a_lab, 1, wl.H, wl.W, wl.K, wl.params)
        eng.wait_group(0)
        if i >= 4:
            ms.append(eng.last_device_times(0)[0])
            modes.add(eng.last_launch_mode(0))
    med = float(np.median(ms))
    return {"device_ms": round(med, 4), "device_ms_min": round(float(np.min(ms)), 4), "megapixels_per_sec": round(wl.H * wl.W / (med * 1e-3) / 1e6, 1),
            "launch": "hipGraph replay" if modes == {2} else "modes %s" % sorted(modes),
            "what": "one %dx%d K=%d frame per group, device-resident, GPU to itself (HIP events around the group)" % (wl.W, wl.H, wl.K)}


def end_to_end_ref_binding(H, W, K, budget_s=3.0, threads=4):
    """The same through the drop-in itself: the REFERENCE's Cython binding with the arch "hip/gfx950" patched in
    (integration/_build, built by integration/build_ref_binding.sh): cfast_slic.SlicModel(K, "hip/gfx950") driven the way
    fast_slic/base_slic.py:18-52 drives it (only the compiled extension travels to the GPU box, not the reference's Python
    package).  Runs in a separate interpreter (the patched module links libfslic_hip.so itself).  None when integration/_build is absent."""
    import glob
    import subprocess
    build = os.path.join(ROOT, "integration", "_build")
    if not glob.glob(os.path.join(build, "cfast_slic*.so")):
        return None
    code = r"""
import sys, time, json, threading
import numpy as np
sys.path.insert(0, %(build)r); sys.path.insert(1, %(root)r)
from fast_slic_amd.synth import variant
H, W, K, budget, threads = %(H)d, %(W)d, %(K)d, %(budget)f, %(threads)d
imgs = [variant("A", H, W, seed=s) for s in range(4)]
import cfast_slic
facade = "cfast_slic.SlicModel('hip/gfx950') driven like fast_slic/base_slic.py"
class _M(object):
    def __init__(self):
        m = cfast_slic.SlicModel(K, "hip/gfx950")
        m.convert_to_lab = True; m.preemptive = False; m.preemptive_thres = 0.05
        m.manhattan_spatial_dist = True; m.num_threads = -1; m.debug_mode = False
        self.slic_model = m
    def iterate(self, image, max_iter=10):
        if not self.slic_model.initialized:
            self.slic_model.initialize(image)
        return self.slic_model.iterate(image, max_iter, 10, 0.25, 3)
def make():
    return _M()
def run(nt):
    counts = [0] * nt
    stop = time.perf_counter() + budget / 2
    def work(t):
        s = make(); s.iterate(imgs[t %% 4]); i = 0
        while time.perf_counter() < stop:
            s.slic_model.initialized = False
            s.iterate(imgs[(t + i) %% 4]); i += 1
        counts[t] = i
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    t0 = time.perf_counter()
    [th.start() for th in ths]; [th.join() for th in ths]
    return H * W * sum(counts) / (time.perf_counter() - t0) / 1e6, sum(counts)
one, n1 = run(1); many, nm = run(threads)
print("RESULT " + json.dumps({"facade": facade, "one_thread": round(one, 1), "threads_%%d" %% threads: round(many, 1), "frames": [n1, nm]}))
""" % {"build": build, "root": ROOT, "H": H, "W": W, "K": K, "budget": budget_s, "threads": threads}
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            return {"error": r.stderr.decode()[-400:]}
        return json.loads(line[-1][7:])
    except Exception as ex:       # noqa: BLE001 - a reported leg, never the bench's failure
        return {"error": repr(ex)[:400]}


def end_to_end(H, W, K, budget_s=4.0, threads=4):
    """Through the Python surface: host numpy in, int16 numpy out (cfast_slic.pyx:150,163,258-260), PCIe both ways."""
    import threading
    from fast_slic_amd import Slic
    from fast_slic_amd.synth import variant
    imgs = [variant("A", H, W, seed=s) for s in range(4)]

    def run(nthreads):
        counts = [0] * nthreads
        stop = time.perf_counter() + budget_s / 2

        def work(t):
            s = Slic(num_components=K, compactness=10)
            s.iterate(imgs[t % 4])                              # seeding + first call (buffers), untimed
            i = 0
            while time.perf_counter() < stop:
                s.slic_model.initialized = False                # cold start every frame, like the device-resident leg
                s.iterate(imgs[(t + i) % 4])
                i += 1
            counts[t] = i

        ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return H * W * sum(counts) / (time.perf_counter() - t0) / 1e6, sum(counts)

    one, n1 = run(1)
    many, nm = run(threads)
    return {"unit": "megapixels/sec", "one_thread": round(one, 1), "threads_%d" % threads: round(many, 1),
            "sample": "%d + %d frames of %dx%d K=%d through fast_slic_amd.Slic(K).iterate(numpy image): seeding + H2D of the frame + "
                      "iterate + D2H of the label map + int16 conversion per call; pageable numpy buffers" % (n1, nm, W, H, K)}


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (groups).  The timed region starts with an empty pipeline and ends with a drain: about 1.5 ms "
                         "of ramp-up and tail, 0.8 %% of the default 1000 steps (200 for the 4K workloads, whose buffer sets are 9 x larger)")
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--workload", default="720p", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-gpu", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0,
                    help="strong scaling: a FIXED batch of this many frames per step split over the GPUs (BASELINE configs[3]: 64); "
                         "default 0 = weak scaling, --frames-per-gpu frames on every GPU")
    ap.add_argument("--slots", type=int, default=6, help="groups in flight per GPU (stream + host thread each)")
    ap.add_argument("--launch-frames", type=int, default=None,
                    help="most frames a launch group gathers from the submit queue (dynamic batching; the library's limit is 16).  Default: 16, and at "
                         "most two submissions per group (every merged form is a buffer set and a recorded graph per slot: small --frames-per-gpu "
                         "would otherwise multiply the setup); given explicitly, as many submissions as fit")
    ap.add_argument("--no-batching", action="store_true", help="one submission per launch group (no dynamic batching in the pipeline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the 4K, K=6000, LSC and end-to-end sub-objects")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--timed-only", action="store_true", help="the timed region and nothing else (profiling runs: no roofline / parity / baseline legs)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank of an N > 1 launch uses GPU 0: the host side of an N-rank launch (submit loops, slot threads, pinned "
                         "blocks, rendezvous) measured on a one-GPU box; the line then says so and is NOT a scaling point")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / timing reduction only (no GPU work): CPU test of the N>1 path")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1000 if args.workload == "720p" else 200
    group_limit = int(os.environ.get("FSLIC_BENCH_MAX_GROUP", "16"))      # the library's group limit (FSLIC_MAX_GROUP of its build)
    if args.launch_frames is not None and not 1 <= args.launch_frames <= group_limit:
        ap.error("--launch-frames must be in 1..%d (the library's group limit, FSLIC_MAX_GROUP)" % group_limit)

    from fast_slic_amd import batch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare: become the launcher of one process per GPU (same thing torch.distributed.run would do)
        sys.exit(batch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank, local_rank, world = batch.env_rank_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one HIP context per rank: the rank's GPU becomes the only visible device before anything initialises HIP
    dev = batch.bind_rank_to_its_gpu(local_rank, share_device=args.share_device) if (world > 1 and not args.dry_run) else local_rank
    pinned = batch.pin_to_gpu_numa_node(0 if args.share_device else local_rank, world, device_index=dev) if not args.dry_run else None
    dist = batch.init_process_group("gloo", rank, world)     # barrier + timing reduction only; no data-path collective

    def barrier(sync=None):
        if sync:
            sync()
        if dist is not None:
            dist.barrier()
        if sync:
            sync()

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001 * (1 + rank))                   # the slowest rank decides
        mine = time.perf_counter() - t0                      # this rank's own clock, before the closing barrier
        barrier()
        elapsed = batch.max_over_ranks(time.perf_counter() - t0, dist)
        per_rank = batch.gather_over_ranks(mine, dist, world)
        units = batch.sum_over_ranks(float(args.steps), dist)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed / args.steps * 1e3, 4), "units_all_ranks": units,
                              "per_rank_ms": [round(t * 1e3, 3) for t in per_rank]}))
            sys.stdout.flush()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import torch
    torch.cuda.set_device(dev)
    from fast_slic_amd import Engine

    cfg = WORKLOADS[args.workload]
    H, W, K = cfg["H"], cfg["W"], cfg["K"]
    strong = args.batch > 0
    if strong and args.batch % world:
        raise SystemExit("--batch %d does not split evenly over %d GPUs" % (args.batch, world))
    fpg = (args.batch // world) if strong else (args.frames_per_gpu or cfg["fpg"])
    if fpg < 1 or fpg > 16:
        raise SystemExit("frames per GPU and step must be in 1..16 (one launch group), got %d" % fpg)
    nslots = max(1, args.slots)
    eng = Engine(dev, nslots)
    # Dynamic batching: a slot thread that finds two submissions waiting serves them as one group (16 frames per launch instead
    # of 8).  Results are per frame and do not change; the parity gate below checks a submission of the timed region.
    if args.no_batching:
        merged_max = 1
    elif args.launch_frames is None:
        merged_max = max(1, min(2, 16 // fpg))
    else:
        merged_max = max(1, args.launch_frames // fpg)
    if merged_max > 1:
        eng.pipeline_batching(merged_max * fpg)
    Workload.merged_max = merged_max
    # this rank's shard of the synthetic frame stream: frames are independent, rank r owns seeds r*fpg .. r*fpg+fpg-1
    wl = Workload(eng, H, W, K, batch.frame_shard(args.batch, rank, world) if strong else batch.weak_shard(fpg, rank))
    n_setup = wl.sets_for_setup(merged_max)
    wl.buffers(n_setup + args.warmup + args.steps)
    sync = torch.cuda.synchronize
    # setup, not a step: every slot allocates and carves its arena on its first group and records (and instantiates) its launch
    # sequence as a hipGraph on its second, for every form of group the pipeline can produce.  The groups are addressed to the
    # slots one by one -- handed to "whichever slot is free" a fast slot takes several and another is still unprepared when
    # a short --warmup ends (the driver's 5 warm-up steps left graph instantiation inside its 20 timed steps: 17 instead of
    # 40 GP/s).
    wl.prepare_slots(0, merged_max)
    for j in range(args.warmup):
        wl.submit(n_setup + j)
    eng.pipeline_drain()
    # The interpreter's cyclic garbage collector stays out of the timed region (as timeit does): a full collection of a process
    # that has imported torch takes about 50 ms, and whether one fell into the region depended on the number of buffer sets
    # created before it (192 - 217 steps: 18 instead of 44 GP/s).
    gc.collect()
    gc.disable()
    barrier(sync)
    t0 = time.perf_counter()
    cpu0 = time.process_time()                               # CPU time of this rank's process: its submit loop + slot threads
    first = n_setup + args.warmup
    for j in range(args.steps):
        wl.submit(first + j)
    totals = eng.pipeline_drain()                            # every step of the timed region is complete here
    mine = time.perf_counter() - t0                          # this rank's own clock over its steps (drained, synchronised)
    cpu_mine = time.process_time() - cpu0
    barrier(sync)
    elapsed = batch.max_over_ranks(time.perf_counter() - t0, dist)
    per_rank = batch.gather_over_ranks(mine, dist, world)
    per_rank_cpu = batch.gather_over_ranks(cpu_mine, dist, world)
    gc.enable()

    if rank == 0:
        total_px = float(H) * W * fpg * world * args.steps
        out = {
            "metric": "megapixels/sec @10 iters (1280x720 & 4K, K=1600); % HBM roofline",
            "value": round(total_px / elapsed / 1e6, 2),
            "unit": "megapixels/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u8/u16/i32 (integer SLIC)", "data": "synthetic",
            "config": {"workload": cfg["name"], "frames_per_gpu": fpg, "global_frames": fpg * world,
                       "slots_per_gpu": eng.n_slots, "submissions_per_launch_group": "up to %d (dynamic batching of the pipeline)" % merged_max,
                       "inputs": "device-resident uint8 RGB, device int16 labels",
                       "parallelism": "frames sharded across GPUs, no collective",
                       "host_cpus_of_this_rank": len(pinned) if pinned else None},
            "device_ms_per_frame": round(totals["device_ms"] / max(totals["frames"], 1), 4),
            "host_topk_frames": totals["host_topk_frames"],
        }
        # the whole path against its own ceiling: 37 N algorithmic bytes per frame (LAB 7N, ten fused passes 6N/3 each, full
        # assign 6N, connectivity 4N) at the 8 TB/s of the specification = 216 GP/s per GPU
        out["roofline_path"] = path_roofline(out["value"] / world, "value (whole job / n_gpus)")
        if strong:
            out["config"]["workload"] = "%s: fixed batch of %d frames per step over %d GPU(s), %d per GPU" % (cfg["name"].split(",")[0] + ", K=%d, 10 iters" % K, args.batch, world, fpg)
        out["host_cpu_ms_per_step"] = {"per_rank": [round(c / args.steps * 1e3, 4) for c in per_rank_cpu],
                                       "what": "process CPU time (submit loop + slot threads) of each rank over the timed region / steps"}
        if world > 1:        # each rank's own time over ITS steps (before the closing barrier) and its own rate
            out["legs_skipped"] = ["roofline_4k", "c3_k6000", "lsc_c5", "single_frame", "end_to_end", "cpu_baseline"]      # printed by the N = 1 line only
            if args.share_device:
                out["config"]["share_device"] = "all %d ranks on GPU 0: a host-side load test, not a scaling point" % world
            out["per_rank_ms"] = {"min": round(min(per_rank) * 1e3, 3), "max": round(max(per_rank) * 1e3, 3), "all": [round(t * 1e3, 3) for t in per_rank]}
            out["per_rank_megapixels_per_sec"] = [round(float(H) * W * fpg * args.steps / t / 1e6, 1) for t in per_rank]
    if rank == 0 and args.timed_only:
        print(json.dumps(out))
        sys.stdout.flush()
    elif rank == 0:
        if not args.no_parity:
            bad = parity_of(wl, first + args.steps - 1)
            out["parity"] = not bad
            if bad:
                out["parity_failures"] = bad[:8]
        # ---- roofline of the dominant kernel: the fused assign+update launch, GPU to itself (reproducible: agrees with the
        # rocprofv3 kernel-trace summary under profiles/), then the same launches inside a pipelined (contended) run
        alone = wl.timed_groups(6)
        per_launch_frames = fpg * merged_max
        big = wl.timed_groups(6, merged=merged_max) if merged_max > 1 else alone      # the launches the timed region is made of
        pmc, pmc_src = pmc_traffic("720p_x%d" % per_launch_frames) if args.workload == "720p" else (None, None)
        fu = big["fused_assign_update"]
        out["roofline"] = {
            "kernel": "k_assign_blk2<R,true,3,false> (R = 16 rows per wavefront in launches of more than 2048 eight-row blocks, else 8; fused assign + centroid sums; %d launches per group, %dx%d x %d frames per launch: "
                      "%s), GPU to itself" % (fu["launches_per_group"], W, H, per_launch_frames,
                                              "%d submissions per group, as the pipelined region runs them" % merged_max if merged_max > 1 else "one submission per group"),
            "bound": "hbm", "achieved": fu["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fu["frac"],
            "traffic": pmc["fused_assign_update"]["bytes"] if pmc else None, "traffic_source": pmc_src,
            "bytes_per_launch": fu["bytes_per_launch"], "avg_launch_ms": fu["avg_launch_ms"],
            "full_assign": big["full_assign"], "all_assign_launches": big["all_assign_launches"],
            "device_ms_per_frame": big["device_ms_per_frame"],
        }
        out["roofline"]["path"] = out.pop("roofline_path")      # the fraction for `value` itself, next to the dominant kernel's
        if merged_max > 1:
            out["roofline"]["launch_of_one_submission"] = {"frames_per_launch": fpg, "fused_assign_update": alone["fused_assign_update"],
                                                           "full_assign": alone["full_assign"], "device_ms_per_frame": alone["device_ms_per_frame"]}
        copy_gbs = measured_copy_bandwidth(eng)
        out["hbm_copy_measured"] = {"value": copy_gbs, "unit": "GB/s", "what": "streaming copy of 1 GiB by the library's k_copy16 (16 B per lane and trip; bytes read + bytes written), best of 5, HIP events"}
        out["roofline"]["frac_of_measured_copy"] = round(fu["achieved"] / copy_gbs, 4) if copy_gbs else None
        # the same launch by rocprofv3's kernel duration (committed summary of `scripts/frames.py 720 1280 1600 6 slic 8` under
        # --kernel-trace): events on the stream also see the launch gap of the directly enqueued (timed) group
        rp, rp_tag = rocprof_avg_us("kernel_stats_1280x720_k1600_slic_x%d.csv" % per_launch_frames, "k_assign_blk2", ", true, 3, ") if args.workload == "720p" else (None, None)      # <8,..> for 8-frame launches, _w6<16,..> for 16
        if rp:
            out["roofline"]["by_rocprofv3_kernel_duration"] = {
                "avg_launch_us": rp, "achieved": round(gbs(fu["bytes_per_launch"], rp * 1e-3), 2),
                "frac": round(gbs(fu["bytes_per_launch"], rp * 1e-3) / HBM_PEAK_GBS, 4),
                "source": "profiles/%s_kernel_stats_1280x720_k1600_slic_x%d.csv" % (rp_tag, per_launch_frames)}
        # contended: slot 0's groups carry per-launch events while the other slots keep the GPU busy
        base = len(wl.sets)
        wl.buffers(6 * nslots)
        rec = []
        for r in range(3):                                   # three bursts; every nslots-th group carries per-launch events
            for j in range(2 * nslots):
                eng.set_launch_timing(j == nslots)
                wl.submit(base + 2 * nslots * r + j)
            eng.pipeline_drain()
            eng.set_launch_timing(False)
            for s in range(nslots):
                lm, lp, nl = eng.last_assign_loop(s)
                if nl:
                    rec.append((lm, lp, nl))
        if rec:
            lm, lp, nl = [float(np.mean([r[q] for r in rec])) for q in range(3)]
            out["roofline"]["inside_the_pipeline"] = {
                "concurrent_groups": nslots, "avg_launch_ms": round(lm / nl, 5),
                "achieved": round(gbs(ASSIGN_BYTES_PER_PX * lp, lm), 2), "frac": round(gbs(ASSIGN_BYTES_PER_PX * lp, lm) / HBM_PEAK_GBS, 4),
                "note": "wall time of a launch on a GPU shared by the groups in flight, not a property of the kernel"}
        if not args.no_4k and args.workload == "720p" and world == 1:
            # the north-star kernel: full assign at 3840x2160, K=1600, same engine, after the timed region
            w4 = Workload(eng, 2160, 3840, 1600, list(range(8)))
            four, one = w4.timed_groups(5), w4.timed_groups(4, n=1)
            pmc4, pmc4_src = pmc_traffic("4k_k1600")
            fa = four["full_assign"]
            mp4 = w4.throughput(60)[0]
            out["roofline_4k"] = {"kernel": "k_assign_blk2_w6<32,false,1,true> (full_assign launch, 32 rows per wavefront, 3840x2160, K=1600), 8 frames per launch",
                                  "bound": "hbm", "achieved": fa["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fa["frac"],
                                  "traffic": pmc4["full_assign"]["bytes"] * 8 if pmc4 else None, "traffic_source": pmc4_src,
                                  "avg_launch_ms": fa["avg_launch_ms"], "fused_assign_update": four["fused_assign_update"],
                                  "all_assign_launches": four["all_assign_launches"], "device_ms_per_frame": four["device_ms_per_frame"],
                                  "megapixels_per_sec_pipelined": mp4, "path": path_roofline(mp4, "megapixels_per_sec_pipelined"), "single_frame_launch": one,
                                  "frac_of_measured_copy": round(fa["achieved"] / copy_gbs, 4) if copy_gbs else None}
            if not args.no_parity:      # the last submission of that pipelined region (6 slots, batching, graph replay) against the oracle
                bad = parity_of(w4, w4.last_timed)
                out["roofline_4k"]["parity"] = not bad
                if bad:
                    out["roofline_4k"]["parity_failures"] = bad[:8]
            out["roofline_4k"]["single_frame"] = single_frame_latency(eng, w4)
            del w4
            # BASELINE configs[2]: 3840x2160 K=6000 (centre pile-up, tie at the cut)
            w6 = Workload(eng, 2160, 3840, 6000, list(range(4)))
            t6 = w6.timed_groups(4)
            mp6 = w6.throughput(60)[0]
            out["c3_k6000"] = {"workload": WORKLOADS["4k6000"]["name"], "megapixels_per_sec": mp6, "path": path_roofline(mp6, "megapixels_per_sec"),
                               "dominant_kernel": "k_assign_blk fused assign+update", "fused_assign_update": t6["fused_assign_update"],
                               "full_assign": t6["full_assign"], "all_assign_launches": t6["all_assign_launches"],
                               "device_ms_per_frame": t6["device_ms_per_frame"]}
            if not args.no_parity:      # frame seed 0 ties at the top-K cut (a[K-1] = a[K] = 374): resolved on the device, inside the pipeline
                bad = parity_of(w6, w6.last_timed)
                out["c3_k6000"]["parity"] = not bad
                if bad:
                    out["c3_k6000"]["parity_failures"] = bad[:8]
            del w6
            # BASELINE configs[4]: LSC variant at 1280x720 K=1600 (f32 path; 6 B per visited pixel-assignment, not its bound)
            wlsc = Workload(eng, 720, 1280, 1600, list(range(8)), variant_id=1)
            tl = wlsc.timed_groups(4, bytes_per_px=LSC_ASSIGN_BYTES_PER_PX)
            out["lsc_c5"] = {"workload": "1280x720 LSC variant, K=1600, 8 frames/GPU (configs[4])", "megapixels_per_sec": wlsc.throughput(100)[0],
                             "dominant_kernel": "k_lsc_assign_sums (features rebuilt from the 4 B LAB word, 10-D distance, the fused update sums; priced at 6 B per visited pixel)",
                             "bound": "instruction issue, not HBM: vector ALU busy 0.46 and LDS busy 0.44 of the launch (PMC, profiles/r03_lsc_pmc_instruction_mix.txt); the materialised-planes design (46 B per pixel) measured against this one in profiles/r03_lsc_recompute.txt",
                             "fused_assign_update": tl["fused_assign_update"],
                             "full_assign": tl["full_assign"], "device_ms_per_frame": tl["device_ms_per_frame"]}
            if not args.no_parity:      # tolerance-based (f32 path): fixture floor of the unmodified reference + run-to-run identity
                measured = {}
                bad = lsc_parity_of(wlsc, wlsc.last_timed, wlsc.last_timed - 3, measured)
                out["lsc_c5"]["parity"] = not bad
                out["lsc_c5"]["measured"] = measured
                out["lsc_c5"]["parity_rule"] = "best overlap (both ways) >= 0.99 and mean centre shift <= 0.1 px against tests/golden/lsc_cases.npz (fast_slic.LSC, 1 thread) on frame 0, as tests/test_gpu_lsc.py (north_star's 1e-3 on centroids holds for the integer paths, where they are exact; this f32 path differs from the reference in the order of its sums); every frame and Cluster byte identical between two submissions"
                if bad:
                    out["lsc_c5"]["parity_failures"] = bad[:8]
            del wlsc
            # BASELINE configs[1] says "single-image": one frame per call, device-resident, and through the Python surface
            out["single_frame"] = single_frame_latency(eng, wl)
            out["end_to_end"] = end_to_end(H, W, K)
            ref_e2e = end_to_end_ref_binding(H, W, K)
            if ref_e2e is not None:
                out["end_to_end"]["through_reference_binding"] = ref_e2e
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
