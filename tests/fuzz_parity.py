#!/usr/bin/env python
"""Parity sweep on the GPU box (longer than the suite's randomized test; same bar: bit-exact against the oracle).
    python tests/fuzz_parity.py [seed=1] [seconds=600]          FUZZ_SKIP=N: start at case N of the seed's sequence; FUZZ_CASES=M: stop after M cases
ONE engine serves the whole sequence on purpose: every case re-carves the arena over what the previous geometries left there (round 6:
a block that bailed out on a candidate-list overflow left the lazily reset label plane undefined, and a later pass summed pixels under
the cluster numbers of an earlier, larger frame -- a memory fault at case 141 of seed 1; tests/test_gpu_parity.py replays that stretch).
Random MID-SIZE shapes (the suite's randomized test stays below 300 px, where the generic kernels run; here the block kernels with
their ragged right / bottom edges, every row count per wavefront and both table forms get random geometry), random options, and every
frame both alone (Engine.iterate) and as one of a group of 2 .. 16 frames of one launch sequence (Engine.iterate_batch, host pointers).
The oracle is the checker; prints two lines per case and a summary, exit code 1 at the first mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fast_slic_amd import Engine, make_params      # noqa: E402
from fast_slic_amd.synth import variant            # noqa: E402
from oracle import oracle as orc                   # noqa: E402
from util import cluster_fields_equal, describe_mismatch      # noqa: E402


def override_shape(rng, H, W, K):
    """FUZZ_SHAPES=tiny: 1 .. 80 px, any K up to the pixel count; =big: up to 2400 x 4000, superpixel side up to 150 (extra draws only when set:
    the default sequences stay what they were)."""
    shapes = os.environ.get("FUZZ_SHAPES", "")
    if shapes == "tiny":
        H, W = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        K = int(rng.integers(1, max(2, int(H * W * float(rng.choice([0.02, 0.2, 0.6, 1.0]))) + 1)))
    elif shapes == "big":
        H, W = int(rng.integers(900, 2400)), int(rng.integers(1500, 4000))
        side = float(rng.choice([8, 16, 24, 40, 72, 100, 150]))
        K = max(1, min(60000, int(H * W / (side * side))))
    return H, W, K


def extreme_options(rng, kw):
    """FUZZ_OPTS=extreme: option values far from the defaults (extra draws only when set)."""
    if os.environ.get("FUZZ_OPTS", "") == "extreme":
        kw["compactness"] = float(rng.choice([0.01, 0.5, 3.0, 150.0, 400.0, 1000.0]))
        kw["subsample_stride"] = int(rng.integers(1, int(os.environ.get("FUZZ_STRIDE_MAX", "12")) + 1))
        kw["max_iter"] = int(rng.choice([0, 1, 2, 13, 25]))
        kw["min_size_factor"] = float(rng.choice([0.0, 0.01, 0.9, 2.0]))
    return kw


def variants_main(seed, budget):
    """Mode `variants` (python tests/fuzz_parity.py SEED SECONDS variants): the same sweep for the paths the plain-C oracle does not restate --
    SlicRealDist / L2 / NoQ, the preemptive mode (Slic and the float family) and the Euclidean spatial table -- against the UNMODIFIED
    reference run live (oracle/_ref, travels as a built library), bit for bit, alone and in groups."""
    from fast_slic_amd import _binding as B
    from oracle import ref
    if not ref.available():
        print("oracle/_ref is not present: nothing to compare with")
        sys.exit(2)
    KIND = {"standard": B.VARIANT_REALDIST, "l2": B.VARIANT_REALDIST_L2, "noq": B.VARIANT_REALDIST_NOQ}
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = n_frames = n_drawn = 0
    skip = int(os.environ.get("FUZZ_SKIP", "0"))
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(60, 900)), int(rng.integers(60, 1300))
        side = float(rng.choice([6, 8, 10, 13, 16, 20, 24, 32, 40, 60]))
        K = max(1, min(60000, int(H * W / (side * side))))
        H, W, K = override_shape(rng, H, W, K)
        kind = str(rng.choice(["slic", "slic", "standard", "l2", "noq"]))
        pre = bool(rng.integers(0, 2))
        thres = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.3]))
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5])),
                  convert_to_lab=bool(rng.integers(0, 4)), manhattan=bool(rng.integers(0, 3)))
        kw = extreme_options(rng, kw)
        if kind == "slic" and not pre and kw["manhattan"]:
            pre = True                                  # (plain Manhattan Slic is the default mode's sweep)
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        cap = 24_000_000 if os.environ.get("FUZZ_SHAPES", "") == "big" else 6_000_000
        if H * W * n > cap:
            n = max(1, cap // (H * W))
        kinds = [("ABCD"[int(rng.integers(0, 4))]) for _ in range(n)]
        seeds = [int(rng.integers(0, 1 << 30)) for _ in kinds]
        n_drawn += 1
        if n_drawn <= skip:
            continue
        tag = "%dx%d K=%d n=%d %s kind=%s preemptive=%s(%.2f) %s" % (W, H, K, n, "".join(kinds), kind, pre, thres, kw)
        print("run case %d %s" % (n_drawn - 1, tag), flush=True)
        frames = [np.ascontiguousarray(variant(k, H, W, seed=sd)) for k, sd in zip(kinds, seeds)]
        p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], kw["manhattan"],
                        preemptive=pre, preemptive_thres=thres, variant=(B.VARIANT_SLIC if kind == "slic" else KIND[kind]))
        exp = []
        ref.set_preemptive(pre, thres)
        try:
            for f in frames:
                cl0 = ref.initialize_clusters(f, K)
                if kind == "slic":
                    r = ref.slic_iterate(f, cl0, num_threads=1, arch="standard", **kw)
                    exp.append((r[0], r[1]))
                else:
                    r = ref.realdist_iterate(f, cl0, kind, num_threads=1, **kw)
                    exp.append((r[0], r[1]))
        finally:
            ref.set_preemptive(False)
        bad = []
        ai = int(os.environ.get("FUZZ_ALONE_INDEX", "0")) % n          # (which frame of the case also runs alone)
        cl = ref.initialize_clusters(frames[ai], K)
        labels = e.iterate(frames[ai], cl, p)
        if not np.array_equal(labels, exp[ai][0]):
            bad.append(describe_mismatch("alone (frame %d)/labels" % ai, labels, exp[ai][0]))
        bad += ["alone (frame %d): %s" % (ai, m) for m in cluster_fields_equal(cl, exp[ai][1])]
        if n > 1:
            cls = [ref.initialize_clusters(f, K) for f in frames]
            outs = [np.empty((H, W), np.uint16) for _ in frames]
            e.iterate_batch([f.ctypes.data for f in frames], cls, [o.ctypes.data for o in outs], H, W, p, device_ptrs=False)
            for i in range(n):
                if not np.array_equal(outs[i], exp[i][0]):
                    bad.append(describe_mismatch("group frame %d/labels" % i, outs[i], exp[i][0]))
                bad += ["group frame %d: %s" % (i, m) for m in cluster_fields_equal(cls[i], exp[i][1])]
        n_cases += 1
        n_frames += n
        print("  %s" % ("ok" if not bad else "BAD"), flush=True)
        if bad:
            print("\n".join(bad[:10]))
            sys.exit(1)
    print("fuzz_parity variants seed %d: %d cases, %d frames, all bit-exact; %.0f s" % (seed, n_cases, n_frames, time.time() - t0))
    e.close()


def cca_main(seed, budget):
    """Mode `cca`: enforce_connectivity alone (src/context.cpp:14-20 / cca.cpp) on adversarial label maps of random size -- noise of few
    labels, blobs, stripes across tile seams, checkerboards (more components than the device sorts: host top-K), unassigned pixels
    (0xFFFF), equal areas at the cut -- against the oracle, bit for bit; ONE engine, so every map lands in what the last one left."""
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = 0
    kinds_seen = {}
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(1, 1100)), int(rng.integers(1, 1700))
        if os.environ.get("FUZZ_SHAPES", "") == "big":
            H, W = int(rng.integers(900, 2400)), int(rng.integers(1500, 4000))
        kind = str(rng.choice(["noise", "blobs", "stripes_v", "stripes_h", "checker", "kron", "ramp", "slic"]))
        K = int(rng.choice([1, 2, 5, 50, 500, 5000, 60000]))
        if kind == "noise":
            lab = rng.integers(0, min(K, int(rng.choice([2, 3, 5, 17]))), (H, W))
        elif kind == "blobs":
            c = int(rng.choice([4, 8, 16, 32, 64]))
            lab = np.kron(rng.integers(0, min(K, 9), (H // c + 1, W // c + 1)), np.ones((c, c), int))[:H, :W]
        elif kind == "stripes_v":
            w = int(rng.integers(1, 9)); lab = (np.arange(W)[None, :] // w % min(K, 3) + np.zeros((H, 1), int))
        elif kind == "stripes_h":
            w = int(rng.integers(1, 9)); lab = (np.arange(H)[:, None] // w % min(K, 3) + np.zeros((1, W), int))
        elif kind == "checker":
            c = int(rng.choice([1, 1, 2, 3])); lab = ((np.arange(H)[:, None] // c + np.arange(W)[None, :] // c) % min(K, 2)) + np.zeros((H, W), int)
        elif kind == "kron":
            c = int(rng.choice([8, 16, 32])); gy, gx = H // c + 1, W // c + 1
            lab = np.kron((np.arange(gy * gx) % K).reshape(gy, gx), np.ones((c, c), int))[:H, :W]
        elif kind == "ramp":
            lab = ((np.arange(H)[:, None] * 7 + np.arange(W)[None, :] * 3) // int(rng.integers(5, 400))) % K + np.zeros((H, W), int)
        else:                         # a SLIC-like map: nearest of K' random seeds by rows of blocks, ragged borders
            c = int(rng.choice([10, 16, 24, 40])); gy, gx = H // c + 1, W // c + 1
            base = np.kron(np.arange(gy * gx).reshape(gy, gx), np.ones((c, c), int))[:H, :W]
            jit = rng.integers(-c // 2, c // 2 + 1, (H, W)) * (rng.random((H, W)) < 0.15)
            yy = np.clip(np.arange(H)[:, None] + jit, 0, H - 1); xx = np.clip(np.arange(W)[None, :] + jit, 0, W - 1)
            lab = base[yy, xx] % K
        lab = np.ascontiguousarray(lab.astype(np.uint16))
        if rng.random() < 0.3:
            lab[rng.random((H, W)) < float(rng.choice([0.001, 0.05, 0.3]))] = 0xFFFF
        thr = int(rng.choice([0, 1, 2, 4, 16, 64, 300, 5000]))
        tag = "%dx%d %s K=%d thr=%d" % (W, H, kind, K, thr)
        print("run case %d %s" % (n_cases, tag), flush=True)
        got = e.enforce_connectivity(lab, K, thr)
        exp = orc.enforce_connectivity(lab, K, thr)
        if not np.array_equal(got, exp):
            print("  BAD")
            print(describe_mismatch(tag, got, exp))
            sys.exit(1)
        n_cases += 1
        kinds_seen[kind] = kinds_seen.get(kind, 0) + 1
    print("fuzz_parity cca seed %d: %d maps, all bit-exact; %s; %.0f s" % (seed, n_cases, kinds_seen, time.time() - t0))
    e.close()


def lsc_main(seed, budget):
    """Mode `lsc`: the LSC variant (f32, order-dependent sums in the reference: no bit parity) on random mid-size shapes -- what must hold
    everywhere: no fault, two runs bit-equal (labels and Cluster bytes), a frame inside a group equals the frame alone; and on the
    structured frames (A, B) agreement with the unmodified reference run live at the fixture floors' level (best overlap >= 0.97)."""
    from fast_slic_amd import _binding as B
    from oracle import ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import best_overlap
    have_ref = ref.available()
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = 0
    worst = 1.0
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(40, 1100)), int(rng.integers(40, 1700))
        side = float(rng.choice([6, 8, 10, 13, 16, 20, 24, 32, 40, 60]))
        K = max(1, min(60000, int(H * W / (side * side))))
        H, W, K = override_shape(rng, H, W, K)
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5])),
                  convert_to_lab=bool(rng.integers(0, 4)))
        kw = extreme_options(rng, kw)
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        cap = 30_000_000 if os.environ.get("FUZZ_SHAPES", "") == "big" else 8_000_000
        if H * W * n > cap:
            n = max(1, cap // (H * W))
        kinds = [("ABCD"[int(rng.integers(0, 4))]) for _ in range(n)]
        seeds = [int(rng.integers(0, 1 << 30)) for _ in kinds]
        tag = "%dx%d K=%d n=%d %s %s" % (W, H, K, n, "".join(kinds), kw)
        print("run case %d %s" % (n_cases, tag), flush=True)
        frames = [np.ascontiguousarray(variant(k, H, W, seed=sd)) for k, sd in zip(kinds, seeds)]
        pre = bool(rng.integers(0, 4) == 0)
        thres = float(rng.choice([0.0, 0.05, 0.3]))
        p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], True,
                        preemptive=pre, preemptive_thres=thres, variant=B.VARIANT_LSC)
        if pre:
            print("  (preemptive, threshold %.2f)" % thres, flush=True)
        def init(f):
            cl = np.zeros(K, B.CLUSTER_DTYPE)
            B._check(B.load_library().fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, cl.ctypes.data))
            return cl
        ai = int(rng.integers(0, n))
        cl_a = init(frames[ai]); lab_a = e.iterate(frames[ai], cl_a, p).copy()
        cl_b = init(frames[ai]); lab_b = e.iterate(frames[ai], cl_b, p).copy()
        bad = []
        if not np.array_equal(lab_a, lab_b) or cl_a.tobytes() != cl_b.tobytes():
            bad.append("two runs of frame %d differ (%d pixels)" % (ai, int((lab_a != lab_b).sum())))
        if n > 1:
            cls = [init(f) for f in frames]
            outs = [np.empty((H, W), np.uint16) for _ in frames]
            e.iterate_batch([f.ctypes.data for f in frames], cls, [o.ctypes.data for o in outs], H, W, p, device_ptrs=False)
            if not np.array_equal(outs[ai], lab_a) or cls[ai].tobytes() != cl_a.tobytes():
                bad.append("frame %d inside the group differs from the frame alone (%d pixels)" % (ai, int((outs[ai] != lab_a).sum())))
        note = ""
        if have_ref and not pre and kinds[ai] in "AB" and H * W <= 600_000 and K >= 4:
            cl0 = ref.initialize_clusters(frames[ai], K)
            r_labels, r_cl, r_pre = ref.lsc_iterate(frames[ai], cl0, num_threads=1, **kw)
            bo = min(best_overlap(lab_a, r_labels), best_overlap(r_labels, lab_a))
            worst = min(worst, bo)
            note = " best overlap with the reference %.4f" % bo
            # (the floor is asserted where the comparison is meaningful: frames of a few hundred pixels or a compactness near zero leave the
            # f32, order-dependent sums of the reference to decide most pixels -- 0.87 on a 68x7 frame with K = 97 at compactness 0.01)
            if bo < 0.97 and 1.0 <= kw["compactness"] <= 40.0 and H * W >= 20000:
                bad.append("best overlap %.4f" % bo)
        n_cases += 1
        print("  %s%s" % ("ok" if not bad else "BAD", note), flush=True)
        if bad:
            print("\n".join(bad))
            sys.exit(1)
    print("fuzz_parity lsc seed %d: %d cases, deterministic, group == alone, worst best-overlap %.4f; %.0f s" % (seed, n_cases, worst, time.time() - t0))
    e.close()


def warm_main(seed, budget):
    """Mode `warm`: arbitrary cluster states as INPUT (the reference accepts whatever SlicModel.clusters was set to, cfast_slic.pyx:68-98):
    random positions (fractional, on the borders, outside the image), piles of identical clusters (bins beyond their eight slots: the
    spill list; block lists beyond 64 candidates: the generic redo), random colours, and the state a previous frame left (warm start)
    -- against the oracle, bit for bit, alone and in groups of different states."""
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = 0
    paths = {}
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(40, 900)), int(rng.integers(40, 1400))
        side = float(rng.choice([8, 10, 13, 16, 20, 24, 32, 40, 60]))
        K = max(1, min(20000, int(H * W / (side * side))))
        H, W, K = override_shape(rng, H, W, K)
        kw = dict(max_iter=int(rng.integers(0, 9)), compactness=float(rng.choice([1.0, 10.0, 40.0])),
                  min_size_factor=float(rng.choice([0.0, 0.25])), subsample_stride=int(rng.choice([1, 2, 3, 3, 5])),
                  convert_to_lab=bool(rng.integers(0, 4)), manhattan=bool(rng.integers(0, 4)))
        kw = extreme_options(rng, kw)
        n = int(rng.choice([1, 2, 4, 8]))
        cap = 24_000_000 if os.environ.get("FUZZ_SHAPES", "") == "big" else 5_000_000
        if H * W * n > cap:
            n = max(1, cap // (H * W))
        frames = [np.ascontiguousarray(variant("ABCD"[int(rng.integers(0, 4))], H, W, seed=int(rng.integers(0, 1 << 30)))) for _ in range(n)]
        states = []
        kinds = []
        for f in frames:
            cl = orc.initialize_clusters(f, K)
            kind = str(rng.choice(["grid", "uniform", "piles", "borders", "warm", "one_cell", "fraction"]))
            if kind == "uniform":
                cl["y"] = rng.uniform(0, H, K).astype(np.float32); cl["x"] = rng.uniform(0, W, K).astype(np.float32)
            elif kind == "piles":
                m = max(1, K // int(rng.choice([2, 9, 40])))
                py = rng.integers(0, H, m); px = rng.integers(0, W, m); pick = rng.integers(0, m, K)
                cl["y"] = py[pick].astype(np.float32); cl["x"] = px[pick].astype(np.float32)
                if rng.random() < 0.5:
                    pc = rng.integers(0, 256, (m, 3)); cl["r"], cl["g"], cl["b"] = (pc[pick, q].astype(np.float32) for q in range(3))
            elif kind == "borders":
                cl["y"] = rng.choice([-3.0, 0.0, 0.4, H - 1.0, H - 0.5, H + 7.0], K).astype(np.float32)
                cl["x"] = rng.uniform(-5, W + 5, K).astype(np.float32)
            elif kind == "warm":
                other = np.ascontiguousarray(variant("AB"[int(rng.integers(0, 2))], H, W, seed=int(rng.integers(0, 1 << 30))))
                cl = orc.slic_iterate(other, cl, **kw)[1]
            elif kind == "one_cell":
                cl["y"] = rng.integers(0, min(H, 12), K).astype(np.float32); cl["x"] = rng.integers(0, min(W, 12), K).astype(np.float32)
            elif kind == "fraction":
                cl["y"] = (cl["y"] + rng.uniform(-0.99, 0.99, K)).astype(np.float32); cl["x"] = (cl["x"] + rng.uniform(-0.99, 0.99, K)).astype(np.float32)
            if kind not in ("grid", "warm") and rng.random() < 0.5:
                cl["r"] = rng.integers(0, 256, K).astype(np.float32); cl["g"] = rng.integers(0, 256, K).astype(np.float32); cl["b"] = rng.integers(0, 256, K).astype(np.float32)
            states.append(cl)
            kinds.append(kind)
        tag = "%dx%d K=%d n=%d %s %s" % (W, H, K, n, kinds, kw)
        print("run case %d %s" % (n_cases, tag), flush=True)
        p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], kw["manhattan"])
        exp = [orc.slic_iterate(f, c, **kw) for f, c in zip(frames, states)]
        bad = []
        cl = states[0].copy()
        labels = e.iterate(frames[0], cl, p)
        path = e.last_path()
        paths[path] = paths.get(path, 0) + 1
        if not np.array_equal(labels, exp[0][0]):
            bad.append(describe_mismatch("alone/labels", labels, exp[0][0]))
        bad += cluster_fields_equal(cl, exp[0][1])
        if n > 1:
            cls = [c.copy() for c in states]
            outs = [np.empty((H, W), np.uint16) for _ in frames]
            e.iterate_batch([f.ctypes.data for f in frames], cls, [o.ctypes.data for o in outs], H, W, p, device_ptrs=False)
            for i in range(n):
                if not np.array_equal(outs[i], exp[i][0]):
                    bad.append(describe_mismatch("group frame %d/labels" % i, outs[i], exp[i][0]))
                bad += ["group frame %d: %s" % (i, m) for m in cluster_fields_equal(cls[i], exp[i][1])]
        n_cases += 1
        print("  %s path=%d" % ("ok" if not bad else "BAD", path), flush=True)
        if bad:
            print("\n".join(bad[:10]))
            sys.exit(1)
    print("fuzz_parity warm seed %d: %d cases, all bit-exact; paths %s; %.0f s" % (seed, n_cases, paths, time.time() - t0))
    e.close()


def pipeline_main(seed, budget):
    """Mode `pipeline`: the submit / drain pipeline the bench runs in (several slots, recorded launch sequences replayed, dynamic batching)
    under a RANDOM sequence of submissions drawn from a pool of fourteen geometries / option sets -- more than the eight launch
    sequences a slot keeps, so entries are recorded, replayed, evicted and recorded again while arenas are re-carved between them --
    device-resident frames, every frame of every submission against the oracle."""
    import torch
    rng = np.random.default_rng(seed)
    n_slots = int(rng.choice([2, 3, 6]))
    batching = int(rng.choice([0, 8, 16]))
    e = Engine(0, n_slots)
    if batching:
        e.pipeline_batching(batching)
    pool = []
    for _ in range(14):
        H, W = int(rng.integers(60, 500)), int(rng.integers(60, 700))
        side = float(rng.choice([8, 10, 16, 24, 40]))
        K = max(1, int(H * W / (side * side)))
        kw = dict(max_iter=int(rng.integers(0, 11)), compactness=float(rng.choice([1.0, 10.0, 40.0])), min_size_factor=float(rng.choice([0.0, 0.25])),
                  subsample_stride=int(rng.choice([1, 2, 3, 3, 4])), convert_to_lab=bool(rng.integers(0, 4)), manhattan=True)
        pool.append((H, W, K, kw, int(rng.choice([1, 2, 4, 8]))))
    print("pipeline: %d slots, batching %d" % (n_slots, batching), flush=True)
    t0 = time.time()
    n_sub = n_frames = 0
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))           # (submissions)
    while time.time() - t0 < budget and (max_cases == 0 or n_sub < max_cases):
        burst = []                       # submissions between two drains
        for _ in range(int(rng.integers(1, 25))):
            H, W, K, kw, n = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.8 else pool[0]
            frames = [np.ascontiguousarray(variant("ABCD"[int(rng.integers(0, 4))], H, W, seed=int(rng.integers(0, 1 << 30)))) for _ in range(n)]
            cls = np.stack([orc.initialize_clusters(f, K) for f in frames])
            exp = [orc.slic_iterate(f, c, **kw) for f, c in zip(frames, cls)]
            d_rgb = [torch.from_numpy(f).cuda() for f in frames]
            d_lab = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in frames]
            torch.cuda.synchronize()
            a = (e.pointer_array([t.data_ptr() for t in d_rgb]), e.pointer_array([cls[i].ctypes.data for i in range(n)]), e.pointer_array([t.data_ptr() for t in d_lab]))
            p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], kw["manhattan"])
            e.pipeline_submit(a[0], a[1], a[2], n, H, W, K, p)
            burst.append((H, W, K, kw, n, cls, exp, d_rgb, d_lab, a, p))
        e.pipeline_drain()
        for (H, W, K, kw, n, cls, exp, d_rgb, d_lab, a, p) in burst:
            for i in range(n):
                got = d_lab[i].cpu().numpy().view(np.uint16)
                msgs = cluster_fields_equal(cls[i], exp[i][1])
                if not np.array_equal(got, exp[i][0]) or msgs:
                    print("BAD after %d submissions: %dx%d K=%d n=%d frame %d %s" % (n_sub, W, H, K, n, i, kw))
                    print(describe_mismatch("labels", got, exp[i][0]))
                    print("\n".join(msgs))
                    sys.exit(1)
            n_sub += 1
            n_frames += n
        print("  drained: %d submissions so far" % n_sub, flush=True)
    print("fuzz_parity pipeline seed %d (%d slots, batching %d): %d submissions, %d frames, all bit-exact; %.0f s" % (seed, n_slots, batching, n_sub, n_frames, time.time() - t0))
    e.close()


def graph_main(seed, budget):
    """Mode `graph`: the superpixel-graph utilities (src/fast-slic.cpp: get_connectivity with its 12-neighbour cut-off, get_mask_density,
    cluster_density_to_mask) on random label maps of random size -- SLIC-like blocks with ragged borders, noise, frames around blocks,
    few labels -- against the unmodified reference run live, bit for bit."""
    from fast_slic_amd import _binding as B
    from oracle import ref
    if not ref.available():
        print("oracle/_ref is not present: nothing to compare with")
        sys.exit(2)
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = 0
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(1, 900)), int(rng.integers(1, 1400))
        if os.environ.get("FUZZ_SHAPES", "") == "big":
            H, W = int(rng.integers(900, 2400)), int(rng.integers(1500, 4000))
        K = int(rng.choice([1, 2, 7, 60, 400, 3000, 20000, 60000]))
        kind = str(rng.choice(["noise", "blocks", "frames", "few", "slic"]))
        if kind == "noise":
            labels = rng.integers(0, K, (H, W))
        elif kind == "blocks":
            labels = ((np.arange(H)[:, None] // 7) * 13 + np.arange(W)[None, :] // 9) % K + np.zeros((H, W), int)
            m = rng.random((H, W)) < 0.1
            labels[m] = rng.integers(0, K, int(m.sum()))
        elif kind == "frames":
            labels = ((np.arange(H)[:, None] // 4) * 50 + np.arange(W)[None, :] // 4) % K + np.zeros((H, W), int)
            labels[::4] = 0
            labels[:, ::4] = 0
        elif kind == "few":
            labels = rng.integers(0, min(K, 3), (H // 8 + 1, W // 8 + 1)).repeat(8, 0).repeat(8, 1)[:H, :W]
        else:
            c = int(rng.choice([6, 10, 16, 24, 40])); gy, gx = H // c + 1, W // c + 1
            base = np.kron(np.arange(gy * gx).reshape(gy, gx), np.ones((c, c), int))[:H, :W]
            jit = rng.integers(-c // 2, c // 2 + 1, (H, W)) * (rng.random((H, W)) < 0.15)
            yy = np.clip(np.arange(H)[:, None] + jit, 0, H - 1); xx = np.clip(np.arange(W)[None, :] + jit, 0, W - 1)
            labels = base[yy, xx] % K
        labels = np.ascontiguousarray(labels.astype(np.uint16))
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        cl["number"] = np.arange(K)
        cl["num_members"] = rng.integers(0, 50, K)
        mask = (rng.integers(0, 256, (H, W)) * (rng.random((H, W)) < float(rng.choice([0.05, 0.6, 1.0])))).astype(np.uint8)
        tag = "%dx%d %s K=%d" % (W, H, kind, K)
        print("run case %d %s" % (n_cases, tag), flush=True)
        t1 = time.time()
        num, nb = e.get_connectivity(labels, H, W, K)
        t2 = time.time()
        rnum, rnb = ref.get_connectivity(labels, K)
        t3 = time.time()
        if t3 - t1 > 2.0:
            print("  (get_connectivity: engine %.2f s, reference %.2f s)" % (t2 - t1, t3 - t2), flush=True)
        ok = np.array_equal(num, rnum) and np.array_equal(nb, rnb)
        dens = e.get_mask_density(cl, labels, mask, H, W)
        ok = ok and np.array_equal(dens, ref.get_mask_density(cl, labels, mask))
        ok = ok and np.array_equal(e.cluster_density_to_mask(labels, dens, H, W), ref.cluster_density_to_mask(cl, labels, dens))
        if rng.random() < 0.3:           # unassigned pixels: the density utilities skip them
            holes = labels.copy()
            holes[rng.random((H, W)) < 0.1] = 0xFFFF
            dens = e.get_mask_density(cl, holes, mask, H, W)
            ok = ok and np.array_equal(dens, ref.get_mask_density(cl, holes, mask))
            ok = ok and np.array_equal(e.cluster_density_to_mask(holes, dens, H, W), ref.cluster_density_to_mask(cl, holes, dens))
        if not ok:
            print("  BAD")
            sys.exit(1)
        n_cases += 1
    print("fuzz_parity graph seed %d: %d maps, all bit-exact; %.0f s" % (seed, n_cases, time.time() - t0))
    e.close()


def threads_main(seed, budget):
    """Mode `threads`: ONE engine of four slots shared by six Python threads that call iterate / enforce_connectivity / get_mask_density
    on random shapes at the same time (what four `Slic.iterate` callers of one process do to the default engine): slot leasing, per-slot
    arenas and launch-sequence caches under contention; every result against the oracle."""
    import threading
    e = Engine(0, 4)
    t0 = time.time()
    errors = []
    counts = [0] * 6
    def worker(tid):
        rng = np.random.default_rng(seed * 100 + tid)
        try:
            while time.time() - t0 < budget and not errors:
                H, W = int(rng.integers(20, 500)), int(rng.integers(20, 700))
                side = float(rng.choice([6, 8, 10, 16, 24, 40]))
                K = max(1, min(60000, int(H * W / (side * side))))
                op = int(rng.integers(0, 4))
                img = np.ascontiguousarray(variant("ABCD"[int(rng.integers(0, 4))], H, W, seed=int(rng.integers(0, 1 << 30))))
                kw = dict(max_iter=int(rng.integers(0, 11)), compactness=float(rng.choice([1.0, 10.0, 80.0])), min_size_factor=float(rng.choice([0.0, 0.25])),
                          subsample_stride=int(rng.choice([1, 2, 3, 3, 5])), convert_to_lab=bool(rng.integers(0, 4)), manhattan=True)
                if op <= 1:
                    p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], True)
                    cl = orc.initialize_clusters(img, K)
                    exp = orc.slic_iterate(img, cl, **kw)
                    labels = e.iterate(img, cl, p)
                    if not np.array_equal(labels, exp[0]) or cluster_fields_equal(cl, exp[1]):
                        errors.append("thread %d: iterate %dx%d K=%d %s" % (tid, W, H, K, kw))
                elif op == 2:
                    lab = np.ascontiguousarray(rng.integers(0, min(K, 5), (H, W)).astype(np.uint16))
                    thr = int(rng.choice([0, 4, 64]))
                    if not np.array_equal(e.enforce_connectivity(lab, K, thr), orc.enforce_connectivity(lab, K, thr)):
                        errors.append("thread %d: enforce_connectivity %dx%d K=%d thr=%d" % (tid, W, H, K, thr))
                else:
                    from fast_slic_amd import _binding as B
                    lab = np.ascontiguousarray(rng.integers(0, K, (H, W)).astype(np.uint16))
                    mask = rng.integers(0, 256, (H, W)).astype(np.uint8)
                    cl = np.zeros(K, B.CLUSTER_DTYPE); cl["number"] = np.arange(K); cl["num_members"] = np.bincount(lab.ravel(), minlength=K)[:K]
                    dens = e.get_mask_density(cl, lab, mask, H, W)
                    sums = np.bincount(lab.ravel(), weights=mask.ravel().astype(np.float64), minlength=K)[:K]
                    # (src/fast-slic.cpp:141-150: integer mean of the mask over the cluster's pixels, 0 for an empty cluster)
                    want = np.where(cl["num_members"] > 0, sums // np.maximum(cl["num_members"], 1), 0).astype(dens.dtype)
                    if not np.array_equal(dens, want):
                        errors.append("thread %d: get_mask_density %dx%d K=%d" % (tid, W, H, K))
                counts[tid] += 1
        except Exception as ex:          # noqa: BLE001
            errors.append("thread %d: %r" % (tid, ex))
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errors:
        print("BAD")
        print("\n".join(errors[:10]))
        sys.exit(1)
    print("fuzz_parity threads seed %d: %d calls from 6 threads on 4 slots, all bit-exact; %.0f s" % (seed, sum(counts), time.time() - t0))
    e.close()


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "threads":
        return threads_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "graph":
        return graph_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "warm":
        return warm_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "pipeline":
        return pipeline_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "variants":
        return variants_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "cca":
        return cca_main(int(sys.argv[1]), float(sys.argv[2]))
    if len(sys.argv) > 3 and sys.argv[3] == "lsc":
        return lsc_main(int(sys.argv[1]), float(sys.argv[2]))
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
    rng = np.random.default_rng(seed)
    e = Engine(0, 1)
    t0 = time.time()
    n_cases = n_frames = n_drawn = 0
    skip = int(os.environ.get("FUZZ_SKIP", "0"))
    max_cases = int(os.environ.get("FUZZ_CASES", "0"))
    only_alone = os.environ.get("FUZZ_ONLY", "") == "alone"
    only_group = os.environ.get("FUZZ_ONLY", "") == "group"
    paths = {}
    shapes = os.environ.get("FUZZ_SHAPES", "")          # "tiny": 1 .. 80 px, any K up to the pixel count and beyond; "big": up to 2400 x 4000, side up to 150
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(120, 1300)), int(rng.integers(120, 2000))
        # superpixel side between ~6 and ~60 pixels: both table forms, every rows-per-wavefront choice
        side = float(rng.choice([6, 8, 10, 13, 16, 20, 24, 32, 40, 48, 60]))
        K = max(1, min(60000, int(H * W / (side * side))))
        if os.environ.get("FUZZ_K"):               # FUZZ_K=65533: the largest cluster count the reference accepts (cfast_slic.pyx:24-27), on every shape that holds it
            K = int(os.environ["FUZZ_K"])
            if H * W < K:
                H, W = max(H, 300), max(W, 300)
        if shapes == "tiny":
            H, W = int(rng.integers(1, 80)), int(rng.integers(1, 80))
            K = int(rng.integers(1, max(2, int(H * W * float(rng.choice([0.02, 0.2, 0.6, 1.0]))) + 1)))
        elif shapes == "big":
            H, W = int(rng.integers(900, 2400)), int(rng.integers(1500, 4000))
            side = float(rng.choice([8, 16, 24, 40, 72, 100, 150]))
            K = max(1, min(60000, int(H * W / (side * side))))
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7])),
                  convert_to_lab=bool(rng.integers(0, 4)), manhattan=True)
        kw = extreme_options(rng, kw)
        if os.environ.get("FUZZ_OPTS", "") == "extreme":
            kw["manhattan"] = bool(rng.integers(0, 2))
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        if H * W * n > (40_000_000 if shapes == "big" else 12_000_000):
            n = max(1, (40_000_000 if shapes == "big" else 12_000_000) // (H * W))
        kinds = [("ABCD"[int(rng.integers(0, 4))]) for _ in range(n)]
        seeds = [int(rng.integers(0, 1 << 30)) for _ in kinds]
        n_drawn += 1
        if n_drawn <= skip:                     # (FUZZ_SKIP=N: replay from case N of the same seed, e.g. the one a fault interrupted)
            continue
        tag = "%dx%d K=%d n=%d %s %s" % (W, H, K, n, "".join(kinds), kw)
        print("run case %d %s" % (n_drawn - 1, tag), flush=True)
        frames = [np.ascontiguousarray(variant(k, H, W, seed=sd)) for k, sd in zip(kinds, seeds)]
        p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], kw["manhattan"])
        exp = []
        for f in frames:
            cl0 = orc.initialize_clusters(f, K)
            exp.append(orc.slic_iterate(f, cl0, **kw))
        # alone
        cl = orc.initialize_clusters(frames[0], K)
        labels = e.iterate(frames[0], cl, p) if not only_group else exp[0][0]
        if only_group:
            cl = exp[0][1]
        path = e.last_path() if not only_group else -1
        paths[path] = paths.get(path, 0) + 1
        bad = []
        if not np.array_equal(labels, exp[0][0]):
            bad.append(describe_mismatch("alone/labels", labels, exp[0][0]))
        bad += cluster_fields_equal(cl, exp[0][1])
        # as a group
        if n > 1 and not only_alone:
            cls = [orc.initialize_clusters(f, K) for f in frames]
            outs = [np.empty((H, W), np.uint16) for _ in frames]
            e.iterate_batch([f.ctypes.data for f in frames], cls, [o.ctypes.data for o in outs], H, W, p, device_ptrs=False)
            for i in range(n):
                if not np.array_equal(outs[i], exp[i][0]):
                    bad.append(describe_mismatch("group frame %d/labels" % i, outs[i], exp[i][0]))
                bad += ["group frame %d: %s" % (i, m) for m in cluster_fields_equal(cls[i], exp[i][1])]
        n_cases += 1
        n_frames += n
        print("  %s path=%d" % ("ok" if not bad else "BAD", path), flush=True)
        if bad:
            print("\n".join(bad[:10]))
            sys.exit(1)
    print("fuzz_parity seed %d: %d cases, %d frames, all bit-exact; paths %s; %.0f s" % (seed, n_cases, n_frames, paths, time.time() - t0))
    e.close()


if __name__ == "__main__":
    main()
