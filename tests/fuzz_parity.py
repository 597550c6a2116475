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
        kind = str(rng.choice(["slic", "slic", "standard", "l2", "noq"]))
        pre = bool(rng.integers(0, 2))
        thres = float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.3]))
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5])),
                  convert_to_lab=bool(rng.integers(0, 4)), manhattan=bool(rng.integers(0, 3)))
        if kind == "slic" and not pre and kw["manhattan"]:
            pre = True                                  # (plain Manhattan Slic is the default mode's sweep)
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        if H * W * n > 6_000_000:
            n = max(1, 6_000_000 // (H * W))
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
    while time.time() - t0 < budget:
        H, W = int(rng.integers(1, 1100)), int(rng.integers(1, 1700))
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
    while time.time() - t0 < budget:
        H, W = int(rng.integers(40, 1100)), int(rng.integers(40, 1700))
        side = float(rng.choice([6, 8, 10, 13, 16, 20, 24, 32, 40, 60]))
        K = max(1, min(60000, int(H * W / (side * side))))
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5])),
                  convert_to_lab=bool(rng.integers(0, 4)))
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        if H * W * n > 8_000_000:
            n = max(1, 8_000_000 // (H * W))
        kinds = [("ABCD"[int(rng.integers(0, 4))]) for _ in range(n)]
        seeds = [int(rng.integers(0, 1 << 30)) for _ in kinds]
        tag = "%dx%d K=%d n=%d %s %s" % (W, H, K, n, "".join(kinds), kw)
        print("run case %d %s" % (n_cases, tag), flush=True)
        frames = [np.ascontiguousarray(variant(k, H, W, seed=sd)) for k, sd in zip(kinds, seeds)]
        p = make_params(kw["max_iter"], kw["compactness"], kw["min_size_factor"], kw["subsample_stride"], kw["convert_to_lab"], True, variant=B.VARIANT_LSC)
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
        if have_ref and kinds[ai] in "AB" and H * W <= 600_000 and K >= 4:
            cl0 = ref.initialize_clusters(frames[ai], K)
            r_labels, r_cl, r_pre = ref.lsc_iterate(frames[ai], cl0, num_threads=1, **kw)
            bo = min(best_overlap(lab_a, r_labels), best_overlap(r_labels, lab_a))
            worst = min(worst, bo)
            note = " best overlap with the reference %.4f" % bo
            if bo < 0.97 and kw["compactness"] <= 40.0:
                bad.append("best overlap %.4f" % bo)
        n_cases += 1
        print("  %s%s" % ("ok" if not bad else "BAD", note), flush=True)
        if bad:
            print("\n".join(bad))
            sys.exit(1)
    print("fuzz_parity lsc seed %d: %d cases, deterministic, group == alone, worst best-overlap %.4f; %.0f s" % (seed, n_cases, worst, time.time() - t0))
    e.close()


def main():
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
    while time.time() - t0 < budget and (max_cases == 0 or n_cases < max_cases):
        H, W = int(rng.integers(120, 1300)), int(rng.integers(120, 2000))
        # superpixel side between ~6 and ~60 pixels: both table forms, every rows-per-wavefront choice
        side = float(rng.choice([6, 8, 10, 13, 16, 20, 24, 32, 40, 48, 60]))
        K = max(1, min(60000, int(H * W / (side * side))))
        kw = dict(max_iter=int(rng.integers(0, 12)), compactness=float(rng.choice([1.0, 5.0, 10.0, 20.0, 40.0, 80.0])),
                  min_size_factor=float(rng.choice([0.0, 0.1, 0.25, 0.5])), subsample_stride=int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7])),
                  convert_to_lab=bool(rng.integers(0, 4)), manhattan=True)
        n = int(rng.choice([1, 2, 3, 5, 8, 16]))
        if H * W * n > 12_000_000:
            n = max(1, 12_000_000 // (H * W))
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
