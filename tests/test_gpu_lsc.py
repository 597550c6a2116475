"""GPU tests of the LSC variant (src/lsc.cpp; BASELINE.json configs[4]) through the C ABI.

The reference's LSC is f32 with order-dependent sequential sums, so parity is tolerance-based:
the partition must agree with `fast_slic.LSC` (arch "standard", one thread; committed in tests/golden/lsc_cases.npz
by tests/golden/make_golden_lsc.py) at least as well as the reference's own AVX2 variant does -- best-overlap >= 0.90
and boundary IoU >= 0.75 in that contract; the thresholds below are tighter, from what this implementation reaches.
The integer stages (LAB, connectivity on equal input) stay bit-exact and are covered by test_gpu_parity.py.
"""
import json
import os

import numpy as np
import pytest

from fast_slic_amd import LSC, SlicModel, make_params
from fast_slic_amd import _binding as B
from fast_slic_amd.synth import variant
from util import best_overlap, boundary_iou

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lsc_cases():
    return np.load(os.path.join(ROOT, "tests", "golden", "lsc_cases.npz"), allow_pickle=False)


def case_names(npz):
    return sorted({k.split("/")[0] for k in npz.files if k.endswith("/shape")})


def run_lsc(engine, img, K, **kw):
    cl = np.zeros(K, B.CLUSTER_DTYPE)
    B._check(B.load_library().fslic_hip_initialize_clusters(img.shape[0], img.shape[1], K, img.ctypes.data, cl.ctypes.data))
    p = make_params(kw.get("max_iter", 10), kw.get("compactness", 10.0), kw.get("min_size_factor", 0.25),
                    kw.get("subsample_stride", 3), kw.get("convert_to_lab", True), variant=B.VARIANT_LSC)
    labels = engine.iterate(np.ascontiguousarray(img), cl, p)
    pre = engine.last_prelabels(img.shape[0], img.shape[1])
    return labels, cl, pre


# per case: (min pre-connectivity pixel agreement, min best-overlap both ways, min boundary IoU)
FLOORS = {"default": (0.99, 0.99, 0.95)}
CENTRE_SHIFT_FLOOR_PX = 0.1      # mean |centre - reference centre| over the K clusters, in pixels


def test_lsc_agrees_with_reference_fixtures(engine, lsc_cases):
    report = []
    for name in case_names(lsc_cases):
        H, W, K = (int(v) for v in lsc_cases[name + "/shape"])
        kw = json.loads(str(lsc_cases[name + "/kwargs"]))
        img = variant(str(lsc_cases[name + "/variant"]), H, W)
        labels, cl, pre = run_lsc(engine, img, K, **kw)
        r_labels, r_pre, r_cl = lsc_cases[name + "/labels"], lsc_cases[name + "/prelabels"], lsc_cases[name + "/clusters"]
        agree = float((pre == r_pre).mean())
        bo = min(best_overlap(labels, r_labels), best_overlap(r_labels, labels))
        iou = boundary_iou(labels, r_labels)
        dyx = float(np.mean(np.hypot(cl["y"] - r_cl["y"], cl["x"] - r_cl["x"])))
        report.append("%s: pre-agree %.4f best-overlap %.4f boundary-IoU %.4f mean centre shift %.3f px" % (name, agree, bo, iou, dyx))
        lo = FLOORS.get(name, FLOORS["default"])
        assert agree >= lo[0] and bo >= lo[1] and iou >= lo[2], "\n".join(report)
        # centroids: `north_star` asks 1e-3 for the integer path (exact there); this f32 path differs from the reference in the
        # order of its sums, which moves a few boundary pixels and with them a few centres by a pixel:
        # measured 0.000 px on eight fixtures and 0.044 px mean at 1280x720 K=1600; the floor is twice the worst case
        assert dyx <= CENTRE_SHIFT_FLOOR_PX, "\n".join(report)
        if name.endswith("_whole_blocks"):      # blocks of 1024 pixels with one label (the count's ten bits wrap, lsc.hip kLscAccW): exact members
            assert np.array_equal(cl["num_members"], r_cl["num_members"]) and np.array_equal(cl["y"], r_cl["y"]) and np.array_equal(cl["x"], r_cl["x"]), (cl, r_cl)
    print("\n".join(report))


def test_lsc_seed_only_pass_is_nearly_exact(engine, lsc_cases):
    # max_iter = 0: feature map + seed centroids + one full assignment, no order-dependent centroid sums involved
    name = "A_150x200_k50_iter0"
    H, W, K = (int(v) for v in lsc_cases[name + "/shape"])
    labels, cl, pre = run_lsc(engine, variant("A", H, W), K, max_iter=0)
    assert float((pre == lsc_cases[name + "/prelabels"]).mean()) >= 0.995


def test_lsc_is_deterministic_and_group_equals_single(engine):
    H, W, K = 240, 320, 150
    imgs = [variant(v, H, W) for v in "ABCA"]
    singles = [run_lsc(engine, im, K) for im in imgs]
    again = run_lsc(engine, imgs[1], K)
    assert np.array_equal(again[0], singles[1][0]) and np.array_equal(again[2], singles[1][2])
    assert again[1].tobytes() == singles[1][1].tobytes()
    # the same frames as one group (frames share every launch)
    cls = []
    for im in imgs:
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        B._check(B.load_library().fslic_hip_initialize_clusters(H, W, K, im.ctypes.data, cl.ctypes.data))
        cls.append(cl)
    p = make_params(10, 10.0, 0.25, 3, True, variant=B.VARIANT_LSC)
    imgs = [np.ascontiguousarray(im) for im in imgs]
    out = [np.zeros((H, W), np.uint16) for _ in imgs]
    engine.iterate_batch([im.ctypes.data for im in imgs], cls, [o.ctypes.data for o in out], H, W, p, False)
    for z in range(len(imgs)):
        assert np.array_equal(out[z], singles[z][0]), "frame %d of the group differs from its single run" % z
        assert cls[z].tobytes() == singles[z][1].tobytes()


def test_lsc_bench_operating_point_8x720p_batching16():
    # bench.py's `lsc_c5` leg (BASELINE.json configs[4]): 8 x 1280x720 K=1600 LSC per submission through the submit / drain pipeline
    # on 6 slots with batching 16 (two submissions per launch group), graph replay, 24 submissions: every frame of every submission
    # identical to its single-frame run (labels and Cluster bytes: the sums are fixed point, so order does not matter), and frame
    # seed 0 within the fixture floors (best overlap >= 0.99, boundary IoU >= 0.95) of the unmodified reference's A_720x1280_k1600
    import torch
    from fast_slic_amd import Engine
    H, W, K = 720, 1280, 1600
    lib = B.load_library()
    p = make_params(10, 10.0, 0.25, 3, True, variant=B.VARIANT_LSC)
    frames = [np.ascontiguousarray(variant("A", H, W, seed=s)) for s in range(8)]
    cl0 = np.zeros((8, K), B.CLUSTER_DTYPE)
    for i, f in enumerate(frames):
        lib.fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, cl0[i].ctypes.data)
    e = Engine(0, 6)
    try:
        singles = []
        for i, f in enumerate(frames):
            cl = cl0[i].copy()
            singles.append((e.iterate(f, cl, p), cl))
        want = np.load(os.path.join(ROOT, "tests", "golden", "lsc_cases.npz"), allow_pickle=False)["A_720x1280_k1600/labels"]
        bo = min(best_overlap(singles[0][0], want), best_overlap(want, singles[0][0]))    # (the connectivity pass renumbers: compare partitions)
        assert bo >= 0.99 and boundary_iou(singles[0][0], want) >= 0.95, (bo, boundary_iou(singles[0][0], want))
        e.pipeline_batching(16)
        d_rgb = [torch.from_numpy(f).cuda() for f in frames]
        torch.cuda.synchronize()
        a_rgb = e.pointer_array([t.data_ptr() for t in d_rgb])
        subs = []
        for r in range(24):
            d_lab = [torch.full((H, W), -7, dtype=torch.int16, device="cuda") for _ in frames]
            cls = cl0.copy()
            arrs = (e.pointer_array([cls[j].ctypes.data for j in range(8)]), e.pointer_array([t.data_ptr() for t in d_lab]))
            torch.cuda.synchronize()
            e.pipeline_submit(a_rgb, arrs[0], arrs[1], 8, H, W, K, p)
            subs.append((d_lab, cls, arrs))
        tot = e.pipeline_drain()
        assert tot["frames"] == 24 * 8 and tot["groups"] <= 24
        for r, (d_lab, cls, _) in enumerate(subs):
            for j in range(8):
                got = d_lab[j].cpu().numpy().view(np.uint16)
                assert np.array_equal(got, singles[j][0]), "submission %d frame %d: %d pixels differ from the single-frame run" % (r, j, int((got != singles[j][0]).sum()))
                assert cls[j].tobytes() == singles[j][1].tobytes(), "submission %d frame %d: Cluster bytes differ" % (r, j)
        assert 2 in [e.last_launch_mode(sl) for sl in range(6)]
    finally:
        e.close()


def test_lsc_facade_matches_reference_surface(engine):
    img = variant("A", 120, 160)
    s = LSC(num_components=40, compactness=10)
    out = s.iterate(img)
    assert out.dtype == np.int16 and out.shape == (120, 160) and out.min() >= 0
    assert s.slic_model.real_dist and s.slic_model.real_dist_type == "lsc"
    m = SlicModel(10)
    m.real_dist = True
    m.real_dist_type = "no-such-type"
    m.initialize(img)
    with pytest.raises(RuntimeError):
        m.iterate(img, 10, 10.0, 0.25, 3)


@pytest.mark.parametrize("H,W,K,kw", [(64, 64, 4, {}), (33, 200, 7, dict(subsample_stride=1)), (300, 41, 9, dict(subsample_stride=5)),
                                      (96, 128, 300, {}), (17, 23, 1, {})])
def test_lsc_ragged_shapes_run_and_cover(engine, H, W, K, kw):
    img = variant("B", H, W)
    labels, cl, pre = run_lsc(engine, img, K, **kw)
    assert labels.shape == (H, W) and (pre != 0xFFFF).all()
    assert np.isfinite(cl["y"]).all() and (cl["y"] >= 0).all() and (cl["y"] <= H - 1).all() and (cl["x"] <= W - 1).all()


def test_lsc_small_superpixels_on_a_constant_frame_are_deterministic(engine):
    # S = 6: a block's window sees more than 256 bin slots, so a thread of the list build can meet two live records -- which two
    # depends on the order the bin pass inserted them, i.e. it varies from run to run.  Round 6's first parallel list build then fell
    # back to the walk over every cluster: exact, but with the roundings of the other distance form, and on a constant frame (every pixel
    # between two seeds is a tie) 6 .. 25 pixels changed sides from run to run (tests/fuzz_parity.py, lsc mode, seed 1 case 146).
    H, W, K = 287, 687, 5476
    img = np.zeros((H, W, 3), np.uint8)
    for max_iter in (1, 3, 11):
        runs = [run_lsc(engine, img, K, max_iter=max_iter, compactness=20.0, min_size_factor=0.1, subsample_stride=2, convert_to_lab=False) for _ in range(5)]
        for r in runs[1:]:
            assert np.array_equal(r[2], runs[0][2]) and np.array_equal(r[0], runs[0][0]), max_iter
            assert r[1].tobytes() == runs[0][1].tobytes(), max_iter
