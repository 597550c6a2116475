"""Generate the preemptive-mode fixtures from the UNMODIFIED reference (oracle/_ref): Slic(preemptive=True) and the
float-distance variants SlicRealDist / SlicRealDistL2 / SlicRealDistNoQ with preemptive=True, src/preemptive.h.

    make -C oracle ref && python tests/golden/make_golden_preemptive.py      (build container only: needs /root/reference)

Output (committed): tests/golden/preemptive_cases.npz.  The mode is deterministic integer arithmetic (thread-count and
arch invariant, checked here), so the GPU tests compare bit for bit, including the is_updatable counters.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
from fast_slic_amd.synth import variant  # noqa: E402

CASES = [
    # name, synthetic variant, H, W, K, preemptive_thres, kwargs
    ("A_120x160_k40", "A", 120, 160, 40, 0.05, {}),
    ("B_200x150_k30_stride2", "B", 200, 150, 30, 0.05, dict(subsample_stride=2, compactness=20.0)),
    ("C_240x320_k150", "C", 240, 320, 150, 0.05, {}),
    ("A_240x320_k150_thres02", "A", 240, 320, 150, 0.2, {}),
    ("A_240x320_k150_thres0", "A", 240, 320, 150, 0.0, {}),
    ("D_128x192_k16", "D", 128, 192, 16, 0.05, {}),
    ("A_480x640_k200_iter20", "A", 480, 640, 200, 0.05, dict(max_iter=20)),
    ("B_480x640_k200", "B", 480, 640, 200, 0.1, {}),
    ("A_720x1280_k1600", "A", 720, 1280, 1600, 0.05, {}),
]


# the float-distance variants in preemptive mode (BaseContext<float>::iterate is the same template, src/context.cpp:152-181):
# name, kind, synthetic variant, H, W, K, preemptive_thres, kwargs
RD_CASES = [
    ("rd_std_A_120x160_k40", "standard", "A", 120, 160, 40, 0.05, {}),
    ("rd_std_B_200x150_k30_stride2", "standard", "B", 200, 150, 30, 0.05, dict(subsample_stride=2, compactness=20.0)),
    ("rd_std_A_240x320_k150_thres02", "standard", "A", 240, 320, 150, 0.2, {}),
    ("rd_std_A_480x640_k3000_pileup", "standard", "A", 480, 640, 3000, 0.05, {}),
    ("rd_l2_C_240x320_k150", "l2", "C", 240, 320, 150, 0.05, {}),
    ("rd_l2_A_480x640_k200_iter20", "l2", "A", 480, 640, 200, 0.05, dict(max_iter=20)),
    ("rd_noq_A_120x160_k40", "noq", "A", 120, 160, 40, 0.05, {}),
    ("rd_noq_B_200x150_k30_stride2", "noq", "B", 200, 150, 30, 0.05, dict(subsample_stride=2, compactness=20.0)),
    ("rd_noq_A_240x320_k150_thres02", "noq", "A", 240, 320, 150, 0.2, {}),
    ("rd_noq_A_200x264_k120_euclid", "noq", "A", 200, 264, 120, 0.1, dict(manhattan=False)),
    ("rd_noq_A_480x640_k200_iter20", "noq", "A", 480, 640, 200, 0.05, dict(max_iter=20)),
    ("rd_std_A_720x1280_k1600", "standard", "A", 720, 1280, 1600, 0.05, {}),
    ("rd_noq_A_720x1280_k1600", "noq", "A", 720, 1280, 1600, 0.05, {}),
]


def realdist_cases(out):
    for name, kind, var, H, W, K, thres, kw in RD_CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        ref.set_preemptive(True, thres)
        try:
            labels, cl, pre = ref.realdist_iterate(img, cl0, kind, num_threads=1, **kw)
            l8, c8, _ = ref.realdist_iterate(img, cl0, kind, num_threads=8, **kw)
        finally:
            ref.set_preemptive(False)
        assert np.array_equal(labels, l8) and cl.tobytes() == c8.tobytes(), "reference not thread-count invariant at " + name
        out[name + "/kind"] = np.array(kind)
        out[name + "/variant"] = np.array(var)
        out[name + "/shape"] = np.array([H, W, K], np.int32)
        out[name + "/thres"] = np.float32(thres)
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/prelabels"] = pre
        out[name + "/kwargs"] = np.array(json.dumps(kw))
        print(name, "segments", len(np.unique(labels)), "is_updatable", np.unique(cl["is_updatable"], return_counts=True))


# the LSC variant in preemptive mode (f32, order-dependent sums: tolerance-based like tests/golden/make_golden_lsc.py; one thread)
LSC_CASES = [
    ("lsc_A_120x160_k40", "A", 120, 160, 40, 0.05, {}),
    ("lsc_C_240x320_k150", "C", 240, 320, 150, 0.05, {}),
    ("lsc_B_200x150_k30_stride2_c20", "B", 200, 150, 30, 0.05, dict(subsample_stride=2, compactness=20.0)),
    ("lsc_A_240x320_k150_thres02", "A", 240, 320, 150, 0.2, {}),
    ("lsc_A_480x640_k200_iter20", "A", 480, 640, 200, 0.05, dict(max_iter=20)),
    ("lsc_A_720x1280_k1600", "A", 720, 1280, 1600, 0.05, {}),
]


def lsc_cases(out):
    for name, var, H, W, K, thres, kw in LSC_CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        ref.set_preemptive(True, thres)
        try:
            labels, cl, pre = ref.lsc_iterate(img, cl0, num_threads=1, arch="standard", **kw)
            again, _, _ = ref.lsc_iterate(img, cl0, num_threads=1, arch="standard", **kw)
        finally:
            ref.set_preemptive(False)
        assert np.array_equal(labels, again), "reference LSC (preemptive) not repeatable at " + name
        out[name + "/variant"] = np.array(var)
        out[name + "/shape"] = np.array([H, W, K], np.int32)
        out[name + "/thres"] = np.float32(thres)
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/prelabels"] = pre
        out[name + "/kwargs"] = np.array(json.dumps(kw))
        print(name, "segments", len(np.unique(labels)), "is_updatable", np.unique(cl["is_updatable"], return_counts=True))


def main():
    out = {}
    realdist_cases(out)
    lsc_cases(out)
    for name, var, H, W, K, thres, kw in CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        ref.set_preemptive(True, thres)
        try:
            labels, cl, lab, pre = ref.slic_iterate(img, cl0, stages=True, num_threads=1, arch="standard", **kw)
            l8, c8 = ref.slic_iterate(img, cl0, num_threads=8, arch="x64/avx2", **kw)
        finally:
            ref.set_preemptive(False)
        assert np.array_equal(labels, l8) and cl.tobytes() == c8.tobytes(), "reference not thread/arch invariant at " + name
        out[name + "/variant"] = np.array(var)
        out[name + "/shape"] = np.array([H, W, K], np.int32)
        out[name + "/thres"] = np.float32(thres)
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/prelabels"] = pre
        out[name + "/kwargs"] = np.array(json.dumps(kw))
        print(name, "segments", len(np.unique(labels)), "is_updatable", np.unique(cl["is_updatable"], return_counts=True))
    np.savez_compressed(os.path.join(HERE, "preemptive_cases.npz"), **out)


if __name__ == "__main__":
    main()
