"""Generate the LSC fixtures from the UNMODIFIED reference (oracle/_ref): `LSC`, arch "standard", one thread --
the contract of SURVEY.md section 8 a13.

    make -C oracle ref && python tests/golden/make_golden_lsc.py      (build container only: needs /root/reference)

Output (committed): tests/golden/lsc_cases.npz -- per case the input frame, the initial clusters, and the reference's
pre-connectivity labels, final labels and Cluster[K].  The path is f32 with order-dependent sums, so the GPU tests
compare partitions through agreement metrics (tests/util.py), not bit for bit.
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
    # name, synthetic variant, H, W, K, kwargs
    ("A_120x160_k40", "A", 120, 160, 40, {}),
    ("C_240x320_k150", "C", 240, 320, 150, {}),
    ("B_200x150_k30_stride2_c20", "B", 200, 150, 30, dict(subsample_stride=2, compactness=20.0)),
    ("A_97x131_k25_rgb", "A", 97, 131, 25, dict(convert_to_lab=False)),
    ("D_128x192_k16_iter3", "D", 128, 192, 16, dict(max_iter=3)),
    ("A_150x200_k50_iter0", "A", 150, 200, 50, dict(max_iter=0)),
    ("A_480x640_k200", "A", 480, 640, 200, {}),
    ("A_720x1280_k1600", "A", 720, 1280, 1600, {}),        # BASELINE.json configs[4]
    ("A_96x128_k300_small_S", "A", 96, 128, 300, {}),      # S = 6: more than 64 clusters reach a block (no block list: every cluster is walked)
    # S = 202 on an all-zero raw frame: blocks of 64 x 16 pixels that are ONE cluster's (a count of 1024 wraps the ten bits it has in the
    # block accumulators, and L = a = b = 0 leaves only the column sum to tell such a block from an empty one: lsc.hip, kLscAccW)
    ("D_256x320_k2_whole_blocks", "D", 256, 320, 2, dict(subsample_stride=1, convert_to_lab=False, max_iter=4)),
]


def main():
    out = {}
    for name, var, H, W, K, kw in CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        labels, cl, pre = ref.lsc_iterate(img, cl0, num_threads=1, arch="standard", **kw)
        again, _, _ = ref.lsc_iterate(img, cl0, num_threads=1, arch="standard", **kw)
        assert np.array_equal(labels, again), "reference LSC not repeatable at " + name
        out[name + "/variant"] = np.array(var)
        out[name + "/shape"] = np.array([H, W, K], np.int32)
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/prelabels"] = pre
        out[name + "/kwargs"] = np.array(json.dumps(kw))
        print(name, "segments", len(np.unique(labels)))
    np.savez_compressed(os.path.join(HERE, "lsc_cases.npz"), **out)


if __name__ == "__main__":
    main()
