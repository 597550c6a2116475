"""Generate the RealDist fixtures from the UNMODIFIED reference (oracle/_ref): SlicRealDist ('standard') and
SlicRealDistL2 ('l2') and SlicRealDistNoQ ('noq'), fast_slic/base_slic.py:63-85.

    make -C oracle ref && python tests/golden/make_golden_realdist.py      (build container only: needs /root/reference)

Output (committed): tests/golden/realdist_cases.npz -- per case the reference's pre-connectivity labels, final labels
and Cluster[K].  These paths are deterministic f32 (one table value + one exactly representable colour term per
candidate), so the GPU tests compare bit for bit.
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
    # name, kind, synthetic variant, H, W, K, kwargs
    ("std_A_120x160_k40", "standard", "A", 120, 160, 40, {}),
    ("std_B_200x150_k30_stride2_c20", "standard", "B", 200, 150, 30, dict(subsample_stride=2, compactness=20.0)),
    ("std_C_240x320_k150", "standard", "C", 240, 320, 150, {}),
    ("std_A_97x131_k25_rgb_euclid", "standard", "A", 97, 131, 25, dict(convert_to_lab=False, manhattan=False)),
    ("std_D_128x192_k16_iter3", "standard", "D", 128, 192, 16, dict(max_iter=3)),
    ("std_A_480x640_k200", "standard", "A", 480, 640, 200, {}),
    ("std_A_480x640_k3000_pileup", "standard", "A", 480, 640, 3000, {}),
    ("l2_A_120x160_k40", "l2", "A", 120, 160, 40, {}),
    ("l2_B_200x150_k30_stride2_c20", "l2", "B", 200, 150, 30, dict(subsample_stride=2, compactness=20.0)),
    ("l2_C_240x320_k150", "l2", "C", 240, 320, 150, {}),
    ("l2_A_97x131_k25_rgb", "l2", "A", 97, 131, 25, dict(convert_to_lab=False)),
    ("l2_A_480x640_k200", "l2", "A", 480, 640, 200, {}),
    ("noq_A_120x160_k40", "noq", "A", 120, 160, 40, {}),
    ("noq_B_200x150_k30_stride2_c20", "noq", "B", 200, 150, 30, dict(subsample_stride=2, compactness=20.0)),
    ("noq_C_240x320_k150", "noq", "C", 240, 320, 150, {}),
    ("noq_A_97x131_k25_rgb_euclid", "noq", "A", 97, 131, 25, dict(convert_to_lab=False, manhattan=False)),
    ("noq_A_200x264_k120_euclid", "noq", "A", 200, 264, 120, dict(manhattan=False)),
    ("noq_D_128x192_k16_iter3", "noq", "D", 128, 192, 16, dict(max_iter=3)),
    ("noq_A_480x640_k200", "noq", "A", 480, 640, 200, {}),
    ("noq_A_480x640_k3000_pileup", "noq", "A", 480, 640, 3000, {}),
    ("std_A_720x1280_k1600", "standard", "A", 720, 1280, 1600, {}),
    ("noq_A_720x1280_k1600", "noq", "A", 720, 1280, 1600, {}),
    ("l2_A_720x1280_k1600", "l2", "A", 720, 1280, 1600, {}),
]


def main():
    out = {}
    for name, kind, var, H, W, K, kw in CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        labels, cl, pre = ref.realdist_iterate(img, cl0, kind, num_threads=1, **kw)
        l8, c8, _ = ref.realdist_iterate(img, cl0, kind, num_threads=8, **kw)
        assert np.array_equal(labels, l8) and cl.tobytes() == c8.tobytes(), "reference not thread-count invariant at " + name
        out[name + "/kind"] = np.array(kind)
        out[name + "/variant"] = np.array(var)
        out[name + "/shape"] = np.array([H, W, K], np.int32)
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/prelabels"] = pre
        out[name + "/kwargs"] = np.array(json.dumps(kw))
        print(name, "segments", len(np.unique(labels)))
    np.savez_compressed(os.path.join(HERE, "realdist_cases.npz"), **out)


if __name__ == "__main__":
    main()
