"""Generate the committed parity fixtures from the UNMODIFIED reference (oracle/_ref).

Run in the build container, where /root/reference exists:
    make -C oracle ref && python tests/golden/make_golden.py
Outputs (committed):
    tests/golden/small_cases.npz   full stage dumps (LAB, pre-connectivity labels, clusters, labels) of
                                   small frames, plus connectivity-only cases
    tests/golden/digests.json      sha256[:16] digests of the reference outputs at BASELINE.json sizes
The GPU box has no /root/reference; tests there compare the HIP path with these files and with the
plain-C oracle (oracle/slic_oracle.c), which tests/test_oracle.py pins against the same files.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref  # noqa: E402
from fast_slic_amd.synth import variant  # noqa: E402


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


SMALL = [
    # name, variant, H, W, K, kwargs
    ("A_96x128_k24", "A", 96, 128, 24, {}),
    ("C_120x160_k40", "C", 120, 160, 40, {}),
    ("D_64x96_k12", "D", 64, 96, 12, {}),
    ("B_100x75_k9_stride2", "B", 100, 75, 9, dict(subsample_stride=2, compactness=20.0)),
    ("A_130x170_k30_msf0", "A", 130, 170, 30, dict(min_size_factor=0.0)),
    ("A_90x120_k20_rgb", "A", 90, 120, 20, dict(convert_to_lab=False)),
    ("A_128x192_k16_iter3", "A", 128, 192, 16, dict(max_iter=3)),
]

DIGEST_CASES = [
    ("C1_480x640_k200", "A", 480, 640, 200),
    ("C2_720x1280_k1600", "A", 720, 1280, 1600),
    ("C3_2160x3840_k6000", "A", 2160, 3840, 6000),
    ("C3b_2160x3840_k1600", "A", 2160, 3840, 1600),
    ("B_720x1280_k1600", "B", 720, 1280, 1600),
    ("C_480x640_k200", "C", 480, 640, 200),
    ("D_480x640_k100", "D", 480, 640, 100),
]


def main():
    out = {}
    for name, var, H, W, K, kw in SMALL:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        labels, cl, lab, pre = ref.slic_iterate(img, cl0, stages=True, num_threads=1, arch="standard", **kw)
        out[name + "/image"] = img
        out[name + "/clusters0"] = cl0
        out[name + "/labels"] = labels
        out[name + "/clusters"] = cl
        out[name + "/lab"] = lab
        out[name + "/prelabels"] = pre
        out[name + "/K"] = np.int32(K)
        out[name + "/kwargs"] = np.array(json.dumps(kw))
    # connectivity-only cases through the reference's standalone entry (cfast_slic.pyx:371-396)
    rng = np.random.default_rng(1234)
    for i in range(12):
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        nl = int(rng.integers(1, 7))
        lab = rng.integers(0, nl, (H, W)).astype(np.uint16)
        if i % 3 == 0:
            lab[rng.random((H, W)) < 0.15] = 0xFFFF
        if i % 4 == 1:   # blocky
            lab = np.kron(rng.integers(0, nl, ((H + 7) // 8, (W + 7) // 8)), np.ones((8, 8), np.int64))[:H, :W].astype(np.uint16)
        K = int(rng.integers(1, 12))
        thr = int(rng.integers(0, 10))
        res = ref.enforce_connectivity(lab, K, thr)
        out["cca%d/in" % i] = lab
        out["cca%d/out" % i] = res
        out["cca%d/K" % i] = np.int32(K)
        out["cca%d/thr" % i] = np.int32(thr)
    # the one live known-answer case of the reference's own tests (src/cpptest/test_cca.cpp:178-204):
    # expectations re-derived from the live reference, as SURVEY.md section 4 prescribes
    grid = np.array([[1, 1, 1, 2, 2], [1, 3, 1, 2, 2], [1, 1, 1, 2, 4], [5, 5, 5, 5, 4], [5, 6, 6, 5, 4]], np.uint16)
    out["kat5x5/in"] = grid
    out["kat5x5/out"] = ref.enforce_connectivity(grid, 10, 0)
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **out)

    dig = {}
    for name, var, H, W, K in DIGEST_CASES:
        img = variant(var, H, W)
        cl0 = ref.initialize_clusters(img, K)
        labels, cl, lab, pre = ref.slic_iterate(img, cl0, stages=True, num_threads=8, arch="x64/avx2")
        labels_s, cl_s = ref.slic_iterate(img, cl0, num_threads=1, arch="standard")
        assert (labels_s == labels).all() and cl_s.tobytes() == cl.tobytes(), "standard != avx2 at " + name
        dig[name] = dict(variant=var, H=H, W=W, K=K, image=h(img), lab=h(lab), prelabels=h(pre),
                         labels_int16=h(ref.to_int16(labels)), clusters_f32=h(ref.cluster_table(cl)),
                         clusters_raw=h(cl))
        print(name, dig[name])
    g, t = ref.lab_tables()
    dig["tables"] = dict(gamma_i32=h(g.astype(np.int32)), lab_tbl_i32=h(t.astype(np.int32)))
    # all-RGB-cube LAB digest is produced by the C oracle after it is pinned; see tests/test_oracle.py
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
