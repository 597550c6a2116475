"""Superpixel-graph utilities (src/fast-slic.cpp), the part that needs no GPU: fslic_hip_knn_connectivity is host
code (K clusters, no pixel data).  Bit-exact against the committed fixtures (tests/golden/graph_cases.npz, generated from
the unmodified reference by make_golden_graph.py) and, where the reference build is present, against it live."""
import os
import numpy as np
import pytest

from fast_slic_amd import _binding as B
from oracle import ref

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_cases.npz"))
KNN_CASES = sorted({k.split("/")[0] for k in GOLD.files if k.endswith("/knn4_num")})


def clusters_of(case):
    raw = GOLD[case + "/clusters"]
    return np.ascontiguousarray(raw).view(B.CLUSTER_DTYPE).reshape(-1)


def shape_of(case):
    return GOLD[case + "/labels"].shape


@pytest.mark.parametrize("case", KNN_CASES)
def test_knn_connectivity_matches_the_golden_vectors(case):
    H, W = shape_of(case)
    cl = clusters_of(case)
    for n in (0, 1, 4, 9):
        num, nb = B.knn_connectivity(H, W, cl, n)
        np.testing.assert_array_equal(num, GOLD["%s/knn%d_num" % (case, n)])
        np.testing.assert_array_equal(nb, GOLD["%s/knn%d_nb" % (case, n)])


def knn_defined(H, W, cl):
    K = cl.shape[0]
    S = max(int(np.sqrt(H * W // K)), 1)
    nh, nw = -(-H // S), -(-W // S)
    v = (cl["y"].astype(np.float64) / S) * nw + cl["x"].astype(np.float64) / S
    return bool(v.max() < nh * nw - 0.5)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref (the reference build) is not present")
def test_knn_connectivity_random_cases_against_the_live_reference():
    rng = np.random.default_rng(4242)
    done = 0
    for trial in range(60):
        H, W = int(rng.integers(16, 400)), int(rng.integers(16, 500))
        K = int(rng.integers(1, max(2, H * W // 40)))
        K = min(K, 3000)
        S = max(int(np.sqrt(H * W // K)), 1)
        y_hi = max((-(-H // S) - 1) * S, 1)
        cl = np.zeros(K, B.CLUSTER_DTYPE)
        cl["number"] = rng.permutation(K)
        frac = rng.integers(0, 3)
        cl["y"] = rng.integers(0, y_hi, K) + (rng.integers(0, 8, K) / 8.0 if frac else 0)
        cl["x"] = rng.integers(0, W, K) + (rng.integers(0, 16, K) / 16.0 if frac == 2 else 0)
        cl["y"] = np.minimum(cl["y"], y_hi - 1)
        cl["x"] = np.minimum(cl["x"], W - 1)
        if not knn_defined(H, W, cl):
            continue
        n = int(rng.integers(0, 14))
        num, nb = B.knn_connectivity(H, W, cl, n)
        rnum, rnb = ref.knn_connectivity(H, W, cl, n)
        np.testing.assert_array_equal(num, rnum, err_msg="trial %d %dx%d K=%d n=%d" % (trial, H, W, K, n))
        np.testing.assert_array_equal(nb, rnb, err_msg="trial %d %dx%d K=%d n=%d" % (trial, H, W, K, n))
        done += 1
    assert done >= 40


def test_knn_connectivity_rejects_what_the_reference_would_index_out_of_range():
    # the reference files a cluster under trunc((y / S) * nw + x / S) unchecked (src/fast-slic.cpp:88); a centre in the last
    # cell row overruns its cell vector.  Here that is an error, not undefined behaviour.
    cl = np.zeros(4, B.CLUSTER_DTYPE)
    cl["number"] = np.arange(4)
    cl["y"] = [1, 2, 3, 99]
    cl["x"] = [1, 2, 3, 99]
    with pytest.raises(ValueError):
        B.knn_connectivity(100, 100, cl, 2)


def test_node_connectivity_tolist():
    num = np.array([2, 0, 1], np.int32)
    nb = np.array([[5, 6, 0], [0, 0, 0], [7, 0, 0]], np.uint32)
    assert B.NodeConnectivity(num, nb).tolist() == [[5, 6], [], [7]]
