"""ctypes wrapper over oracle/_ref/libfslic_ref.so (the unmodified reference, see ref_harness.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, tests/golden/make_golden.py, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  Never imported by fast_slic_amd.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libfslic_ref.so")

CLUSTER_DTYPE = np.dtype([
    ("y", "<f4"), ("x", "<f4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("a", "<f4"),
    ("number", "<u2"), ("is_active", "u1"), ("is_updatable", "u1"), ("num_members", "<u4"),
])  # src/fast-slic-common.h:10-23 (32 bytes)
assert CLUSTER_DTYPE.itemsize == 32


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.ref_last_timing_report.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def initialize_clusters(image, K, arch="standard"):
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    cl = np.zeros(K, CLUSTER_DTYPE)
    rc = lib().ref_initialize_clusters(arch.encode(), H, W, K, _p(image), _p(cl))
    assert rc == 0
    return cl


def slic_iterate(image, clusters, max_iter=10, compactness=10.0, min_size_factor=0.25,
                 subsample_stride=3, convert_to_lab=True, manhattan=True, num_threads=1,
                 arch="standard", stages=False):
    """Returns (labels_u16[H,W], clusters) or, with stages=True, also (lab[H,W,4], prelabels_u16[H,W])."""
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    K = clusters.shape[0]
    cl = clusters.copy()
    labels = np.zeros((H, W), np.uint16)
    lab = np.zeros((H, W, 4), np.uint8) if stages else None
    pre = np.zeros((H, W), np.uint16) if stages else None
    rc = lib().ref_slic_iterate(arch.encode(), H, W, K, _p(image), _p(cl), int(max_iter),
                                C.c_float(compactness), C.c_float(min_size_factor), int(subsample_stride),
                                int(bool(convert_to_lab)), int(bool(manhattan)), int(num_threads),
                                _p(labels), _p(lab), _p(pre))
    assert rc == 0
    if stages:
        return labels, cl, lab, pre
    return labels, cl


def lsc_iterate(image, clusters, max_iter=10, compactness=10.0, min_size_factor=0.25,
                subsample_stride=3, convert_to_lab=True, num_threads=1, arch="standard"):
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    K = clusters.shape[0]
    cl = clusters.copy()
    labels = np.zeros((H, W), np.uint16)
    pre = np.zeros((H, W), np.uint16)
    rc = lib().ref_lsc_iterate(arch.encode(), H, W, K, _p(image), _p(cl), int(max_iter),
                               C.c_float(compactness), C.c_float(min_size_factor), int(subsample_stride),
                               int(bool(convert_to_lab)), 1, int(num_threads), _p(labels), _p(pre))
    assert rc == 0
    return labels, cl, pre


def set_preemptive(on, thres=0.05):
    """Preemptive mode (cfast_slic.pyx:183-184) of the following *_iterate calls made from this thread."""
    lib().ref_set_preemptive(int(bool(on)), C.c_float(thres))


REALDIST_KINDS = {"standard": 0, "l2": 1, "noq": 2}


def realdist_iterate(image, clusters, kind="standard", max_iter=10, compactness=10.0, min_size_factor=0.25,
                     subsample_stride=3, convert_to_lab=True, manhattan=True, num_threads=1, float_color=True):
    """SlicRealDist / SlicRealDistL2 / SlicRealDistNoQ (fast_slic/base_slic.py:63-85) of the unmodified reference."""
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    K = clusters.shape[0]
    cl = clusters.copy()
    labels = np.zeros((H, W), np.uint16)
    pre = np.zeros((H, W), np.uint16)
    rc = lib().ref_realdist_iterate(REALDIST_KINDS[kind], H, W, K, _p(image), _p(cl), int(max_iter),
                                    C.c_float(compactness), C.c_float(min_size_factor), int(subsample_stride),
                                    int(bool(convert_to_lab)), int(bool(manhattan)), int(num_threads), int(bool(float_color)),
                                    _p(labels), _p(pre))
    assert rc == 0
    return labels, cl, pre


def enforce_connectivity(labels_u16, K, min_threshold, num_threads=1):
    out = np.ascontiguousarray(labels_u16, dtype=np.uint16).copy()
    H, W = out.shape
    lib().ref_enforce_connectivity(_p(out), H, W, int(K), int(min_threshold), int(num_threads))
    return out


def get_connectivity(labels_u16, K):
    """fast_slic_get_connectivity (src/fast-slic.cpp:16-78) -> (num[K] int32, neighbors[K,12] uint32)"""
    a = np.ascontiguousarray(labels_u16, dtype=np.uint16)
    H, W = a.shape
    num = np.zeros(K, np.int32)
    nb = np.zeros((K, 12), np.uint32)
    assert lib().ref_get_connectivity(H, W, int(K), _p(a), _p(num), _p(nb)) == 0
    return num, nb


def knn_connectivity(H, W, clusters, num_neighbors, num_threads=1):
    """fast_slic_knn_connectivity (src/fast-slic.cpp:80-130) -> (num[K] int32, neighbors[K,n] uint32)"""
    cl = np.ascontiguousarray(clusters)
    K = cl.shape[0]
    num = np.zeros(K, np.int32)
    nb = np.zeros((K, max(int(num_neighbors), 1)), np.uint32)
    assert lib().ref_knn_connectivity(int(H), int(W), K, _p(cl), int(num_neighbors), int(num_threads), _p(num), _p(nb)) == 0
    return num, nb[:, :int(num_neighbors)]


def get_mask_density(clusters, labels_u16, mask):
    """fast_slic_get_mask_density (src/fast-slic.cpp:141-154)"""
    a = np.ascontiguousarray(labels_u16, dtype=np.uint16)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    cl = np.ascontiguousarray(clusters)
    H, W = a.shape
    d = np.zeros(cl.shape[0], np.uint8)
    assert lib().ref_get_mask_density(H, W, cl.shape[0], _p(cl), _p(a), _p(m), _p(d)) == 0
    return d


def cluster_density_to_mask(clusters, labels_u16, densities):
    """fast_slic_cluster_density_to_mask (src/fast-slic.cpp:156-168)"""
    a = np.ascontiguousarray(labels_u16, dtype=np.uint16)
    d = np.ascontiguousarray(densities, dtype=np.uint8)
    cl = np.ascontiguousarray(clusters)
    H, W = a.shape
    out = np.zeros((H, W), np.uint8)
    assert lib().ref_cluster_density_to_mask(H, W, cl.shape[0], _p(cl), _p(a), _p(d), _p(out)) == 0
    return out


def lab_tables():
    g = np.zeros(256, np.int32)
    t = np.zeros(8193, np.int32)
    lib().ref_lab_tables(_p(g), _p(t))
    return g, t


def last_timing_report():
    return lib().ref_last_timing_report().decode()


def supports_arch(arch):
    return bool(lib().ref_supports_arch(arch.encode()))


def to_int16(labels_u16):
    """cfast_slic.pyx:258-260"""
    r = labels_u16.astype(np.int16)
    r[labels_u16 == 0xFFFF] = -1
    return r


def cluster_table(cl):
    """[K,6] float32 (y,x,L,a,b,num_members) as hashed in SURVEY 8c."""
    return np.stack([cl["y"], cl["x"], cl["r"], cl["g"], cl["b"],
                     cl["num_members"].astype(np.float32)], 1).astype(np.float32)
