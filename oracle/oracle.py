"""ctypes wrapper over oracle/libslic_oracle.so (plain-C restatement, see slic_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Never imported by fast_slic_amd.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libslic_oracle.so")

CLUSTER_DTYPE = np.dtype([
    ("y", "<f4"), ("x", "<f4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("a", "<f4"),
    ("number", "<u2"), ("is_active", "u1"), ("is_updatable", "u1"), ("num_members", "<u4"),
])
assert CLUSTER_DTYPE.itemsize == 32

_lib = None


def build(force=False):
    src = os.path.join(_HERE, "slic_oracle.c")
    if force or not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def lab_tables():
    g = np.zeros(256, np.int32)
    t = np.zeros(8193, np.int32)
    cb = np.zeros(9, np.int32)
    lib().orc_lab_tables(_p(g), _p(t), _p(cb))
    return g, t, cb


def rgb_to_lab(image):
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    out = np.zeros((H, W, 4), np.uint8)
    lib().orc_rgb_to_lab(_p(image), H, W, _p(out))
    return out


def initialize_clusters(image, K):
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    cl = np.zeros(K, CLUSTER_DTYPE)
    lib().orc_initialize_clusters(H, W, K, _p(image), _p(cl))
    return cl


def slic_iterate(image, clusters, max_iter=10, compactness=10.0, min_size_factor=0.25,
                 subsample_stride=3, convert_to_lab=True, manhattan=True, stages=False):
    H, W, _ = image.shape
    image = np.ascontiguousarray(image)
    K = clusters.shape[0]
    cl = clusters.copy()
    labels = np.zeros((H, W), np.uint16)
    lab = np.zeros((H, W, 4), np.uint8) if stages else None
    pre = np.zeros((H, W), np.uint16) if stages else None
    rc = lib().orc_slic_iterate(H, W, K, _p(image), _p(cl), int(max_iter), C.c_float(compactness),
                                C.c_float(min_size_factor), int(subsample_stride),
                                int(bool(convert_to_lab)), int(bool(manhattan)),
                                _p(labels), _p(lab), _p(pre))
    assert rc == 0
    if stages:
        return labels, cl, lab, pre
    return labels, cl


def enforce_connectivity(labels_u16, K, min_threshold):
    out = np.ascontiguousarray(labels_u16, dtype=np.uint16).copy()
    H, W = out.shape
    lib().orc_enforce_connectivity(_p(out), H, W, int(K), int(min_threshold))
    return out


def heap_select(comps, middle, area):
    comps = np.ascontiguousarray(comps, dtype=np.int32).copy()
    area = np.ascontiguousarray(area, dtype=np.int32)
    lib().orc_heap_select(_p(comps), int(middle), int(comps.shape[0]), _p(area))
    return comps


def S_of(H, W, K):
    return lib().orc_S(H, W, K)


def to_int16(labels_u16):
    """cfast_slic.pyx:258-260"""
    r = labels_u16.astype(np.int16)
    r[labels_u16 == 0xFFFF] = -1
    return r


def cluster_table(cl):
    return np.stack([cl["y"], cl["x"], cl["r"], cl["g"], cl["b"],
                     cl["num_members"].astype(np.float32)], 1).astype(np.float32)


def fnv1a64(values):
    """FNV-1a-64 over uint32 elements as defined in SURVEY.md 8(c)."""
    h = 0xcbf29ce484222325
    for v in np.asarray(values).astype(np.uint32).tolist():
        h = ((h ^ v) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h
