"""ctypes binding of the C ABI in include/fslic_hip.h.

Plays the role cfast_slic.pyx plays in the reference: `SlicModel` owns the Cluster[K] block, forwards
`initialize` / `iterate` to the engine and converts the uint16 label map to the int16 map the
reference returns (cfast_slic.pyx:258-260).  Only the arch "hip/gfx950" exists here; the reference's
CPU archs stay in the reference.  There is deliberately NO CPU fallback: if the HIP library is
missing or no GPU is visible, constructing a SlicModel raises.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

ARCH_NAME = "hip/gfx950"
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfslic_hip.so")

CLUSTER_DTYPE = np.dtype([
    ("y", "<f4"), ("x", "<f4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("a", "<f4"),
    ("number", "<u2"), ("is_active", "u1"), ("is_updatable", "u1"), ("num_members", "<u4"),
])  # src/fast-slic-common.h:10-23
assert CLUSTER_DTYPE.itemsize == 32

FSLIC_OK, FSLIC_E_INVALID, FSLIC_E_UNSUPPORTED, FSLIC_E_HIP, FSLIC_E_INTERNAL = range(5)


class Params(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32), ("compactness", C.c_float), ("min_size_factor", C.c_float),
        ("subsample_stride", C.c_int32), ("convert_to_lab", C.c_int32),
        ("manhattan_spatial_dist", C.c_int32), ("preemptive", C.c_int32),
        ("preemptive_thres", C.c_float), ("num_threads", C.c_int32), ("debug_mode", C.c_int32),
        ("abi", C.c_int32), ("variant", C.c_int32), ("reserved", C.c_int32 * 4),
    ]


EXPORTS = [
    "fslic_hip_device_count", "fslic_hip_create", "fslic_hip_destroy", "fslic_hip_initialize_clusters",
    "fslic_hip_iterate", "fslic_hip_iterate_device", "fslic_hip_iterate_batch", "fslic_hip_submit_group",
    "fslic_hip_wait_group", "fslic_hip_rgb_to_lab",
    "fslic_hip_enforce_connectivity", "fslic_hip_last_prelabels", "fslic_hip_last_timing_report",
    "fslic_hip_last_device_times", "fslic_hip_set_launch_timing", "fslic_hip_last_assign_loop", "fslic_hip_last_group_frames", "fslic_hip_last_path", "fslic_hip_last_launch_mode", "fslic_hip_group_done", "fslic_hip_last_error", "fslic_hip_version",
    "fslic_hip_get_connectivity", "fslic_hip_knn_connectivity", "fslic_hip_get_mask_density", "fslic_hip_cluster_density_to_mask",
    "fslic_hip_pipeline_submit", "fslic_hip_pipeline_drain", "fslic_hip_pipeline_batching", "fslic_hip_last_host_topk_frames",
    "fslic_hip_separate_pass_redos", "fslic_hip_copy_bandwidth",
]

_lib = None
_lib_lock = threading.Lock()


def build_library(force=False):
    """Compile fast_slic_amd/libfslic_hip.so for gfx950 with hipcc (in-tree, see csrc/Makefile)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-s", "-C", csrc]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return _LIB_PATH


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.  If this library pulled in the system runtime first
    and torch were imported afterwards, the process would hold two HIP runtimes and torch would report
    "No HIP GPUs are available".  When torch is installed (not necessarily imported) bind to ITS runtime, so
    device pointers from torch tensors and this engine share one runtime whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        _preload_torch_hip_runtime()
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                "fast_slic_amd: %s is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fast_slic_amd/csrc` (there is no CPU fallback)" % _LIB_PATH)
        lib = C.CDLL(os.environ.get("FSLIC_LIB") or _LIB_PATH)      # FSLIC_LIB: A/B a second build of the same ABI (scripts/gpu_ab.sh)
        vp, i32 = C.c_void_p, C.c_int
        lib.fslic_hip_device_count.restype = i32
        lib.fslic_hip_create.argtypes = [i32, i32, C.POINTER(vp)]
        lib.fslic_hip_destroy.argtypes = [vp]
        lib.fslic_hip_destroy.restype = None
        lib.fslic_hip_initialize_clusters.argtypes = [i32, i32, i32, vp, vp]
        lib.fslic_hip_iterate.argtypes = [vp, C.POINTER(Params), i32, i32, i32, vp, vp, vp]
        lib.fslic_hip_iterate_device.argtypes = [vp, i32, C.POINTER(Params), i32, i32, i32, vp, vp, vp]
        lib.fslic_hip_iterate_batch.argtypes = [vp, C.POINTER(Params), i32, i32, i32, i32, vp, vp, vp, i32]
        lib.fslic_hip_submit_group.argtypes = [vp, i32, C.POINTER(Params), i32, i32, i32, i32, vp, vp, vp]
        lib.fslic_hip_wait_group.argtypes = [vp, i32]
        lib.fslic_hip_pipeline_submit.argtypes = [vp, C.POINTER(Params), i32, i32, i32, i32, vp, vp, vp]
        lib.fslic_hip_pipeline_drain.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        lib.fslic_hip_pipeline_batching.argtypes = [vp, i32]
        lib.fslic_hip_last_host_topk_frames.argtypes = [vp, i32]
        lib.fslic_hip_separate_pass_redos.argtypes = [vp, i32]
        if hasattr(lib, "fslic_hip_copy_bandwidth"):      # (an A/B build of an earlier round, FSLIC_LIB, may lack it)
            lib.fslic_hip_copy_bandwidth.argtypes = [vp, C.c_size_t, i32, C.POINTER(C.c_double)]
        lib.fslic_hip_group_done.argtypes = [vp, i32]
        lib.fslic_hip_rgb_to_lab.argtypes = [vp, i32, i32, vp, i32, vp]
        lib.fslic_hip_enforce_connectivity.argtypes = [vp, vp, i32, i32, i32, i32]
        lib.fslic_hip_last_prelabels.argtypes = [vp, i32, vp]
        lib.fslic_hip_last_timing_report.argtypes = [vp]
        lib.fslic_hip_last_timing_report.restype = C.c_char_p
        lib.fslic_hip_last_device_times.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.fslic_hip_set_launch_timing.argtypes = [vp, i32]
        lib.fslic_hip_last_assign_loop.argtypes = [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        lib.fslic_hip_last_group_frames.argtypes = [vp, i32]
        lib.fslic_hip_last_path.argtypes = [vp, i32]
        lib.fslic_hip_last_launch_mode.argtypes = [vp, i32]
        if hasattr(lib, "fslic_hip_lab_force_generic"):           # testing aid, not in include/fslic_hip.h: a library given through
            lib.fslic_hip_lab_force_generic.argtypes = [vp, i32]  # FSLIC_LIB may lack it (only the tests that use it then fail)
        lib.fslic_hip_last_error.restype = C.c_char_p
        lib.fslic_hip_version.restype = C.c_char_p
        lib.fslic_hip_get_connectivity.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        lib.fslic_hip_knn_connectivity.argtypes = [i32, i32, i32, vp, C.c_size_t, vp, vp]
        lib.fslic_hip_get_mask_density.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
        lib.fslic_hip_cluster_density_to_mask.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        _lib = lib
        return lib


def _raise(rc):
    msg = load_library().fslic_hip_last_error().decode("utf-8", "replace")
    if rc == FSLIC_E_INVALID:
        raise ValueError(msg)
    if rc == FSLIC_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def _check(rc):
    if rc != FSLIC_OK:
        _raise(rc)


class Engine(object):
    """One GPU, `n_slots` frames in flight.  Thin owner of an fslic_engine handle."""

    def __init__(self, device=0, n_slots=1):
        lib = load_library()
        h = C.c_void_p()
        _check(lib.fslic_hip_create(int(device), int(n_slots), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.n_slots = int(n_slots)

    def close(self):
        if getattr(self, "_h", None):
            load_library().fslic_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stage entry points -------------------------------------------------------------
    def rgb_to_lab(self, image, convert=True):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        H, W, ch = image.shape
        if ch != 3:
            raise ValueError("nchan != 3")
        out = np.empty((H, W, 4), np.uint8)
        _check(load_library().fslic_hip_rgb_to_lab(self._h, H, W, image.ctypes.data, int(bool(convert)), out.ctypes.data))
        return out

    def enforce_connectivity(self, labels_u16, K, min_threshold):
        out = np.ascontiguousarray(labels_u16, dtype=np.uint16).copy()
        H, W = out.shape
        _check(load_library().fslic_hip_enforce_connectivity(self._h, out.ctypes.data, H, W, int(K), int(min_threshold)))
        return out

    # ---- superpixel-graph utilities (src/fast-slic.cpp); labels / mask / result: numpy arrays or raw device pointers (int) ----
    @staticmethod
    def _plane(a, dtype):
        """(pointer, keep-alive) of a host array or of device memory given as an integer address."""
        if isinstance(a, (int, np.integer)):
            return C.c_void_p(int(a)), None
        arr = np.ascontiguousarray(a, dtype=dtype)
        return C.c_void_p(arr.ctypes.data), arr

    def get_connectivity(self, labels, H, W, K):
        ptr, keep = self._plane(labels, np.uint16)
        num = np.zeros(K, np.int32)
        nb = np.zeros((K, 12), np.uint32)
        _check(load_library().fslic_hip_get_connectivity(self._h, H, W, int(K), ptr, num.ctypes.data, nb.ctypes.data))
        return num, nb

    def get_mask_density(self, clusters, labels, mask, H, W):
        lp, k1 = self._plane(labels, np.uint16)
        mp, k2 = self._plane(mask, np.uint8)
        cl = np.ascontiguousarray(clusters)
        dens = np.zeros(cl.shape[0], np.uint8)
        _check(load_library().fslic_hip_get_mask_density(self._h, H, W, cl.shape[0], cl.ctypes.data, lp, mp, dens.ctypes.data))
        return dens

    def cluster_density_to_mask(self, labels, densities, H, W, out=None):
        lp, k1 = self._plane(labels, np.uint16)
        d = np.ascontiguousarray(densities, dtype=np.uint8)
        if out is None:
            res = np.empty((H, W), np.uint8)
            op = C.c_void_p(res.ctypes.data)
        else:
            res, op = out, C.c_void_p(int(out))
        _check(load_library().fslic_hip_cluster_density_to_mask(self._h, H, W, d.shape[0], lp, d.ctypes.data, op))
        return res

    def iterate(self, image, clusters, params):
        """Host numpy in, host numpy out (uint16 labels); clusters (CLUSTER_DTYPE[K]) updated in place."""
        H, W, _ = image.shape
        labels = np.empty((H, W), np.uint16)
        _check(load_library().fslic_hip_iterate(self._h, C.byref(params), H, W, clusters.shape[0],
                                                image.ctypes.data, clusters.ctypes.data, labels.ctypes.data))
        return labels

    def iterate_device(self, d_rgb_ptr, d_labels_ptr, H, W, clusters, params, slot=0):
        """Frame and label map resident in HBM (raw device pointers, e.g. torch tensor.data_ptr())."""
        _check(load_library().fslic_hip_iterate_device(self._h, int(slot), C.byref(params), H, W, clusters.shape[0],
                                                       C.c_void_p(d_rgb_ptr), clusters.ctypes.data, C.c_void_p(d_labels_ptr)))

    def iterate_batch(self, rgb_ptrs, cluster_arrays, label_ptrs, H, W, params, device_ptrs):
        n = len(rgb_ptrs)
        K = cluster_arrays[0].shape[0]
        PA = C.c_void_p * n
        a_rgb = PA(*[C.c_void_p(int(p)) for p in rgb_ptrs])
        a_cl = PA(*[C.c_void_p(c.ctypes.data) for c in cluster_arrays])
        a_lab = PA(*[C.c_void_p(int(p)) for p in label_ptrs])
        _check(load_library().fslic_hip_iterate_batch(self._h, C.byref(params), H, W, K, n,
                                                      C.cast(a_rgb, C.c_void_p), C.cast(a_cl, C.c_void_p),
                                                      C.cast(a_lab, C.c_void_p), int(bool(device_ptrs))))

    @staticmethod
    def pointer_array(ptrs):
        """A C array of pointers (kept alive by the caller) for submit_group / iterate_batch."""
        PA = C.c_void_p * len(ptrs)
        return PA(*[C.c_void_p(int(p)) for p in ptrs])

    def submit_group(self, slot, a_rgb, a_clusters, a_labels, n, H, W, K, params):
        """Enqueue one group of n device-resident frames on `slot` and return immediately (pointer arrays from
        pointer_array(); the cluster blocks they point to are updated by wait_group)."""
        _check(load_library().fslic_hip_submit_group(self._h, int(slot), C.byref(params), H, W, K, n,
                                                     C.cast(a_rgb, C.c_void_p), C.cast(a_clusters, C.c_void_p),
                                                     C.cast(a_labels, C.c_void_p)))

    def wait_group(self, slot):
        _check(load_library().fslic_hip_wait_group(self._h, int(slot)))

    def pipeline_submit(self, a_rgb, a_clusters, a_labels, n, H, W, K, params):
        """Hand one group of n device-resident frames to whichever slot is free (blocks only while all are occupied)."""
        _check(load_library().fslic_hip_pipeline_submit(self._h, C.byref(params), H, W, K, n,
                                                        C.cast(a_rgb, C.c_void_p), C.cast(a_clusters, C.c_void_p),
                                                        C.cast(a_labels, C.c_void_p)))

    def pipeline_batching(self, max_frames_per_group):
        """Let a slot thread serve several waiting submissions of identical geometry / options as one group of up to
        `max_frames_per_group` (<= 16) frames; 0 switches it off (the default).  Call before the first pipeline_submit."""
        _check(load_library().fslic_hip_pipeline_batching(self._h, int(max_frames_per_group)))

    def pipeline_drain(self):
        """Wait for every submitted group; dict of the totals since the previous drain."""
        ms, g, f, h = C.c_double(), C.c_longlong(), C.c_longlong(), C.c_longlong()
        _check(load_library().fslic_hip_pipeline_drain(self._h, C.byref(ms), C.byref(g), C.byref(f), C.byref(h)))
        return {"device_ms": ms.value, "groups": g.value, "frames": f.value, "host_topk_frames": h.value}

    def last_host_topk_frames(self, slot=0):
        return int(load_library().fslic_hip_last_host_topk_frames(self._h, int(slot)))

    def separate_pass_redos(self):
        """Frames (all slots, since the engine was created) redone with the separate cluster pass because of a stale pixel."""
        return sum(int(load_library().fslic_hip_separate_pass_redos(self._h, s)) for s in range(self.n_slots))

    def copy_bandwidth(self, nbytes=1 << 30, reps=5):
        """Measured HBM rate of this GPU in GB/s: bytes read + written per second of a plain streaming copy of `nbytes`."""
        v = C.c_double()
        _check(load_library().fslic_hip_copy_bandwidth(self._h, C.c_size_t(int(nbytes)), int(reps), C.byref(v)))
        return v.value

    def group_done(self, slot):
        """True when wait_group(slot) would not block."""
        return load_library().fslic_hip_group_done(self._h, int(slot)) != 0

    def last_launch_mode(self, slot=0):
        """0 direct enqueue, 1 recorded as a hipGraph during the call, 2 graph replay."""
        return int(load_library().fslic_hip_last_launch_mode(self._h, int(slot)))

    def last_prelabels(self, H, W, slot=0):
        out = np.empty((H, W), np.uint16)
        _check(load_library().fslic_hip_last_prelabels(self._h, int(slot), out.ctypes.data))
        return out

    def last_timing_report(self):
        return load_library().fslic_hip_last_timing_report(self._h).decode("utf-8")

    def last_device_times(self, slot=0):
        t, f = C.c_float(), C.c_float()
        _check(load_library().fslic_hip_last_device_times(self._h, int(slot), C.byref(t), C.byref(f)))
        return t.value, f.value

    def set_launch_timing(self, on):
        _check(load_library().fslic_hip_set_launch_timing(self._h, int(bool(on))))

    def last_assign_loop(self, slot=0):
        """(sum of the fused assign launches' durations in ms, pixels they visited, number of launches)."""
        ms, px, n = C.c_float(), C.c_double(), C.c_int()
        _check(load_library().fslic_hip_last_assign_loop(self._h, int(slot), C.byref(ms), C.byref(px), C.byref(n)))
        return ms.value, px.value, n.value

    def last_group_frames(self, slot=0):
        return load_library().fslic_hip_last_group_frames(self._h, int(slot))

    def last_path(self, slot=0):
        return load_library().fslic_hip_last_path(self._h, int(slot))

    def lab_force_generic(self, on):
        """Testing aid (not in include/fslic_hip.h): the following Slic groups of this engine take the brute-force gather kernel."""
        _check(load_library().fslic_hip_lab_force_generic(self._h, int(bool(on))))


_default_engines = {}
_engine_lock = threading.Lock()


DEFAULT_ENGINE_SLOTS = 8      # concurrent iterate() calls (different SlicModels, different threads) per GPU before they queue


def default_engine(device=0):
    """The process-wide engine of a GPU, shared by every SlicModel on it.  Its entry points are thread-safe: a call takes
    one of the engine's slots for its duration (cfast_slic.pyx:188-193 runs iterate() without the GIL, and so does ctypes)."""
    with _engine_lock:
        e = _default_engines.get(device)
        if e is None:
            e = Engine(device, DEFAULT_ENGINE_SLOTS)
            _default_engines[device] = e
        return e


def device_count():
    return load_library().fslic_hip_device_count()


def is_supported_arch(arch_name):
    """cfast_slic.is_supported_arch (cfast_slic.pyx:354-357) for the one arch this package provides."""
    return arch_name == ARCH_NAME


def get_supported_archs():
    """cfast_slic.get_supported_archs (cfast_slic.pyx:359-369)."""
    return [ARCH_NAME]


def make_params(max_iter, compactness, min_size_factor, subsample_stride, convert_to_lab=True,
                manhattan_spatial_dist=True, preemptive=False, preemptive_thres=0.05, num_threads=-1,
                debug_mode=False, variant=0):
    p = Params()
    p.abi = PARAMS_ABI
    p.variant = int(variant)        # 0: Slic (Context), 1: LSC (ContextLSC)
    p.max_iter = int(max_iter)
    p.compactness = float(compactness)
    p.min_size_factor = float(min_size_factor)
    p.subsample_stride = int(subsample_stride)
    p.convert_to_lab = int(bool(convert_to_lab))
    p.manhattan_spatial_dist = int(bool(manhattan_spatial_dist))
    p.preemptive = int(bool(preemptive))
    p.preemptive_thres = float(preemptive_thres)
    p.num_threads = int(num_threads)
    p.debug_mode = int(bool(debug_mode))
    return p


PARAMS_ABI = 0x46533033         # FSLIC_PARAMS_ABI, include/fslic_hip.h
VARIANT_SLIC, VARIANT_LSC, VARIANT_REALDIST, VARIANT_REALDIST_L2, VARIANT_REALDIST_NOQ = 0, 1, 2, 3, 4     # fslic_params.variant
_REAL_DIST_VARIANTS = {"standard": VARIANT_REALDIST, "l2": VARIANT_REALDIST_L2, "noq": VARIANT_REALDIST_NOQ, "lsc": VARIANT_LSC}


class SlicModel(object):
    """Mirror of cfast_slic.SlicModel (cfast_slic.pyx:14-328) for arch "hip/gfx950"."""

    def __init__(self, num_components, arch_name=ARCH_NAME, real_dist=False, device=0):
        if isinstance(arch_name, bytes):
            arch_name = arch_name.decode("utf-8")
        if not is_supported_arch(arch_name):
            raise NotImplementedError("Unsupported arch " + repr(arch_name))      # pyx:21-22
        if num_components >= 65534:
            raise ValueError("num_components cannot exceed 65534")                # pyx:24-25
        elif num_components <= 0:
            raise ValueError("num_components should be a non-negative integer")   # pyx:26-27
        self.num_components = int(num_components)
        self.num_threads = -1
        self.arch_name = arch_name
        self.real_dist = real_dist
        self.real_dist_type = "standard"
        self.convert_to_lab = False
        self.float_color = True
        self.debug_mode = False
        self.initialized = False
        self.preemptive = False
        self.preemptive_thres = 0.05
        self.manhattan_spatial_dist = True
        self.last_timing_report = ""
        self.last_recorder_report = ""
        self.device = int(device)
        self._clusters = np.zeros(self.num_components, CLUSTER_DTYPE)              # pyx:38-39
        self._engine = default_engine(self.device)   # raises when no GPU / library: no CPU fallback

    def copy(self):
        # cfast_slic.pyx:45-49 drops arch_name; here the arch is preserved.
        result = SlicModel(self.num_components, self.arch_name, device=self.device)
        result.real_dist = self.real_dist
        result.real_dist_type = self.real_dist_type
        result._clusters[:] = self._clusters
        result.initialized = self.initialized
        return result

    @property
    def clusters(self):                                                            # pyx:51-66
        out = []
        for c in self._clusters:
            out.append(dict(
                number=int(c["number"]),
                yx=(float(c["y"]), float(c["x"])),
                color=(float(c["r"]), float(c["g"]), float(c["b"])),
                num_members=int(c["num_members"]),
            ))
        return out

    @clusters.setter
    def clusters(self, clusters):                                                  # pyx:68-98
        n = len(clusters)
        new = np.zeros(n, CLUSTER_DTYPE)
        for i, d in enumerate(clusters):
            y, x = d["yx"]
            r, g, b = d["color"]
            new[i]["number"] = i
            new[i]["y"] = np.uint16(y)
            new[i]["x"] = np.uint16(x)
            new[i]["r"] = np.uint8(r)
            new[i]["g"] = np.uint8(g)
            new[i]["b"] = np.uint8(b)
            new[i]["num_members"] = np.uint32(d["num_members"])
        self._clusters = new
        self.num_components = n
        self.initialized = True

    @property
    def cluster_array(self):
        """The raw Cluster[K] block (structured numpy view); not part of the reference surface."""
        return self._clusters

    def to_yxmrgb(self):                                                           # pyx:100-113
        c = self._clusters
        return np.stack([c["y"], c["x"], c["num_members"].astype(np.float64), c["r"], c["g"], c["b"]], 1).astype(np.float64)

    @staticmethod
    def _check_image(image):
        image = np.asarray(image)
        if image.dtype != np.uint8 or image.ndim != 3 or not image.flags["C_CONTIGUOUS"]:
            raise ValueError("image must be a C-contiguous uint8 array of shape [H, W, 3]")
        if image.shape[2] != 3:
            raise ValueError("nchan != 3")                                         # pyx:125-126
        return image

    def initialize(self, image):                                                   # pyx:124-147
        image = self._check_image(image)
        H, W, _ = image.shape
        _check(load_library().fslic_hip_initialize_clusters(H, W, self.num_components, image.ctypes.data,
                                                            self._clusters.ctypes.data))
        self.initialized = True

    @staticmethod
    def _check_assignments(assignments):
        a = np.asarray(assignments)
        if a.dtype != np.int16 or a.ndim != 2 or not a.flags["C_CONTIGUOUS"]:
            raise ValueError("assignments must be a C-contiguous int16 array of shape [H, W]")
        return a

    def get_connectivity(self, assignments):                                       # pyx:262-273
        a = self._check_assignments(assignments)
        H, W = a.shape
        num, nb = self._engine.get_connectivity(a.view(np.uint16), H, W, self.num_components)
        return NodeConnectivity(num, nb)

    def get_knn_connectivity(self, assignments, num_neighbors):                    # pyx:275-282
        a = self._check_assignments(assignments)
        H, W = a.shape
        num, nb = knn_connectivity(H, W, self._clusters, num_neighbors)
        return NodeConnectivity(num, nb)

    def get_mask_density(self, mask, assignments):                                 # pyx:284-304
        a = self._check_assignments(assignments)
        m = np.asarray(mask)
        if m.dtype != np.uint8 or m.ndim != 2 or not m.flags["C_CONTIGUOUS"]:
            raise ValueError("mask must be a C-contiguous uint8 array of shape [H, W]")
        H, W = a.shape
        if m.shape[0] != H or m.shape[1] != W:
            raise ValueError("The shape of mask does not match the one of assignments")          # pyx:290-291
        return self._engine.get_mask_density(self._clusters, a.view(np.uint16), m, H, W)

    def broadcast_density_to_mask(self, densities, assignments):                   # pyx:306-324
        a = self._check_assignments(assignments)
        d = np.asarray(densities)
        if d.dtype != np.uint8 or d.ndim != 1:
            raise ValueError("densities must be a uint8 array of shape [K]")
        if d.shape[0] != self.num_components:
            raise ValueError("The shape of densities should match the number of clusters")       # pyx:310-311
        H, W = a.shape
        return self._engine.cluster_density_to_mask(a.view(np.uint16), d, H, W)

    def iterate(self, image, max_iter, compactness, min_size_factor, subsample_stride):   # pyx:150-260
        if not self.initialized:
            raise RuntimeError("Slic model is not initialized")                    # pyx:151-152
        image = self._check_image(image)
        variant = VARIANT_SLIC
        if self.real_dist:                                                         # pyx:198-235
            if self.real_dist_type not in _REAL_DIST_VARIANTS:
                raise RuntimeError("No such real_dist_type " + repr(self.real_dist_type))      # pyx:236
            variant = _REAL_DIST_VARIANTS[self.real_dist_type]
        p = make_params(max_iter, compactness, min_size_factor, subsample_stride, self.convert_to_lab,
                        self.manhattan_spatial_dist, self.preemptive, self.preemptive_thres,
                        self.num_threads, self.debug_mode, variant)
        labels = self._engine.iterate(image, self._clusters, p)
        self.last_timing_report = self._engine.last_timing_report()
        # pyx:258-260 makes an int16 copy and turns 0xFFFF into -1; the int16 VIEW of the same bytes is that array (0xFFFF is -1
        # already), without three passes over the map (250 us at 1280x720, more than the device-side work)
        return labels.view(np.int16)


class NodeConnectivity(object):
    """cfast_slic.NodeConnectivity (cfast_slic.pyx:330-351): per-cluster neighbour lists."""

    def __init__(self, num_neighbors, neighbors):
        self.num_neighbors = num_neighbors
        self.neighbors = neighbors

    def tolist(self):                                                              # pyx:337-347
        return [[int(v) for v in self.neighbors[k, :self.num_neighbors[k]]] for k in range(self.num_neighbors.shape[0])]


def knn_connectivity(H, W, clusters, num_neighbors):
    """fast_slic_knn_connectivity (src/fast-slic.cpp:80-130): host only, no engine needed."""
    cl = np.ascontiguousarray(clusters)
    K = cl.shape[0]
    n = int(num_neighbors)
    if n < 0:
        raise OverflowError("num_neighbors must be non-negative")
    num = np.zeros(K, np.int32)
    nb = np.zeros((K, max(n, 1)), np.uint32)
    _check(load_library().fslic_hip_knn_connectivity(int(H), int(W), K, cl.ctypes.data, n, num.ctypes.data, nb.ctypes.data))
    return num, nb[:, :n]


def enforce_connectivity(assignments, min_threshold, device=0):
    """cfast_slic.enforce_connectivity (cfast_slic.pyx:371-396): in place on an int16 [H, W] array."""
    a = np.asarray(assignments)
    if a.dtype != np.int16 or a.ndim != 2 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("assignments must be a C-contiguous int16 array of shape [H, W]")
    u = a.view(np.uint16)
    valid = u[u != 0xFFFF]
    K = (int(valid.max()) if valid.size else 0) + 1
    out = default_engine(device).enforce_connectivity(u, K, min_threshold)
    u[...] = out
    return assignments
