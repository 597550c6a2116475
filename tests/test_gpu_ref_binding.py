"""The REFERENCE's own Cython binding with the arch "hip/gfx950" patched in (integration/hip_arch.patch, built by
integration/build_ref_binding.sh from a scratch copy of the reference; only the built module, integration/_build/cfast_slic*.so,
ships to the GPU box, like oracle/_ref): through the real cfast_slic.SlicModel (cfast_slic.pyx:14-328) the arch
"hip/gfx950" must give what the arch "standard" / "x64/avx2" give.  The models are driven the way fast_slic/base_slic.py:18-52
drives them (attributes, initialize on first use, iterate)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build")

pytestmark = pytest.mark.gpu

_CODE = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(build)r)
sys.path.insert(1, %(root)r)
import cfast_slic
from fast_slic_amd.synth import variant

class Model(object):
    """What fast_slic.base_slic.BaseSlic does with a SlicModel (base_slic.py:18-52, :55-56, :64-72, :84-88)."""
    def __init__(self, arch, K, lsc=False, num_threads=-1, compactness=10):
        m = cfast_slic.SlicModel(K, arch)
        if lsc:
            m.real_dist = True
            m.real_dist_type = "lsc"
        m.convert_to_lab = True; m.preemptive = False; m.preemptive_thres = 0.05
        m.manhattan_spatial_dist = True; m.num_threads = num_threads; m.debug_mode = False
        self.slic_model, self.compactness = m, compactness
    def iterate(self, image, max_iter=10):
        if not self.slic_model.initialized:
            self.slic_model.initialize(image)
        return self.slic_model.iterate(image, max_iter, self.compactness, 0.25, 3)

out = {"archs": cfast_slic.get_supported_archs()}
for (H, W, K, kind) in [(240, 320, 150, "A"), (480, 640, 200, "A"), (720, 1280, 1600, "A"), (200, 264, 120, "C")]:
    img = variant(kind, H, W)
    a, b = Model("standard", K, num_threads=8), Model("hip/gfx950", K)
    la, lb = a.iterate(img), b.iterate(img)
    ca, cb = a.slic_model.clusters, b.slic_model.clusters
    out["%%dx%%d_k%%d_%%s" %% (W, H, K, kind)] = {"labels_equal": bool(np.array_equal(la, lb)), "dtype": str(lb.dtype),
        "clusters_equal": ca == cb, "report": '"iterate"' in b.slic_model.last_timing_report}
    la2, lb2 = a.iterate(img), b.iterate(img)                      # warm start: the Cluster block carried over
    out["%%dx%%d_k%%d_%%s" %% (W, H, K, kind)]["second_call_equal"] = bool(np.array_equal(la2, lb2)) and a.slic_model.clusters == b.slic_model.clusters
img = variant("A", 720, 1280)
out["avx2_equal"] = bool(np.array_equal(Model("x64/avx2", 1600, num_threads=8).iterate(img), Model("hip/gfx950", 1600).iterate(img)))
l1, l2 = Model("standard", 400, lsc=True, num_threads=1).iterate(img), Model("hip/gfx950", 400, lsc=True).iterate(img)
sys.path.insert(2, %(root)r + "/tests")
from util import best_overlap, boundary_iou                       # labels are renumbered by the connectivity pass: compare partitions
out["lsc_best_overlap"] = min(best_overlap(l1, l2), best_overlap(l2, l1))
out["lsc_boundary_iou"] = boundary_iou(l1, l2)
try:
    Model("hip/gfx950", 10).iterate(np.zeros((4, 4, 4), np.uint8))
    out["nchan_error"] = False
except ValueError:
    out["nchan_error"] = True
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not glob.glob(os.path.join(BUILD, "cfast_slic*.so")), reason="integration/_build not built (no /root/reference where build() ran)")
def test_reference_binding_with_hip_arch_equals_reference_cpu_path():
    # a separate interpreter: the patched cfast_slic links libfslic_hip.so itself and must work without fast_slic_amd's binding
    r = subprocess.run([sys.executable, "-c", _CODE % {"build": BUILD, "root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    import json
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    assert "hip/gfx950" in out["archs"] and "standard" in out["archs"]
    for key, v in out.items():
        if isinstance(v, dict):
            assert v["labels_equal"] and v["clusters_equal"] and v["second_call_equal"] and v["report"] and v["dtype"] == "int16", (key, v)
    assert out["avx2_equal"] and out["nchan_error"]
    # the f32 LSC path is tolerance-based: no worse than the reference's own AVX2 variant against its scalar one
    assert out["lsc_best_overlap"] >= 0.90 and out["lsc_boundary_iou"] >= 0.75, (out["lsc_best_overlap"], out["lsc_boundary_iou"])


_GRAPH_UTILS = r"""
import sys, json, os
import numpy as np
sys.path.insert(0, %(build)r)
sys.path.insert(1, %(root)r)
import cfast_slic
from fast_slic_amd.synth import variant

def model(arch, K):
    m = cfast_slic.SlicModel(K, arch)
    m.convert_to_lab = True; m.preemptive = False; m.preemptive_thres = 0.05
    m.manhattan_spatial_dist = True; m.num_threads = 4; m.debug_mode = False
    return m

out = {}
for (H, W, K, kind) in [(240, 320, 150, "A"), (720, 1280, 1600, "A"), (200, 264, 120, "C")]:
    img = variant(kind, H, W)
    a, b = model("standard", K), model("hip/gfx950", K)
    for m in (a, b):
        m.initialize(img)
    la, lb = a.iterate(img, 10, 10, 0.25, 3), b.iterate(img, 10, 10, 0.25, 3)
    assert np.array_equal(la, lb)
    rng = np.random.default_rng(H + K)
    mask = (rng.integers(0, 256, size=(H, W)) * (img[:, :, 0] > 96)).astype(np.uint8)
    dens = rng.integers(0, 256, size=K).astype(np.uint8)
    key = "%%dx%%d_k%%d_%%s" %% (W, H, K, kind)
    out[key] = {
        # SlicModel.get_connectivity, cfast_slic.pyx:262-273 -> fslic_hip_get_connectivity
        "connectivity": a.get_connectivity(la).tolist() == b.get_connectivity(lb).tolist(),
        # SlicModel.get_mask_density, cfast_slic.pyx:285-306 -> fslic_hip_get_mask_density
        "mask_density": bool(np.array_equal(a.get_mask_density(mask, la), b.get_mask_density(mask, lb))),
        # SlicModel.broadcast_density_to_mask, cfast_slic.pyx:308-328 -> fslic_hip_cluster_density_to_mask
        "broadcast": bool(np.array_equal(a.broadcast_density_to_mask(dens, la), b.broadcast_density_to_mask(dens, lb))),
        "n_edges": sum(len(r) for r in b.get_connectivity(lb).tolist()),
    }
    # module-level enforce_connectivity, cfast_slic.pyx:371-396 -> fslic_hip_enforce_connectivity (in place, returns its argument)
    raw = (rng.integers(0, 40, size=(H, W)) + 40 * (np.arange(W)[None, :] // 64)).astype(np.int16)
    r1 = np.asarray(cfast_slic.enforce_connectivity(raw.copy(), 12))
    r2 = np.asarray(cfast_slic.enforce_connectivity(raw.copy(), 12, "hip/gfx950"))
    r3 = np.asarray(cfast_slic.enforce_connectivity(raw.copy(), 12, arch_name="standard"))
    out[key]["enforce"] = bool(np.array_equal(r1, r2)) and bool(np.array_equal(r1, r3)) and not np.array_equal(r1, raw)
# SlicModel.get_knn_connectivity, cfast_slic.pyx:275-283 -> fslic_hip_knn_connectivity.  The reference files a centre under
# trunc((y / S) * nw + x / S) unchecked (src/fast-slic.cpp:88) and overruns its cell vector for centres in the last cell row --
# which is where iterate() leaves some -- so both models get the same centres kept above that row through the `clusters` setter
# (cfast_slic.pyx:76-97); with the centres iterate() left, the hip arch answers ValueError instead of undefined behaviour.
H, W, K = 300, 400, 200
S = int(np.sqrt(H * W // K)); y_hi = (-(-H // S) - 1) * S
rng = np.random.default_rng(7)
cl = [dict(yx=(int(rng.integers(0, y_hi - 1)), int(rng.integers(0, W))), color=(1, 2, 3), num_members=5) for k in range(K)]
a, b = model("standard", K), model("hip/gfx950", K)
a.clusters = cl; b.clusters = cl
dummy = np.zeros((H, W), np.int16)
out["knn"] = all(a.get_knn_connectivity(dummy, n).tolist() == b.get_knn_connectivity(dummy, n).tolist() for n in (1, 4, 9))
out["knn_nonempty"] = sum(len(r) for r in b.get_knn_connectivity(dummy, 4).tolist()) > 0
b.clusters = [dict(yx=(H - 1, W - 1), color=(1, 2, 3), num_members=5)] * K
try:
    b.get_knn_connectivity(dummy, 4)
    out["knn_range_error"] = False
except ValueError:
    out["knn_range_error"] = True
try:
    a.get_mask_density(np.zeros((3, 3), np.uint8), la)
    out["shape_error"] = False
except ValueError:
    out["shape_error"] = True
# device choice: "hip/gfx950" is FSLIC_DEVICE (default 0), "hip/gfx950:N" is device N
img = variant("A", 240, 320)
m0 = model("hip/gfx950:0", 150); m0.initialize(img)
mA = model("standard", 150); mA.initialize(img)
out["device0_equal"] = bool(np.array_equal(m0.iterate(img, 10, 10, 0.25, 3), mA.iterate(img, 10, 10, 0.25, 3)))
out["supported"] = [cfast_slic.is_supported_arch(s) for s in ("hip/gfx950", "hip/gfx950:0", "hip/gfx950:63", "hip/gfx950:x")]
for name, arch in (("no_device_63", "hip/gfx950:63"), ("bad_device", "hip/gfx950:x")):
    try:
        cfast_slic.SlicModel(10, arch)
        out[name] = False
    except NotImplementedError:
        out[name] = True
os.environ["FSLIC_DEVICE"] = "63"
try:
    cfast_slic.SlicModel(10, "hip/gfx950")
    out["env_device_63"] = False
except NotImplementedError:
    out["env_device_63"] = True
print("RESULT " + json.dumps(out))
"""


@pytest.mark.skipif(not glob.glob(os.path.join(BUILD, "cfast_slic*.so")), reason="integration/_build not built (no /root/reference where build() ran)")
def test_reference_binding_graph_utilities_connectivity_and_device_choice():
    """The hunks VERDICT r05 found missing: get_connectivity / get_knn_connectivity / get_mask_density /
    broadcast_density_to_mask (cfast_slic.pyx:262-324) and module-level enforce_connectivity (cfast_slic.pyx:371-396) reach the
    fslic_hip_* entry points through the reference's own binding, and give what the arch "standard" gives."""
    r = subprocess.run([sys.executable, "-c", _GRAPH_UTILS % {"build": BUILD, "root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    import json
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1][7:])
    for key, v in out.items():
        if isinstance(v, dict):
            assert v["connectivity"] and v["mask_density"] and v["broadcast"] and v["enforce"] and v["n_edges"] > 0, (key, v)
    assert out["shape_error"] and out["device0_equal"] and out["knn"] and out["knn_nonempty"] and out["knn_range_error"]
    assert out["supported"] == [True, True, False, False]
    assert out["no_device_63"] and out["bad_device"] and out["env_device_63"]


_HIP_FACADE = r"""
import sys, json
import numpy as np
sys.path.insert(0, %(build)r)
sys.path.insert(1, %(root)r)
import cfast_slic
from fast_slic_amd.base_slic import BaseSlic          # the facade surface of fast_slic/base_slic.py:3-59 (the reference's package itself does not travel)
from fast_slic_amd.synth import variant

class OverRef(BaseSlic):
    def make_slic_model(self, num_components):         # the reference's SlicModel, not fast_slic_amd's ctypes one
        return cfast_slic.SlicModel(num_components, self.arch_name)
class Slic(OverRef):
    arch_name = "standard"
class SlicHip(OverRef):                                # what the patch's fast_slic/hip.py declares
    arch_name = "hip/gfx950"
class LSCHip(OverRef):
    arch_name = "hip/gfx950"
    def make_slic_model(self, num_components):         # fast_slic/base_slic.py:64-72,87-89
        m = cfast_slic.SlicModel(num_components, self.arch_name)
        m.real_dist = True; m.real_dist_type = "lsc"
        return m

img = variant("A", 480, 640)
a = Slic(num_components=200, compactness=10, num_threads=4)
b = SlicHip(num_components=200, compactness=10)
la, lb = a.iterate(img), b.iterate(img)
out = {"arch": SlicHip.arch_name, "labels_equal": bool(np.array_equal(la, lb)), "clusters_equal": a.slic_model.clusters == b.slic_model.clusters,
       "dtype": str(lb.dtype), "model": type(b.slic_model).__module__}
l = LSCHip(num_components=100, compactness=10).iterate(img)
out["lsc_shape"] = list(l.shape)
print("RESULT " + json.dumps(out))
"""


@pytest.mark.skipif(not glob.glob(os.path.join(BUILD, "cfast_slic*.so")), reason="integration/_build not built (no /root/reference where build() ran)")
def test_facade_classes_over_the_reference_binding():
    # BaseSlic-shaped classes over the REAL cfast_slic.SlicModel with arch "hip/gfx950", next to the same facade over arch "standard".
    # (`from fast_slic.hip import SlicHip` itself -- the patch's hunk over the reference's own package -- is imported where the reference
    # exists, by integration/build_ref_binding.sh; no source or bytecode of the reference's Python package is shipped to the GPU box.)
    r = subprocess.run([sys.executable, "-c", _HIP_FACADE % {"build": BUILD, "root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    import json
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["arch"] == "hip/gfx950" and out["model"] == "cfast_slic"
    assert out["labels_equal"] and out["clusters_equal"] and out["dtype"] == "int16" and out["lsc_shape"] == [480, 640]
