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


_HIP_FACADE = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(build)r)
sys.path.insert(1, %(root)r)
from fast_slic.hip import SlicHip, LSCHip              # the patch's fast_slic/hip.py (mirrors fast_slic/avx2.py:1-14) over the reference's facade
from fast_slic import Slic
from fast_slic_amd.synth import variant
img = variant("A", 480, 640)
a = Slic(num_components=200, compactness=10, num_threads=4)
b = SlicHip(num_components=200, compactness=10)
la, lb = a.iterate(img), b.iterate(img)
out = {"arch": SlicHip.arch_name, "labels_equal": bool(np.array_equal(la, lb)), "clusters_equal": a.slic_model.clusters == b.slic_model.clusters,
       "dtype": str(lb.dtype), "lsc_arch": LSCHip.arch_name}
l = LSCHip(num_components=100, compactness=10).iterate(img)
out["lsc_shape"] = list(l.shape)
print("RESULT " + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(BUILD, "fast_slic", "hip.pyc")), reason="integration/_build holds no compiled fast_slic package")
def test_fast_slic_hip_facade_classes():
    # the drop-in as a user of the reference types it: `from fast_slic.hip import SlicHip` (hunk fast_slic/hip.py of
    # integration/hip_arch.patch on top of the reference's own facade, fast_slic/base_slic.py:18-88), next to fast_slic.Slic
    r = subprocess.run([sys.executable, "-c", _HIP_FACADE % {"build": BUILD, "root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    import json
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["arch"] == "hip/gfx950" and out["lsc_arch"] == "hip/gfx950"
    assert out["labels_equal"] and out["clusters_equal"] and out["dtype"] == "int16" and out["lsc_shape"] == [480, 640]
