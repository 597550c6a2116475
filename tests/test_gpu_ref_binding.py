"""The REFERENCE's own Cython binding with the arch "hip/gfx950" patched in (integration/hip_arch.patch, built by
integration/build_ref_binding.sh into integration/_build/fast-slic and shipped to the GPU box like oracle/_ref):
`fast_slic.hip.SlicHip(K).iterate(img)` must equal `fast_slic.Slic(K).iterate(img)` through the real cfast_slic
(cfast_slic.pyx:150-260, fast_slic/avx2.py:1-14 is the pattern fast_slic/hip.py follows)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "integration", "_build", "fast-slic")

pytestmark = pytest.mark.gpu

_CODE = r'''
import sys, json
import numpy as np
sys.path.insert(0, %(build)r)
sys.path.insert(1, %(root)r)
import cfast_slic
from fast_slic import Slic, LSC
from fast_slic.avx2 import SlicAvx2
from fast_slic.hip import SlicHip, LSCHip
from fast_slic_amd.synth import variant
out = {"archs": cfast_slic.get_supported_archs()}
for (H, W, K, kind) in [(240, 320, 150, "A"), (480, 640, 200, "A"), (720, 1280, 1600, "A"), (200, 264, 120, "C")]:
    img = variant(kind, H, W)
    a, b = Slic(num_components=K, compactness=10, num_threads=8), SlicHip(num_components=K, compactness=10)
    la, lb = a.iterate(img), b.iterate(img)
    ca, cb = a.slic_model.clusters, b.slic_model.clusters
    out["%%dx%%d_k%%d_%%s" %% (W, H, K, kind)] = {"labels_equal": bool(np.array_equal(la, lb)), "dtype": str(lb.dtype),
        "clusters_equal": ca == cb, "report": '"iterate"' in b.slic_model.last_timing_report}
    la2, lb2 = a.iterate(img), b.iterate(img)                      # warm start: the Cluster block carried over
    out["%%dx%%d_k%%d_%%s" %% (W, H, K, kind)]["second_call_equal"] = bool(np.array_equal(la2, lb2)) and a.slic_model.clusters == b.slic_model.clusters
img = variant("A", 720, 1280)
out["avx2_equal"] = bool(np.array_equal(SlicAvx2(num_components=1600, compactness=10, num_threads=8).iterate(img), SlicHip(num_components=1600, compactness=10).iterate(img)))
l1, l2 = LSC(num_components=400, compactness=10, num_threads=1).iterate(img), LSCHip(num_components=400, compactness=10).iterate(img)
sys.path.insert(2, %(root)r + "/tests")
from util import best_overlap, boundary_iou                       # labels are renumbered by the connectivity pass: compare partitions
out["lsc_best_overlap"] = min(best_overlap(l1, l2), best_overlap(l2, l1))
out["lsc_boundary_iou"] = boundary_iou(l1, l2)
try:
    SlicHip(num_components=10).iterate(np.zeros((4, 4, 4), np.uint8))
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
