"""Bring-up aid: per-phase cycle stamps of the full-assign launch (FSLIC_DEBUG_TIMING=1)."""
import os, sys
os.environ.setdefault("FSLIC_DEBUG_TIMING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
e = Engine(0, 1)
for (H, W, K) in [(720, 1280, 1600), (2160, 3840, 1600), (2160, 3840, 6000)]:
    img = variant("A", H, W)
    cl = np.zeros(K, CLUSTER_DTYPE)
    B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
    for i in range(2):
        e.iterate(img, cl.copy(), make_params(10, 10.0, 0.25, 3))
    print(H, W, K, e.last_device_times(), e.last_timing_report())
