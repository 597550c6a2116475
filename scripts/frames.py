"""Profiling workload: n single-stream frames of H x W, K clusters, device-resident, through the C ABI.
   python scripts/frames.py H W K [n]      (FSLIC_DEBUG_TIMING=1|2 prints per-phase cycle stamps of the assign kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
e = Engine(0, 1)
img = variant("A", H, W)
d = torch.from_numpy(img).cuda()
l = torch.empty((H, W), dtype=torch.int16, device="cuda")
cl = np.zeros(K, CLUSTER_DTYPE)
B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
torch.cuda.synchronize()
for i in range(n):
    e.iterate_device(d.data_ptr(), l.data_ptr(), H, W, cl.copy(), make_params(10, 10.0, 0.25, 3))
print(H, W, K, e.last_device_times(), e.last_timing_report())
