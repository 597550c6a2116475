"""Profiling workload: a few 3840x2160 K=1600 frames through the C ABI (device-resident), nothing else."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K = 2160, 3840, int(os.environ.get("FSLIC_K", "1600"))
n = int(os.environ.get("FSLIC_FRAMES", "4"))
e = Engine(0, 1)
img = variant("A", H, W)
d = torch.from_numpy(img).cuda()
l = torch.empty((H, W), dtype=torch.int16, device="cuda")
cl = np.zeros(K, CLUSTER_DTYPE)
B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
torch.cuda.synchronize()
for i in range(n):
    e.iterate_device(d.data_ptr(), l.data_ptr(), H, W, cl.copy(), make_params(10, 10.0, 0.25, 3))
print(e.last_device_times(), e.last_timing_report())
