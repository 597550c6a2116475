"""Profiling workload: n single-stream steps of G frames of H x W, K clusters, device-resident, through the C ABI.
   python scripts/frames.py H W K [n] [slic|lsc|pre] [G]    """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
var = B.VARIANT_LSC if len(sys.argv) > 5 and sys.argv[5] == "lsc" else B.VARIANT_SLIC
G = int(sys.argv[6]) if len(sys.argv) > 6 else 1
e = Engine(0, 1)
imgs = [variant("A", H, W, seed=g) for g in range(G)]
d = [torch.from_numpy(im).cuda() for im in imgs]
l = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in imgs]
cl0 = []
for im in imgs:
    cl = np.zeros(K, CLUSTER_DTYPE)
    B.load_library().fslic_hip_initialize_clusters(H, W, K, im.ctypes.data, cl.ctypes.data)
    cl0.append(cl)
torch.cuda.synchronize()
p = make_params(10, 10.0, 0.25, 3, variant=var, preemptive=len(sys.argv) > 5 and sys.argv[5] == "pre", preemptive_thres=0.05)      # "pre": Slic(preemptive=True)
for i in range(n):
    e.iterate_batch([t.data_ptr() for t in d], [c.copy() for c in cl0], [t.data_ptr() for t in l], H, W, p, True)
print(H, W, K, G, e.last_device_times(), e.last_timing_report())
