"""Timing of the superpixel-graph utilities (graph.hip; src/fast-slic.cpp) on device-resident label maps, next to the unmodified
reference on the host (oracle/_ref, serial code).   python scripts/graph_utils_throughput.py [reps]   -> one JSON line per size"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
from oracle import ref

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = Engine(0, 1)
for H, W, K in ((720, 1280, 1600), (2160, 3840, 1600)):
    img = variant("A", H, W, seed=0)
    cl = np.zeros(K, CLUSTER_DTYPE)
    B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
    labels = e.iterate(np.ascontiguousarray(img), cl, make_params(10, 10.0, 0.25, 3))
    rng = np.random.default_rng(0)
    mask = rng.integers(0, 256, (H, W)).astype(np.uint8)
    d_lab = torch.from_numpy(labels.view(np.int16).copy()).cuda()
    d_mask = torch.from_numpy(mask).cuda()
    d_out = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dens = e.get_mask_density(cl, d_lab.data_ptr(), d_mask.data_ptr(), H, W)

    def timed(fn, n):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n * 1e3

    out = {"workload": "%dx%d K=%d, device-resident label map / mask / result" % (W, H, K), "reps": reps, "unit": "ms per call (wall, incl. the host part)",
           "get_connectivity": round(timed(lambda: e.get_connectivity(d_lab.data_ptr(), H, W, K), reps), 4),
           "get_mask_density": round(timed(lambda: e.get_mask_density(cl, d_lab.data_ptr(), d_mask.data_ptr(), H, W), reps), 4),
           "cluster_density_to_mask": round(timed(lambda: e.cluster_density_to_mask(d_lab.data_ptr(), dens, H, W, out=d_out.data_ptr()), reps), 4)}
    if ref.available():
        out["reference_host"] = {"get_connectivity": round(timed(lambda: ref.get_connectivity(labels, K), 3), 3),
                                 "get_mask_density": round(timed(lambda: ref.get_mask_density(cl, labels, mask), 3), 3),
                                 "cluster_density_to_mask": round(timed(lambda: ref.cluster_density_to_mask(cl, labels, dens), 3), 3)}
    print(json.dumps(out))
e.close()
