"""Reproduce one variant case: python scripts/repro_variant.py H W K kind max_iter stride lab manhattan compactness"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_slic_amd import Engine, make_params, _binding as B
from fast_slic_amd.synth import variant
H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = {"slic": 0, "lsc": 1, "standard": 2, "l2": 3, "noq": 4}[sys.argv[4]]
mi, st, lab, man, comp = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), float(sys.argv[9])
img = np.ascontiguousarray(variant("B", H, W, seed=1000))
cl = np.zeros(K, B.CLUSTER_DTYPE)
B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
if os.environ.get("REPRO_REF"):
    from oracle import ref
    names = {0: None, 2: "standard", 3: "l2", 4: "noq"}
    c0 = ref.initialize_clusters(img, K)
    r = ref.realdist_iterate(img, c0, names[kind], num_threads=int(os.environ.get("REPRO_REF")), max_iter=mi, subsample_stride=st, convert_to_lab=bool(lab), manhattan=bool(man), compactness=comp, min_size_factor=0.1)
    print("reference done", len(np.unique(r[0])), flush=True)
e = Engine(0, 4)
p = make_params(mi, comp, 0.1, st, bool(lab), bool(man), variant=kind)
print("calling", flush=True)
out = e.iterate(img, cl, p)
print("ok", out.shape, len(np.unique(out)), flush=True)
