"""Lab build only (FSLIC_LIB=fast_slic_amd/libfslic_hip_lab.so): time stamps of k_cca_select per frame of one 8 x 720p group."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K, G = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (720, 1280, 1600, 8)      # python scripts/lab_select_stamps.py [H W K G]
e = Engine(0, 1)
imgs = [variant("A", H, W, seed=g) for g in range(G)]
d = [torch.from_numpy(im).cuda() for im in imgs]
l = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in imgs]
cl0 = []
for im in imgs:
    cl = np.zeros(K, CLUSTER_DTYPE); B.load_library().fslic_hip_initialize_clusters(H, W, K, im.ctypes.data, cl.ctypes.data); cl0.append(cl)
torch.cuda.synchronize()
p = make_params(10, 10.0, 0.25, 3)
lib = B.load_library()
lib.fslic_hip_debug_status_words.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
for i in range(4):
    e.iterate_batch([t.data_ptr() for t in d], [c.copy() for c in cl0], [t.data_ptr() for t in l], H, W, p, True)
names = ["start", "selected", "tie path in", "ranked+permuted", "make_heap", "replacements", "labels"]
for f in range(G):
    w = np.zeros(16, np.uint32)
    lib.fslic_hip_debug_status_words(e._h, 0, f, w.ctypes.data)
    st = w[8:15].astype(np.int64)
    print("frame %d roots %d M %d status %d:" % (f, w[4], w[5], w[7]), " ".join("%s +%.1fus" % (names[i], ((st[i] - st[0]) & 0xFFFFFFFF) / 100.0) for i in range(1, 7) if st[i]))
