"""Lab build 2 (FSLIC_LIB=fast_slic_amd/libfslic_hip_lab2.so): per-phase time of k_cca_local, averaged over the blocks of a frame."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
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
for i in range(3):
    e.iterate_batch([t.data_ptr() for t in d], [c.copy() for c in cl0], [t.data_ptr() for t in l], H, W, p, True)
nblk = ((W + 63) // 64) * ((H + 31) // 32)
lib.fslic_hip_debug_cand_tail.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
names = ["loads", "A rows", "B seams", "C flatten+areas", "D count", "D global atomic", "D stores issued"]
g0 = None
for f in range(G):
    w = np.zeros(8 * nblk, np.int32)
    lib.fslic_hip_debug_cand_tail(e._h, 0, f, 8 * nblk, w.ctypes.data)
    st = w.reshape(nblk, 8)[:, :7].astype(np.float64) / 100.0          # us since the block's start
    d = np.diff(np.concatenate([np.zeros((nblk, 1)), st], 1), axis=1)
    t0 = w.reshape(nblk, 8)[:, 7].astype(np.uint32).astype(np.int64)
    g0 = t0.min() if g0 is None else g0
    print("frame %d: first block starts %.1f us after the kernel's first, last block ends at %.1f us" % (f, (t0.min() - g0) / 100.0, ((t0 - g0) / 100.0 + w.reshape(nblk, 8)[:, 6] / 100.0).max()))
    rel = np.sort((t0 - t0.min()) / 100.0)
    alive = [(int(((t0 - t0.min()) / 100.0 <= t).sum() - ((t0 - t0.min()) / 100.0 + st[:, 6] <= t).sum())) for t in (1, 3, 6, 10, 15, 20, 30)]
    print("frame %d: block starts (us after the first): p10 %.1f p50 %.1f p90 %.1f max %.1f; blocks of this frame alive at t=1,3,6,10,15,20,30 us: %s" % (f, rel[nblk // 10], rel[nblk // 2], rel[nblk * 9 // 10], rel[-1], alive))
    print("frame %d (%d blocks): block life mean %.2f us (p10 %.2f, p90 %.2f); phases: " % (f, nblk, st[:, 6].mean(), np.percentile(st[:, 6], 10), np.percentile(st[:, 6], 90))
          + "  ".join("%s %.2f" % (names[i], d[:, i].mean()) for i in range(7)))
