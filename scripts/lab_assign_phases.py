"""Lab build 4 (FSLIC_LIB=fast_slic_amd/libfslic_hip_lab4.so): per-phase time of the block assign kernel (k_assign_blk*),
from the 100 MHz time stamps the kernel leaves per (pass, block).  usage: lab_assign_phases.py H W K G [pass ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
H, W, K, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
passes = [int(a) for a in sys.argv[5:]] or [5, 10]
e = Engine(0, 1)
imgs = [variant("A", H, W, seed=g) for g in range(G)]
d = [torch.from_numpy(im).cuda() for im in imgs]
l = [torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in imgs]
lib = B.load_library()
cl0 = []
for im in imgs:
    cl = np.zeros(K, CLUSTER_DTYPE); lib.fslic_hip_initialize_clusters(H, W, K, im.ctypes.data, cl.ctypes.data); cl0.append(cl)
torch.cuda.synchronize()
p = make_params(10, 10.0, 0.25, 3)
for i in range(3):
    e.iterate_batch([t.data_ptr() for t in d], [c.copy() for c in cl0], [t.data_ptr() for t in l], H, W, p, True)
lib.fslic_hip_debug_cand_area_head.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
names = ["records kept (wave 0)", "first barrier", "ranks known", "argmin done", "labels stored", "sums flushed", "barrier before the flush", "end"]
S = int(np.sqrt(H * W // K))
for ps in passes:
    fused = ps < 10
    Hv = (H - (ps % 3) + 2) // 3 if fused else H
    # rows per wavefront as launch_assign picks them
    tiles_x = (W + 63) // 64
    blocks8 = G * tiles_x * ((Hv + 31) // 32)
    vmode = S > 40          # tables.cpp: row-vector table when the 2-D table exceeds 12 KB (S = 72 yes, S = 24 / 37 no)
    R = 16 if ((not fused and blocks8 > 3072) or (fused and vmode and blocks8 > 3072)) else 8
    if not fused and vmode and blocks8 > 4 * 6144:
        R = 32                      # launch_assign: 32 rows per wavefront in the full pass of big launches
    nby = (Hv + 4 * R - 1) // (4 * R)
    nblk = tiles_x * nby
    allw = []
    for f in range(G):
        w = np.zeros(11 * 2048 * 16, np.uint32)
        lib.fslic_hip_debug_cand_area_head(e._h, 0, f, w.size, w.ctypes.data)
        allw.append(w.reshape(11, 2048, 16)[ps, :nblk])
    w = np.concatenate(allw, 0).astype(np.int64)
    t0 = w[:, 0]; t0 = (t0 - t0.min()) & 0xFFFFFFFF
    st = w[:, 1:9] / 100.0
    life = st[:, 7]
    start = t0 / 100.0
    end = start + life
    print("pass %d (%s, R=%d): %d blocks (%d per frame x %d frames); kernel span by stamps %.2f us" % (ps, "fused" if fused else "full", R, w.shape[0], nblk, G, end.max()))
    print("  block starts (us after the first): p10 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f" % tuple(np.percentile(start, [10, 50, 75, 90, 100])))
    print("  block life: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us" % ((life.mean(),) + tuple(np.percentile(life, [10, 50, 90, 100]))))
    alive = [(int((start <= t).sum() - (end <= t).sum())) for t in (0.5, 1, 2, 4, 6, 8, 10, 12, 15, 20, 30, 40, 50, 60)]
    print("  blocks alive at t=0.5,1,2,4,6,8,10,12,15,20,30,40,50,60 us:", alive)
    print("  stamps since block start (mean us): " + "  ".join("%s %.2f" % (names[i], st[:, i].mean()) for i in range(8)))
    print("  candidates of the block %.1f, of wavefront 0 %.1f" % (w[:, 10].mean(), w[:, 9].mean()))
    # early blocks (first round) vs late
    first = start < 1.0
    if first.any() and (~first).any():
        for nm, m in (("blocks started in the first us", first), ("blocks started later", ~first)):
            print("  %s (%d): " % (nm, m.sum()) + "  ".join("%.2f" % st[m, i].mean() for i in range(8)))
