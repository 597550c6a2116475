"""Throughput of a variant (lsc = BASELINE.json configs[4]; standard / l2 = the RealDist family) at 1280x720, K=1600 on
one MI355X, next to the unmodified reference on this box's host cores (LSCAvx2 + OpenMP for lsc; the scalar contexts +
OpenMP for the others, which have no SIMD specialisation).   python scripts/variant_throughput.py [lsc|standard|l2|noq] [steps] [slots]
Same pipeline shape as bench.py: device-resident frames, groups of 8 frames, groups collected in completion order."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant

H, W, K, G = 720, 1280, 1600, 8
kind = sys.argv[1] if len(sys.argv) > 1 else "lsc"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nslots = int(sys.argv[3]) if len(sys.argv) > 3 else 4
eng = Engine(0, nslots)
params = make_params(10, 10.0, 0.25, 3, variant={"lsc": B.VARIANT_LSC, "standard": B.VARIANT_REALDIST, "l2": B.VARIANT_REALDIST_L2, "noq": B.VARIANT_REALDIST_NOQ}[kind])
frames = [variant("A", H, W, seed=s) for s in range(G)]
d_rgb = [torch.from_numpy(f).cuda() for f in frames]
cl0 = np.zeros((G, K), CLUSTER_DTYPE)
for i, f in enumerate(frames):
    B.load_library().fslic_hip_initialize_clusters(H, W, K, f.ctypes.data, cl0[i].ctypes.data)
d_lab = [[torch.empty((H, W), dtype=torch.int16, device="cuda") for _ in frames] for _ in range(nslots)]
cls = [np.zeros((G, K), CLUSTER_DTYPE) for _ in range(nslots)]
a_rgb = eng.pointer_array([t.data_ptr() for t in d_rgb])
a_lab = [eng.pointer_array([t.data_ptr() for t in d_lab[s]]) for s in range(nslots)]
a_cls = [eng.pointer_array([cls[s][i].ctypes.data for i in range(G)]) for s in range(nslots)]
pending = [False] * nslots
torch.cuda.synchronize()


def free_slot():
    for s in range(nslots):
        if not pending[s]:
            return s
    while True:
        for s in range(nslots):
            if eng.group_done(s):
                eng.wait_group(s); pending[s] = False
                return s


def step():
    s = free_slot()
    np.copyto(cls[s].view(np.uint8).reshape(-1), cl0.view(np.uint8).reshape(-1))
    eng.submit_group(s, a_rgb, a_cls[s], a_lab[s], G, H, W, K, params)
    pending[s] = True


def drain():
    for s in range(nslots):
        if pending[s]:
            eng.wait_group(s); pending[s] = False


for _ in range(3 * nslots):
    step()
drain()
t0 = time.perf_counter()
for _ in range(steps):
    step()
drain()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = {"workload": "%s 1280x720 K=1600, 10 iters, %d frames per group, %d slots" % (kind, G, nslots),
       "megapixels_per_sec": round(H * W * G * steps / dt / 1e6, 1), "ms_per_group": round(dt / steps * 1e3, 4)}
try:
    from oracle import ref
    if ref.available():
        img = frames[0]
        c0 = ref.initialize_clusters(img, K)
        best = None
        if kind == "lsc":
            run_ref = lambda nt: ref.lsc_iterate(img, c0, num_threads=nt, arch="x64/avx2")
        else:
            run_ref = lambda nt: ref.realdist_iterate(img, c0, kind, num_threads=nt)
        for nt in (1, 8, 16, 32):
            run_ref(nt)
            ts = []
            for _ in range(3):
                t = time.perf_counter(); run_ref(nt); ts.append(time.perf_counter() - t)
            ms = min(ts) * 1e3
            if best is None or ms < best[1]:
                best = (nt, ms)
        out["cpu_reference"] = {"implementation": "LSCAvx2 + OpenMP" if kind == "lsc" else "ContextRealDist%s + OpenMP" % {"l2": "L2", "noq": "NoQ"}.get(kind, ""), "threads": best[0], "ms_per_frame": round(best[1], 2), "megapixels_per_sec": round(H * W / best[1] / 1e3, 1)}
except Exception as e:      # the reference build is test infrastructure; absent on some boxes
    out["cpu_reference"] = "unavailable: %s" % e
print(json.dumps(out))
eng.close()
