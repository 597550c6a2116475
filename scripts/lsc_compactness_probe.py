"""LSC at large compactness against the live reference (oracle/_ref, test infrastructure): pre-connectivity agreement, best overlap,
boundary IoU, largest centre shift, by compactness and iteration count; the reference's own AVX2 arch beside it.
   python scripts/lsc_compactness_probe.py [max_iter ...]      honours FSLIC_LIB"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import ref
from fast_slic_amd import Engine, make_params, _binding as B
from fast_slic_amd.synth import variant
from util import best_overlap, boundary_iou
iters = [int(a) for a in sys.argv[1:]] or [10]
e = Engine(0, 1)
for mi in iters:
    for compactness in (10.0, 300.0, 500.0, 700.0, 1000.0):
        for (H, W, K, seed) in [(240, 320, 100, 3), (200, 264, 48, 4)]:
            img = variant("A", H, W, seed=seed)
            cl0 = ref.initialize_clusters(img, K)
            r_labels, r_cl, r_pre = ref.lsc_iterate(img, cl0, num_threads=1, arch="standard", compactness=compactness, max_iter=mi)
            a_labels, a_cl, a_pre = ref.lsc_iterate(img, cl0, num_threads=1, arch="x64/avx2", compactness=compactness, max_iter=mi)
            cl = cl0.copy().view(B.CLUSTER_DTYPE)
            labels = e.iterate(np.ascontiguousarray(img), cl, make_params(mi, compactness, 0.25, 3, variant=B.VARIANT_LSC))
            pre = e.last_prelabels(H, W)
            d = float(np.abs(np.stack([cl["y"] - r_cl["y"], cl["x"] - r_cl["x"]])).max())
            print("iters %2d c=%6.0f %dx%d K=%3d  hip vs ref: pre %.4f overlap %.4f iou %.3f maxshift %.1f | ref avx2 vs ref: pre %.4f overlap %.4f" % (
                mi, compactness, W, H, K, (pre == r_pre).mean(), min(best_overlap(labels, r_labels), best_overlap(r_labels, labels)), boundary_iou(labels, r_labels), d,
                (a_pre == r_pre).mean(), min(best_overlap(a_labels, r_labels), best_overlap(r_labels, a_labels))))
e.close()
