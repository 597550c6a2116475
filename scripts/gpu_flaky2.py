"""Alternate between geometries on ONE engine slot (arena re-carved each time) and compare with the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_slic_amd import Engine, make_params
from fast_slic_amd.synth import variant
from oracle import oracle as orc
cfgs = [(2160, 3840, 6000), (2160, 3840, 1600), (720, 1280, 1600), (480, 640, 200)]
ref = {}
for (H, W, K) in cfgs:
    img = variant("A", H, W); cl0 = orc.initialize_clusters(img, K)
    ref[(H, W, K)] = (img, cl0) + tuple(orc.slic_iterate(img, cl0, stages=True))
e = Engine(0, 4)
p = make_params(10, 10.0, 0.25, 3)
bad = 0; tot = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for c in cfgs:
        img, cl0, o_labels, o_cl, o_lab, o_pre = ref[c]
        cl = cl0.copy()
        labels = e.iterate(img, cl, p)
        pre = e.last_prelabels(c[0], c[1])
        d = int((labels != o_labels).sum()); dp = int((pre != o_pre).sum()); dc = cl.tobytes() != o_cl.tobytes()
        tot += 1
        if d or dp or dc:
            bad += 1
            ys, xs = np.nonzero(labels != o_labels)
            print("rep %d cfg %s: labels %d prelabels %d clusters %s first (%s) got %s" % (rep, c, d, dp, dc,
                  (ys[0], xs[0]) if d else None, np.unique(labels[labels != o_labels])[:6] if d else None))
print("bad runs: %d of %d" % (bad, tot))
