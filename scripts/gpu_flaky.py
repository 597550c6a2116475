"""Repeat one configuration many times and report label mismatches against the oracle (race hunting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_slic_amd import Engine, make_params
from fast_slic_amd.synth import variant
from oracle import oracle as orc
H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
img = variant("A", H, W)
cl0 = orc.initialize_clusters(img, K)
o_labels, o_cl, o_lab, o_pre = orc.slic_iterate(img, cl0, stages=True)
e = Engine(0, 1)
p = make_params(10, 10.0, 0.25, 3)
bad = 0
for i in range(n):
    cl = cl0.copy()
    labels = e.iterate(img, cl, p)
    pre = e.last_prelabels(H, W)
    d = int((labels != o_labels).sum()); dp = int((pre != o_pre).sum())
    if d or dp:
        bad += 1
        ys, xs = np.nonzero(labels != o_labels)
        print("run %d: %d label mismatches, %d prelabel mismatches; first at (%d,%d) got %d exp %d; distinct got=%s" % (
            i, d, dp, ys[0] if d else -1, xs[0] if d else -1, labels[ys[0], xs[0]] if d else -1, o_labels[ys[0], xs[0]] if d else -1,
            np.unique(labels[labels != o_labels])[:8] if d else []))
print("bad runs: %d of %d" % (bad, n))
