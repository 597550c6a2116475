"""Verbose stage-by-stage bring-up check (not a pytest file): prints where the HIP path first departs
from the oracle so that one GPU round trip yields a diagnosis.  python tests/gpu_stage_check.py"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc          # noqa: E402
from fast_slic_amd import Engine, make_params   # noqa: E402
from fast_slic_amd.synth import variant   # noqa: E402
from util import describe_mismatch, cluster_fields_equal   # noqa: E402


def step(name, fn):
    t = time.time()
    try:
        msg = fn()
        print("[%s] %s (%.2fs)" % ("ok" if not msg else "MISMATCH", name, time.time() - t))
        if msg:
            print(msg)
    except Exception:
        print("[EXC] %s" % name)
        traceback.print_exc()
    sys.stdout.flush()


def main():
    e = Engine(0, 2)

    def lab_small():
        img = variant("C", 97, 131)
        return describe_mismatch("lab", e.rgb_to_lab(img), orc.rgb_to_lab(img))
    step("lab 97x131", lab_small)

    def cca(case):
        rng = np.random.default_rng(5)
        if case == 0:
            lab = np.kron(rng.integers(0, 5, (7, 9)), np.ones((16, 16), int))[:100, :130].astype(np.uint16)
            K, thr = 20, 10
        elif case == 1:
            lab = rng.integers(0, 3, (150, 200)).astype(np.uint16); K, thr = 30, 3
        else:
            lab = np.zeros((70, 70), np.uint16); K, thr = 3, 5
        return describe_mismatch("cca%d" % case, e.enforce_connectivity(lab, K, thr), orc.enforce_connectivity(lab, K, thr))
    for c in range(3):
        step("cca case %d" % c, lambda c=c: cca(c))

    def pipeline(H, W, K, var, generic, **kw):
        img = variant(var, H, W)
        cl0 = orc.initialize_clusters(img, K)
        o_labels, o_cl, o_lab, o_pre = orc.slic_iterate(img, cl0, stages=True, **kw)
        cl = cl0.copy()
        p = make_params(kw.get("max_iter", 10), kw.get("compactness", 10.0), kw.get("min_size_factor", 0.25),
                        kw.get("subsample_stride", 3), force_generic=generic)
        labels = e.iterate(img, cl, p)
        pre = e.last_prelabels(H, W)
        msgs = [describe_mismatch("prelabels", pre, o_pre)] + cluster_fields_equal(cl, o_cl) + [describe_mismatch("labels", labels, o_labels)]
        msgs = [m for m in msgs if m]
        print("    path=%d device_ms=%s" % (e.last_path(), e.last_device_times()))
        return "\n".join(msgs)

    for generic in (True, False):
        for mi in (0, 1, 2, 10):
            step("pipeline 96x128 K=24 generic=%s max_iter=%d" % (generic, mi),
                 lambda g=generic, mi=mi: pipeline(96, 128, 24, "A", g, max_iter=mi))
    for generic in (True, False):
        step("pipeline 480x640 K=200 generic=%s" % generic, lambda g=generic: pipeline(480, 640, 200, "A", g))
    step("pipeline 720p K=1600", lambda: pipeline(720, 1280, 1600, "A", False))
    step("pipeline 4K K=1600", lambda: pipeline(2160, 3840, 1600, "A", False))
    step("pipeline 4K K=6000", lambda: pipeline(2160, 3840, 6000, "A", False))
    print(e.last_timing_report())


if __name__ == "__main__":
    main()
