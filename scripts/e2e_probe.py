"""Where the time of Slic(K).iterate(numpy) goes: per call wall time through fast_slic_amd for 1 and 4 caller threads, and the host-side
split the library prints with FSLIC_HOST_TIMING=1.  Modes through the environment: FSLIC_HOST_STAGING=0|1|2, FSLIC_COPY_THREADS=n."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_slic_amd import Slic
from fast_slic_amd.synth import variant
H, W, K = 720, 1280, 1600
imgs = [variant("A", H, W, seed=s) for s in range(4)]
def run(nt, secs=1.5):
    counts = [0] * nt
    stop = time.perf_counter() + secs
    def work(t):
        s = Slic(num_components=K, compactness=10); s.iterate(imgs[t % 4]); i = 0
        while time.perf_counter() < stop:
            s.slic_model.initialized = False
            s.iterate(imgs[(t + i) % 4]); i += 1
        counts[t] = i
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    t0 = time.perf_counter(); [th.start() for th in ths]; [th.join() for th in ths]
    el = time.perf_counter() - t0
    return sum(counts) / el
tag = "staging=%s copy_threads=%s" % (os.environ.get("FSLIC_HOST_STAGING", "1"), os.environ.get("FSLIC_COPY_THREADS", "3"))
for nt in (1, 2, 4, 8):
    fps = run(nt)
    print("%s threads=%d: %.0f frames/s = %.0f MP/s, %.1f us per frame per thread" % (tag, nt, fps, fps * H * W / 1e6, nt / fps * 1e6))
