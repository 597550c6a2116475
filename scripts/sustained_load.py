"""The bench's timed loop kept running for a number of seconds over a ring of buffer sets (clock / power probes).
   python scripts/sustained_load.py [seconds] [slots]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from fast_slic_amd import Engine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 6
eng = Engine(0, slots)
w = bench.Workload(eng, 720, 1280, 1600, list(range(8)))
RING = 64
w.buffers(RING)
for j in range(12): w.submit(j)
eng.pipeline_drain()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for j in range(RING): w.submit(j)
    n += RING
eng.pipeline_drain()
el = time.perf_counter() - t0
print("%.1f s, %d groups, %.1f MP/s" % (el, n, 720 * 1280 * 8 * n / el / 1e6), flush=True)
eng.close()
