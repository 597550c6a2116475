"""Pipelined throughput of the bench workload with parts of the path switched off through the parameters (fewer
iterations, no connectivity pass): what each part costs when groups overlap.   python scripts/ablate_throughput.py [slots] [case]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from fast_slic_amd import Engine, make_params
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eng = Engine(0, slots)
H, W, K = 720, 1280, 1600
w = bench.Workload(eng, H, W, K, list(range(8)))
cases = [("full path (10 iterations, connectivity)", make_params(10, 10.0, 0.25, 3)),
         ("5 iterations", make_params(5, 10.0, 0.25, 3)),
         ("0 iterations", make_params(0, 10.0, 0.25, 3)),
         ("10 iterations, no connectivity pass", make_params(10, 10.0, 0.0, 3)),
         ("0 iterations, no connectivity pass", make_params(0, 10.0, 0.0, 3))]
if len(sys.argv) > 2:
    cases = [cases[int(sys.argv[2])]]
for name, p in cases:
    w.params = p
    w.sets = []
    runs = [w.throughput(120, warm=12) for _ in range(3)]
    best, tot = max(runs, key=lambda r: r[0])
    us = H * W * 8 / best          # us per group of 8 frames
    # device time of a group (first kernel's start to last kernel's end on its slot's stream) against slots x the time per
    # group: equal when every slot always has a group on the device, smaller when slots wait for the host
    lat = 1e3 * tot["device_ms"] / max(tot["groups"], 1)
    print("%-45s %9.1f MP/s   %7.1f us per group   device time of a group %7.1f us = %.2f x (slots x time per group)"
          % (name, best, us, lat, lat / (slots * us)), flush=True)
eng.close()
