import os, sys, json, time, gc
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np, torch
import bench
from fast_slic_amd import Engine
seeds = [int(x) for x in sys.argv[1].split(",")]
eng = Engine(0, 6)
eng.pipeline_batching(16)
bench.Workload.merged_max = 2
wl = bench.Workload(eng, 720, 1280, 1600, seeds)
for steps in (20, 600):
    v, tot = wl.throughput(steps)
    print("seeds", seeds, "steps", steps, "MP/s", v, "device ms/frame", round(tot["device_ms"] / tot["frames"], 4))
