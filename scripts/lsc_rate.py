"""Pipelined rate of the LSC variant (1280x720 K=1600, 8 frames per GPU: BASELINE configs[4]) on the build FSLIC_LIB selects -- the
figure bench.py reports as lsc_c5.megapixels_per_sec, without the other legs.   python scripts/lsc_rate.py [steps] [reps]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from fast_slic_amd._binding import Engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
eng = Engine(0, 6)                     # bench.py's default slot count
bench.Workload.merged_max = 2          # as bench.py's default run: two submissions of 8 frames per launch group
eng.pipeline_batching(16)
w = bench.Workload(eng, 720, 1280, 1600, list(range(8)), variant_id=1)
for _ in range(reps):
    print("LSC 720p pipelined, %d steps: %.0f MP/s" % (steps, w.throughput(steps)[0]))
