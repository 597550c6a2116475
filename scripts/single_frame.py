"""One frame per group, device-resident, GPU to itself: device time by HIP events (bench.py's `single_frame`), for A/B runs.
   python scripts/single_frame.py [H W K] ...   (default: 720 1280 1600 and 2160 3840 1600); honours FSLIC_LIB"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from fast_slic_amd import Engine
args = [int(a) for a in sys.argv[1:]] or [720, 1280, 1600, 2160, 3840, 1600]
eng = Engine(0, 1)
for i in range(0, len(args), 3):
    H, W, K = args[i:i + 3]
    wl = bench.Workload(eng, H, W, K, [0])
    r = bench.single_frame_latency(eng, wl)
    print("%dx%d K=%d: device %.4f ms (min %.4f), %s" % (W, H, K, r["device_ms"], r["device_ms_min"], r["launch"]))
eng.close()
