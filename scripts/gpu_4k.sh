#!/bin/bash
# 4K K=1600 single-frame timing under a few knobs: prints full-assign / fused-assign event times
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cfg in ${CFGS:-"FSLIC_NOP=1" "FSLIC_ASSIGN_R=8" "FSLIC_NO_VTAB=1"}; do
  echo "== $cfg"
  env $cfg python - <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np, torch
from fast_slic_amd import Engine, make_params, CLUSTER_DTYPE, _binding as B
from fast_slic_amd.synth import variant
for (H, W, K) in [(2160, 3840, 1600), (2160, 3840, 6000)]:
    e = Engine(0, 1)
    e.set_launch_timing(True)
    img = variant("A", H, W); d = torch.from_numpy(img).cuda(); l = torch.empty((H, W), dtype=torch.int16, device="cuda")
    cl = np.zeros(K, CLUSTER_DTYPE); B.load_library().fslic_hip_initialize_clusters(H, W, K, img.ctypes.data, cl.ctypes.data)
    torch.cuda.synchronize()
    fa, lm, tt = [], [], []
    for i in range(6):
        e.iterate_device(d.data_ptr(), l.data_ptr(), H, W, cl.copy(), make_params(10, 10.0, 0.25, 3))
        if i >= 2:
            t, f = e.last_device_times(0); fa.append(f); tt.append(t); lm.append(e.last_assign_loop(0)[0] / 10)
    print("K=%d full_assign %.1f us  fused %.1f us  frame %.1f us" % (K, np.mean(fa) * 1e3, np.mean(lm) * 1e3, np.mean(tt) * 1e3))
    e.close()
PY
done
