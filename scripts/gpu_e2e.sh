#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
: > $OUT/e2e_probe.txt
for m in "1 3" "1 0" "1 7" "2 3" "0 0"; do
  set -- $m
  FSLIC_HOST_STAGING=$1 FSLIC_COPY_THREADS=$2 timeout 120 python scripts/e2e_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/e2e_probe.txt
done
for m in 1 2 0; do
  echo "== host timing, staging=$m" | tee -a $OUT/e2e_probe.txt
  FSLIC_HOST_STAGING=$m FSLIC_HOST_TIMING=1 timeout 60 python - 2>&1 <<PY | grep "iterate:\|group_begin\|group_finish" | tail -6 | tee -a $OUT/e2e_probe.txt
import sys; sys.path.insert(0, "$R")
from fast_slic_amd import Slic
from fast_slic_amd.synth import variant
img = variant("A", 720, 1280)
s = Slic(num_components=1600, compactness=10)
for i in range(8):
    s.slic_model.initialized = False; s.iterate(img)
PY
done
