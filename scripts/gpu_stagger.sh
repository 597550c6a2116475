#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for st in 0 16 32 48 64 96; do
  echo "#### FSLIC_STAGGER=$st"
  FSLIC_GROUP=16 FSLIC_STAGGER=$st KSTATS_CFGS="720 1280 1600 6 slic 16;720 1280 1600 6 slic 8;2160 3840 1600 4 slic 8" bash scripts/gpu_kstats.sh 2>&1 | grep -E "^==|assign"
done
