#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for r in 1 0; do
  echo "#### FSLIC_R16=$r"
  FSLIC_R16=$r KSTATS_CFGS="2160 3840 1600 4 slic 8;2160 3840 1600 4 slic 1;720 1280 1600 6 slic 8" bash scripts/gpu_kstats.sh 2>&1 | grep -E "^==|assign"
done
