#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
bash scripts/gpu_e2e.sh
KSTATS_CFGS="720 1280 1600 6 slic 8;2160 3840 6000 4 slic 4" bash scripts/gpu_kstats.sh 2>&1 | grep -E "^==|assign"
PMC_G=8 bash scripts/gpu_pmc_group.sh 2>&1 | tail -16
