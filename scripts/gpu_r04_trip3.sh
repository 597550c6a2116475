#!/bin/bash
# Round-4 trip: the -m gpu suite, slot-count sweep of the driver's command line, round-4 profile set (kernel statistics of the BASELINE configurations)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
for sl in 5 6 7 8; do
  for rep in 1 2 3; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --slots $sl --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots $sl rep $rep: value', d['value'])" | tee -a $OUT/slots_sweep.txt
  done
done
KSTATS_CFGS="720 1280 1600 6 slic 8;720 1280 1600 6 slic 1;2160 3840 1600 4 slic 8;2160 3840 1600 4 slic 1;2160 3840 6000 4 slic 4;720 1280 1600 6 lsc 8" bash scripts/gpu_kstats.sh > $OUT/kstats_all.txt 2>&1
grep -E "^==|k_assign|k_cca_local|k_lsc_assign" $OUT/kstats_all.txt | cut -c1-160
