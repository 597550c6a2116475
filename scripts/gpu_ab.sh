#!/bin/bash
# A/B of an environment switch on the bench: AB_VAR=FSLIC_GRAPH AB_VALS="0 1" bash scripts/gpu_ab.sh
# (AB_VAR=FSLIC_LIB with two library paths compares two builds of the same ABI on one box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
i=0
for v in ${AB_VALS:-0 1}; do
  i=$((i+1))
  for rep in 1 2 3; do
    env ${AB_VAR:-FSLIC_GRAPH}=$v timeout 300 python bench.py --no-cpu-baseline --no-4k ${BENCH_ARGS:-} 2> $OUT/ab_$i.err | python -c "
import sys,json
b=json.loads(sys.stdin.read()); print('${AB_VAR:-FSLIC_GRAPH}=$(basename $v)', 'value', b['value'], 'ms/step', b['ms_per_step'], 'device ms/frame', b.get('device_ms_per_frame'))"
  done
done
