#!/bin/bash
# round 2: does the number of hardware queues the runtime multiplexes the streams onto bound the pipeline?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for q in 4 8 16; do
  for s in 6 8 12; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 100 --warmup 24 --slots $s --no-4k --no-cpu-baseline 2> $OUT/bench_q${q}_s$s.err > $OUT/bench_q${q}_s$s.json
    python - <<PY
import json
d = json.load(open("$OUT/bench_q${q}_s$s.json"))
print("hw queues $q slots $s: value", d["value"], "ms_per_step", d["ms_per_step"], "dev ms/frame", d["device_ms_per_frame"])
PY
  done
done
