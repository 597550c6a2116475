#!/bin/bash
# round 2, first box visit: barrier microbenchmark + slot-count sweep of the unchanged round-1 library
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 120 scripts/microbench/frame_barrier.out > $OUT/frame_barrier.txt 2>&1; echo "barrier rc=$?"
cat $OUT/frame_barrier.txt
for s in 4 6 8 12; do
  timeout 300 python bench.py --steps 100 --warmup 24 --slots $s --no-4k --no-cpu-baseline 2> $OUT/bench_s$s.err > $OUT/bench_s$s.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_s$s.json"))
print("slots $s: value", d["value"], "ms_per_step", d["ms_per_step"], "dev ms/frame", d["device_ms_per_frame"])
PY
done
FSLIC_HOST_TIMING=1 timeout 300 python bench.py --steps 40 --warmup 12 --slots 6 --no-4k --no-cpu-baseline 2> $OUT/bench_host_timing.err > $OUT/bench_host_timing.json
python scripts/host_timing_summary.py < $OUT/bench_host_timing.err | tail -30; python scripts/host_timeline.py < $OUT/bench_host_timing.err | tail -30
