#!/bin/bash
# Round 4, last trip: the -m gpu suite, smoke, and the driver's bench line on the final build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_last.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu_last.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmdline_last.json 2> $OUT/bench_driver_last.err; echo "bench (driver) rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_driver_cmdline_last.json")); r = d["roofline"]; k = d["roofline_4k"]
print("value", d["value"], "parity", d["parity"], "single", d["single_frame"]["device_ms"], "e2e", d["end_to_end"]["one_thread"], d["end_to_end"]["threads_4"])
print("roofline frac", r["frac"], r["avg_launch_ms"], "4k full", k["frac"], "fused", k["fused_assign_update"]["frac"], k["fused_assign_update"]["avg_launch_ms"], "all", k["all_assign_launches"]["frac"], "parity", k["parity"], "pipelined", k["megapixels_per_sec_pipelined"])
print("c3", d["c3_k6000"]["parity"], d["c3_k6000"]["megapixels_per_sec"], "lsc", d["lsc_c5"]["parity"], d["lsc_c5"]["megapixels_per_sec"])
PY
