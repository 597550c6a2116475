#!/bin/bash
# one trip: the whole -m gpu suite, the LSC fixture report, a bench line (driver's command line + a longer one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
timeout 300 python -m pytest tests/test_gpu_lsc.py -m gpu -q -s -k agrees_with_reference 2>&1 | grep -E "pre-agree|passed|failed" | tee $OUT/lsc_report.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
tail -3 $OUT/bench_driver.err; python - <<PY
import json
d = json.load(open("$OUT/bench_driver.json"))
def show(k, v, ind=0):
    if isinstance(v, dict):
        print(" " * ind + k + ":")
        for a, b in v.items(): show(a, b, ind + 2)
    else:
        print(" " * ind + "%s: %s" % (k, str(v)[:160]))
for k in ("value", "ms_per_step", "parity", "device_ms_per_frame", "single_frame", "end_to_end"): show(k, d.get(k))
r = d.get("roofline", {}); print("roofline frac", r.get("frac"), "avg_launch_ms", r.get("avg_launch_ms"), "full", r.get("full_assign", {}).get("frac"))
for k in ("roofline_4k", "c3_k6000", "lsc_c5"):
    o = d.get(k, {}); print(k, "frac", o.get("frac"), "parity", o.get("parity"), o.get("parity_failures"), "mp/s", o.get("megapixels_per_sec", o.get("megapixels_per_sec_pipelined")), "fused", o.get("fused_assign_update", {}).get("frac"), "single", o.get("single_frame"))
PY
