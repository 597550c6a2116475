#!/bin/bash
# end-of-round check: the whole -m gpu suite, smoke(), the driver's bench command (wall time), the default bench line, the driver's command under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -10 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | cut -c1-200
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench_driver_cmdline.json 2> $OUT/final_bench_driver_cmdline.err; echo "driver cmd rc=$? wall $(( $(date +%s) - t0 )) s"
t0=$(date +%s); timeout 900 python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err; echo "default bench rc=$? wall $(( $(date +%s) - t0 )) s"
bash scripts/gpu_driver_trace.sh 2>&1 | grep -E "window|depth"
python - <<PY
import json
for fn in ("final_bench_driver_cmdline","final_bench"):
    d=json.load(open("$OUT/%s.json" % fn)); r=d["roofline"]
    print(fn, "value", d["value"], "parity", d["parity"], "frac", r["frac"], "rocprof", (r.get("by_rocprofv3_kernel_duration") or {}).get("frac"), "traffic", r["traffic"], r["traffic_source"], "4k", d["roofline_4k"]["frac"], d["roofline_4k"]["parity"], "c3", d["c3_k6000"]["parity"], "lsc", d["lsc_c5"]["parity"], "single", d["single_frame"]["device_ms"], "e2e", d["end_to_end"]["one_thread"], d["end_to_end"]["threads_4"], d["end_to_end"].get("through_reference_binding"))
PY
