#!/bin/bash
# Round-4 closing trip: the -m gpu suite, smoke, both bench lines, the profile set (kernel statistics and timelines of the BASELINE
# configurations, the 16-frame launch of the pipeline, stream overlap of the pipelined region)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -10 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmdline.json 2> $OUT/bench_driver.err; echo "bench (driver) rc=$?"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench (default) rc=$?"
python - <<PY
import json
for fn in ("bench_driver_cmdline.json", "bench.json"):
    d = json.load(open("$OUT/" + fn))
    print(fn, "value", d["value"], "parity", d.get("parity"), "single", d.get("single_frame", {}).get("device_ms"), "e2e", d.get("end_to_end", {}).get("one_thread"), d.get("end_to_end", {}).get("threads_4"))
    r = d.get("roofline", {}); print("  roofline frac", r.get("frac"), "avg_launch_ms", r.get("avg_launch_ms"), "full", r.get("full_assign", {}).get("frac"), "rocprof", r.get("by_rocprofv3_kernel_duration"))
    o = d.get("roofline_4k", {}); print("  4k full", o.get("frac"), o.get("avg_launch_ms"), "fused", o.get("fused_assign_update", {}).get("frac"), "all", o.get("all_assign_launches", {}).get("frac"), "parity", o.get("parity"), "pipelined", o.get("megapixels_per_sec_pipelined"), "single", o.get("single_frame", {}).get("device_ms"))
    for k in ("c3_k6000", "lsc_c5"):
        o = d.get(k, {}); print(" ", k, "parity", o.get("parity"), "mp/s", o.get("megapixels_per_sec"), o.get("measured"))
    print("  cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
KSTATS_CFGS="720 1280 1600 6 slic 8;720 1280 1600 6 slic 1;2160 3840 1600 4 slic 8;2160 3840 1600 4 slic 1;2160 3840 6000 4 slic 4;720 1280 1600 6 lsc 8" KSTATS_BENCH=1 bash scripts/gpu_kstats.sh > $OUT/kstats_all.txt 2>&1
FSLIC_GROUP=16 KSTATS_CFGS="720 1280 1600 6 slic 16" bash scripts/gpu_kstats.sh >> $OUT/kstats_all.txt 2>&1
grep -E "^==|k_assign|k_cca_local|k_lsc_assign|k_bin_clusters<1>" $OUT/kstats_all.txt | cut -c1-160
tail -25 $OUT/bench_stream_overlap.txt
