#!/bin/bash
# Round-4 validation trip: the -m gpu suite, an experiment build A/B, the driver's bench line, eight ranks on one GPU, phase stamps at 4K
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
if [ -n "$TRIP_EXP" ]; then
  EXP_LIBS="$TRIP_EXP libfslic_hip.so" EXP_PARITY="$TRIP_EXP" EXP_4K="$TRIP_EXP libfslic_hip.so" EXP_TESTS="${TRIP_EXP_TESTS:-baseline_sizes or connectivity or one_group_of_eight_4k or area_ties}" bash scripts/gpu_exp.sh
fi
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_driver.json"))
print("value", d["value"], "parity", d.get("parity"), "single", d.get("single_frame"), "e2e", {k: v for k, v in d.get("end_to_end", {}).items() if k != "through_reference_binding"})
r = d.get("roofline", {}); print("roofline frac", r.get("frac"), "avg_launch_ms", r.get("avg_launch_ms"), "full", r.get("full_assign", {}).get("frac"))
o = d.get("roofline_4k", {}); print("4k full", o.get("frac"), o.get("avg_launch_ms"), "fused", o.get("fused_assign_update", {}).get("frac"), "all", o.get("all_assign_launches", {}).get("frac"), "parity", o.get("parity"), "pipelined", o.get("megapixels_per_sec_pipelined"), "single", o.get("single_frame"))
for k in ("c3_k6000", "lsc_c5"):
    o = d.get(k, {}); print(k, "parity", o.get("parity"), "mp/s", o.get("megapixels_per_sec"))
PY
timeout 300 python bench.py --steps 200 --timed-only > $OUT/bench_1rank_200.json 2>/dev/null; cat $OUT/bench_1rank_200.json | cut -c1-400
timeout 600 python bench.py --gpus 8 --share-device --steps 200 --timed-only > $OUT/bench_8ranks_one_gpu.json 2> $OUT/bench_8ranks_one_gpu.err; echo "8 ranks rc=$?"; cat $OUT/bench_8ranks_one_gpu.json; tail -3 $OUT/bench_8ranks_one_gpu.err
nproc; lscpu | grep -E "Model name|NUMA node|Socket" | head -6
export FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab4.so
: > $OUT/lab4_assign_phases.txt
for cfg in "2160 3840 1600 8" "2160 3840 1600 1" "720 1280 1600 16" "720 1280 1600 1"; do
  echo "== $cfg" | tee -a $OUT/lab4_assign_phases.txt
  timeout 300 python scripts/lab_assign_phases.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lab4_assign_phases.txt
done
