#!/bin/bash
# End-of-round refresh after the LSC work: the whole -m gpu suite, smoke(), both bench lines, and the LSC profiles (kernel
# statistics / timeline of one 8-frame group, instruction mix, phase stamps).  Outputs under gpurun_out/r/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | cut -c1-200
t0=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmdline.json 2> $OUT/bench_driver_cmdline.err; echo "driver cmd rc=$? wall $(( $(date +%s) - t0 )) s"
t0=$(date +%s); timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default bench rc=$? wall $(( $(date +%s) - t0 )) s"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
FSLIC_GROUP=8 timeout 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python $R/scripts/frames.py 720 1280 1600 4 lsc 8 > /dev/null 2>&1
DB=$(find $OUT/p -name '*_results.db' | head -1)
python $R/scripts/rocpd_stats.py timeline $DB | cut -c1-170 > $OUT/timeline_1280x720_k1600_lsc_x8.txt
python $R/scripts/rocpd_stats.py stats $DB | cut -c1-170 > $OUT/kernel_stats_1280x720_k1600_lsc_x8.csv
rm -rf $OUT/p
cd $R
PMC_VAR=lsc bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_group.txt $OUT/lsc_pmc_instruction_mix.txt
[ -f fast_slic_amd/libfslic_hip_lab4.so ] && FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab4.so timeout 200 python scripts/lab_lsc_phases.py 720 1280 1600 8 0 4 2>&1 | grep -v amdgpu.ids > $OUT/lab_lsc_phases.txt
python - <<PY
import json
for fn in ("bench_driver_cmdline","bench"):
    d=json.load(open("$OUT/%s.json" % fn)); r=d["roofline"]
    print(fn, "value", d["value"], "parity", d["parity"], "frac", r["frac"], "4k", d["roofline_4k"]["frac"], d["roofline_4k"]["parity"], "c3", d["c3_k6000"]["megapixels_per_sec"], d["c3_k6000"]["parity"], "lsc", d["lsc_c5"]["megapixels_per_sec"], d["lsc_c5"]["parity"], d["lsc_c5"]["fused_assign_update"], "single", d["single_frame"]["device_ms"], "e2e", d["end_to_end"]["one_thread"], d["end_to_end"]["threads_4"])
PY
head -4 $OUT/lsc_pmc_instruction_mix.txt; grep -E "lsc" $OUT/kernel_stats_1280x720_k1600_lsc_x8.csv | cut -c1-110
