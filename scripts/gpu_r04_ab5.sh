#!/bin/bash
# Round 4: 64-entry label table indexed by rank: parity subset, kernel durations against libfslic_hip_var_old.so, LDS counters of the fused launch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu -k "baseline_sizes or group_of_eight or one_group_of_eight_4k or stale_pixels or stride or sixteen_row or always_fused or small_golden or option" > gpurun_out/exp/pytest_ab5.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/exp/pytest_ab5.log
[ $rc -ne 0 ] && exit 1
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 4 slic 8;720 1280 1600 4 slic 8;720 1280 1600 4 slic 1" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
  FSLIC_GROUP=16 KSTATS_CFGS="720 1280 1600 4 slic 16" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
done
unset FSLIC_LIB
PMC_H=2160 PMC_W=3840 bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "kernel  |k_assign" | cut -c1-140
python - <<PY
import csv
for name in ('k_assign_blk2_w6<16, true', 'k_assign_blk2_w6<32, false'):
    rows=[r for r in csv.reader(open('gpurun_out/pmc_group.csv')) if len(r)>=4 and name in r[0]]
    d={r[1]:float(r[2]) for r in rows}; w=d.get('SQ_WAVES',1)
    print(name, "per wavefront: LDS instructions %.1f, index-active %.0f, bank-conflict %.0f, vector instructions %.0f" % (d['SQ_INSTS_LDS']/w, d['SQ_LDS_IDX_ACTIVE']/w, d['SQ_LDS_BANK_CONFLICT']/w, d['SQ_INSTS_VALU']/w))
PY
