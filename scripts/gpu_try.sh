#!/bin/bash
# quick trip: parity tests of the integer path, then phase stamps (lab build 4) and per-kernel statistics of both forms of the block kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/try_pytest.log
rm -f $OUT/lab4_assign_phases.txt
for v in 2 1; do
  export FSLIC_ASSIGN=$v
  for cfg in "720 1280 1600 8" "720 1280 1600 1" "2160 3840 1600 8" "2160 3840 6000 4"; do
    echo "== FSLIC_ASSIGN=$v $cfg" | tee -a $OUT/lab4_assign_phases.txt
    FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab4.so timeout 300 python scripts/lab_assign_phases.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lab4_assign_phases.txt
  done
  KSTATS_CFGS="720 1280 1600 6 slic 16;720 1280 1600 6 slic 8;2160 3840 1600 4 slic 8;720 1280 1600 6 slic 1" bash scripts/gpu_kstats.sh > $OUT/kstats_v$v.txt 2>&1
  grep -E "^==|assign|bin_clusters" $OUT/kstats_v$v.txt
done
