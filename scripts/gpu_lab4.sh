#!/bin/bash
# lab build 4: phase stamps of the block assign kernel at the bench's operating points
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab4.so
for cfg in "720 1280 1600 16" "720 1280 1600 8" "720 1280 1600 1" "2160 3840 1600 8" "2160 3840 1600 1" "2160 3840 6000 4"; do
  set -- $cfg
  echo "== $cfg" | tee -a $OUT/lab4_assign_phases.txt
  timeout 300 python scripts/lab_assign_phases.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lab4_assign_phases.txt
done
