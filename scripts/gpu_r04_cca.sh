#!/bin/bash
# Round 4, tile connectivity kernel rewritten on wavefront masks: parity first, then A/B against the previous build
# (libfslic_hip_var_old.so), then the group's instruction mix.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_graph_utils.py -x -q -m gpu \
  -k "connectivity or baseline_sizes or group_of_eight_frames or one_group_of_eight_4k or ties or small_golden or adversarial or poisoned" > gpurun_out/exp/pytest_cca.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/exp/pytest_cca.log
[ $rc -ne 0 ] && exit 1
EXP_LIBS="libfslic_hip.so libfslic_hip_var_old.so" EXP_PARITY="none" EXP_4K="libfslic_hip.so libfslic_hip_var_old.so" bash scripts/gpu_exp.sh
unset FSLIC_LIB
bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp gpurun_out/pmc_group.txt gpurun_out/exp/pmc_group_720p_x8.txt; cat gpurun_out/pmc_group.txt | cut -c1-140
