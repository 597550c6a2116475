#!/bin/bash
# Round 4: connectivity subset, then A/B of the connectivity kernels against libfslic_hip_var_old.so (single-stream groups and pipelined rates).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_graph_utils.py -x -q -m gpu \
  -k "connectivity or baseline_sizes or group_of_eight_frames or one_group_of_eight_4k or ties or small_golden or adversarial or poisoned" > gpurun_out/exp/pytest_cca.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/exp/pytest_cca.log
[ $rc -ne 0 ] && exit 1
EXP_CFGS="2160 3840 1600 4 slic 8;720 1280 1600 4 slic 8;720 1280 1600 4 slic 1" EXP_LIBS="libfslic_hip.so libfslic_hip_var_old.so" EXP_PARITY="none" EXP_4K="libfslic_hip.so libfslic_hip_var_old.so" bash scripts/gpu_exp.sh 2>&1 | grep -E "^#|^==|pipelined"
for t in _product _var_old; do echo "-- $t"; grep -E "^==|k_cca" gpurun_out/exp/kstats$t.txt | cut -c1-100; done
