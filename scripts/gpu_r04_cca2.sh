#!/bin/bash
# Round 4: connectivity subset first, then A/B of the tile kernel against libfslic_hip_var_old.so, then the LDS microbenchmark.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_graph_utils.py -x -q -m gpu \
  -k "connectivity or baseline_sizes or group_of_eight_frames or one_group_of_eight_4k or ties or small_golden or adversarial or poisoned" > gpurun_out/exp/pytest_cca.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/exp/pytest_cca.log
[ $rc -ne 0 ] && exit 1
EXP_LIBS="libfslic_hip.so libfslic_hip_var_old.so" EXP_PARITY="none" EXP_4K="libfslic_hip.so libfslic_hip_var_old.so" bash scripts/gpu_exp.sh 2>&1 | grep -E "^#|^==|k_cca_local|pipelined"
unset FSLIC_LIB
PMC_G=8 bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "kernel  |k_cca_local|group total"
(cd scripts/microbench && hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomics lds_atomics.hip 2>/dev/null && /tmp/lds_atomics > $R/gpurun_out/lds_atomics.out; tail -14 $R/gpurun_out/lds_atomics.out)
