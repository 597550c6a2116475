#!/bin/bash
# Round 4: connectivity subset, then kernel durations of the connectivity kernels against libfslic_hip_var_old.so.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_graph_utils.py -x -q -m gpu \
  -k "connectivity or baseline_sizes or group_of_eight_frames or one_group_of_eight_4k or ties or small_golden or adversarial or poisoned or host" > gpurun_out/exp/pytest_cca.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/exp/pytest_cca.log
[ $rc -ne 0 ] && exit 1
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 3 slic 8;720 1280 1600 3 slic 8;720 1280 1600 3 slic 1" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_cca" | cut -c1-100
  cd $R
done
