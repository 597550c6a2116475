#!/bin/bash
# Round 4: parity of the assign-related tests on the product build, A/B against libfslic_hip_var_old.so (kernel durations; pipelined rates,
# two rounds), instruction mix of the fused launch at 8 x 4K.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu -k "not lsc and not realdist and not preemptive" > gpurun_out/exp/pytest_ab2.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/exp/pytest_ab2.log
[ $rc -ne 0 ] && exit 1
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 4 slic 8;720 1280 1600 4 slic 8" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | cut -c1-120
  cd $R
  FSLIC_GROUP=16 KSTATS_CFGS="720 1280 1600 4 slic 16" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | cut -c1-120
  cd $R
done
for rep in 1 2; do for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  python bench.py --timed-only --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '720p pipelined', d['value'])"
  python bench.py --workload 4k --timed-only --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '4K pipelined', d['value'])"
done; done
unset FSLIC_LIB
PMC_H=2160 PMC_W=3840 bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "kernel  |k_assign|group total" | cut -c1-140
