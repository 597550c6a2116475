#!/bin/bash
# Round 4: kept-boundary form of the fused epilogue: parity of the assign-related tests, A/B against libfslic_hip_var_old.so (kernel durations,
# pipelined rates), instruction mix of the fused launch at 8 x 4K and 8 x 720p.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu -k "baseline_sizes or group_of_eight or one_group_of_eight_4k or stale_pixels or stride or sixteen_row or always_fused or small_golden or option or poisoned or bench_operating" > gpurun_out/exp/pytest_ab4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/exp/pytest_ab4.log
[ $rc -ne 0 ] && exit 1
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 4 slic 8;720 1280 1600 4 slic 8;720 1280 1600 4 slic 1" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
  FSLIC_GROUP=16 KSTATS_CFGS="720 1280 1600 4 slic 16" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
  python bench.py --timed-only --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720p pipelined', d['value'])"
  python bench.py --workload 4k --timed-only --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K pipelined', d['value'])"
done
unset FSLIC_LIB
PMC_H=2160 PMC_W=3840 bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "kernel  |k_assign" | cut -c1-140
bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "k_assign|group total" | cut -c1-140
