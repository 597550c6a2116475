#!/bin/bash
# Round 4: whole GPU suite on the product build (LAB multiply-adds, division-free cluster pass, batched finaliser loads), then A/B against
# libfslic_hip_var_old.so: kernel durations of single-stream groups (1 and 8 x 720p, 8 x 4K) and one-frame device time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/exp/pytest_all.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/exp/pytest_all.log
[ $rc -ne 0 ] && exit 1
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="720 1280 1600 4 slic 1;720 1280 1600 4 slic 8;2160 3840 1600 3 slic 8" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_assign_bin|k_bin_clusters<1>|k_assign_blk2<8, true|k_assign_blk2_w6<16, true|rgb_to_lab" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-110
  cd $R
  python scripts/single_frame.py 720 1280 1600 2>/dev/null | tail -2
  python scripts/single_frame.py 2160 3840 1600 2>/dev/null | tail -1
done
