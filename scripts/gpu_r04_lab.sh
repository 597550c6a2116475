#!/bin/bash
# Round 4: LAB prepass with two quads per thread in flight: parity (whole RGB cube, baseline sizes, options), then the kernel's duration
# against libfslic_hip_var_old.so in single-stream groups.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lab or baseline_sizes or option or small_golden" 2>&1 | tail -2
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 4 slic 8;720 1280 1600 4 slic 8;720 1280 1600 4 slic 1" bash scripts/gpu_kstats.sh > /dev/null 2>&1
  cd $R
  for t in 3840x2160_k1600_slic_x8 1280x720_k1600_slic_x8 1280x720_k1600_slic_x1; do grep -m1 "k_rgb_to_lab" gpurun_out/timeline_$t.txt | cut -c1-62; done
  grep "k_rgb_to_lab" gpurun_out/kernel_stats_3840x2160_k1600_slic_x8.csv | sed 's/.*unsigned int)",//'
done
