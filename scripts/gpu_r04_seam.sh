#!/bin/bash
# Round 4: seam pixels per thread of k_cca_border (1, 2, 4 = product, 8) and the previous kernel (var_old): kernel durations of single-stream
# groups and pipelined rates, two rounds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
for lib in libfslic_hip_var_px1.so libfslic_hip_var_px2.so libfslic_hip.so libfslic_hip_var_px8.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "connectivity or baseline_sizes" 2>&1 | tail -1
  KSTATS_CFGS="2160 3840 1600 3 slic 8;720 1280 1600 3 slic 8;720 1280 1600 3 slic 1" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "^==|k_cca_border" | cut -c1-90
  cd $R
done
for rep in 1 2; do
for lib in libfslic_hip_var_px1.so libfslic_hip_var_px2.so libfslic_hip.so libfslic_hip_var_px8.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  python bench.py --timed-only --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '720p pipelined', d['value'], 'MP/s')"
done
done
