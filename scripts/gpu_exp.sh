#!/bin/bash
# A/B of experiment builds (make -C fast_slic_amd/csrc VAR=<name> DEFS=-DFSLIC_EXP=<bits>): per build a parity subset against the
# oracle, kernel durations by rocprofv3 --kernel-trace (8 x 4K and 16 x 720p groups, single stream) and the pipelined rate.
#   EXP_LIBS="libfslic_hip.so libfslic_hip_var_e1.so ..."   EXP_PARITY="<libs that run the parity subset>"  EXP_4K="<libs that run the 4K pipelined rate>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/exp
mkdir -p $OUT
cd $R
LIBS=${EXP_LIBS:-libfslic_hip.so}
for lib in $LIBS; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  tag=${lib%.so}; tag=${tag#libfslic_hip}; tag=${tag:-_product}
  echo "######## $lib"
  if [[ " ${EXP_PARITY:-$LIBS} " == *" $lib "* ]]; then
    timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -x -q -m gpu \
      -k "${EXP_TESTS:-baseline_sizes or group_of_eight_frames or one_group_of_eight_4k or stale_pixels}" \
      > $OUT/pytest$tag.log 2>&1
    echo "pytest rc=$?"; tail -3 $OUT/pytest$tag.log
  fi
  KSTATS_CFGS="${EXP_CFGS:-2160 3840 1600 4 slic 8;720 1280 1600 4 slic 16}" bash scripts/gpu_kstats.sh > $OUT/kstats$tag.txt 2>&1
  grep -E "^==|k_assign|k_cca_local|k_bin" $OUT/kstats$tag.txt | cut -c1-150
  cd $R
  timeout 300 python bench.py --timed-only --steps ${EXP_STEPS:-400} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720p pipelined', d['value'], 'MP/s', d['ms_per_step'], 'ms/step')" | tee $OUT/bench$tag.txt
  if [ "${EXP_NOBATCH:-0}" = "1" ]; then
    timeout 300 python bench.py --timed-only --no-batching --steps ${EXP_STEPS:-400} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720p pipelined, one submission per group', d['value'], 'MP/s')" | tee -a $OUT/bench$tag.txt
    for rep in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command line (20 steps)', d['value'], 'MP/s')" | tee -a $OUT/bench$tag.txt; done
  fi
  if [[ " ${EXP_4K:-} " == *" $lib "* ]]; then
    timeout 300 python bench.py --workload 4k --timed-only --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K pipelined', d['value'], 'MP/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT/bench$tag.txt
  fi
done
