#!/bin/bash
# Round-4 trip: the -m gpu suite, the 4-row fused pass (experiment build) on one-frame groups, N ranks on one GPU, instruction mix of a group
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT/exp
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -14 $OUT/pytest_gpu.log
for lib in libfslic_hip.so $TRIP_EXP; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  if [ "$lib" != "libfslic_hip.so" ]; then
    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "baseline_sizes or small_golden or ragged or warm or stale or option or randomized or poisoned" > $OUT/exp/pytest_$lib.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/exp/pytest_$lib.log
  fi
  timeout 200 python scripts/single_frame.py 2>/dev/null | tee $OUT/exp/single_frame_$lib.txt
done
unset FSLIC_LIB
for n in 2 4 8; do
  timeout 400 python bench.py --gpus $n --share-device --steps 100 --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ranks', d['n_gpus'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'cpu ms/step per rank', d['host_cpu_ms_per_step']['per_rank'], 'device ms/frame', d['device_ms_per_frame'])" | tee -a $OUT/ranks_one_gpu.txt
done
for q in 1 2; do
  GPU_MAX_HW_QUEUES=$q timeout 400 python bench.py --gpus 8 --share-device --steps 100 --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q ranks', d['n_gpus'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'cpu ms/step per rank', d['host_cpu_ms_per_step']['per_rank'], 'device ms/frame', d['device_ms_per_frame'])" | tee -a $OUT/ranks_one_gpu.txt
done
timeout 400 python bench.py --gpus 8 --share-device --slots 1 --no-batching --steps 100 --timed-only 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots=1 ranks', d['n_gpus'], 'value', d['value'], 'ms/step', d['ms_per_step'], 'cpu ms/step per rank', d['host_cpu_ms_per_step']['per_rank'], 'device ms/frame', d['device_ms_per_frame'])" | tee -a $OUT/ranks_one_gpu.txt
bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $OUT/pmc_group.txt $OUT/pmc_group_720p_x8.txt; head -16 $OUT/pmc_group.txt
