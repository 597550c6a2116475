#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
for fb in 1 0; do
  export FSLIC_FUSEBIN=$fb
  echo "#### FSLIC_FUSEBIN=$fb"
  KSTATS_CFGS="720 1280 1600 6 slic 1;720 1280 1600 6 slic 8;2160 3840 1600 4 slic 8" bash scripts/gpu_kstats.sh 2>&1 | grep -E "^==|assign|bin_clusters"
  for t in 1280x720_k1600_slic_x1 1280x720_k1600_slic_x8 3840x2160_k1600_slic_x8; do echo "$t total: $(tail -1 $OUT/timeline_$t.txt | cut -c1-24)"; cp $OUT/timeline_$t.txt $OUT/timeline_${t}_fb$fb.txt; done
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-4k --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('driver cmdline value', d['value'], 'parity', d['parity'], 'roofline frac', r['frac'], r['avg_launch_ms'], 'dev ms/frame', d['device_ms_per_frame'])"
  timeout 600 python bench.py --gpus 1 --steps 600 --warmup 24 --no-4k --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('600 steps value', d['value'])"
done
