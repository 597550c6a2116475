#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
KSTATS_CFGS="720 1280 1600 6 slic 8;720 1280 1600 6 slic 1;2160 3840 1600 4 slic 8" bash scripts/gpu_kstats.sh 2>&1 | grep -E "^==|assign|bin_clusters"
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/p; timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o p -- python $R/scripts/frames.py 720 1280 1600 3 slic 8 > /dev/null 2>&1
  python $R/scripts/rocpd_pmc.py $(find $OUT/p -name '*_results.db' | head -1) assign; rm -rf $OUT/p
done
