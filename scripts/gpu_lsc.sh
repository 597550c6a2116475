#!/bin/bash
# LSC variant: tests, then single-stream kernel timelines (rocprofv3 --kernel-trace) at 720p K=1600, 1 and 8 frames per group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 600 python -m pytest tests/test_gpu_lsc.py -q -s -x > $OUT/pytest_lsc.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_lsc.log
  tail -20 $OUT/pytest_lsc.log
fi
cd /tmp && export TMPDIR=/tmp
for G in 1 8; do
  tag=lsc_720x1280_k1600_g$G
  rm -rf $OUT/prof_$tag
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$tag -o t -- python $R/scripts/frames.py 720 1280 1600 4 lsc $G > $OUT/prof_$tag.log 2>&1
  tail -1 $OUT/prof_$tag.log
  python $R/scripts/rocpd_stats.py timeline $OUT/prof_$tag/t_results.db > $OUT/timeline_$tag.txt
  python $R/scripts/rocpd_stats.py stats $OUT/prof_$tag/t_results.db > $OUT/stats_$tag.csv
  rm -rf $OUT/prof_$tag
done
cat $OUT/timeline_lsc_720x1280_k1600_g8.txt | cut -c1-150
