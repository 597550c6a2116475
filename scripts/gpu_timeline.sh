#!/bin/bash
# Kernel timeline of one bench step (8 x 720p in one group) + stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_tl
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_tl -o t -- python $R/bench.py --slots ${SLOTS:-1} --no-cpu-baseline --no-4k --steps 3 --warmup 2 ${BENCH_ARGS:-} > $OUT/prof_tl.log 2>&1
python $R/scripts/rocpd_stats.py timeline $OUT/prof_tl/t_results.db > $OUT/timeline_bench.txt
python $R/scripts/rocpd_stats.py stats $OUT/prof_tl/t_results.db > $OUT/stats_bench.csv
rm -rf $OUT/prof_tl
cat $OUT/timeline_bench.txt
