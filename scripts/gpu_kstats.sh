#!/bin/bash
# Per-kernel statistics (rocprofv3 --kernel-trace) of single-stream groups: KSTATS_CFGS="H W K n variant G; ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra CFGS <<< "${KSTATS_CFGS:-720 1280 1600 6 slic 8}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  tag=${2}x${1}_k${3}_${5}_x${6}
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python $R/scripts/frames.py $cfg > /dev/null 2>&1
  DB=$(find $OUT/p -name '*_results.db' | head -1)
  python $R/scripts/rocpd_stats.py timeline $DB | cut -c1-170 > $OUT/timeline_$tag.txt
  python $R/scripts/rocpd_stats.py stats $DB > $OUT/kernel_stats_$tag.csv
  echo "== $tag"; cat $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/p
done
if [ "${KSTATS_BENCH:-0}" = "1" ]; then
  rm -rf $OUT/p
  timeout 600 rocprofv3 --kernel-trace -d $OUT/p -o b -- python $R/bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-4k --no-parity > $OUT/bench_profiled.json 2>$OUT/bench_profiled.err
  DB=$(find $OUT/p -name '*_results.db' | head -1)
  python $R/scripts/rocpd_stats.py overlap $DB 0.4 > $OUT/bench_stream_overlap.txt
  cat $OUT/bench_stream_overlap.txt
  rm -rf $OUT/p
fi
