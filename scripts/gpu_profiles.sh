#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box trip (outputs under gpurun_out/r/, copy into profiles/):
#   bench line, rocprofv3 kernel statistics + stream-overlap analysis of the bench, single-stream timelines,
#   FETCH_SIZE / WRITE_SIZE passes, cycle-level SQ counters, LSC throughput, microbenchmarks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r
rm -rf $OUT; mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Max Clock' | head -8 > $OUT/rocminfo.txt
nproc >> $OUT/rocminfo.txt; grep -m1 'model name' /proc/cpuinfo >> $OUT/rocminfo.txt
timeout 600 python bench.py 2> $OUT/bench.err > $OUT/bench.json; echo "bench rc=$?"
for v in lsc standard l2 noq; do timeout 300 python scripts/variant_throughput.py $v 24 4 2>/dev/null | tail -1 >> $OUT/variant_throughput.json; done
cd /tmp && export TMPDIR=/tmp
# 1. the bench under rocprofv3: per-kernel statistics and how the streams overlap
rm -rf $OUT/p
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/p -o b -- python $R/bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-4k > $OUT/bench_profiled.json 2>$OUT/bench_profiled.err
DB=$(find $OUT/p -name '*_results.db' | head -1)
python $R/scripts/rocpd_stats.py stats $DB $OUT/bench_kernel_stats.csv > /dev/null
python $R/scripts/rocpd_stats.py overlap $DB 0.4 > $OUT/bench_stream_overlap.txt
python $R/scripts/rocpd_stats.py window $DB 5000 700 > $OUT/bench_all_streams_window.txt
rm -rf $OUT/p
# 2. single-stream timelines
for cfg in "720 1280 1600 4 slic 8" "2160 3840 1600 4 slic 1" "2160 3840 1600 4 slic 4" "2160 3840 1600 4 slic 8" "2160 3840 6000 4 slic 1" "720 1280 1600 4 lsc 8"; do
  set -- $cfg
  tag=${2}x${1}_k${3}_${5}_x${6}
  rm -rf $OUT/p
  timeout 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python $R/scripts/frames.py $cfg > /dev/null 2>&1
  DB=$(find $OUT/p -name '*_results.db' | head -1)
  python $R/scripts/rocpd_stats.py timeline $DB | cut -c1-160 > $OUT/timeline_$tag.txt
  python $R/scripts/rocpd_stats.py stats $DB | cut -c1-160 > $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/p
done
# 3. HBM traffic of the assign launches (separate --pmc passes, as the MI355X guide prescribes)
for cfg in "720 1280 1600 3 slic 8" "2160 3840 1600 3 slic 1"; do
  set -- $cfg
  tag=${2}x${1}_x${6}
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/p
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o p -- python $R/scripts/frames.py $cfg > /dev/null 2>&1
    DB=$(find $OUT/p -name '*_results.db' | head -1)
    python $R/scripts/rocpd_pmc.py $DB >> $OUT/pmc_fetch_write_$tag.csv
    rm -rf $OUT/p
  done
done
# 4. cycle-level counters of the assign kernels
cd $R
GRAFT_REPO_ROOT=$R PMC_G=8 bash scripts/gpu_pmc_cycles.sh > $OUT/pmc_cycles_4k_x8.txt 2>&1
GRAFT_REPO_ROOT=$R PMC_H=720 PMC_W=1280 PMC_G=8 bash scripts/gpu_pmc_cycles.sh > $OUT/pmc_cycles_720p_x8.txt 2>&1
# 5. microbenchmarks
for m in ${MICROBENCH:-kernel_boundary graph_launch lds_unaligned dispatch_rate valu_rate pk_mad_clamp}; do
  [ -x scripts/microbench/$m.out ] && timeout 200 scripts/microbench/$m.out > $OUT/microbench_$m.txt 2>&1
done
ls $OUT
