#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the assign launches of the BENCH workload (8 x 720p per launch), separate passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_b
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_b -o p -- python $R/bench.py --steps 4 --warmup 2 --slots 1 --no-cpu-baseline --no-4k > $OUT/pmc_b.log 2>&1
  DB=$(find $OUT/pmc_b -name '*_results.db' | head -1)
  python $R/scripts/rocpd_pmc.py $DB > $OUT/pmc_bench_$set.csv
  grep -E "assign_blk|rgb_to_lab|cca_local|cca_relabel" $OUT/pmc_bench_$set.csv
  rm -rf $OUT/pmc_b
done
