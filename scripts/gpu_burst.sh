#!/bin/bash
# The driver's command line (20 steps after 5 warm-up steps) under rocprofv3 --kernel-trace: what every stream does during the timed
# burst (scripts/rocpd_stats.py burst).  BURST_ARGS: further bench.py arguments; BURST_LIB: a library to select through FSLIC_LIB.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "${BURST_LIB:-}" ] && export FSLIC_LIB=$R/fast_slic_amd/$BURST_LIB
for rep in $(seq 1 ${BURST_REPS:-1}); do
  rm -rf $OUT/p
  timeout 600 rocprofv3 --kernel-trace -d $OUT/p -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --timed-only ${BURST_ARGS:-} > $OUT/burst_bench.json 2>$OUT/burst_bench.err
  DB=$(find $OUT/p -name '*_results.db' | head -1)
  cut -c1-140 $OUT/burst_bench.json
  python $R/scripts/rocpd_stats.py burst $DB | tee $OUT/burst_${BURST_TAG:-product}_$rep.txt
  rm -rf $OUT/p
done
