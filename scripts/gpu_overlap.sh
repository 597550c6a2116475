#!/bin/bash
# What the GPU does while the pipeline runs: rocprofv3 kernel trace of the timed region only, concurrency histogram and a window of all streams.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout 600 rocprofv3 --kernel-trace -d $OUT/p -o b -- python $R/bench.py --steps ${OV_STEPS:-60} --warmup 12 --timed-only ${OV_ARGS:-} > $OUT/bench_profiled.json 2>$OUT/bench_profiled.err
DB=$(find $OUT/p -name '*_results.db' | head -1)
python $R/scripts/rocpd_stats.py overlap $DB 0.4 > $OUT/bench_stream_overlap.txt
python $R/scripts/rocpd_stats.py window $DB 4000 600 > $OUT/bench_all_streams_window.txt
cut -c1-160 $OUT/bench_profiled.json; cat $OUT/bench_stream_overlap.txt
rm -rf $OUT/p
