#!/bin/bash
# One GPU-box round trip: parity tests, bench line, rocprofv3 kernel-trace summary.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh'
# Everything lands under gpurun_out/ (merged back); copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Max Clock' | head -8 > $OUT/rocminfo.txt
nproc >> $OUT/rocminfo.txt; grep -m1 'model name' /proc/cpuinfo >> $OUT/rocminfo.txt
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py ${BENCH_ARGS:-} 2> $OUT/bench.err > $OUT/bench.json; echo "bench rc=$?"
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_bench
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-4k > $OUT/prof_bench.log 2>&1
echo "rocprof rc=$?"
DB=$(find $OUT/prof_bench -name '*_results.db' | head -1)
if [ -n "$DB" ]; then
  python $R/scripts/rocpd_stats.py stats $DB $OUT/bench_kernel_stats.csv | head -40
  python $R/scripts/rocpd_stats.py timeline $DB > $OUT/bench_last_frame_timeline.txt
fi
rm -rf $OUT/prof_bench
