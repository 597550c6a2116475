#!/bin/bash
# Quick GPU check: parity tests (-x), then bench variants.  Usage via gpurun; env TESTS=0 skips tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -25 $OUT/pytest_gpu.log
fi
for sl in ${SLOTS:-1 2}; do
  timeout 300 python bench.py --slots $sl --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_s$sl.json 2> $OUT/bench_s$sl.err; echo "bench slots=$sl rc=$?"
  cat $OUT/bench_s$sl.json; tail -3 $OUT/bench_s$sl.err
done
