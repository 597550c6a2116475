#!/bin/bash
# GPU box: the parity suite (all -m gpu tests, or the files given), then a short bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout ${TEST_TIMEOUT:-1500} python -m pytest ${TESTS:-tests} -m gpu -x -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 600 python bench.py ${BENCH_ARGS:---steps 100 --warmup 24 --no-4k --no-cpu-baseline} 2> $OUT/bench.err > $OUT/bench.json; echo "bench rc=$?"
  tail -3 $OUT/bench.err; cut -c1-600 $OUT/bench.json
fi
