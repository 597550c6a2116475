#!/bin/bash
# Diagnosis: what the shader clock reads under (a) a pure vector-ALU microbenchmark, (b) the pipelined loop, (c) the pipelined
# loop with the performance level forced to "high" (diagnosis only: the product never touches power settings).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
smp() { rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed -e 's/^GPU\[0\][[:space:]]*: //' | paste -sd'|'; }
echo "perf level:"; rocm-smi --showperflevel 2>/dev/null | grep -i "level"
echo "== (a) valu_rate microbenchmark"; (for i in 1 2 3; do scripts/microbench/valu_rate.out > /dev/null 2>&1; done) & BP=$!
sleep 1.5; smp; smp; smp; wait $BP
echo "== (b) pipelined loop, auto"; python scripts/sustained_load.py 8 > $OUT/cp_b.txt 2>/dev/null & BP=$!
sleep 6; smp; smp; smp; wait $BP; cat $OUT/cp_b.txt
echo "== (c) pipelined loop, perf level high"; rocm-smi --setperflevel high 2>&1 | grep -v "^$" | head -3
python scripts/sustained_load.py 8 > $OUT/cp_c.txt 2>/dev/null & BP=$!
sleep 6; smp; smp; smp; wait $BP; cat $OUT/cp_c.txt
rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | head -2
