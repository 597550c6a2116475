#!/bin/bash
# Is the pipelined rate bound by the device or by one process's host side?  The bench's timed loop alone, then the same loop
# in two (three) processes at once on the one GPU: a device-bound rate stays put in sum, a host-bound one scales.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
one() { python bench.py --steps 400 --warmup 24 --timed-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"; }
echo "one process:"; one
for n in 2 3; do
  echo "$n processes at once:"
  for i in $(seq $n); do one > $OUT/probe_$i.txt & done
  wait
  cat $(for i in $(seq $n); do echo $OUT/probe_$i.txt; done) | tr '\n' ' '; echo
done
