#!/bin/bash
# How often does the driver-sized timed region (20 steps after 5 warm-up steps: 3.8 ms) catch a stall?  N runs, value of each.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-30}
for i in $(seq $N); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"
done | tr '\n' ' '
echo
