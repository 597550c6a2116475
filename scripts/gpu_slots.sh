#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for sl in 4 5 6 8 10 12; do
  for rep in 1 2 3; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --slots $sl --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots $sl rep $rep: value', d['value'])"
  done
done
for sl in 5 6 10; do
  timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 24 --slots $sl --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slots $sl 1000 steps: value', d['value'])"
done
