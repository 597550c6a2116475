#!/bin/bash
# Slot counts x builds: the driver's command line (SLOT_REPS runs) and a steady-state run per combination, then the burst trace of the product.
#   SLOT_LIBS="libfslic_hip.so ..."  SLOT_COUNTS="6 8"  SLOT_REPS=4  SLOT_ENV="GPU_MAX_HW_QUEUES=8" (optional, applied to every run)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
[ -n "${SLOT_ENV:-}" ] && export $SLOT_ENV
for lib in ${SLOT_LIBS:-libfslic_hip.so}; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  for sl in ${SLOT_COUNTS:-6 8}; do
    echo "== $lib slots $sl ${SLOT_ENV:-}"
    for rep in $(seq 1 ${SLOT_REPS:-4}); do python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  driver cmdline', d['value'], d['config'].get('hw_queue_of_slot'))"; done
    python bench.py --timed-only --steps 400 --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  steady 400', d['value'], d['host_cpu_ms_per_step']['per_rank'])"
  done
done
unset FSLIC_LIB
for sl in ${SLOT_BURST:-}; do BURST_ARGS="--slots $sl" BURST_TAG=slots$sl bash scripts/gpu_burst.sh; cd $R; done
