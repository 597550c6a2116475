cd $GRAFT_REPO_ROOT
for lib in libfslic_hip.so libfslic_hip_var_old.so; do
  export FSLIC_LIB=$PWD/fast_slic_amd/$lib
  for sl in 4 6 8; do
    echo "== $lib slots $sl"
    for rep in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  driver cmdline', d['value'])"; done
    python bench.py --timed-only --steps 400 --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  steady 400', d['value'], d['host_cpu_ms_per_step']['per_rank'])"
  done
done
unset FSLIC_LIB
BURST_ARGS="--slots 8" BURST_TAG=slots8 bash scripts/gpu_burst.sh
cd $GRAFT_REPO_ROOT
BURST_ARGS="--slots 4" BURST_TAG=slots4 bash scripts/gpu_burst.sh
