#!/bin/bash
# Round 4 experiment: up to 32 frames per launch group (libfslic_hip_var_g32.so, -DFSLIC_MAX_GROUP=32) against 16.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
run() { python bench.py --timed-only --steps ${2:-400} --launch-frames $1 ${3:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'MP/s', d['ms_per_step'], 'ms/step')"; }
echo "product, 16 frames per launch:"; run 16; run 16 20
export FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_var_g32.so
echo "g32 build, 16 frames per launch:"; run 16
echo "g32 build, 24 frames per launch:"; run 24
echo "g32 build, 32 frames per launch:"; run 32; run 32 20; run 32 20
echo "g32 build, 32 frames per launch, 4K:"; python bench.py --workload 4k --timed-only --steps 100 --launch-frames 32 2>/dev/null | tail -1 | cut -c1-200
echo "g32 build, parity of the timed output:"; python bench.py --steps 100 --launch-frames 32 --no-cpu-baseline --no-4k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k: v for k, v in d.items() if 'parity' in k})"
