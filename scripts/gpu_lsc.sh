#!/bin/bash
# LSC round: the tolerance tests, per-kernel statistics of one 8-frame LSC group and the pipelined rate; with LSC_AB=<lib> the
# statistics and the rate again for a second build of the library.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
[ "${LSC_TEST:-0}" = "1" ] && python -m pytest tests/test_gpu_lsc.py -q -x 2>&1 | tail -3
for lib in "" $LSC_AB; do
  [ -n "$lib" ] && export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "=== lib ${lib:-default}"
  KSTATS_CFGS="720 1280 1600 6 lsc 8" bash $R/scripts/gpu_kstats.sh 2>&1 | grep -E "lsc_assign|lsc_feat|lsc_seed" | head -12
  [ "${LSC_RATE:-1}" = "1" ] && python $R/scripts/variant_throughput.py lsc 48 4 2>/dev/null | tail -1 | cut -c1-200
done
