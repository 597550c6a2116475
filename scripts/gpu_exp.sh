#!/bin/bash
# Timing experiments: kernel statistics of several builds of the library (fast_slic_amd/libfslic_<tag>.so) on one box.
#   EXP_TAGS="hip exp1 exp2" EXP_CFG="2160 3840 1600 4 slic 4" bash scripts/gpu_exp.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for t in ${EXP_TAGS:-hip}; do
  rm -rf $OUT/prof_t
  FSLIC_LIB=$R/fast_slic_amd/libfslic_$t.so rocprofv3 --kernel-trace -d $OUT/prof_t -o t -- python $R/scripts/frames.py ${EXP_CFG:-2160 3840 1600 4 slic 4} > /dev/null 2>&1
  echo "== $t (${EXP_CFG:-2160 3840 1600 4 slic 4})"
  python $R/scripts/rocpd_stats.py stats $OUT/prof_t/t_results.db | grep "${EXP_FILTER:-assign_blk}" | cut -c1-120
  rm -rf $OUT/prof_t
done
