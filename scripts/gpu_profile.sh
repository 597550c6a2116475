#!/bin/bash
# Single-stream kernel timelines (rocprofv3 --kernel-trace) + in-kernel phase stamps of the assign kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "720 1280 1600" "2160 3840 1600" "2160 3840 6000"; do
  set -- $cfg
  tag=${1}x${2}_k${3}
  rm -rf $OUT/prof_$tag
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$tag -o t -- python $R/scripts/frames.py $1 $2 $3 4 > $OUT/prof_$tag.log 2>&1
  python $R/scripts/rocpd_stats.py timeline $OUT/prof_$tag/t_results.db > $OUT/timeline_$tag.txt
  FSLIC_DEBUG_TIMING=1 timeout 120 python $R/scripts/frames.py $1 $2 $3 2 > $OUT/phase1_$tag.txt 2>&1
  FSLIC_DEBUG_TIMING=2 timeout 120 python $R/scripts/frames.py $1 $2 $3 2 > $OUT/phase2_$tag.txt 2>&1
  rm -rf $OUT/prof_$tag
done
cat $OUT/timeline_720x1280_k1600.txt
grep dbg $OUT/phase*_*.txt
