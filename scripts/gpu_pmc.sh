#!/bin/bash
# PMC counters of the assign kernel (separate passes, --kernel-trace only, as the MI355X guide prescribes).
# Workload: single-stream 3840x2160 K=1600 frames (scripts/frames.py).  Output: gpurun_out/pmc_<set>.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
H=${PMC_H:-2160}; W=${PMC_W:-3840}; K=${PMC_K:-1600}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $OUT/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i -o p -- python $R/scripts/frames.py $H $W $K 3 > $OUT/pmc_$i.log 2>&1
  DB=$(find $OUT/pmc_$i -name '*_results.db' | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_pmc.py $DB > $OUT/pmc_$i.csv; cat $OUT/pmc_$i.csv | head -30; else tail -5 $OUT/pmc_$i.log; fi
  rm -rf $OUT/pmc_$i
done
