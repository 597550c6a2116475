#!/bin/bash
# Round 6, VERDICT r05 item 2: why is a chip whose vector ALUs are 54 % busy the limit of the pipelined region?
#   (1) SQ occupancy / stall counters of every kernel of a group (separate --pmc passes; the profiler serialises the kernels, so these are
#       the kernels ALONE on the chip): STALL_CFG="H W K G" (default 720 1280 1600 16)
#   (2) the GPU_MAX_HW_QUEUES x slots grid on the current build (STALL_GRID=1)
#   (3) PC sampling INSIDE the pipelined region (nothing is serialised): which kernels the sampled wavefronts belong to and, with the
#       stochastic method, what they were stalled on (STALL_PCS=1; beta feature of rocprofv3, run under its own timeout)
# Output: gpurun_out/stalls/*
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
set -- ${STALL_CFG:-720 1280 1600 16}
H=$1; W=$2; K=$3; G=$4
if [ "${STALL_PMC:-1}" = "1" ]; then
  : > $OUT/pmc.csv
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
             "SPI_RA_RES_STALL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN"; do
    rm -rf $OUT/p
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o p -- python $R/scripts/frames.py $H $W $K 3 slic $G > $OUT/pmc.log 2>&1 || { echo "pmc set failed: $set"; tail -3 $OUT/pmc.log; }
    DB=$(find $OUT/p -name '*_results.db' | head -1)
    [ -n "$DB" ] && python $R/scripts/rocpd_pmc.py $DB fslic >> $OUT/pmc.csv
    rm -rf $OUT/p
  done
  python $R/scripts/stall_table.py $OUT/pmc.csv > $OUT/pmc_stalls_${W}x${H}_k${K}_x${G}.txt
  cat $OUT/pmc_stalls_${W}x${H}_k${K}_x${G}.txt | cut -c1-200
fi
cd $R
if [ "${STALL_GRID:-0}" = "1" ]; then
  for q in ${STALL_QUEUES:-4 6 8}; do for sl in ${STALL_SLOTS:-6 8}; do
    for rep in 1 2; do
      GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --timed-only --steps ${STALL_STEPS:-1000} --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q slots=$sl steady ${STALL_STEPS:-1000}:', d['value'], 'MP/s')"
    done
    for rep in 1 2 3; do
      GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only --slots $sl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q slots=$sl driver 20:', d['value'], 'MP/s')"
    done
  done; done | tee $OUT/hw_queues_x_slots.txt
fi
if [ "${STALL_PCS:-0}" = "1" ]; then
  cd /tmp
  rocprofv3 -L 2>/dev/null | grep -i -B2 -A12 "pc.sampl" | head -60 > $OUT/pcs_avail.txt
  for method in ${STALL_PCS_METHODS:-stochastic host_trap}; do
    rm -rf $OUT/pcs
    if [ $method = stochastic ]; then unit=cycles; iv=${STALL_PCS_CYCLES:-1048576}; else unit=time; iv=${STALL_PCS_US:-20}; fi
    timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $iv --kernel-trace \
        --output-format csv -d $OUT/pcs -o s -- python $R/bench.py --timed-only --steps ${STALL_PCS_STEPS:-300} > $OUT/pcs_$method.log 2>&1
    echo "pc sampling ($method) rc=$?"; tail -2 $OUT/pcs_$method.log | cut -c1-300
    find $OUT/pcs -type f | xargs ls -la 2>/dev/null | head
    F=$(find $OUT/pcs -name '*pc_sampling*.csv' | head -1)
    if [ -n "$F" ]; then
      head -5 $F > $OUT/pcs_${method}_head.txt
      python $R/scripts/pcs_table.py $F $(find $OUT/pcs -name '*kernel_trace.csv' | head -1) > $OUT/pcs_${method}_table.txt 2>&1
      cut -c1-220 $OUT/pcs_${method}_table.txt | head -70
    fi
    rm -rf $OUT/pcs
  done
fi
