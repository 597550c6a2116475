#!/bin/bash
# Cycle-level SQ counters of the assign kernels (separate --pmc passes): where do the wavefront cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_cycles.csv
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rm -rf $OUT/pmc_c
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_c -o p -- python $R/scripts/frames.py ${PMC_H:-2160} ${PMC_W:-3840} ${PMC_K:-1600} 2 ${PMC_VAR:-slic} ${PMC_G:-1} > $OUT/pmc_c.log 2>&1
  DB=$(find $OUT/pmc_c -name '*_results.db' | head -1)
  python $R/scripts/rocpd_pmc.py $DB ${PMC_FILTER:-assign} >> $OUT/pmc_cycles.csv
  rm -rf $OUT/pmc_c
done
python - <<PY
import csv,collections
d=collections.defaultdict(dict)
for r in csv.reader(open("$OUT/pmc_cycles.csv")):
    if len(r)<4 or r[0]=="kernel": continue
    d[r[0]][r[1]]=float(r[2])
for k,v in d.items():
    print("==",k)
    print("  "+"  ".join("%s=%.3g"%(c[3:],x) for c,x in sorted(v.items())))
PY
