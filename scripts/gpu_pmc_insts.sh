#!/bin/bash
# Instruction counters of the assign kernels under the experiment flags (FSLIC_EXP) -> per-phase dynamic instruction counts
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in ${EXPS:-0}; do
  rm -rf $OUT/pmc_i
  FSLIC_EXP=$e timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_i -o p -- python $R/scripts/frames.py ${PMC_H:-2160} ${PMC_W:-3840} ${PMC_K:-1600} 2 > $OUT/pmc_i.log 2>&1
  DB=$(find $OUT/pmc_i -name '*_results.db' | head -1)
  echo "== FSLIC_EXP=$e"
  python $R/scripts/rocpd_pmc.py $DB assign_blk | python -c "
import sys,csv,collections
d=collections.defaultdict(dict)
for r in csv.DictReader(sys.stdin): d[r['kernel']][r['counter']]=float(r['mean'])
for k,v in d.items():
    w=v['SQ_WAVES']; print(k[18:52], 'waves=%d'%w, ' '.join('%s=%.0f'%(c[8:],v[c]/w) for c in sorted(v) if c!='SQ_WAVES'))
"
  rm -rf $OUT/pmc_i
done
