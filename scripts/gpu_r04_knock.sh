#!/bin/bash
# Round 4: knock-out builds (results wrong on purpose, only duration and SQ counters are read): what a kernel's LDS time is made of.
# Fused assign kernel at 8 x 3840x2160: the guards were FSLIC_EXP 1 (no LDS atomics in the fast epilogue), 2 (no whole-column atomics), 4 (no
# label look-ups); they found the label table's bank conflicts and were removed with the fix (profiles/r04_assign_experiments.txt; restore them
# from commit 7126983's parent to repeat).  LSC: FSLIC_EXP 8 (no block-accumulator atomics in k_lsc_assign_sums) is in lsc.hip, not yet run:
#   make -C fast_slic_amd/csrc VAR=k8 DEFS=-DFSLIC_EXP=8;  FSLIC_LIB=.../libfslic_hip_var_k8.so PMC_VAR=lsc bash scripts/gpu_pmc_group.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in libfslic_hip.so libfslic_hip_var_k1.so libfslic_hip_var_k2.so libfslic_hip_var_k4.so; do
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="2160 3840 1600 3 slic 8" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "k_assign_blk2_w6<16" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
  PMC_H=2160 PMC_W=3840 bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "k_assign_blk2_w6<16" | cut -c1-140
  python - <<PY
import csv
rows=[r for r in csv.reader(open('gpurun_out/pmc_group.csv')) if len(r)>=4 and 'k_assign_blk2_w6<16, true' in r[0]]
d={r[1]:float(r[2]) for r in rows}; w=d.get('SQ_WAVES',1)
print("  per wavefront: LDS instructions %.1f, index-active %.0f, bank-conflict %.0f, vector instructions %.0f" % (d['SQ_INSTS_LDS']/w, d['SQ_LDS_IDX_ACTIVE']/w, d['SQ_LDS_BANK_CONFLICT']/w, d['SQ_INSTS_VALU']/w))
PY
done
