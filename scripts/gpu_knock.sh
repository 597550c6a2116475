#!/bin/bash
# Knock-out builds (results wrong on purpose, only duration and SQ counters are read): what a kernel's LDS time is made of.
#   KNOCK_LIBS="libfslic_hip.so libfslic_hip_var_k8.so"  KNOCK_KERNEL="k_lsc_assign_sums"  KNOCK_CFG="720 1280 1600 3 lsc 8"  PMC_* as gpu_pmc_group.sh
# Round 4, fused assign kernel at 8 x 3840x2160: the guards were FSLIC_EXP 1 (no LDS atomics in the fast epilogue), 2 (no whole-column atomics), 4
# (no label look-ups); they found the label table's bank conflicts and were removed with the fix (profiles/r04_assign_experiments.txt; the
# guarded source is in commit 7126983's tree of experiments: `git log -S"FSLIC_EXP & 4"`).  LSC: FSLIC_EXP 8 (no block-accumulator atomics in
# k_lsc_assign_sums), 16 (no packed FMAs in its walk), 32 (no feature rebuild) are in lsc.hip: profiles/r05_lsc_knockout.txt.  make -C fast_slic_amd/csrc VAR=k8 DEFS=-DFSLIC_EXP=8
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
KERNEL=${KNOCK_KERNEL:-k_lsc_assign_sums}
CFG=${KNOCK_CFG:-720 1280 1600 3 lsc 8}
set -- $CFG
export PMC_H=$1 PMC_W=$2 PMC_K=$3 PMC_VAR=$5 PMC_G=$6
for lib in ${KNOCK_LIBS:-libfslic_hip.so libfslic_hip_var_k8.so}; do
  [ -f $R/fast_slic_amd/$lib ] || { echo "missing $lib"; continue; }
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  echo "######## $lib"
  KSTATS_CFGS="$CFG" bash scripts/gpu_kstats.sh 2>/dev/null | grep -E "$KERNEL" | sed 's/(fslic::FrameDev[^"]*"/"/' | cut -c1-100
  cd $R
  bash scripts/gpu_pmc_group.sh 2>/dev/null | grep -E "$KERNEL" | cut -c1-140
  python - <<PY
import csv
rows=[r for r in csv.reader(open('gpurun_out/pmc_group.csv')) if len(r)>=4 and '$KERNEL' in r[0]]
d={r[1]:float(r[2]) for r in rows}; w=d.get('SQ_WAVES',1)
print("  per wavefront: LDS instructions %.1f, index-active %.0f, bank-conflict %.0f, vector instructions %.0f" % (d['SQ_INSTS_LDS']/w, d['SQ_LDS_IDX_ACTIVE']/w, d['SQ_LDS_BANK_CONFLICT']/w, d['SQ_INSTS_VALU']/w))
PY
done
