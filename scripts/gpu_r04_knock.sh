#!/bin/bash
# Round 4: knock-out builds of the fused assign kernel (results wrong on purpose; FSLIC_EXP 1: no LDS atomics in the fast epilogue, 2: no
# whole-column atomics, 4: no label look-ups): duration and SQ counters of the fused launch at 8 x 3840x2160, to see what its LDS time is made of.
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
