#!/bin/bash
# Instruction mix of EVERY kernel of a group (separate --pmc passes, kernels serialised by the profiler): which unit
# carries the group's work?   PMC_H/PMC_W/PMC_K/PMC_VAR/PMC_G as in gpu_pmc_cycles.sh.  Output: gpurun_out/pmc_group.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_group.csv
for set in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"; do
  rm -rf $OUT/pmc_c
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_c -o p -- python $R/scripts/frames.py ${PMC_H:-720} ${PMC_W:-1280} ${PMC_K:-1600} 3 ${PMC_VAR:-slic} ${PMC_G:-8} > $OUT/pmc_c.log 2>&1
  DB=$(find $OUT/pmc_c -name '*_results.db' | head -1)
  python $R/scripts/rocpd_pmc.py $DB fslic >> $OUT/pmc_group.csv
  rm -rf $OUT/pmc_c
done
python - <<PY > $OUT/pmc_group.txt
import csv, collections
d = collections.defaultdict(dict); calls = {}
for r in csv.reader(open("$OUT/pmc_group.csv")):
    if len(r) < 4 or r[0] == "kernel": continue
    d[r[0]][r[1][3:]] = float(r[2]); calls[r[0]] = int(r[3]) / 3.0
SIMDS, CUS, GHZ = 1024.0, 256.0, 2.4
print("per GROUP (calls x mean), in us at %.1f GHz if that unit alone were the limit: VALU = INSTS_VALU x 4 cycles / %d SIMDs; LDS = LDS_IDX_ACTIVE / %d CUs" % (GHZ, SIMDS, CUS))
print("%-52s %6s %9s %9s %9s %9s %9s %9s %9s" % ("kernel", "calls", "VALU us", "SALU us", "LDS us", "VMEM/wave", "waves", "busyCU us", "bank-cf"))      # bank-cf: LDS_BANK_CONFLICT / LDS_IDX_ACTIVE
tot = collections.Counter()
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("INSTS_VALU", 0) * calls[kv[0]]):
    n = calls[k]
    valu = n * v.get("INSTS_VALU", 0) * 4 / SIMDS / GHZ / 1e3
    salu = n * v.get("INSTS_SALU", 0) / SIMDS / GHZ / 1e3
    lds = n * v.get("LDS_IDX_ACTIVE", 0) / CUS / GHZ / 1e3
    vm = (v.get("INSTS_VMEM_RD", 0) + v.get("INSTS_VMEM_WR", 0)) / max(v.get("WAVES", 1), 1)
    busy = n * v.get("BUSY_CU_CYCLES", 0) / CUS / GHZ / 1e3
    tot.update({"valu": valu, "salu": salu, "lds": lds, "busy": busy})
    print("%-52s %6.1f %9.2f %9.2f %9.2f %9.1f %9.0f %9.2f %9.2f" % (k[:52], n, valu, salu, lds, vm, v.get("WAVES", 0), busy, v.get("LDS_BANK_CONFLICT", 0) / max(v.get("LDS_IDX_ACTIVE", 0), 1)))
print("%-52s %6s %9.2f %9.2f %9.2f %9s %9s %9.2f" % ("group total", "", tot["valu"], tot["salu"], tot["lds"], "", "", tot["busy"]))
PY
cat $OUT/pmc_group.txt
