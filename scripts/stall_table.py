#!/usr/bin/env python
"""Occupancy / stall table per kernel from the csv of scripts/gpu_stalls.sh (rocpd_pmc.py rows: kernel,counter,mean,calls).
Units (counter_defs.yaml, gfx950): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_BUSY_CU_CYCLES, SQ_ACTIVE_INST_* in quad-cycles summed over the
chip; SQ_BUSY_CYCLES in cycles per shader engine (summed: 32 SEs x XCD instances report)."""
import collections
import csv
import sys

d = collections.defaultdict(dict)
for r in csv.reader(open(sys.argv[1])):
    if len(r) < 4 or r[0] == "kernel":
        continue
    d[r[0]][r[1]] = float(r[2])
SIMDS, CUS = 1024.0, 256.0
hdr = ("kernel", "waves", "life us", "res/SIMD", "wait-any", "wait-inst", "wait-lds", "act VALU", "act LDS", "act VMEM", "act SCA", "VALU/SIMD", "LDS/CU", "bankcf", "ldsfifo", "tafifo", "spi-stall", "lds-full", "vgpr-full", "wave-full")
print("per launch, kernels alone on the chip (the profiler serialises them).  life = SQ_WAVE_CYCLES x 4 / waves (mean wavefront life, at 2.4 GHz); res/SIMD = "
      "SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES / 4 (resident wavefronts per SIMD while the CU is busy); wait-* and act-* as fractions of SQ_WAVE_CYCLES "
      "(of a resident wavefront's time: waiting for anything / for an instruction to issue / for the LDS; executing VALU / LDS / VMEM / scalar); VALU/SIMD = "
      "SQ_ACTIVE_INST_VALU x 4 / (SIMDs x busy cycles): how busy a SIMD's vector ALU is while its CU is busy; LDS/CU = SQ_LDS_IDX_ACTIVE / (CUs x busy); "
      "spi-stall = SPI_RA_RES_STALL_CSN (arbiter cycles a workgroup was ready and did not fit) with the resource that was full")
print(("%-44s" + " %9s" * (len(hdr) - 1)) % hdr)
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    g = lambda n: v.get(n, 0.0)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    waves = max(g("SQ_WAVES"), 1.0)
    busy_cu = max(g("SQ_BUSY_CU_CYCLES"), 1.0)          # quad-cycles summed over the CUs
    busy_cycles = busy_cu * 4.0 / CUS                    # cycles a CU was busy, mean over the CUs
    name = k.replace("void fslic::", "").split("(")[0][:44]
    print(("%-44s %9.0f %9.2f %9.2f" + " %9.3f" * 10 + " %9.0f %9.0f %9.0f %9.0f %9.0f %9.0f") % (
        name, waves, wc * 4 / waves / 2400.0, wc / busy_cu / 4.0,
        g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc,
        g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc,
        g("SQ_ACTIVE_INST_VALU") * 4 / SIMDS / busy_cycles, g("SQ_LDS_IDX_ACTIVE") / CUS / busy_cycles,
        g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
        g("SQ_LDS_DATA_FIFO_FULL") + g("SQ_LDS_CMD_FIFO_FULL"), g("SQ_VMEM_TA_ADDR_FIFO_FULL") + g("SQ_VMEM_TA_CMD_FIFO_FULL") + g("SQ_VMEM_WR_TA_DATA_FIFO_FULL"),
        g("SPI_RA_RES_STALL_CSN"), g("SPI_RA_LDS_CU_FULL_CSN"), g("SPI_RA_VGPR_SIMD_FULL_CSN"), g("SPI_RA_WAVE_SIMD_FULL_CSN")))
print()
print("raw means per launch:")
for k, v in sorted(d.items()):
    print(k.replace("void fslic::", "").split("(")[0][:60], " ".join("%s=%.0f" % kv for kv in sorted(v.items())))
