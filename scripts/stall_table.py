#!/usr/bin/env python
"""Occupancy / stall table per kernel from the csv of scripts/gpu_stalls.sh (rocpd_pmc.py rows: kernel,counter,mean,calls).
Units, calibrated on the data (profiles/r06_pipeline_stalls.txt): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed
over the chip (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU: one issue slot of four cycles per vector instruction); SQ_BUSY_CU_CYCLES in
CYCLES summed over the 256 CUs (k_cca_local: 11.07 M / 256 = 43 258 cycles = 18.0 us at 2.4 GHz, the launch's duration)."""
import collections
import csv
import sys

d = collections.defaultdict(dict)
for r in csv.reader(open(sys.argv[1])):
    if len(r) < 4 or r[0] == "kernel":
        continue
    d[r[0]][r[1]] = float(r[2])
SIMDS, CUS = 1024.0, 256.0
hdr = ("kernel", "waves", "busy us", "life us", "res/SIMD", "wait-any", "wait-inst", "wait-lds", "act VALU", "act LDS", "act VMEM", "act SCA", "VALU/SIMD", "SALU/SIMD", "LDS/CU", "bankcf", "ldsfifo", "tafifo", "spi-stall", "lds-full", "vgpr-full", "wave-full")
print("per launch, kernels alone on the chip (the profiler serialises them).  life = SQ_WAVE_CYCLES x 4 / waves (mean wavefront life, at 2.4 GHz); res/SIMD = "
      "SQ_WAVE_CYCLES x 4 / SQ_BUSY_CU_CYCLES / 4 (resident wavefronts per SIMD while the CU is busy); wait-* and act-* as fractions of SQ_WAVE_CYCLES "
      "(of a resident wavefront's time: waiting for anything / for an instruction to issue / for the LDS; executing VALU / LDS / VMEM / scalar); VALU/SIMD = "
      "SQ_ACTIVE_INST_VALU x 4 / (SIMDs x busy cycles): how busy a SIMD's vector ALU is while its CU is busy (at 4 cycles per instruction; v_sad_u8 / v_pk_* take 4.9); "
      "SALU/SIMD = SQ_INSTS_SALU x 4 / (SIMDs x busy cycles): a SIMD's share of scalar issue slots (one per four cycles); LDS/CU = SQ_LDS_IDX_ACTIVE / (CUs x busy); busy us = SQ_BUSY_CU_CYCLES / CUs / 2400; "
      "spi-stall = SPI_RA_RES_STALL_CSN (arbiter cycles a workgroup was ready and did not fit) with the resource that was full")
print(("%-44s" + " %9s" * (len(hdr) - 1)) % hdr)
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    g = lambda n: v.get(n, 0.0)
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    waves = max(g("SQ_WAVES"), 1.0)
    busy_cu = max(g("SQ_BUSY_CU_CYCLES"), 1.0)          # cycles summed over the CUs
    busy_cycles = busy_cu / CUS                          # cycles a CU was busy, mean over the CUs
    name = k.replace("void fslic::", "").split("(")[0][:44]
    print(("%-44s %9.0f %9.2f %9.2f %9.2f" + " %9.3f" * 11 + " %9.0f %9.0f %9.0f %9.0f %9.0f %9.0f") % (
        name, waves, busy_cycles / 2400.0, wc * 4 / waves / 2400.0, wc / busy_cu,
        g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_WAIT_INST_LDS") / wc,
        g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc,
        g("SQ_ACTIVE_INST_VALU") * 4 / SIMDS / busy_cycles, g("SQ_INSTS_SALU") * 4 / SIMDS / busy_cycles, g("SQ_LDS_IDX_ACTIVE") / CUS / busy_cycles,
        g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1.0),
        g("SQ_LDS_DATA_FIFO_FULL") + g("SQ_LDS_CMD_FIFO_FULL"), g("SQ_VMEM_TA_ADDR_FIFO_FULL") + g("SQ_VMEM_TA_CMD_FIFO_FULL") + g("SQ_VMEM_WR_TA_DATA_FIFO_FULL"),
        g("SPI_RA_RES_STALL_CSN"), g("SPI_RA_LDS_CU_FULL_CSN"), g("SPI_RA_VGPR_SIMD_FULL_CSN"), g("SPI_RA_WAVE_SIMD_FULL_CSN")))
print()
print("raw means per launch:")
for k, v in sorted(d.items()):
    print(k.replace("void fslic::", "").split("(")[0][:60], " ".join("%s=%.0f" % kv for kv in sorted(v.items())))
