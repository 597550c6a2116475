#!/bin/bash
# Per-kernel resource usage of one kernel file (default assign.hip) (VGPRs, SGPRs, scratch, LDS, occupancy) from hipcc's remarks.
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -c -o /dev/null \
  ${1:-$(dirname $0)/../fast_slic_amd/csrc/assign.hip} -Wno-unused-value -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k.split(" ")[0]]=v
print("%-70s %5s %5s %7s %6s %4s"%("kernel","vgpr","sgpr","scratch","lds","occ"))
for r in rows:
    print("%-70s %5s %5s %7s %6s %4s"%(r["name"][:70],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize"),r.get("LDS"),r.get("Occupancy")))
'
