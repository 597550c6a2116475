#!/bin/bash
# Round 4: the whole GPU suite on the product build, then A/B against libfslic_hip_var_old.so (kernel durations, pipelined rates),
# then the instruction mix of the fused launch at 8 x 4K and of a 720p group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/exp/pytest_all.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/exp/pytest_all.log
[ $rc -ne 0 ] && exit 1
EXP_LIBS="libfslic_hip.so libfslic_hip_var_old.so" EXP_PARITY="none" EXP_4K="libfslic_hip.so libfslic_hip_var_old.so" EXP_NOBATCH=${AB_DRIVER:-0} bash scripts/gpu_exp.sh
unset FSLIC_LIB
bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp gpurun_out/pmc_group.txt gpurun_out/exp/pmc_group_720p_x8.txt; cut -c1-140 gpurun_out/pmc_group.txt
PMC_H=2160 PMC_W=3840 bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp gpurun_out/pmc_group.txt gpurun_out/exp/pmc_group_4k_x8.txt; cut -c1-140 gpurun_out/pmc_group.txt | head -8
