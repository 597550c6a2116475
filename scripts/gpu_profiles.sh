#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box trip (outputs under gpurun_out/r/, copy into profiles/ as rNN_*):
#   the bench line, rocprofv3 kernel statistics + stream-overlap analysis of the pipelined timed region, single-stream
#   kernel statistics / timelines of the BASELINE configurations, FETCH_SIZE / WRITE_SIZE passes of the assign launches
#   (separate --pmc passes, --kernel-trace only), in-kernel time stamps (lab builds), microbenchmarks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r
rm -rf $OUT; mkdir -p $OUT
cd $R
# the commit this tree was built from (the GPU box has no .git: the caller passes it, scripts/pmc_traffic_json.py stamps the traffic json with it)
echo "${BUILD_COMMIT:-unrecorded}" > $OUT/build_commit.txt
rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Max Clock' | head -8 > $OUT/rocminfo.txt
nproc >> $OUT/rocminfo.txt; grep -m1 'model name' /proc/cpuinfo >> $OUT/rocminfo.txt
timeout 900 python bench.py 2> $OUT/bench.err > $OUT/bench.json; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver_cmdline.err > $OUT/bench_driver_cmdline.json; echo "bench (driver command line) rc=$?"
cd /tmp && export TMPDIR=/tmp
# 1. the pipelined timed region under rocprofv3: per-kernel statistics and how the streams overlap
rm -rf $OUT/p
timeout 600 rocprofv3 --kernel-trace -d $OUT/p -o b -- python $R/bench.py --steps 1500 --warmup 12 --timed-only > $OUT/bench_profiled.json 2>$OUT/bench_profiled.err
DB=$(find $OUT/p -name '*_results.db' | head -1)
python $R/scripts/rocpd_stats.py stats $DB $OUT/bench_kernel_stats.csv > /dev/null
python $R/scripts/rocpd_stats.py overlap $DB 0.1 > $OUT/bench_stream_overlap.txt      # (the last tenth of the trace: all of it inside the timed region)
python $R/scripts/rocpd_stats.py window $DB 4000 500 > $OUT/bench_all_streams_window.txt
rm -rf $OUT/p
# 2. single-stream groups of the BASELINE configurations
for cfg in "720 1280 1600 6 slic 8" "720 1280 1600 6 slic 16" "720 1280 1600 6 slic 1" "2160 3840 1600 4 slic 8" "2160 3840 1600 4 slic 1" "2160 3840 6000 4 slic 4" "720 1280 1600 4 lsc 8"; do
  set -- $cfg
  tag=${2}x${1}_k${3}_${5}_x${6}
  rm -rf $OUT/p
  FSLIC_GROUP=$6 timeout 300 rocprofv3 --kernel-trace -d $OUT/p -o t -- python $R/scripts/frames.py $cfg > /dev/null 2>&1      # (one launch group of all G frames)
  DB=$(find $OUT/p -name '*_results.db' | head -1)
  python $R/scripts/rocpd_stats.py timeline $DB | cut -c1-170 > $OUT/timeline_$tag.txt
  python $R/scripts/rocpd_stats.py stats $DB > $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/p
done
# 3. HBM traffic of the assign launches (separate --pmc passes, as the MI355X guide prescribes)
for cfg in "720 1280 1600 3 slic 8" "720 1280 1600 3 slic 16" "2160 3840 1600 3 slic 1"; do
  set -- $cfg
  tag=${2}x${1}_x${6}
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/p
    FSLIC_GROUP=$6 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p -o p -- python $R/scripts/frames.py $cfg > /dev/null 2>&1
    DB=$(find $OUT/p -name '*_results.db' | head -1)
    python $R/scripts/rocpd_pmc.py $DB >> $OUT/pmc_fetch_write_$tag.csv
    rm -rf $OUT/p
  done
done
cd $R
# 4. in-kernel time stamps (lab builds of the same sources: make -C fast_slic_amd/csrc LAB=1 / LAB=2)
[ -f fast_slic_amd/libfslic_hip_lab1.so ] && FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab1.so timeout 200 python scripts/lab_select_stamps.py 2>/dev/null > $OUT/lab_select_stamps.txt
[ -f fast_slic_amd/libfslic_hip_lab1.so ] && { echo "== 3840x2160 K=6000, 4 frames" >> $OUT/lab_select_stamps.txt; FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab1.so timeout 200 python scripts/lab_select_stamps.py 2160 3840 6000 4 2>/dev/null >> $OUT/lab_select_stamps.txt; }
[ -f fast_slic_amd/libfslic_hip_lab2.so ] && FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab2.so timeout 200 python scripts/lab_local_phases.py 720 1280 1600 8 > $OUT/lab_tile_ccl_phases_720p.txt 2>&1
[ -f fast_slic_amd/libfslic_hip_lab2.so ] && FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab2.so timeout 200 python scripts/lab_local_phases.py 2160 3840 1600 8 > $OUT/lab_tile_ccl_phases_4k.txt 2>&1
[ -f fast_slic_amd/libfslic_hip_lab4.so ] && for cfg in "720 1280 1600 8" "720 1280 1600 1" "2160 3840 1600 8" "2160 3840 6000 4"; do
  echo "== $cfg" >> $OUT/lab_assign_phases.txt
  FSLIC_LIB=$R/fast_slic_amd/libfslic_hip_lab4.so timeout 200 python scripts/lab_assign_phases.py $cfg 2>&1 | grep -v amdgpu.ids >> $OUT/lab_assign_phases.txt
done
# 4b. instruction mix of every kernel of an 8 x 1280x720 group (SQ counters, separate --pmc passes)
PMC_G=8 bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_group.txt $OUT/pmc_group_instruction_mix_720p_x8.txt; cd $R
FSLIC_GROUP=16 PMC_G=16 bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_group.txt $OUT/pmc_group_instruction_mix_720p_x16.txt; cd $R
PMC_H=2160 PMC_W=3840 PMC_G=8 bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_group.txt $OUT/pmc_group_instruction_mix_4k_x8.txt; cd $R
# 5. microbenchmarks
for m in ${MICROBENCH:-copy_rate}; do
  [ -x scripts/microbench/$m.out ] && timeout 200 scripts/microbench/$m.out > $OUT/microbench_$m.txt 2>&1
done
# 6. the driver's command line under rocprofv3: what every stream does during the timed burst
BURST_REPS=1 bash scripts/gpu_burst.sh > /dev/null 2>&1; cp $R/gpurun_out/burst_product_1.txt $OUT/burst_driver_cmdline.txt 2>/dev/null
cd $R
# 7. the LSC group: SQ counters of every kernel (vector ALU / LDS time, bank conflicts of k_lsc_assign_sums)
PMC_VAR=lsc PMC_G=8 bash scripts/gpu_pmc_group.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_group.txt $OUT/lsc_pmc_instruction_mix.txt; cd $R
# 8. occupancy / stall counters of every kernel of the benched group shape (scripts/gpu_stalls.sh, round 6)
FSLIC_GROUP=16 STALL_CFG="720 1280 1600 16" bash scripts/gpu_stalls.sh > /dev/null 2>&1; cp $R/gpurun_out/stalls/pmc_stalls_1280x720_k1600_x16.txt $OUT/ 2>/dev/null; cd $R
FSLIC_GROUP=8 STALL_CFG="2160 3840 1600 8" bash scripts/gpu_stalls.sh > /dev/null 2>&1; cp $R/gpurun_out/stalls/pmc_stalls_3840x2160_k1600_x8.txt $OUT/ 2>/dev/null; cd $R
python scripts/pmc_traffic_json.py $OUT/ > /dev/null 2>&1
ls $OUT
