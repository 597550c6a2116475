#!/bin/bash
# One parametrised GPU trip (round 5; replaces the single-purpose gpu_r04_*.sh trip files): every section is switched on by a variable.
#   gpurun --timeout 1500 -- 'TRIP_SUITE=1 TRIP_LIBS="libfslic_hip.so libfslic_hip_var_old.so" bash scripts/gpu_trip.sh'
#
#   TRIP_SUITE=1            the whole -m gpu suite on the product build (no -x: every failure is listed), TRIP_TESTS / TRIP_K narrow it
#   TRIP_LIBS="a.so b.so"   A/B over builds of the same ABI (fast_slic_amd/<name>, selected through FSLIC_LIB); per build:
#     TRIP_PARITY="..."       pytest -k expression of a parity subset run FIRST on that build (empty: none)
#     TRIP_CFGS="H W K n variant G; ..."   rocprofv3 kernel statistics + timelines of single-stream groups (scripts/gpu_kstats.sh)
#     TRIP_KERNELS="regex"    which rows of those statistics are echoed (the files keep all rows)
#     TRIP_STEADY=N           pipelined rate over N steps (bench.py --timed-only), TRIP_STEADY_REPS times
#     TRIP_DRIVER=N           the driver's command line (--steps 20 --warmup 5), N times
#     TRIP_4K=1               pipelined 4K rate (100 steps)
#     TRIP_PMC=1              instruction mix of every kernel of one group (scripts/gpu_pmc_group.sh; PMC_* select the group)
#   TRIP_SELECT="lab1.so .."  in-kernel time stamps of k_cca_select per frame (lab build 1), per library
#   TRIP_STAMPS4="lab4.so .." phase stamps of the block assign kernel (lab build 4, scripts/lab_assign_phases.py), per library
#   TRIP_BENCH=1            the complete bench line of the product build (python bench.py, all legs) -> gpurun_out/trip/bench_full.json
#   TRIP_BENCH_DRIVER=1     ... and the driver's command line with all legs -> bench_driver_cmdline.json
#   TRIP_OVERLAP=1          rocprofv3 kernel trace of the pipelined region: stream overlap + per-kernel share (scripts/gpu_kstats.sh KSTATS_BENCH)
#   TRIP_EXTRA="cmd"        anything else, run last from the repository root
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trip
mkdir -p $OUT
cd $R
tagof() { local t=${1%.so}; t=${t#libfslic_hip}; echo ${t:-_product}; }

if [ "${TRIP_SUITE:-0}" = "1" ]; then
  echo "######## suite"
  timeout ${TRIP_SUITE_TIMEOUT:-1500} python -m pytest ${TRIP_TESTS:-tests} -m gpu -q --durations=8 ${TRIP_K:+-k "$TRIP_K"} > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
  grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -30
fi

for lib in ${TRIP_LIBS:-}; do
  [ -f $R/fast_slic_amd/$lib ] || { echo "missing $lib"; continue; }
  export FSLIC_LIB=$R/fast_slic_amd/$lib
  tag=$(tagof $lib)
  echo "######## $lib"
  if [ -n "${TRIP_PARITY:-}" ]; then
    timeout 900 python -m pytest tests -q -m gpu -k "$TRIP_PARITY" > $OUT/pytest$tag.log 2>&1
    echo "parity subset rc=$?"; tail -2 $OUT/pytest$tag.log
  fi
  if [ -n "${TRIP_CFGS:-}" ]; then
    KSTATS_CFGS="$TRIP_CFGS" bash scripts/gpu_kstats.sh > $OUT/kstats$tag.txt 2>&1
    for f in $R/gpurun_out/kernel_stats_*.csv $R/gpurun_out/timeline_*.txt; do [ -f $f ] && mv $f $OUT/$(basename ${f%.*})$tag.${f##*.}; done
    grep -E "^==|${TRIP_KERNELS:-k_}" $OUT/kstats$tag.txt | sed 's/(fslic::[A-Za-z]*Dev[^"]*"/"/' | cut -c1-120
    cd $R
  fi
  for rep in $(seq 1 ${TRIP_STEADY_REPS:-1}); do
    [ "${TRIP_STEADY:-0}" -gt 0 ] && timeout 300 python bench.py --timed-only --steps $TRIP_STEADY 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('720p pipelined, $TRIP_STEADY steps', d['value'], 'MP/s; host CPU ms per step', d['host_cpu_ms_per_step']['per_rank'])" | tee -a $OUT/bench$tag.txt
  done
  for rep in $(seq 1 ${TRIP_DRIVER:-0}); do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver command line (20 steps)', d['value'], 'MP/s; host CPU ms per step', d['host_cpu_ms_per_step']['per_rank'])" | tee -a $OUT/bench$tag.txt
  done
  if [ "${TRIP_4K:-0}" = "1" ]; then
    timeout 300 python bench.py --workload 4k --timed-only --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K pipelined', d['value'], 'MP/s')" | tee -a $OUT/bench$tag.txt
    timeout 300 python bench.py --workload 4k6000 --timed-only --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K K=6000 pipelined', d['value'], 'MP/s')" | tee -a $OUT/bench$tag.txt
  fi
  if [ "${TRIP_PMC:-0}" = "1" ]; then
    bash scripts/gpu_pmc_group.sh > $OUT/pmc_group$tag.txt 2>&1; cd $R
    [ -f gpurun_out/pmc_group.csv ] && mv gpurun_out/pmc_group.csv $OUT/pmc_group$tag.csv
    grep -E "kernel  |${TRIP_KERNELS:-k_}|group total" $OUT/pmc_group$tag.txt | cut -c1-150
  fi
  unset FSLIC_LIB
done

for lib in ${TRIP_SELECT:-}; do
  echo "######## select stamps: $lib"
  FSLIC_LIB=$R/fast_slic_amd/$lib timeout 300 python scripts/lab_select_stamps.py ${TRIP_SELECT_ARGS:-} 2>/dev/null | tee $OUT/select_stamps$(tagof $lib).txt
done
for lib in ${TRIP_STAMPS4:-}; do
  echo "######## assign phase stamps: $lib"
  FSLIC_LIB=$R/fast_slic_amd/$lib timeout 600 python scripts/lab_assign_phases.py ${TRIP_STAMPS4_ARGS:-} 2>/dev/null | tee $OUT/assign_phases$(tagof $lib).txt | cut -c1-200
done

if [ "${TRIP_BENCH:-0}" = "1" ]; then
  echo "######## bench line (all legs)"
  timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench_full.json
fi
if [ "${TRIP_BENCH_DRIVER:-0}" = "1" ]; then
  echo "######## bench line, the driver's command"
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmdline.json 2> $OUT/bench_driver_cmdline.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_driver_cmdline.json
fi
if [ "${TRIP_OVERLAP:-0}" = "1" ]; then
  echo "######## stream overlap of the pipelined region"
  KSTATS_CFGS="720 1280 1600 2 slic 1" KSTATS_BENCH=1 bash scripts/gpu_kstats.sh > $OUT/overlap.txt 2>&1; cd $R
  for f in bench_stream_overlap.txt bench_profiled.json; do [ -f gpurun_out/$f ] && mv gpurun_out/$f $OUT/$f; done
  cat $OUT/bench_stream_overlap.txt 2>/dev/null | cut -c1-160
fi
if [ -n "${TRIP_EXTRA:-}" ]; then
  echo "######## extra"
  cd $R; eval "$TRIP_EXTRA"
fi
