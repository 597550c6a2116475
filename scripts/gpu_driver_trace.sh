#!/bin/bash
# the driver's command line under rocprofv3: what the streams do in its 3.4 ms timed region
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/p
timeout 600 rocprofv3 --kernel-trace -d $OUT/p -o b -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --timed-only > $OUT/driver_trace_bench.json 2>/dev/null
DB=$(find $OUT/p -name '*_results.db' | head -1)
cat $OUT/driver_trace_bench.json | cut -c1-200
python $R/scripts/rocpd_stats.py overlap $DB 3400 > $OUT/driver_trace_overlap.txt
python $R/scripts/rocpd_stats.py window $DB 4200 4200 | awk '{print $1, $2, $3, $5}' > $OUT/driver_trace_window.txt
cat $OUT/driver_trace_overlap.txt
python - <<PY
# per stream: first kernel start and last kernel end inside the window, group boundaries (rgb_to_lab starts)
rows=[l.split() for l in open("$OUT/driver_trace_window.txt")]
import collections
st=collections.defaultdict(list)
for t,d,s,n in rows: st[s].append((float(t),float(d),n))
for s,v in sorted(st.items()):
    labs=[t for t,d,n in v if "rgb_to_lab" in n]
    print(s, "kernels", len(v), "groups start at", [round(x) for x in labs], "last end", round(v[-1][0]+v[-1][1]))
PY
rm -rf $OUT/p
