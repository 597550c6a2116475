#!/bin/bash
# The driver-sized timed region (20 steps after 5 warm-up steps, ~3.3 ms), N times with FSLIC_HOST_TIMING=1: the value of every run, and
# for the runs that fall more than 12 % below the median the host-side timeline of their groups (where did the stall sit?).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/hiccup; mkdir -p $OUT; rm -f $OUT/*
N=${1:-16}
for i in $(seq $N); do
  FSLIC_HOST_TIMING=1 python bench.py --gpus 1 --steps 20 --warmup 5 --timed-only 2>$OUT/err_$i.txt | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])" > $OUT/val_$i.txt
  echo -n "$(cat $OUT/val_$i.txt) "
done
echo
python - <<PY
import glob, re, statistics
vals = {int(re.search(r"val_(\d+)", f).group(1)): float(open(f).read()) for f in glob.glob("$OUT/val_*.txt")}
med = statistics.median(vals.values())
print("median", med, "min", min(vals.values()), "max", max(vals.values()))
for i, v in sorted(vals.items()):
    if v < 0.88 * med:
        print("---- run", i, "value", v)
        lines = [l for l in open("$OUT/err_%d.txt" % i) if l.startswith("[fslic host]")]
        # the timed region = the last 10 .. 20 groups: print the tail
        for l in lines[-44:]:
            print(l.rstrip()[:230])
PY
