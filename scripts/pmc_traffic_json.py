"""profiles/rNN_pmc_fetch_write_*.csv (scripts/gpu_profiles.sh, step 3) -> profiles/rNN_pmc_assign_traffic.json, the per-launch
HBM traffic of the assign kernels that bench.py reports as roofline.traffic.
   python scripts/pmc_traffic_json.py [profiles/r03]"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03")
SETS = {"720p_x8": ("1280x720_x8", 1280 * 720 * 8, 3), "720p_x16": ("1280x720_x16", 1280 * 720 * 16, 3), "4k_k1600": ("3840x2160_x1", 3840 * 2160, 3)}   # tag, pixels per launch, subsample stride


def read(path):
    d = {}
    for r in csv.reader(open(path)):
        if len(r) < 4 or r[0] == "kernel":
            continue
        d[(r[0], r[1])] = float(r[2])
    return d


out = {"note": "bytes per launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 correction on the read side only); mean over the "
               "dispatches of separate rocprofv3 --pmc passes (profiles/%s_pmc_fetch_write_*.csv)" % os.path.basename(prefix)}
# which build the counters belong to: the commit of the tree the trip ran on (scripts/gpu_profiles.sh writes it next to the csv files, the GPU
# box has no .git), else this checkout's HEAD -- bench.py prints it beside roofline.traffic so that a stale figure is visible
stamp = prefix + ("" if prefix.endswith("/") else "_") + "build_commit.txt"
try:
    out["build_commit"] = open(stamp).read().strip()
except OSError:
    try:
        out["build_commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        out["build_commit"] = None
for key, (tag, px, stride) in SETS.items():
    path = "%s%spmc_fetch_write_%s.csv" % (prefix, "" if prefix.endswith("/") else "_", tag)      # (a directory prefix: files without the round tag)
    if not os.path.exists(path):
        continue
    d = read(path)
    ent = {}
    for name, fused in (("full_assign", False), ("fused_assign_update", True)):
        sel = [k for k in d if "k_assign_blk" in k[0] and (", true," in k[0]) == fused]
        f = [d[k] for k in sel if k[1] == "FETCH_SIZE"]
        w = [d[k] for k in sel if k[1] == "WRITE_SIZE"]
        if not f or not w:
            continue
        ent[name] = {"fetch_kib": round(f[0], 1), "write_kib": round(w[0], 1), "bytes": round(f[0] * 1024 * 2 + w[0] * 1024, 1),
                     "algorithmic_bytes": 6 * px // (stride if fused else 1)}
    out[key] = ent
json.dump(out, open(prefix + ("" if prefix.endswith("/") else "_") + "pmc_assign_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
