#!/usr/bin/env python
"""Static instruction budget of one kernel of assign.hip by phase and basic block.

    python scripts/isa_budget.py 'k_assign_blk2_w6<16, true, 3, false>' [--blocks] [--defs=-DFOO]

Compiles fast_slic_amd/csrc/assign.hip for gfx950 with -DFSLIC_MARKS (a comment line in the assembly at every phase boundary of
assign_blk2_body; the product build emits nothing there), cuts the kernel's listing into basic blocks (labels / branches), and
prints, phase by phase, the number of vector-ALU, scalar, LDS, vector-memory, wait (s_waitcnt / s_nop) and branch instructions --
straight-line code and every loop body (a block range closed by a backward branch) separately, so that dynamic trip counts can be
applied to them (scripts/assign_trip_counts.py)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "L"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "M"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "W"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_sleep")):
        return "B"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SM"
    if op.startswith("s_"):
        return "S"
    return "?"


def listing(defs):
    out = "/tmp/isa_budget_assign.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", "-DFSLIC_MARKS",
           "-Wno-unused-value", "-o", out, os.path.join(ROOT, "fast_slic_amd", "csrc", "assign.hip")] + defs
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def main():
    want = sys.argv[1]
    show_blocks = "--blocks" in sys.argv
    defs = [a[len("--defs="):] for a in sys.argv[2:] if a.startswith("--defs=")]
    src = listing(defs)
    names = [l.split(":")[0] for l in src if l.startswith("_Z") and ":" in l]
    demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    sym = None
    for m, d in zip(names, demangled):
        if want in d:
            sym = m
            print("kernel:", d.split("(")[0])
            break
    if sym is None:
        raise SystemExit("no kernel matches %r" % want)
    i0 = next(i for i, l in enumerate(src) if l.startswith(sym + ":"))
    i1 = next(i for i in range(i0, len(src)) if src[i].startswith(".Lfunc_end"))
    # basic blocks
    blocks = []          # dict(label, mark, counts, branches)
    cur = {"label": "<entry>", "phase": None, "n": collections.Counter(), "targets": [], "ops": collections.Counter()}
    phase = "pre"
    for l in src[i0 + 1:i1]:
        s = l.strip()
        if not s or s.startswith(";;#ASM"):
            continue
        if s.startswith("; @@MARK"):
            phase = s.split()[2]
            # a mark closes the current block so that phases are separated even inside straight-line code
            blocks.append(cur)
            cur = {"label": "<mark %s>" % phase, "phase": phase, "n": collections.Counter(), "targets": [], "ops": collections.Counter()}
            continue
        if s.startswith(";") or s.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", s):
                blocks.append(cur)
                cur = {"label": s.split(":")[0], "phase": phase, "n": collections.Counter(), "targets": [], "ops": collections.Counter()}
            continue
        if cur["phase"] is None:
            cur["phase"] = phase
        op = s.split()[0]
        k = classify(op)
        cur["n"][k] += 1
        cur["ops"][op] += 1
        if k == "B" and "branch" in op:
            cur["targets"].append(s.split()[1])
    blocks.append(cur)
    blocks = [b for b in blocks if sum(b["n"].values()) or b["label"].startswith("<mark")]
    index = {b["label"]: i for i, b in enumerate(blocks)}
    # loops: a backward branch from block j to block i <= j
    loops = []
    for j, b in enumerate(blocks):
        for t in b["targets"]:
            if t in index and index[t] <= j:
                loops.append((index[t], j))
    in_loop = {}
    for (a, z) in sorted(loops, key=lambda az: az[1] - az[0]):      # innermost first
        for i in range(a, z + 1):
            in_loop.setdefault(i, (a, z))
    cols = ["V", "S", "L", "M", "W", "B", "SM"]
    print("%-16s %-34s" % ("phase", "part") + "".join("%6s" % c for c in cols) + "   total")
    agg = collections.OrderedDict()
    for i, b in enumerate(blocks):
        lp = in_loop.get(i)
        part = "straight-line" if lp is None else "loop %s..%s" % (blocks[lp[0]]["label"], blocks[lp[1]]["label"])
        key = (blocks[lp[0]]["phase"] if lp else b["phase"], part)
        agg.setdefault(key, collections.Counter()).update(b["n"])
        if show_blocks:
            print("    %-12s %-22s %-20s" % (b["phase"], b["label"], part[:20]) + "".join("%6d" % b["n"][c] for c in cols) + "   " +
                  " ".join("%s:%d" % kv for kv in b["ops"].most_common(6)))
    tot = collections.Counter()
    for (ph, part), n in agg.items():
        tot.update(n)
        print("%-16s %-34s" % (ph, part) + "".join("%6d" % n[c] for c in cols) + "  %6d" % sum(n.values()))
    print("%-16s %-34s" % ("TOTAL (static)", "") + "".join("%6d" % tot[c] for c in cols) + "  %6d" % sum(tot.values()))


if __name__ == "__main__":
    main()
