#!/usr/bin/env python
"""Dynamic trip counts of the block assign kernel's loops, computed on the CPU from the oracle's state:
    python scripts/assign_trip_counts.py H W K [iters_before=6] [R=16] [stride=3] [rem=0]
The oracle (test infrastructure; this is a measurement script, not the product) runs `iters_before` iterations of the frame
synth.variant("A", H, W); its cluster centres are what the next subsampled pass would find in the bins, and its pre-connectivity
label map stands in for that pass's labels.  The geometry below restates assign_blk2_body (fast_slic_amd/csrc/assign.hip): block =
64 columns x 4 wavefronts x R visited rows, the block's candidate list (window tests of the gather), the wavefront's `todo` mask,
the quarter coverage of the 2-D table walk (R = 16) or the half walks, and the row gaps in which some lane changes its label.
Prints means per wavefront; scripts/isa_budget.py supplies the static instruction counts these multiply."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fast_slic_amd.synth import variant      # noqa: E402
from oracle import oracle as orc             # noqa: E402


def main():
    H, W, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    it0 = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    R = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    stride = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    rem = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    img = variant("A", H, W)
    cl0 = orc.initialize_clusters(img, K)
    _, cl, _, pre = orc.slic_iterate(img, cl0, max_iter=it0, stages=True)
    S = orc.S_of(H, W, K)
    cy = cl["y"].astype(np.int64)
    cx = cl["x"].astype(np.int64)
    # duplicates on one pixel with identical colour are shadowed out of the bins by the cluster pass: drop exact duplicates of (y, x, colour)
    key = np.stack([cy, cx, cl["r"].astype(np.int64), cl["g"].astype(np.int64), cl["b"].astype(np.int64)], 1)
    _, first = np.unique(key, axis=0, return_index=True)
    keep = np.zeros(K, bool)
    keep[first] = True
    cy, cx = cy[keep], cx[keep]
    Hv = (H - rem + stride - 1) // stride
    BH = 4 * R
    nbx, nby = (W + 63) // 64, (Hv + BH - 1) // BH
    rows = rem + stride * np.arange(Hv)
    ys_all = np.concatenate([rows, rows[-1] + stride * np.arange(1, 4 * 64)])      # (rows past the bottom edge: never used for full wavefronts)
    lab_rows = pre[rows]                       # labels of the visited rows
    acc = {k: 0.0 for k in ("waves", "bn", "todo", "q1", "q2", "q3", "q4", "pairs", "gaps", "fast", "half1", "half2", "covered_px_cand", "change_lanes")}
    for by in range(nby):
        v0b = by * BH
        vlast = min(v0b + BH - 1, Hv - 1)
        wy_lo, wy_hi = rem + v0b * stride - S, rem + vlast * stride + S
        in_y = (cy >= wy_lo) & (cy <= wy_hi)
        for bx in range(nbx):
            x0 = bx * 64
            wx_lo, wx_hi = x0 - S, min(x0 + 63, W - 1) + S
            blk = in_y & (cx >= wx_lo) & (cx <= wx_hi)
            bn = int(blk.sum())
            bcy = cy[blk]
            bcx = cx[blk]
            for w in range(4):
                v0 = v0b + w * R
                nrows = max(0, min(R, Hv - v0))
                if nrows == 0:
                    continue
                acc["waves"] += 1
                acc["bn"] += bn
                yw_lo = rem + v0 * stride
                yw_hi = yw_lo + (nrows - 1) * stride
                todo = (bcy >= yw_lo - S) & (bcy <= yw_hi + S)
                acc["todo"] += int(todo.sum())
                tcy = bcy[todo]
                tcx = bcx[todo]
                if R == 16:
                    cov = []
                    for q in range(4):
                        ylo = yw_lo + q * 4 * stride
                        yhi = ylo + 3 * stride
                        cov.append((tcy >= ylo - S) & (tcy <= yhi + S))
                    nq = sum(c.astype(int) for c in cov)
                    # a window is an interval: the quarters it reaches are contiguous
                    for n in (1, 2, 3, 4):
                        acc["q%d" % n] += int((nq == n).sum())
                    acc["pairs"] += int(nq.sum()) * 2          # two row pairs per quarter
                else:
                    yh = yw_lo + (R // 2) * stride
                    up = tcy <= yw_lo + (R // 2 - 1) * stride + S
                    lo = (tcy >= yh - S) & (tcy <= yh + (R // 2 - 1) * stride + S)
                    both = up & lo
                    acc["half2"] += int(both.sum())
                    acc["half1"] += int((up ^ lo).sum())
                    acc["pairs"] += int(both.sum()) * (R // 2) + int((up ^ lo).sum()) * (R // 4)
                # the row-vector form (VT, 3840x2160 K=1600): rows above the centre use table index k0 - |dx|, rows at / below it k0 + |dx|; a
                # candidate whose rows of one side are all sixteen and inside the window takes a mask-free loop, the others the masked
                # loops by halves of eight rows (assign_blk2_body, `masked`)
                if R == 16:
                    for c_y in tcy:
                        dy = ys_all[v0:v0 + R] - c_y if nrows == R else None
                        if dy is None:
                            continue
                        inwin = np.abs(dy) <= S
                        va, vb = inwin & (dy < 0), inwin & (dy >= 0)
                        for side, v in (("a", va), ("b", vb)):
                            if v.all():
                                acc["vt_whole_" + side] = acc.get("vt_whole_" + side, 0) + 1
                            elif v.any():
                                lo, hi = v[:8].any(), v[8:].any()
                                kind = "both" if (lo and hi) else ("lo" if lo else "hi")
                                acc["vt_%s_%s" % (side, kind)] = acc.get("vt_%s_%s" % (side, kind), 0) + 1
                # exact pixel-candidate pairs of the wavefront (what an ideal machine would evaluate)
                xs = np.arange(x0, min(x0 + 64, W))
                ys = rows[v0:v0 + nrows]
                ncov = (np.abs(ys[:, None] - tcy[None, :]) <= S).sum(0) * (np.abs(xs[:, None] - tcx[None, :]) <= S).sum(0)
                acc["covered_px_cand"] += int(ncov.sum())
                lr = lab_rows[v0:v0 + nrows, x0:min(x0 + 64, W)]
                chg = lr[1:] != lr[:-1]
                acc["gaps"] += int(chg.any(1).sum())
                acc["change_lanes"] += int(chg.sum())
                acc["fast"] += int(nrows == R and not (lr == 0xFFFF).any())
    n = acc["waves"]
    print("%dx%d K=%d S=%d, state after %d iterations, pass rem=%d stride=%d, R=%d rows per wavefront: %d wavefronts with rows (%d x %d blocks)" % (W, H, K, S, it0, rem, stride, R, n, nbx, nby))
    print("per wavefront: block list %.2f, todo (window reaches the wavefront's rows) %.2f" % (acc["bn"] / n, acc["todo"] / n))
    if R == 16:
        print("  walk iterations by quarters reached: 1: %.2f  2: %.2f  3: %.2f  4: %.2f   (row pairs evaluated %.1f = %.2f quarters per iteration)" % (
            acc["q1"] / n, acc["q2"] / n, acc["q3"] / n, acc["q4"] / n, acc["pairs"] / n, acc["pairs"] / 2 / max(acc["todo"], 1)))
    else:
        print("  walk iterations: both halves %.2f, one half %.2f (row pairs evaluated %.1f)" % (acc["half2"] / n, acc["half1"] / n, acc["pairs"] / n))
    if R == 16:
        keys = ["vt_whole_a", "vt_whole_b", "vt_a_lo", "vt_a_hi", "vt_a_both", "vt_b_lo", "vt_b_hi", "vt_b_both"]
        print("  row-vector form, walk iterations per wavefront (interior wavefronts): " + "  ".join("%s %.2f" % (k[3:], acc.get(k, 0) / n) for k in keys))
    print("  pixel-candidate pairs inside a window %.0f of %.0f evaluated lane-slots (%.1f %%): %.2f candidates per pixel" % (
        acc["covered_px_cand"] / n, acc["pairs"] / n * 128, 100.0 * acc["covered_px_cand"] / max(acc["pairs"] * 128, 1), acc["covered_px_cand"] / n / (64.0 * R)))
    print("  row gaps with a label change in some lane %.2f of %d; lanes changing per such gap %.1f; wavefronts on the fast epilogue %.3f" % (
        acc["gaps"] / n, R - 1, acc["change_lanes"] / max(acc["gaps"], 1), acc["fast"] / n))


if __name__ == "__main__":
    main()
