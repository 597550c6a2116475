// device_common.h -- helpers shared by the gfx950 kernel translation units (lab.hip, clusters.hip, assign.hip, cca.hip).
// Internal to the library.
#pragma once
#include "kernels.h"
#include <cstdlib>

namespace fslic {

#define LANE() (threadIdx.x & 63)

// Kernels index the double-buffered members with the constants 0 (this pass) and 1 (the other parity): a
// run-time index into the by-value FrameDev argument would push the whole struct into scratch memory once
// select() has modified it.  The launchers rotate the arrays accordingly.
static inline FrameDev rotated(const FrameDev& f, int buf, int sbuf) {
    FrameDev g = f;
    g.cell_cnt[0] = f.cell_cnt[buf]; g.cell_cnt[1] = f.cell_cnt[buf ^ 1];
    g.cell_items[0] = f.cell_items[buf]; g.cell_items[1] = f.cell_items[buf ^ 1];
    g.ovf_cnt[0] = f.ovf_cnt[buf]; g.ovf_cnt[1] = f.ovf_cnt[buf ^ 1];
    g.ovf_items[0] = f.ovf_items[buf]; g.ovf_items[1] = f.ovf_items[buf ^ 1];
    g.sums[0] = f.sums[sbuf]; g.sums[1] = f.sums[sbuf ^ 1];
    return g;
}

static __device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Which assign blocks (64 px x BH visited rows) can hold pixels of a cluster centred at (cy, cx): the blocks its
// (2S+1)^2 window overlaps, in block columns [bx0, bx1] and block rows [by0, by1] (any == false: no visited row).
// The block kernel and the cluster pass evaluate this identically, which is what pairs writer and reader of
// FrameDev::cpart.
struct SlotGeom { int bx0, bx1, by0, by1; bool any; };
static __device__ __forceinline__ SlotGeom slot_geom(int cy, int cx, int S, int H, int W, int rem, int stride, int BH, int Hv) {
    SlotGeom g;
    g.bx0 = max(cx - S, 0) >> 6;
    g.bx1 = min(cx + S, W - 1) >> 6;
    const int t = max(cy - S, 0) - rem;
    const int v0 = t <= 0 ? 0 : (t + stride - 1) / stride;
    const int u = min(cy + S, H - 1) - rem;
    const int v1 = u < 0 ? -1 : min(u / stride, Hv - 1);
    g.any = v0 <= v1;
    g.by0 = v0 / BH;
    g.by1 = g.any ? v1 / BH : -1;
    return g;
}

}  // namespace fslic
