// clusters.hip -- cluster pass: centroid finalisation and coarse binning (src/context.cpp:128-135, 356-373)
// Hand-written gfx950 (CDNA4, wave64) kernels of the SLIC hot path; see kernels.h for the launch interface and
// DESIGN.md for the design.  All arithmetic on this path is integer, so results are bit-identical to the reference.
#include "device_common.h"

namespace fslic {

// =============================================================================================
// Cluster pass (K threads): centroid finalisation + coarse binning.
//   mode 0: colour resample at the centre, src/context.cpp:128-135
//   mode 1: integer mean of the accumulated sums, src/context.cpp:356-373 (round_int,
//           src/fast-slic-common.h:62-65), num_members written even when it is 0 (:362-364)
// Then every cluster record {yx, lab, phase<<16|k} is appended to the coarse cell containing
// its centre (spill list when a cell is full).  The other parity's counters are cleared for the
// next iteration, so no per-iteration memset launch is needed.
// =============================================================================================
// Sums of cluster k over the pass: the block kernel's partial sums (one entry per overlapping block) plus whatever
// went through global atomics.  out[0..5] = count, sum y, sum x, sum L, sum a, sum b.
// The entries of a cluster sit at (block row mod NY, block column mod nbx) of its NY x nbx entry grid (kernels.h, FrameDev::nbx), so
// their addresses do not depend on where the cluster is: they are requested TOGETHER with the cluster's position, colour and
// neighbours -- one memory round trip for everything this pass reads -- and which of them are live (written by a block of this
// pass) is decided from the position when both have arrived.  (Until round 5 the entries were addressed relative to the first
// block the window overlaps: position -> geometry -> entries, a dependent round trip more on a launch that is nothing but a chain of
// them and sits between every two assign passes.)
template <int NB> struct EntryLoads { uint2 v[NB][3]; };
template <int NB> static __device__ __forceinline__ void request_entries(const FrameDev& f, int k, uint32_t e0, uint32_t ne, EntryLoads<NB>& L) {
    const char* cp = reinterpret_cast<const char*>(f.cpart + (size_t)k * f.cpart_ns * 6);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t e = min(e0 + (uint32_t)u, ne - 1u);        // (past the grid: re-read its last entry, masked out below)
        const uint2* p = reinterpret_cast<const uint2*>(cp + __umul24(e, 24u));
        L.v[u][0] = p[0]; L.v[u][1] = p[1]; L.v[u][2] = p[2];
    }
}
// the entry of grid position e is live when its block lies inside the window's block range (offsets modulo the grid)
static __device__ __forceinline__ bool entry_live(const FrameDev& f, const SlotGeom& g, uint32_t e, uint32_t ne, uint32_t ymask) {
    const uint32_t nbx = (uint32_t)f.nbx, xsh = 31u - (uint32_t)__clz((int)nbx);
    const uint32_t ry = ((e >> xsh) - (uint32_t)g.by0) & ymask, rx = (e - (uint32_t)g.bx0) & (nbx - 1u);
    return e < ne && ry <= (uint32_t)(g.by1 - g.by0) && rx <= (uint32_t)(g.bx1 - g.bx0);
}
template <int NB> static __device__ __forceinline__ void request_live_entries(const FrameDev& f, const SlotGeom& g, bool take, int k, uint32_t e0, uint32_t ne, uint32_t ymask, EntryLoads<NB>& L) {
    const char* cp = reinterpret_cast<const char*>(f.cpart + (size_t)k * f.cpart_ns * 6);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const uint32_t e = e0 + (uint32_t)u;
        L.v[u][0] = L.v[u][1] = L.v[u][2] = make_uint2(0u, 0u);
        if (take && entry_live(f, g, e, ne, ymask)) {
            const uint2* p = reinterpret_cast<const uint2*>(cp + __umul24(e, 24u));
            L.v[u][0] = p[0]; L.v[u][1] = p[1]; L.v[u][2] = p[2];
        }
    }
}
template <int NB> static __device__ __forceinline__ void add_live_entries(const FrameDev& f, const SlotGeom& g, bool take, uint32_t e0, uint32_t ne, uint32_t ymask, const EntryLoads<NB>& L, uint32_t* out) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const bool livee = take && entry_live(f, g, e0 + (uint32_t)u, ne, ymask);
        out[0] += livee ? L.v[u][0].x : 0u; out[1] += livee ? L.v[u][0].y : 0u; out[2] += livee ? L.v[u][1].x : 0u;
        out[3] += livee ? L.v[u][1].y : 0u; out[4] += livee ? L.v[u][2].x : 0u; out[5] += livee ? L.v[u][2].y : 0u;
    }
}

template <int MODE, int NB>
static __device__ __forceinline__ void bin_clusters_body(FrameDev& f, const PassGeom& pg, const uint32_t* __restrict__ src_yx, size_t src_stride) {
    f.select(blockIdx.y);
    // mode 0: the centres of a group start are read where the host staged them (pinned memory), no upload command
    const uint32_t* __restrict__ yx_in = (MODE == 0 && src_yx) ? src_yx + (size_t)blockIdx.y * src_stride : f.cl_yx;
    // mode 1: consecutive blocks overlap by one cluster: thread 0 of a block only computes the member count of the
    // cluster before the block's first one (the shadow test below needs the predecessor's count)
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi < f.ncell) f.cell_cnt[1][gi] = 0;
    if (gi == 0) *f.ovf_cnt[1] = 0;
    const int k = MODE == 0 ? gi : (int)(blockIdx.x * (blockDim.x - 1) + threadIdx.x) - 1;
    const bool live = k >= 0 && k < f.K;
    uint32_t yx = live ? yx_in[k] : 0u;
    int y = (int)(yx >> 16), x = (int)(yx & 0xFFFF);
    uint32_t lab;
    // A cluster whose integer state (y, x, L, a, b) equals that of cluster k-1 has the same distance to every
    // pixel and the same phase, and k-1 is visited first: it can never win a pixel.  Such runs come from
    // initialize_clusters piling its leftover seeds on the image centre (src/context.cpp:80-86; 225 of 6000
    // at 4K) and stay identical for ever because they never receive members.  They are kept out of the bins
    // (exact), which keeps every candidate list short.  Detected only where it is race-free: both clusters
    // unchanged by this pass (mode 1), or equal uploaded centres (mode 0).
    bool shadowed = false;
    if (MODE == 0) {
        if (!live) return;
        lab = f.lab[(size_t)y * f.W + x] & 0x00FFFFFFu;
        f.cl_lab[k] = lab;
        f.cl_yx_out[k] = yx;
        if (src_yx) f.cl_yx[k] = yx;      // device copy of the staged centres (the LSC seeding reads it)
        if (f.cl_f) {      // 'noq': the centre keeps its (clamped) float position, the colour starts as the pixel's (src/context.cpp:128-135)
            float* c = f.cl_f + 8 * (size_t)k;
            c[2] = (float)(lab & 255u); c[3] = (float)((lab >> 8) & 255u); c[4] = (float)((lab >> 16) & 255u);
        }
        shadowed = !f.cl_f && k > 0 && yx_in[k - 1] == yx;
        f.cl_shadow[k] = shadowed ? 1u : 0u;
    } else {
        // everything the pass reads, requested in one go (see request_entries)
        const int kc = min(max(k, 0), f.K - 1);              // (dead lanes re-read a valid cluster: no predicated loads)
        const uint32_t lab_own = f.cl_lab[kc];
        const uint32_t yx_prev = yx_in[max(kc - 1, 0)], lab_prev = f.cl_lab[max(kc - 1, 0)];
        const uint32_t shadow_word = f.cl_shadow[kc];        // (loaded whatever the pass geometry: a load inside a branch is waited for at the branch's end)
        const uint2* const sp = reinterpret_cast<const uint2*>(f.sums[0] + 6 * (size_t)kc);
        const uint2 s01 = sp[0], s23 = sp[1], s45 = sp[2];
        const uint32_t ymask = pg.use_slots ? cpart_rows_mask(f.S, pg.stride, max(pg.BH, 1)) : 0u, ne = pg.use_slots ? (ymask + 1u) * (uint32_t)f.nbx : 0u;
        // NB == 4 (grids of four entries: 1280x720 K=1600): the entries are requested before the position is known (above).  Larger
        // grids (NB == 8): only the LIVE entries are requested, after the geometry -- measured both ways in round 5
        // (profiles/r05_experiments.txt, section `Cluster pass k_bin_clusters<1>`): requesting a whole grid of eight costs more than the dependent round trip saves (the pass
        // is bound by its scattered 24-byte loads, not by the length of its chain: 7.25 -> 7.89 us at 3840x2160 K=6000), a grid of
        // four gains a little (5.38 -> 5.14 us for 2 x 8 frames of 1280x720).
        EntryLoads<NB> L;
        if (NB == 4 && pg.use_slots) request_entries<NB>(f, kc, 0u, ne, L);
        // a shadowed cluster is in no candidate list: it received nothing and its entries are stale
        const bool take = live && !(pg.use_slots && shadow_word);
        uint32_t s[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        if (take) { s[0] = s01.x; s[1] = s01.y; s[2] = s23.x; s[3] = s23.y; s[4] = s45.x; s[5] = s45.y; }
        if (pg.use_slots) {
            const SlotGeom g = slot_geom((int)(yx >> 16), (int)(yx & 0xFFFFu), f.S, f.H, f.W, pg.rem, pg.stride, pg.BH, pg.Hv);
            if (NB == 4) {
                add_live_entries<NB>(f, g, take && g.any, 0u, ne, ymask, L, s);
            } else {
                for (uint32_t e0 = 0u; e0 < ne; e0 += (uint32_t)NB) {
                    request_live_entries<NB>(f, g, take && g.any, kc, e0, ne, ymask, L);
                    add_live_entries<NB>(f, g, take && g.any, e0, ne, ymask, L, s);
                }
            }
        }
        // member count of cluster k-1 (needed below): the previous thread's
        __shared__ uint32_t s_n[256];
        s_n[threadIdx.x] = s[0];
        __syncthreads();
        if (!live || threadIdx.x == 0) return;
        const uint32_t n_prev = s_n[threadIdx.x - 1];
        lab = lab_own;
        const uint32_t yx_prev_k = k > 0 ? yx_prev : 0u, lab_prev_k = k > 0 ? lab_prev : 0u;
        uint32_t* z0 = f.sums[1] + 6 * (size_t)k;    // the next pass accumulates here (sums[0] is still being read by neighbours)
        const int32_t n = (int32_t)s[0];
        f.cl_n[k] = (uint32_t)n;
        if (n != 0) {
            // (sums of non-negative terms below 2^31: the unsigned quotients through one reciprocal are the reference's signed ones)
            const uint32_t h = (uint32_t)n >> 1, rcp = 0xFFFFFFFFu / (uint32_t)n;
            y = (int)div_by(s[1] + h, (uint32_t)n, rcp);
            x = (int)div_by(s[2] + h, (uint32_t)n, rcp);
            // A centroid of in-image pixels is in the image, so this is a no-op on valid sums.  It only matters for a
            // frame whose tiled attempt overflowed a candidate list (some partial sums were then never written and
            // the frame is about to be redone by the generic path): positions must stay addressable meanwhile.
            y = min(max(y, 0), f.H - 1);
            x = min(max(x, 0), f.W - 1);
            const int32_t L = (int32_t)div_by(s[3] + h, (uint32_t)n, rcp);
            const int32_t a = (int32_t)div_by(s[4] + h, (uint32_t)n, rcp);
            const int32_t b = (int32_t)div_by(s[5] + h, (uint32_t)n, rcp);
            if (f.cl_f) {
                // centroid_quantization_enabled() == false (src/context.cpp:449-452): (float)sum / n per component
                // (src/context.cpp:374-380); bins, phases and windows use the integer parts (src/context.cpp:214-217)
                float* c = f.cl_f + 8 * (size_t)k;
                const float fn = (float)n;
                const float fy = (float)(int32_t)s[1] / fn, fx = (float)(int32_t)s[2] / fn;
                c[0] = fy; c[1] = fx;
                c[2] = (float)(int32_t)s[3] / fn; c[3] = (float)(int32_t)s[4] / fn; c[4] = (float)(int32_t)s[5] / fn;
                y = min(max((int)fy, 0), f.H - 1);
                x = min(max((int)fx, 0), f.W - 1);
            }
            yx = ((uint32_t)y << 16) | (uint32_t)(x & 0xFFFF);
            lab = (uint32_t)(L & 255) | ((uint32_t)(a & 255) << 8) | ((uint32_t)(b & 255) << 16);
            f.cl_lab[k] = lab;
            f.cl_moved[k] = 1;
        } else if (!f.cl_f && k > 0 && n_prev == 0u) {
            shadowed = yx_prev_k == yx && lab_prev_k == lab;
        }
        f.cl_yx_out[k] = yx;
        f.cl_shadow[k] = shadowed ? 1u : 0u;
        z0[0] = 0; z0[1] = 0; z0[2] = 0; z0[3] = 0; z0[4] = 0; z0[5] = 0;
    }
    // visit order of the reference scatter: phase = 2*((cy/T)&1) + ((cx/T)&1), then k
    const uint32_t phase = 2u * (div_by((uint32_t)y, (uint32_t)f.T, f.T_rcp) & 1u) + (div_by((uint32_t)x, (uint32_t)f.T, f.T_rcp) & 1u);
    ClusterRec rec;
    rec.yx = yx; rec.lab = lab; rec.tag = (phase << 16) | (uint32_t)k; rec.pad = f.generation();
    f.cl_tag[k] = rec.tag;
    if (shadowed) return;
    const int cell = (y >> f.cs_shift) * f.ncx + (x >> f.cs_shift);
    const uint32_t slot = atomicAdd(&f.cell_cnt[0][cell], 1u);
    if (slot < (uint32_t)kCellCap) {
        f.cell_items[0][(size_t)cell * kCellCap + slot] = rec;
    } else {
        const uint32_t o = atomicAdd(f.ovf_cnt[0], 1u);
        f.ovf_items[0][o] = rec;
    }
}

template <int MODE, int NB = 8>
__global__ __launch_bounds__(256) void k_bin_clusters(FrameDev f, PassGeom pg, const uint32_t* __restrict__ src_yx, size_t src_stride) {
    bin_clusters_body<MODE, NB>(f, pg, src_yx, src_stride);
}

// LSC: the cluster pass (blocks 0 .. nbin-1) and, in the blocks behind them, the second half of the variant's update --
// C_f = sum w*F / sum w per cluster (src/lsc.cpp:256-269 then :309-316; 0/0 = NaN for a memberless cluster exactly as the
// reference produces it) from the fixed-point accumulators the assign pass filled, which are cleared for the next pass.
// The integer sums of the shared centroid update are in FrameDev::sums, where the cluster pass reads them: the two halves
// touch different data.  16 threads per cluster: 10 features, the weight, 5 idle.
__global__ __launch_bounds__(256) void k_bin_clusters_lsc(FrameDev f, PassGeom pg, LscDev l, int nbin) {
    if ((int)blockIdx.x < nbin) {
        bin_clusters_body<1, 8>(f, pg, nullptr, (size_t)0);
        return;
    }
    l.select(blockIdx.y);
    const int t = ((int)blockIdx.x - nbin) * (int)blockDim.x + (int)threadIdx.x;
    const int k = t >> 4, q = t & 15;
    if (k >= f.K) return;
    long long* e = l.extra + (size_t)k * kLscExPitch;
    const long long mine = q <= kLscFeat ? e[q] : 0ll;
    const long long wsum = e[kLscFeat];
    const float acc = (float)((double)mine / (double)kLscFixF), tot_w = (float)((double)wsum / (double)kLscFixW);
    if (q < kLscFeat) l.cfeat[(size_t)k * kLscCfPitch + q] = acc / tot_w;
    // every thread of the cluster has read what it needs before any of them clears (same wavefront: 16 consecutive lanes)
    __builtin_amdgcn_wave_barrier();
    if (q <= kLscFeat) e[q] = 0;
}

void launch_bin_clusters(const FrameDev& f_, int nframes, int mode, int buf, int sbuf, const PassGeom& pg, hipStream_t st,
                         const uint32_t* src_yx, size_t src_stride) {
    const FrameDev f = rotated(f_, buf, sbuf);
    if (mode == 0) {
        const int n = f.K > f.ncell ? f.K : f.ncell;
        launch((k_bin_clusters<0, 8>), dim3((n + 255) / 256, nframes), dim3(256), 0, st, f, pg, src_yx, src_stride);
    } else {
        const int bk = (f.K + 254) / 255, bc = (f.ncell + 255) / 256;    // 255 clusters per block (see the kernel)
        // entries per cluster in this pass's grid (device_common.h, cpart_rows_mask, on the host)
        int ne = 0;
        if (pg.use_slots) {
            const int n = (2 * f.S / pg.stride) / pg.BH + 2;
            int ny = 1;
            while (ny < n) ny <<= 1;
            ne = ny * f.nbx;
        }
        if (ne > 0 && ne <= 4) launch((k_bin_clusters<1, 4>), dim3(bk > bc ? bk : bc, nframes), dim3(256), 0, st, f, pg, nullptr, (size_t)0);
        else launch((k_bin_clusters<1, 8>), dim3(bk > bc ? bk : bc, nframes), dim3(256), 0, st, f, pg, nullptr, (size_t)0);
    }
}

void launch_bin_clusters_lsc(const FrameDev& f_, const LscDev& l, int nframes, int buf, int sbuf, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int bk = (f.K + 254) / 255, bc = (f.ncell + 255) / 256;
    const int nbin = bk > bc ? bk : bc;
    launch(k_bin_clusters_lsc, dim3(nbin + (f.K * 16 + 255) / 256, nframes), dim3(256), 0, st, f, PassGeom{}, l, nbin);
}

}  // namespace fslic
