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
static __device__ __forceinline__ void gather_sums(const FrameDev& f, const PassGeom& pg, int k, uint32_t yx, uint32_t* out) {
    const uint32_t* s = f.sums[0] + 6 * (size_t)k;
#pragma unroll
    for (int q = 0; q < 6; ++q) out[q] = s[q];
    if (!pg.use_slots) return;
    const SlotGeom g = slot_geom((int)(yx >> 16), (int)(yx & 0xFFFFu), f.S, f.H, f.W, pg.rem, pg.stride, pg.BH, pg.Hv);
    if (!g.any) return;
    const uint32_t* cp = f.cpart + (size_t)k * f.cpart_ns * 6;
    // entries (by, bx) in [by0, by1] x [bx0, bx1], eight at a time with all their loads in flight (one memory round trip
    // per eight entries instead of one per entry: this pass is a chain of dependent loads on a handful of blocks, and
    // it sits on the critical path of every iteration)
    const int nx = g.bx1 - g.bx0 + 1, ne = nx * (g.by1 - g.by0 + 1);
    // e / nx for the handful of entries of a cluster through a 16-bit reciprocal (guarded): eight general divisions per batch were a
    // sixth of this kernel's instructions, and it sits on the critical path of every iteration
    const uint32_t inv_nx = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)nx)) + 1u;
    constexpr int kBatch = 8;
    for (int e0 = 0; e0 < ne; e0 += kBatch) {
        uint2 v[kBatch][3];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int e = e0 + u;
            uint32_t ey = __umul24((uint32_t)e, inv_nx) >> 16;
            if (__umul24(ey, (uint32_t)nx) > (uint32_t)e) ey--;
            const uint32_t ex = (uint32_t)e - __umul24(ey, (uint32_t)nx);
            v[u][0] = v[u][1] = v[u][2] = make_uint2(0u, 0u);
            if (e < ne) {
                const uint2* p = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(cp) + __umul24(__umul24(ey, (uint32_t)f.nbx) + ex, 24u));   // 24-byte entries
                v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2];
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            out[0] += v[u][0].x; out[1] += v[u][0].y; out[2] += v[u][1].x; out[3] += v[u][1].y; out[4] += v[u][2].x; out[5] += v[u][2].y;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bin_clusters(FrameDev f, PassGeom pg, const uint32_t* __restrict__ src_yx, size_t src_stride) {
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
        // what the tail needs, requested before the gather so that it does not cost another round trip afterwards
        const uint32_t lab_own = live ? f.cl_lab[k] : 0u;
        const uint32_t yx_prev = (live && k > 0) ? yx_in[k - 1] : 0u, lab_prev = (live && k > 0) ? f.cl_lab[k - 1] : 0u;
        // a shadowed cluster is in no candidate list: it received nothing and its cpart entries are stale
        uint32_t s[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        if (live && !(pg.use_slots && f.cl_shadow[k])) gather_sums(f, pg, k, yx, s);
        // member count of cluster k-1 (needed below): the previous thread's
        __shared__ uint32_t s_n[256];
        s_n[threadIdx.x] = s[0];
        __syncthreads();
        if (!live || threadIdx.x == 0) return;
        const uint32_t n_prev = s_n[threadIdx.x - 1];
        lab = lab_own;
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
            shadowed = yx_prev == yx && lab_prev == lab;
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

void launch_bin_clusters(const FrameDev& f_, int nframes, int mode, int buf, int sbuf, const PassGeom& pg, hipStream_t st,
                         const uint32_t* src_yx, size_t src_stride) {
    const FrameDev f = rotated(f_, buf, sbuf);
    if (mode == 0) {
        const int n = f.K > f.ncell ? f.K : f.ncell;
        launch(k_bin_clusters<0>, dim3((n + 255) / 256, nframes), dim3(256), 0, st, f, pg, src_yx, src_stride);
    } else {
        const int bk = (f.K + 254) / 255, bc = (f.ncell + 255) / 256;    // 255 clusters per block (see the kernel)
        launch(k_bin_clusters<1>, dim3(bk > bc ? bk : bc, nframes), dim3(256), 0, st, f, pg, nullptr, (size_t)0);
    }
}

}  // namespace fslic
