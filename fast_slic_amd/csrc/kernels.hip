// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the SLIC hot path.
//
// Nothing here is a translation of the reference's AVX2/NEON code.  The reference *scatters*:
// each cluster min-updates a (2S+1)^2 window, clusters visited in (phase, k) order
// (src/context.cpp:199-298, src/arch/x64/avx2.h:89-187).  These kernels *gather*: every pixel
// takes the lexicographic minimum of (distance, phase, k) over the clusters whose window covers
// it, which is the same function of the inputs and needs no min_dists plane in memory.
//
// All arithmetic on this path is integer, so the results are bit-identical to the reference.
#include "kernels.h"
#include <cstdlib>

namespace fslic {

#define LANE() (threadIdx.x & 63)

// Kernels index the double-buffered members with the constants 0 (this pass) and 1 (the other parity): a
// run-time index into the by-value FrameDev argument would push the whole struct into scratch memory once
// select() has modified it.  The launchers rotate the arrays accordingly.
static FrameDev rotated(const FrameDev& f, int buf, int sbuf) {
    FrameDev g = f;
    g.cell_cnt[0] = f.cell_cnt[buf]; g.cell_cnt[1] = f.cell_cnt[buf ^ 1];
    g.cell_items[0] = f.cell_items[buf]; g.cell_items[1] = f.cell_items[buf ^ 1];
    g.ovf_cnt[0] = f.ovf_cnt[buf]; g.ovf_cnt[1] = f.ovf_cnt[buf ^ 1];
    g.ovf_items[0] = f.ovf_items[buf]; g.ovf_items[1] = f.ovf_items[buf ^ 1];
    g.sums[0] = f.sums[sbuf]; g.sums[1] = f.sums[sbuf ^ 1];
    return g;
}

static __device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// =============================================================================================
// RGB -> CIELAB prepass.  Restates FastCIELabCvt::convert (src/cielab.h:309-325) with both lookup
// tables staged in LDS (16.5 KB); 4 pixels per thread: 3 dword loads in, one 16-byte store out.
// Algorithmic traffic: 3 B read + 4 B written per pixel.
// =============================================================================================
struct LabConst { int cb[9]; };

static __device__ __forceinline__ uint32_t lab_px(uint32_t R, uint32_t G, uint32_t B,
                                                  const uint16_t* sg, const uint16_t* sl, const LabConst& k) {
    const int sr = sg[R], sgv = sg[G], sb = sg[B];
    const int xr = (k.cb[0] * sr + k.cb[1] * sgv + k.cb[2] * sb) >> 16;
    const int yr = (k.cb[3] * sr + k.cb[4] * sgv + k.cb[5] * sb) >> 16;
    const int zr = (k.cb[6] * sr + k.cb[7] * sgv + k.cb[8] * sb) >> 16;
    const int fx = sl[xr], fy = sl[yr], fz = sl[zr];
    const int ciel = 116 * fy - (16 << 13);
    const int ciea = 500 * (fx - fy) + (128 << 13);
    const int cieb = 200 * (fy - fz) + (128 << 13);
    int l = (int)((unsigned)ciel >> 12);
    int a = (int)((unsigned)ciea >> 12) - 128;
    int b = (int)((unsigned)cieb >> 12) - 128;
    l = min(max(l, 0), 255);
    a = min(max(a, 0), 255);
    b = min(max(b, 0), 255);
    return (uint32_t)l | ((uint32_t)a << 8) | ((uint32_t)b << 16);
}

template <bool CONVERT>
__global__ __launch_bounds__(1024) void k_rgb_to_lab(FrameDev f, const uint16_t* __restrict__ g_gamma,
                                                     const uint16_t* __restrict__ g_labtbl, LabConst kc, int init_labels) {
    __shared__ uint16_t s_gamma[256];
    __shared__ uint16_t s_lab[8194];
    f.select(blockIdx.y);
    const uint8_t* __restrict__ rgb = f.rgb;
    uint32_t* __restrict__ lab = f.lab;
    uint16_t* __restrict__ labels_init = init_labels ? f.labels : nullptr;
    const int N = f.N;
    const int aligned4 = ((reinterpret_cast<uintptr_t>(rgb) & 3) == 0) ? 1 : 0;
    if (CONVERT) {
        for (int i = threadIdx.x; i < 128; i += blockDim.x)
            reinterpret_cast<uint32_t*>(s_gamma)[i] = reinterpret_cast<const uint32_t*>(g_gamma)[i];
        for (int i = threadIdx.x; i < 4097; i += blockDim.x)   // 8194 entries (last one padding)
            reinterpret_cast<uint32_t*>(s_lab)[i] = reinterpret_cast<const uint32_t*>(g_labtbl)[i];
        __syncthreads();
    }
    const int nquad = aligned4 ? (N >> 2) : 0;
    const int gstride = gridDim.x * blockDim.x;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += gstride) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb) + 3 * (size_t)q;
        const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
        uint32_t r0 = w0 & 255, g0 = (w0 >> 8) & 255, b0 = (w0 >> 16) & 255;
        uint32_t r1 = w0 >> 24, g1 = w1 & 255, b1 = (w1 >> 8) & 255;
        uint32_t r2 = (w1 >> 16) & 255, g2 = w1 >> 24, b2 = w2 & 255;
        uint32_t r3 = (w2 >> 8) & 255, g3 = (w2 >> 16) & 255, b3 = w2 >> 24;
        uint4 o;
        if (CONVERT) {
            o.x = lab_px(r0, g0, b0, s_gamma, s_lab, kc);
            o.y = lab_px(r1, g1, b1, s_gamma, s_lab, kc);
            o.z = lab_px(r2, g2, b2, s_gamma, s_lab, kc);
            o.w = lab_px(r3, g3, b3, s_gamma, s_lab, kc);
        } else {   // raw copy, src/context.cpp:117-127
            o.x = r0 | (g0 << 8) | (b0 << 16);
            o.y = r1 | (g1 << 8) | (b1 << 16);
            o.z = r2 | (g2 << 8) | (b2 << 16);
            o.w = r3 | (g3 << 8) | (b3 << 16);
        }
        reinterpret_cast<uint4*>(lab)[q] = o;
        // assignment plane <- 0xFFFF once per call (src/context.cpp:138-145)
        if (labels_init) reinterpret_cast<uint2*>(labels_init)[q] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    // tail (and the whole image when the source pointer is not 4-byte aligned)
    for (int p = (nquad << 2) + blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gstride) {
        const uint32_t r = rgb[3 * (size_t)p], g = rgb[3 * (size_t)p + 1], b = rgb[3 * (size_t)p + 2];
        lab[p] = CONVERT ? lab_px(r, g, b, s_gamma, s_lab, kc) : (r | (g << 8) | (b << 16));
        if (labels_init) labels_init[p] = 0xFFFFu;
    }
}

void launch_rgb_to_lab(const FrameDev& f, int nframes, int convert, const LabTables& t, bool init_labels, hipStream_t st) {
    LabConst kc;
    for (int i = 0; i < 9; i++) kc.cb[i] = t.cb[i];
    int blocks = (f.N / 4 + 1023) / 1024;
    const int cap = (512 + nframes - 1) / nframes;   // about two resident blocks per CU over the whole group, grid-stride
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (convert)
        hipLaunchKernelGGL(k_rgb_to_lab<true>, dim3(blocks, nframes), dim3(1024), 0, st, f, t.gamma, t.labtbl, kc, init_labels ? 1 : 0);
    else
        hipLaunchKernelGGL(k_rgb_to_lab<false>, dim3(blocks, nframes), dim3(1024), 0, st, f, t.gamma, t.labtbl, kc, init_labels ? 1 : 0);
}

// =============================================================================================
// Cluster pass (K threads): centroid finalisation + coarse binning.
//   mode 0: colour resample at the centre, src/context.cpp:128-135
//   mode 1: integer mean of the accumulated sums, src/context.cpp:356-373 (round_int,
//           src/fast-slic-common.h:62-65), num_members written even when it is 0 (:362-364)
// Then every cluster record {yx, lab, phase<<16|k} is appended to the coarse cell containing
// its centre (spill list when a cell is full).  The other parity's counters are cleared for the
// next iteration, so no per-iteration memset launch is needed.
// =============================================================================================
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
    // entries (by, bx) in [by0, by1] x [bx0, bx1], four at a time with all their loads in flight (one memory round trip
    // per four entries instead of one per entry: this pass is a chain of dependent loads on a handful of blocks)
    const int nx = g.bx1 - g.bx0 + 1, ne = nx * (g.by1 - g.by0 + 1);
    for (int e0 = 0; e0 < ne; e0 += 4) {
        uint2 v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u;
            const int ey = e / nx, ex = e - ey * nx;
            v[u][0] = v[u][1] = v[u][2] = make_uint2(0u, 0u);
            if (e < ne) {
                const uint2* p = reinterpret_cast<const uint2*>(cp + (ey * f.nbx + ex) * 6);   // 24-byte entries
                v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            out[0] += v[u][0].x; out[1] += v[u][0].y; out[2] += v[u][1].x; out[3] += v[u][1].y; out[4] += v[u][2].x; out[5] += v[u][2].y;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bin_clusters(FrameDev f, PassGeom pg) {
    f.select(blockIdx.y);
    // mode 1: consecutive blocks overlap by one cluster: thread 0 of a block only computes the member count of the
    // cluster before the block's first one (the shadow test below needs the predecessor's count)
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi < f.ncell) f.cell_cnt[1][gi] = 0;
    if (gi == 0) *f.ovf_cnt[1] = 0;
    const int k = MODE == 0 ? gi : (int)(blockIdx.x * (blockDim.x - 1) + threadIdx.x) - 1;
    const bool live = k >= 0 && k < f.K;
    uint32_t yx = live ? f.cl_yx[k] : 0u;
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
        shadowed = k > 0 && f.cl_yx[k - 1] == yx;
        f.cl_shadow[k] = shadowed ? 1u : 0u;
    } else {
        // a shadowed cluster is in no candidate list: it received nothing and its cpart entries are stale
        uint32_t s[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        if (live && !(pg.use_slots && f.cl_shadow[k])) gather_sums(f, pg, k, yx, s);
        // member count of cluster k-1 (needed below): the previous thread's
        __shared__ uint32_t s_n[256];
        s_n[threadIdx.x] = s[0];
        __syncthreads();
        if (!live || threadIdx.x == 0) return;
        const uint32_t n_prev = s_n[threadIdx.x - 1];
        lab = f.cl_lab[k];
        uint32_t* z0 = f.sums[1] + 6 * (size_t)k;    // the next pass accumulates here (sums[0] is still being read by neighbours)
        const int32_t n = (int32_t)s[0];
        f.cl_n[k] = (uint32_t)n;
        if (n != 0) {
            const int32_t h = n / 2;
            y = ((int32_t)s[1] + h) / n;
            x = ((int32_t)s[2] + h) / n;
            // A centroid of in-image pixels is in the image, so this is a no-op on valid sums.  It only matters for a
            // frame whose tiled attempt overflowed a candidate list (some partial sums were then never written and
            // the frame is about to be redone by the generic path): positions must stay addressable meanwhile.
            y = min(max(y, 0), f.H - 1);
            x = min(max(x, 0), f.W - 1);
            const int32_t L = ((int32_t)s[3] + h) / n;
            const int32_t a = ((int32_t)s[4] + h) / n;
            const int32_t b = ((int32_t)s[5] + h) / n;
            yx = ((uint32_t)y << 16) | (uint32_t)(x & 0xFFFF);
            lab = (uint32_t)(L & 255) | ((uint32_t)(a & 255) << 8) | ((uint32_t)(b & 255) << 16);
            f.cl_lab[k] = lab;
            f.cl_moved[k] = 1;
        } else if (k > 0 && n_prev == 0u) {
            shadowed = f.cl_yx[k - 1] == yx && f.cl_lab[k - 1] == lab;
        }
        f.cl_yx_out[k] = yx;
        f.cl_shadow[k] = shadowed ? 1u : 0u;
        z0[0] = 0; z0[1] = 0; z0[2] = 0; z0[3] = 0; z0[4] = 0; z0[5] = 0;
    }
    // visit order of the reference scatter: phase = 2*((cy/T)&1) + ((cx/T)&1), then k
    const uint32_t phase = (uint32_t)(2 * ((y / f.T) & 1) + ((x / f.T) & 1));
    ClusterRec rec;
    rec.yx = yx; rec.lab = lab; rec.tag = (phase << 16) | (uint32_t)k; rec.pad = f.gen;
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

void launch_bin_clusters(const FrameDev& f_, int nframes, int mode, int buf, int sbuf, const PassGeom& pg, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    if (mode == 0) {
        const int n = f.K > f.ncell ? f.K : f.ncell;
        hipLaunchKernelGGL(k_bin_clusters<0>, dim3((n + 255) / 256, nframes), dim3(256), 0, st, f, pg);
    } else {
        const int bk = (f.K + 254) / 255, bc = (f.ncell + 255) / 256;    // 255 clusters per block (see the kernel)
        hipLaunchKernelGGL(k_bin_clusters<1>, dim3(bk > bc ? bk : bc, nframes), dim3(256), 0, st, f, pg);
    }
}

// =============================================================================================
// Assign kernel (the roofline kernel).  One WAVEFRONT = one tile of 64 columns x R visited rows; the four
// wavefronts of a block are independent (no barrier, no shared list), so they drift apart in time and the
// loads of one overlap the argmin of another.
// Lane = column, so |x - cx| and the x window test are per-lane-per-candidate, while |y - cy| and the y window
// test are wave-uniform per row and live in SGPRs.  Per pixel-candidate the VALU does
//     a   = A + B                      (spatial term, 16 fractional bits, window folded in as kHuge)
//     w   = v_sad_hi_u8(pix, clab, a)  ((|dL|+|da|+|db|) << 16) + a
//     key = v_and_or(w, 0xffff0000, idx)   idx = rank of the candidate in (phase, k) order
//     best = min(best, key)
// so `best` ends up as the lexicographic minimum of (distance, phase, k).  Pixels that no window covers are
// not written: the previous label persists (src/context.cpp:138-145 clears the assignment plane once per
// call, not per iteration).
// Candidate list per wavefront: every slot of every coarse cell the tile's pixels can see (a slot is live
// when its generation stamp equals this pass's: one round trip fetches count and payload) plus the spill
// list; ballot compaction into 64 LDS slots; exact prune of clusters with identical (y,x,L,a,b) to an
// earlier-ordered one (same distance everywhere, loses every tie); rank by (phase,k) with readlane
// broadcasts.  More than 64 live candidates (a pile-up of seeds on one pixel): the records are streamed
// again and compared unsorted with a two-register (distance, phase<<16|k) key; no capacity limit.
// Algorithmic traffic: 4 B read + 2 B written per visited pixel; min-distances never leave VGPRs.
// With FUSE the centroid sums of update() (src/context.cpp:301-354) for the same rows are accumulated in a
// wave-private LDS region (privatised copies) and flushed with one global atomic per (candidate, quantity).
// =============================================================================================
static __device__ __forceinline__ void global_accumulate(uint32_t* sums, uint32_t k, uint32_t y, uint32_t x, uint32_t pix) {
    uint32_t* s = sums + 6 * (size_t)k;
    atomicAdd(s + 0, 1u);
    atomicAdd(s + 1, y);
    atomicAdd(s + 2, x);
    atomicAdd(s + 3, pix & 255u);
    atomicAdd(s + 4, (pix >> 8) & 255u);
    atomicAdd(s + 5, (pix >> 16) & 255u);
}

struct TileWindow { int wy_lo, wy_hi, wx_lo, wx_hi; };
static __device__ __forceinline__ bool in_window(const ClusterRec& r, const TileWindow& w) {
    const int cy = (int)(r.yx >> 16), cx = (int)(r.yx & 0xFFFF);
    return cy >= w.wy_lo && cy <= w.wy_hi && cx >= w.wx_lo && cx <= w.wx_hi;
}

template <int R, bool FUSE, bool LUT>
__global__ __launch_bounds__(256) void k_assign(FrameDev f, int rem, int stride, int Hv, int buf, int sbuf) {
    extern __shared__ uint32_t s_lut[];
    __shared__ uint32_t s_rawbuf[kWavesPerBlock][3 * 64];
    __shared__ uint4 s_listbuf[kWavesPerBlock][64];
    __shared__ uint32_t s_accbuf[FUSE ? kWavesPerBlock * kAccSlots * 6 * kAccCopies : 1];

    f.select(blockIdx.z);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    uint32_t* const s_raw = s_rawbuf[wave];
    uint4* const s_list = s_listbuf[wave];
    uint32_t* const s_acc = s_accbuf + (FUSE ? wave * (kAccSlots * 6 * kAccCopies) : 0);
    const int W = f.W, S = f.S;
    const int x0 = blockIdx.x * kTileW;
    const int x = x0 + lane;
    const int vrow_w0 = (blockIdx.y * kWavesPerBlock + wave) * R;
    const bool okx = x < W;
#define FSLIC_STAMP(slot_)                                                                              \
    do {                                                                                                 \
        if (f.dbg && lane == 0)                                                                          \
            f.dbg[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave) * 8 + (slot_)] = __builtin_readcyclecounter(); \
    } while (0)
    if (LUT) {
        for (int i = tid; i < f.lut_words; i += 256) s_lut[i] = f.lut[i];
        __syncthreads();                     // the only barrier of the kernel
    }
    if (vrow_w0 >= Hv) return;               // wave-uniform
    FSLIC_STAMP(0);
    const int yw_lo = rem + vrow_w0 * stride;
    const int nrows = min(R, Hv - vrow_w0);
    const uint32_t novf = *f.ovf_cnt[0];   // issued early, consumed after the cell gather

    // 1. pixel loads first; they do not depend on the candidate list
    uint32_t pix[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = okx && r < nrows;
        pix[r] = ok ? f.lab[(size_t)(yw_lo + r * stride) * W + x] : 0u;
    }
    if (FUSE) {
#pragma unroll
        for (int i = 0; i < kAccSlots * 6 * kAccCopies / 64; ++i) s_acc[i * 64 + lane] = 0u;
    }

    // 2. gather
    TileWindow win;
    win.wy_lo = yw_lo - S; win.wy_hi = yw_lo + (nrows - 1) * stride + S;
    win.wx_lo = x0 - S;    win.wx_hi = min(x0 + kTileW - 1, W - 1) + S;
    const int sh = f.cs_shift;
    const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, W - 1) >> sh;
    const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
    const int ncxr = cxb - cxa + 1, ncyr = cyb - cya + 1;
    const int total = ncxr * ncyr * kCellCap;
    const uint32_t inv = (uint32_t)(4294967296.0f / (float)ncxr) + 1u;   // exact quotient for the small operands here
    const ClusterRec* __restrict__ citems = f.cell_items[0];
    const ClusterRec* __restrict__ oitems = f.ovf_items[0];
    const unsigned long long lowmask = (1ull << lane) - 1ull;
    auto cell_slot = [&](int e) -> size_t {
        const uint32_t ce = (uint32_t)e >> 3;
        uint32_t cyi = __umulhi(ce, inv);
        if (cyi * (uint32_t)ncxr > ce) cyi--;                // guard the float reciprocal
        const uint32_t cxi = ce - cyi * (uint32_t)ncxr;
        return ((size_t)(cya + (int)cyi) * f.ncx + cxa + (int)cxi) * kCellCap + (e & 7);
    };
    int nraw = 0;
    for (int c0 = 0; c0 < total; c0 += 256) {
        ClusterRec rec[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = c0 + u * 64 + lane;
            live[u] = e < total;
            if (live[u]) rec[u] = citems[cell_slot(e)];
            else { rec[u].yx = 0; rec[u].lab = 0; rec[u].tag = 0; rec[u].pad = 0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool keep = live[u] && rec[u].pad == f.gen && in_window(rec[u], win);
            const unsigned long long m = __ballot(keep);
            const int pos = nraw + __popcll(m & lowmask);
            if (keep && pos < 64) { s_raw[pos] = rec[u].yx; s_raw[64 + pos] = rec[u].lab; s_raw[128 + pos] = rec[u].tag; }
            nraw += __popcll(m);
        }
    }
    for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
        const uint32_t i = i0 + lane;
        ClusterRec rec;
        rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
        if (i < novf) rec = oitems[i];
        const bool keep = i < novf && in_window(rec, win);
        const unsigned long long m = __ballot(keep);
        const int pos = nraw + __popcll(m & lowmask);
        if (keep && pos < 64) { s_raw[pos] = rec.yx; s_raw[64 + pos] = rec.lab; s_raw[128 + pos] = rec.tag; }
        nraw += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();   // staging writes above, cross-lane reads below: keep program order
    FSLIC_STAMP(1);

    const uint32_t M = f.M, Coff = f.C;
    uint32_t hi16;   // 0xffff0000 in a VGPR so that (w & hi16) | c is one v_and_or_b32 with c in an SGPR
    asm volatile("v_mov_b32 %0, 0xffff0000" : "=v"(hi16));
    // spatial + colour term of one candidate for row r, window folded in: < kHuge iff the pixel is covered
    auto spatial_A = [&](int cx) -> uint32_t {
        const int dx = x - cx;
        const uint32_t ux = (uint32_t)(dx < 0 ? -dx : dx);
        if (LUT) return ux <= (uint32_t)S ? ux : (uint32_t)f.lut_oow;
        return ux <= (uint32_t)S ? __umul24(ux, M) + Coff : kHuge;
    };
    auto dist_w = [&](int r, int cy, uint32_t clab, uint32_t A) -> uint32_t {
        const int dy = yw_lo + r * stride - cy;
        const uint32_t uy = (uint32_t)(dy < 0 ? -dy : dy);                   // SALU
        if (LUT) {
            const uint32_t B = (uy <= (uint32_t)S ? uy : (uint32_t)f.lut_oow) * (uint32_t)f.lut_row;
            return __builtin_amdgcn_sad_hi_u8(pix[r], clab, s_lut[A + B]);
        }
        const uint32_t B = uy <= (uint32_t)S ? uy * M : kHuge;
        return __builtin_amdgcn_sad_hi_u8(pix[r], clab, A + B);
    };
    const int yw_hi = yw_lo + (R - 1) * stride;

    if (nraw > 64) {
        // ---- unsorted streaming path (no capacity limit): lexicographic (distance, phase<<16|k) in two registers
        uint32_t bw[R], bt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { bw[r] = 0xFFFFFFFFu; bt[r] = 0xFFFFFFFFu; }
        auto stream = [&](const ClusterRec& rec, bool keep) {
            unsigned long long m = __ballot(keep);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t cyx = __builtin_amdgcn_readlane(rec.yx, src);
                const uint32_t clab = __builtin_amdgcn_readlane(rec.lab, src);
                const uint32_t tag = __builtin_amdgcn_readlane(rec.tag, src);
                const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
                if (cy + S < yw_lo || cy - S > yw_hi) continue;
                const uint32_t A = spatial_A(cx);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t w = dist_w(r, cy, clab, A) & hi16;
                    const bool better = (w < bw[r]) || (w == bw[r] && tag < bt[r]);
                    bw[r] = better ? w : bw[r];
                    bt[r] = better ? tag : bt[r];
                }
            }
        };
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int e = c0 + lane;
            ClusterRec rec;
            rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
            if (e < total) rec = citems[cell_slot(e)];
            stream(rec, e < total && rec.pad == f.gen && in_window(rec, win));
        }
        for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
            const uint32_t i = i0 + lane;
            ClusterRec rec;
            rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
            if (i < novf) rec = oitems[i];
            stream(rec, i < novf && in_window(rec, win));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!(okx && r < nrows)) continue;
            const int y = yw_lo + r * stride;
            const size_t p = (size_t)y * W + x;
            uint32_t lbl;
            if (bw[r] < kHuge) { lbl = bt[r] & 0xFFFFu; f.labels[p] = (uint16_t)lbl; }
            else lbl = f.labels[p];
            if (FUSE && lbl != 0xFFFFu) global_accumulate(f.sums[0], lbl, (uint32_t)y, (uint32_t)x, pix[r]);
        }
        return;
    }

    // 3. exact prune + rank (one record per lane, pair tests by readlane broadcast)
    int n;
    {
        const bool act = lane < nraw;
        const uint32_t yx_j = act ? s_raw[lane] : 0u;
        const uint32_t lab_j = act ? s_raw[64 + lane] : 0u;
        const uint32_t tag_j = act ? s_raw[128 + lane] : 0xFFFFFFFFu;
        bool dup = false;
        for (int i = 0; i < nraw; ++i) {
            const uint32_t ty = __builtin_amdgcn_readlane(yx_j, i);
            const uint32_t tl = __builtin_amdgcn_readlane(lab_j, i);
            const uint32_t tt = __builtin_amdgcn_readlane(tag_j, i);
            dup = dup || (act && (tt < tag_j) && (ty == yx_j) && (tl == lab_j));
        }
        const unsigned long long dupmask = __ballot(dup);
        int rank = 0;
        for (int i = 0; i < nraw; ++i) {
            const uint32_t tt = __builtin_amdgcn_readlane(tag_j, i);
            rank += (int)((((dupmask >> i) & 1ull) == 0ull) && tt < tag_j);
        }
        if (act && !dup) s_list[rank] = make_uint4(yx_j, lab_j, tag_j & 0xFFFFu, 0u);
        n = nraw - __popcll(dupmask);
    }
    __builtin_amdgcn_wave_barrier();
    FSLIC_STAMP(2);

    // 4. the argmin over the sorted list
    uint32_t best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = 0xFFFFFFFFu;
    uint2 nxt = *reinterpret_cast<const uint2*>(&s_list[0]);
#pragma unroll 1
    for (int c = 0; c < n; ++c) {
        const uint2 rec = nxt;
        nxt = *reinterpret_cast<const uint2*>(&s_list[min(c + 1, 63)]);   // prefetch: hides the LDS latency
        const uint32_t cyx = rfl(rec.x);
        const uint32_t clab = rfl(rec.y);
        const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
        const uint32_t A = spatial_A(cx);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t w = dist_w(r, cy, clab, A);
            best[r] = min(best[r], LUT ? (w | (uint32_t)c) : ((w & hi16) | (uint32_t)c));
        }
    }
    FSLIC_STAMP(3);

    // 5. write labels (+ fused centroid sums)
    uint32_t slot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = yw_lo + r * stride;
        const bool ok = okx && r < nrows;
        const bool found = best[r] < kHuge;
        const uint32_t idx = found ? (best[r] & 0xFFFFu) : 0u;
        const uint32_t k = s_list[idx].z;
        slot[r] = 0xFFFFFFFFu;
        if (ok) {
            const size_t p = (size_t)y * W + x;
            if (found) {
                f.labels[p] = (uint16_t)k;
                if (FUSE) {
                    if (idx < (uint32_t)kAccSlots) slot[r] = idx;
                    else global_accumulate(f.sums[0], k, (uint32_t)y, (uint32_t)x, pix[r]);
                }
            } else if (FUSE) {
                const uint32_t old = f.labels[p];     // stale label still takes part in update()
                if (old != 0xFFFFu) global_accumulate(f.sums[0], old, (uint32_t)y, (uint32_t)x, pix[r]);
            }
        }
    }
    FSLIC_STAMP(4);
    if (FUSE) {
        bool uni = slot[0] != 0xFFFFFFFFu;
#pragma unroll
        for (int r = 1; r < R; ++r) uni = uni && (slot[r] == slot[0]);
        const uint32_t copy = (uint32_t)lane & (kAccCopies - 1);
        if (uni) {   // whole column in one superpixel: one set of LDS atomics for R pixels
            uint32_t sL = 0, sa = 0, sb = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) { sL += pix[r] & 255u; sa += (pix[r] >> 8) & 255u; sb += (pix[r] >> 16) & 255u; }
            uint32_t* a = s_acc + (slot[0] * 6) * kAccCopies + copy;
            atomicAdd(a + 0 * kAccCopies, (uint32_t)R);
            atomicAdd(a + 1 * kAccCopies, (uint32_t)(R * yw_lo + stride * (R * (R - 1) / 2)));
            atomicAdd(a + 2 * kAccCopies, (uint32_t)(R * x));
            atomicAdd(a + 3 * kAccCopies, sL);
            atomicAdd(a + 4 * kAccCopies, sa);
            atomicAdd(a + 5 * kAccCopies, sb);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (slot[r] != 0xFFFFFFFFu) {
                    uint32_t* a = s_acc + (slot[r] * 6) * kAccCopies + copy;
                    atomicAdd(a + 0 * kAccCopies, 1u);
                    atomicAdd(a + 1 * kAccCopies, (uint32_t)(yw_lo + r * stride));
                    atomicAdd(a + 2 * kAccCopies, (uint32_t)x);
                    atomicAdd(a + 3 * kAccCopies, pix[r] & 255u);
                    atomicAdd(a + 4 * kAccCopies, (pix[r] >> 8) & 255u);
                    atomicAdd(a + 5 * kAccCopies, (pix[r] >> 16) & 255u);
                }
            }
        }
        // flush (wave-private region: the LDS executes one wavefront's operations in order, no barrier needed)
        __builtin_amdgcn_wave_barrier();
        const int nacc = min(n, kAccSlots) * 6;
        for (int i = lane; i < nacc; i += 64) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < kAccCopies; ++c) v += s_acc[i * kAccCopies + c];
            if (v) {
                const int sl = i / 6;
                atomicAdd(f.sums[0] + 6 * (size_t)s_list[sl].z + (i - sl * 6), v);
            }
        }
    }
    FSLIC_STAMP(5);
    if (f.dbg && lane == 0) {
        const size_t o = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave) * 8;
        f.dbg[o + 6] = (unsigned long long)nraw;
        f.dbg[o + 7] = (unsigned long long)n;
    }
#undef FSLIC_STAMP
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// =============================================================================================
// Block-level packed assign kernel (the default, and the roofline kernel).
// One block = 64 columns x (4 wavefronts x R) visited rows.  The key of a pixel-candidate is 16 bits,
// (distance << 6) | idx, idx = rank of the candidate in (phase, k) order within the block's list (<= 64 entries),
// distance < 1024; two rows share one VALU instruction:
//     s   = v_sad_u8 (pix[r],   clab, 0)
//     s   = v_sad_hi_u8(pix[r+1], clab, s)           two colour distances packed
//     t   = v_lshl_add_u32(s, 6, idx|idx<<16)
//     key = v_pk_add_u16(t, spatial pair) clamp      0xffff outside the window: saturates, never wins
//     best= v_pk_min_u16(best, key)
// i.e. 2.5 VALU + one aligned LDS read per pixel-candidate pair.  The spatial pair comes from an LDS copy of the
// reference's patch (src/context.cpp:22-40) laid out so that the rows of a wavefront are immediate offsets from one
// per-lane address (two layouts, see VT below).  The prologue is paid once per BLOCK and costs one memory round trip:
//   wavefront 0   fetches every bin slot the block's pixels can see (all loads in flight at once), compacts the
//                 live records by ballot, ranks them by (phase, k) with readlane broadcasts and writes the sorted
//                 block list to LDS: idx = position in that list = visit order of the reference's scatter
//   wavefronts 1-3 stage the spatial table (and clear the centroid accumulators) meanwhile
//   one barrier
//   every wavefront takes the list into registers (lane j = candidate j), ballots the candidates whose window
//   reaches its own rows and walks that bit mask: cluster fields arrive by v_readlane, no LDS latency per candidate.
// Fused update (src/context.cpp:301-354): block-level LDS accumulators (privatised copies, LDS atomics), then ONE
// plain store per (candidate, quantity) into FrameDev::cpart at the entry reserved for (cluster, this block).
// No global atomic on the common path: device-scope atomics cost 21 of the 39 us of the per-wavefront-flush
// version of this pass (8 x 720p frames).  Pixels that no window covers keep their label and reach the sums through
// global atomics (rare).
// Algorithmic traffic: 4 B read + 2 B written per visited pixel.
// =============================================================================================
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

// s_bitreplicate_b64_b32: every bit of the source twice, in order (bit i -> bits 2i, 2i+1)
static __device__ __forceinline__ unsigned long long sbitrep(uint32_t v) {
    unsigned long long r;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(v));
    return r;
}
// One 32-bit word per pair of rows: 0xffff in the half of every row whose bit is set in `inv` (R bits), each bit
// replicated 16 times by four rounds of s_bitreplicate (scalar unit only).
template <int R>
static __device__ __forceinline__ void row_masks(uint32_t inv, uint32_t (&m)[R / 2]) {
    static_assert(R == 8 || R == 16, "row masks are built for 8 or 16 rows");
    const unsigned long long r1 = sbitrep(inv);                 // x2
    const unsigned long long r2 = sbitrep((uint32_t)r1);        // x4  (R <= 16: 32 significant bits of r1)
    if (R == 8) {
        const unsigned long long r3 = sbitrep((uint32_t)r2);    // x8: 64 bits
        const unsigned long long a = sbitrep((uint32_t)r3), b = sbitrep((uint32_t)(r3 >> 32));
        m[0] = (uint32_t)a; m[1] = (uint32_t)(a >> 32); m[2] = (uint32_t)b; m[3] = (uint32_t)(b >> 32);
    } else {
        const unsigned long long r3a = sbitrep((uint32_t)r2), r3b = sbitrep((uint32_t)(r2 >> 32));   // x8: 128 bits
        const unsigned long long q0 = sbitrep((uint32_t)r3a), q1 = sbitrep((uint32_t)(r3a >> 32));
        const unsigned long long q2 = sbitrep((uint32_t)r3b), q3 = sbitrep((uint32_t)(r3b >> 32));
        m[0] = (uint32_t)q0; m[1] = (uint32_t)(q0 >> 32); m[2] = (uint32_t)q1; m[3] = (uint32_t)(q1 >> 32);
        m[R / 2 - 4] = (uint32_t)q2; m[R / 2 - 3] = (uint32_t)(q2 >> 32); m[R / 2 - 2] = (uint32_t)q3; m[R / 2 - 1] = (uint32_t)(q3 >> 32);
    }
}

// VT (Manhattan patch only): the spatial term is a function of d = |dx| + |dy| alone, so the LDS table derives from the
// one-dimensional, V-shaped  V[t] = patch(|t - Cc|) << 6  (0xffff beyond 2S) instead of the (S+2) x (2S+2pad+1)
// column-major patch copy (26 KB at S = 72, re-read by every block).  Row r of a candidate reads V[Cc + dy_r + ux]
// when dy_r >= 0 and V[Cc + dy_r - ux] when dy_r < 0: either way the index advances with the row, so the rows are
// still fixed offsets from ONE per-lane base as long as they lie on one side of the centre; a candidate whose centre
// row lies inside the wavefront's rows is walked twice (rows above, rows below).  Lanes outside the x window use
// ux = 2S+1, which lands beyond 2S for every row; rows outside the y window are switched off by a per-pair scalar
// mask OR-ed in together with idx (v_lshl_or_b32: no extra VALU).
template <int R, bool FUSE, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_assign_blk(FrameDev f, int rem, int Hv) {
    extern __shared__ uint32_t s_dyn[];                  // spatial table (u16), f.tab_words dwords
    __shared__ uint32_t s_braw[3 * 64];
    __shared__ uint4 s_list[64];                         // {yx, lab, k, cpart entry | kNoSlot}
    __shared__ uint32_t s_bn;
    // centroid accumulators: per (candidate, copy) two 64-bit words of three 20-bit-spaced fields each,
    //   A = sum L | sum a << 20 | sum b << 40        B = sum (x - x0) | sum (y - y0) << 20 | count << 40
    // so a pixel (or a whole column) costs two ds_add_u64 instead of six ds_add_u32.  No field can carry into its
    // neighbour: a block has 2048 pixels, channels <= 255, x - x0 <= 63, y - y0 <= 93.
    __shared__ unsigned long long s_acc[FUSE ? 64 * kBlkCopies * 2 : 1];
    static_assert(R % 2 == 0, "rows are processed in pairs");
    constexpr int BH = kWavesPerBlock * R;
    static_assert(!FUSE || (BH * 64 * 255 < (1 << 20) && (BH - 1) * STRIDE * BH * 64 < (1 << 20)), "accumulator fields overflow");

    f.select(blockIdx.z);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    const int W = f.W, S = f.S;
    const int x0 = blockIdx.x * kTileW;
    const int x = x0 + lane;
    const int vrow_b0 = blockIdx.y * BH;
    const int vrow_w0 = vrow_b0 + wave * R;
    const bool okx = x < W;
    const int nrows = max(0, min(R, Hv - vrow_w0));      // 0: this wavefront has no rows (bottom edge)
    const int yw_lo = rem + vrow_w0 * STRIDE;
#define FSLIC_STAMP(slot_)                                                                              \
    do {                                                                                                 \
        if (f.dbg && lane == 0)                                                                          \
            f.dbg[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave) * 8 + (slot_)] = __builtin_readcyclecounter(); \
    } while (0)
    FSLIC_STAMP(0);

    // 1. pixel loads first; nothing below depends on them until the argmin
    // Out-of-image lanes and rows re-read the last valid column / row instead of being masked off: no exec juggling
    // around the loads (their results are never stored and never reach the centroid sums).
    uint32_t pix[R];
    if (nrows > 0) {
        const uint32_t xc = (uint32_t)min(x, W - 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t* rowp = f.lab + (size_t)(yw_lo + min(r, nrows - 1) * STRIDE) * W;   // wave-uniform base
            pix[r] = rowp[xc];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) pix[r] = 0u;
    }

    if (wave == 0) {
        // 2a. the block's candidate list
        const uint32_t novf = *f.ovf_cnt[0];
        const int sh = f.cs_shift;
        const ClusterRec* __restrict__ citems = f.cell_items[0];
        const ClusterRec* __restrict__ oitems = f.ovf_items[0];
        const unsigned long long lowmask = (1ull << lane) - 1ull;
        TileWindow win;
        const int vlast = min(vrow_b0 + BH - 1, Hv - 1);
        win.wy_lo = rem + vrow_b0 * STRIDE - S; win.wy_hi = rem + vlast * STRIDE + S;
        win.wx_lo = x0 - S;                     win.wx_hi = min(x0 + kTileW - 1, W - 1) + S;
        const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, W - 1) >> sh;
        const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
        const int ncxr = cxb - cxa + 1;
        int cnt = 0;
        auto visit = [&](const ClusterRec& rec, bool keep) {
            const unsigned long long m = __ballot(keep);
            const int pos = cnt + __popcll(m & lowmask);
            if (keep && pos < 64) { s_braw[pos] = rec.yx; s_braw[64 + pos] = rec.lab; s_braw[128 + pos] = rec.tag; }
            cnt += __popcll(m);
        };
        // lane -> (cell column, slot); a cell row of up to 8 cells is one 16-byte load per lane, six cell rows in flight
        for (int cx0 = 0; cx0 < ncxr; cx0 += 8) {
            const int cxi = cx0 + (lane >> 3);
            const bool colok = cxi < ncxr;
            const ClusterRec* base = citems + ((size_t)cya * f.ncx + cxa + min(cxi, ncxr - 1)) * kCellCap + (lane & 7);
            for (int cy0 = cya; cy0 <= cyb; cy0 += 6) {
                ClusterRec rr[6];
                bool okr[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {      // unconditional loads (rows past the range re-read the last one)
                    okr[u] = colok && (cy0 + u) <= cyb;
                    rr[u] = base[(size_t)min(u, cyb - cy0) * f.ncx * kCellCap];
                }
                base += 6 * (size_t)f.ncx * kCellCap;
#pragma unroll
                for (int u = 0; u < 6; ++u) visit(rr[u], okr[u] && rr[u].pad == f.gen && in_window(rr[u], win));
            }
        }
        for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
            const uint32_t i = i0 + lane;
            ClusterRec rec;
            rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
            if (i < novf) rec = oitems[i];
            visit(rec, i < novf && in_window(rec, win));
        }
        __builtin_amdgcn_wave_barrier();
        // rank by (phase, k): list order == visit order of the reference's scatter (src/context.cpp:214-242)
        const int bn = cnt;
        if (bn <= 64) {
            const bool act = lane < bn;
            const uint32_t yx_j = act ? s_braw[lane] : 0u;
            const uint32_t lab_j = act ? s_braw[64 + lane] : 0u;
            const uint32_t tag_j = act ? s_braw[128 + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int i = 0; i < bn; ++i) rank += (int)(__builtin_amdgcn_readlane(tag_j, i) < tag_j);
            uint32_t entry = kNoSlot;
            if (FUSE) {
                const uint32_t k = tag_j & 0xFFFFu;
                const SlotGeom g = slot_geom((int)(yx_j >> 16), (int)(yx_j & 0xFFFFu), S, f.H, W, rem, STRIDE, BH, Hv);
                const int by = (int)blockIdx.y, bx = (int)blockIdx.x;
                if (g.any && by >= g.by0 && by <= g.by1)
                    entry = (k * (uint32_t)f.cpart_ns + (uint32_t)((by - g.by0) * f.nbx + (bx - g.bx0))) * 6u;
            }
            if (act) s_list[rank] = make_uint4(yx_j, lab_j, tag_j & 0xFFFFu, entry);
        }
        if (lane == 0) s_bn = (uint32_t)bn;
    } else {
        // 2b. spatial table -> LDS, four 16-byte loads in flight per thread; centroid accumulators <- 0
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(f.tab);
        uint4* dst = reinterpret_cast<uint4*>(s_dyn);
        const int nq = f.tab_words >> 2;
        const int t = tid - 64;
        for (int b0 = 0; b0 < nq; b0 += 4 * 192) {     // indices past the end are clamped: the last quad is rewritten, no masking
            uint4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = src[min(b0 + u * 192 + t, nq - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[min(b0 + u * 192 + t, nq - 1)] = q[u];
        }
        if (FUSE) {
#pragma unroll
            for (int i = 0; i < (64 * kBlkCopies * 2 + 191) / 192; ++i) s_acc[min(i * 192 + t, 64 * kBlkCopies * 2 - 1)] = 0ull;
        }
    }
    __syncthreads();
    FSLIC_STAMP(1);
    const int bn = (int)rfl(s_bn);
    if (bn > 64) {                            // more live candidates than idx bits: the host redoes the frame generically
        if (tid == 0) *f.err_flag = 1u;
        return;
    }

    // 3. the block list in registers: lane j = candidate j; candidates whose window reaches this wavefront's rows
    const uint4 mine = s_list[min(lane, 63)];
    const uint32_t yx_j = mine.x, lab_j = mine.y;
    const int yw_hi = yw_lo + (nrows - 1) * STRIDE;
    unsigned long long todo;
    {
        const int cyj = (int)(yx_j >> 16);
        todo = __ballot(lane < bn && nrows > 0 && cyj + S >= yw_lo && cyj - S <= yw_hi);
        if (f.exp_flags & 4) todo = 0;        // timing experiment: no argmin
        if (f.exp_flags & 8) todo &= 1ull;    // timing experiment: one candidate
    }
    FSLIC_STAMP(2);

    // 4. the argmin, two rows per instruction
    us2 best[R / 2];
#pragma unroll
    for (int q = 0; q < R / 2; ++q) best[q] = (us2){0xFFFF, 0xFFFF};
    const uint32_t nrpad = (uint32_t)f.tab_nrpad;
    const int dyoff = f.tab_dyoff;
    if (VT) {
        const int Cc = f.tab_dyoff;                 // centre index of the V table
        const uint32_t xout = (uint32_t)f.tab_xout;
        constexpr int BIAS = 32768;                 // makes the numerators of the row-range divisions positive
        constexpr uint32_t full = (1u << R) - 1u;
        // Per-candidate row ranges, computed ONCE in the vector unit with lane j = candidate j (the scalar unit is
        // shared by the whole CU and issues about as many instructions per clock as the four SIMDs together).
        //   rows r with dy_r = dy0 + r*STRIDE: above the centre r < rs, inside the window ra <= r <= rb
        uint32_t vab_j;                             // rows above (low half) | rows at/below (high half) that are in the window
        {
            const int dy0 = yw_lo - (int)(yx_j >> 16);
            const int rs = min(max((-dy0 + STRIDE * BIAS + STRIDE - 1) / STRIDE - BIAS, 0), R);
            const int ra = min(max((-S - dy0 + STRIDE * BIAS + STRIDE - 1) / STRIDE - BIAS, 0), R);
            const int rb = min(max((S - dy0 + STRIDE * BIAS) / STRIDE - BIAS, -1), R - 1);
            const uint32_t upto_rs = (1u << rs) - 1u;
            const uint32_t in_win = rb < 0 ? 0u : (((2u << rb) - 1u) & ~((1u << ra) - 1u) & full);
            vab_j = (in_win & upto_rs) | ((in_win & ~upto_rs) << 16);
        }
        // row-vector table: entry t = (V[t], V[t+STRIDE], .. V[t+7*STRIDE]) as one 16-byte word, so the spatial terms of
        // eight rows are ONE 16-byte-ALIGNED ds_read_b128 (a 2-byte-aligned b128 read costs 64 clk of the CU's LDS pipe
        // instead of 8, scripts/microbench/lds_unaligned.hip)
        const uint4* const s_rv = reinterpret_cast<const uint4*>(s_dyn);
        auto pass = [&](uint32_t clab, uint32_t base, uint32_t idx01, const uint32_t* mq) {
            uint32_t spw[R / 2];
            {
                const uint4 e0 = s_rv[base];
                spw[0] = e0.x; spw[1] = e0.y; spw[2] = e0.z; spw[3] = e0.w;
                if (R == 16) {
                    const uint4 e1 = s_rv[base + 8 * STRIDE];
                    spw[R / 2 - 4] = e1.x; spw[R / 2 - 3] = e1.y; spw[R / 2 - 2] = e1.z; spw[R / 2 - 1] = e1.w;
                }
            }
#pragma unroll
            for (int q = 0; q < R / 2; ++q) {
                const us2 sp = __builtin_bit_cast(us2, spw[q]);
                const uint32_t s0 = __builtin_amdgcn_sad_u8(pix[2 * q], clab, 0u);
                const uint32_t s01 = __builtin_amdgcn_sad_hi_u8(pix[2 * q + 1], clab, s0);
                const uint32_t t = (s01 << 6) | (mq ? (mq[q] | idx01) : idx01);                // v_lshl_or_b32, scalar 3rd operand
                const us2 key = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, t), sp);
                best[q] = __builtin_elementwise_min(best[q], key);
            }
        };
#pragma unroll 1
        while (todo) {
            const int c = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t cyx = __builtin_amdgcn_readlane(yx_j, c);
            const uint32_t clab = __builtin_amdgcn_readlane(lab_j, c);
            const uint32_t vab = __builtin_amdgcn_readlane(vab_j, c);
            const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
            const uint32_t ux = __builtin_amdgcn_sad_u16((uint32_t)x, (uint32_t)cx, 0u);      // |x - cx|
            const uint32_t uxc = ux <= (uint32_t)S ? ux : xout;
            const uint32_t k0 = (uint32_t)(Cc + yw_lo - cy);                                   // wave-uniform
            const uint32_t idx01 = (uint32_t)c | ((uint32_t)c << 16);
            if (vab == (full << 16)) {                    // every row at/below the centre and inside the window
                pass(clab, k0 + uxc, idx01, nullptr);
            } else if (vab == full) {                     // every row above the centre and inside the window
                pass(clab, k0 - uxc, idx01, nullptr);
            } else {                                      // window edge and / or centre row inside this wavefront's rows
                const uint32_t va = vab & 0xFFFFu, vb = vab >> 16;
                uint32_t mq[R / 2];
                if (va) { row_masks<R>(~va & full, mq); pass(clab, k0 - uxc, idx01, mq); }
                if (vb) { row_masks<R>(~vb & full, mq); pass(clab, k0 + uxc, idx01, mq); }
            }
        }
    } else {
#pragma unroll 1
    while (todo) {
        const int c = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t cyx = __builtin_amdgcn_readlane(yx_j, c);
        const uint32_t clab = __builtin_amdgcn_readlane(lab_j, c);
        const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
        const uint32_t ux = __builtin_amdgcn_sad_u16((uint32_t)x, (uint32_t)cx, 0u);      // |x - cx|
        const uint32_t uxc = min(ux, (uint32_t)(S + 1));
        const uint32_t a0 = __umul24(uxc, nrpad) + (uint32_t)(yw_lo - cy + dyoff);        // dword index of the pair (row 0, row 1)
        const uint32_t idx01 = (uint32_t)c | ((uint32_t)c << 16);
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            const us2 sp = __builtin_bit_cast(us2, s_dyn[a0 + (2 * q) * STRIDE]);            // one aligned ds_read_b32 per row pair
            const uint32_t s0 = __builtin_amdgcn_sad_u8(pix[2 * q], clab, 0u);
            const uint32_t s01 = __builtin_amdgcn_sad_hi_u8(pix[2 * q + 1], clab, s0);
            const uint32_t t = (s01 << 6) + idx01;
            const us2 key = __builtin_elementwise_add_sat(__builtin_bit_cast(us2, t), sp);
            best[q] = __builtin_elementwise_min(best[q], key);
        }
    }
    }
    FSLIC_STAMP(3);

    // 5. write labels (+ fused centroid sums)
    uint32_t slot[R];
    {
        uint32_t kk[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t key = (r & 1) ? (uint32_t)best[r / 2].y : (uint32_t)best[r / 2].x;
            kk[r] = s_list[key & 63u].z;                            // idx 63 when not found: read, never used
        }
        uint16_t* rowp = f.labels + (size_t)yw_lo * W;
        const uint32_t xo = (uint32_t)x;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = okx && r < nrows;
            const uint32_t key = (r & 1) ? (uint32_t)best[r / 2].y : (uint32_t)best[r / 2].x;
            const bool found = key != 0xFFFFu;
            slot[r] = kNoSlot;
            if (ok && found) rowp[xo] = (uint16_t)kk[r];
            if (FUSE && ok) {
                if (found) slot[r] = key & 63u;
                else {
                    const uint32_t old = rowp[xo];     // stale label still takes part in update()
                    if (old != 0xFFFFu) global_accumulate(f.sums[0], old, (uint32_t)(yw_lo + r * STRIDE), xo, pix[r]);
                }
            }
            rowp += (size_t)STRIDE * W;
        }
    }
    FSLIC_STAMP(4);
    if (FUSE) {
        if (!(f.exp_flags & 2)) {
            bool uni = slot[0] != kNoSlot;
#pragma unroll
            for (int r = 1; r < R; ++r) uni = uni && (slot[r] == slot[0]);
            const uint32_t copy = (uint32_t)lane & (kBlkCopies - 1);
            const uint32_t dyw = (uint32_t)(wave * R * STRIDE);            // first row of this wavefront relative to the block's
            if (uni) {   // whole column in one superpixel: one pair of LDS atomics for R pixels
                uint32_t sL = 0, sa = 0, sb = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) { sL += pix[r] & 255u; sa += (pix[r] >> 8) & 255u; sb += (pix[r] >> 16) & 255u; }
                unsigned long long* a = s_acc + (slot[0] * kBlkCopies + copy) * 2;
                const unsigned long long A = (unsigned long long)(sL | (sa << 20)) | ((unsigned long long)(sb << 8) << 32);
                const uint32_t sdy = (uint32_t)R * dyw + (uint32_t)(STRIDE * (R * (R - 1) / 2));
                const unsigned long long B = (unsigned long long)((uint32_t)R * (uint32_t)lane | (sdy << 20)) | ((unsigned long long)((uint32_t)R << 8) << 32);
                atomicAdd(a, A);
                atomicAdd(a + 1, B);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (slot[r] != kNoSlot) {
                        unsigned long long* a = s_acc + (slot[r] * kBlkCopies + copy) * 2;
                        const uint32_t p = pix[r];
                        const unsigned long long A = (unsigned long long)((p & 255u) | ((p & 0xFF00u) << 12)) | ((unsigned long long)((p >> 8) & 0xFF00u) << 32);
                        const unsigned long long B = (unsigned long long)((uint32_t)lane | ((dyw + (uint32_t)(r * STRIDE)) << 20)) | (256ull << 32);
                        atomicAdd(a, A);
                        atomicAdd(a + 1, B);
                    }
                }
            }
        }
        __syncthreads();
        // one plain store per (candidate, quantity): entry reserved for (cluster, this block), zeros included
        const uint32_t yb0 = (uint32_t)(rem + vrow_b0 * STRIDE);
        for (int i = tid; i < bn * 6; i += 256) {
            const int c = i / 6, q = i - c * 6;
            const uint32_t entry = s_list[c].w;
            if (entry != kNoSlot) {
                unsigned long long A = 0, B = 0;
#pragma unroll
                for (int cpy = 0; cpy < kBlkCopies; ++cpy) { A += s_acc[(c * kBlkCopies + cpy) * 2]; B += s_acc[(c * kBlkCopies + cpy) * 2 + 1]; }
                const uint32_t cnt = (uint32_t)(B >> 40);
                uint32_t v;
                if (q == 0) v = cnt;
                else if (q == 1) v = ((uint32_t)(B >> 20) & 0xFFFFFu) + yb0 * cnt;
                else if (q == 2) v = ((uint32_t)B & 0xFFFFFu) + (uint32_t)x0 * cnt;
                else if (q == 3) v = (uint32_t)A & 0xFFFFFu;
                else if (q == 4) v = (uint32_t)(A >> 20) & 0xFFFFFu;
                else v = (uint32_t)(A >> 40);
                f.cpart[(size_t)entry + (uint32_t)q] = v;
            }
        }
    }
    FSLIC_STAMP(5);
    if (f.dbg && lane == 0) {
        const size_t o = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWavesPerBlock + wave) * 8;
        f.dbg[o + 6] = (unsigned long long)bn;
        f.dbg[o + 7] = (unsigned long long)bn;
    }
#undef FSLIC_STAMP
}

template <int R, bool FUSE, int STRIDE>
static void launch_assign_blk_t(const FrameDev& f_, int nframes, int rem, int Hv, hipStream_t st) {
    FrameDev f = f_;
    if (FUSE) { f.tab = f.tabs; f.tab_words = f.tabs_words; f.tab_dyoff = f.tabs_dyoff; f.tab_nrpad = f.tabs_nrpad; }   // table of the subsampled stride
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * R - 1) / (kWavesPerBlock * R), nframes);
    static const size_t lds_pad = getenv("FSLIC_LDS_PAD") ? (size_t)atoi(getenv("FSLIC_LDS_PAD")) : 0;   // experiment: caps blocks per CU
    const size_t lds = (size_t)f.tab_words * sizeof(uint32_t) + lds_pad;
    if (lds_pad) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assign_blk<R, FUSE, STRIDE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assign_blk<R, FUSE, STRIDE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (f.tab_vmode) hipLaunchKernelGGL((k_assign_blk<R, FUSE, STRIDE, true>), grid, dim3(256), lds, st, f, rem, Hv);
    else hipLaunchKernelGGL((k_assign_blk<R, FUSE, STRIDE, false>), grid, dim3(256), lds, st, f, rem, Hv);
}

static int assign_rows_per_wave(const FrameDev& f, int nframes, int Hv, bool fuse_update) {
    const char* e = getenv("FSLIC_ASSIGN_R");
    if (e && (atoi(e) == 8 || atoi(e) == 16)) return atoi(e);
    const int tiles_x = (f.W + kTileW - 1) / kTileW;
    const int blocks8 = nframes * tiles_x * ((Hv + kWavesPerBlock * 8 - 1) / (kWavesPerBlock * 8));
    return (!fuse_update && blocks8 > 3072) ? 16 : 8;
}

template <int R>
static void launch_assign_r(const FrameDev& f, int nframes, int rem, int stride, int Hv, int buf, int sbuf, bool fuse_update, hipStream_t st) {
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * R - 1) / (kWavesPerBlock * R), nframes);
    const size_t lds = (size_t)f.lut_words * sizeof(uint32_t);
    if (f.lut_words > 0) {
        if (fuse_update) hipLaunchKernelGGL((k_assign<R, true, true>), grid, dim3(256), lds, st, f, rem, stride, Hv, buf, sbuf);
        else hipLaunchKernelGGL((k_assign<R, false, true>), grid, dim3(256), lds, st, f, rem, stride, Hv, buf, sbuf);
    } else {
        if (fuse_update) hipLaunchKernelGGL((k_assign<R, true, false>), grid, dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
        else hipLaunchKernelGGL((k_assign<R, false, false>), grid, dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
    }
}

PassGeom launch_assign(const FrameDev& f_, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    PassGeom pg;
    pg.rem = rem; pg.stride = stride; pg.Hv = Hv; pg.BH = kWavesPerBlock * 8; pg.use_slots = 0;
    if (Hv <= 0) return pg;
    static const int which = getenv("FSLIC_ASSIGN") ? atoi(getenv("FSLIC_ASSIGN")) : 0;   // 32: force the 32-bit kernel (A/B timing)
    if (f.tab_words > 0 && stride >= 1 && stride <= 3 && which == 0) {       // block-level packed kernel
        if (!fuse_update && stride == 1) {
            if (assign_rows_per_wave(f, nframes, Hv, false) == 16) launch_assign_blk_t<16, false, 1>(f, nframes, rem, Hv, st);
            else launch_assign_blk_t<8, false, 1>(f, nframes, rem, Hv, st);
            return pg;
        }
        if (fuse_update) {
            pg.use_slots = 1;
            if (stride == 1) launch_assign_blk_t<8, true, 1>(f, nframes, rem, Hv, st);
            else if (stride == 2) launch_assign_blk_t<8, true, 2>(f, nframes, rem, Hv, st);
            else launch_assign_blk_t<8, true, 3>(f, nframes, rem, Hv, st);
            return pg;
        }
    }
    if (assign_rows_per_wave(f, nframes, Hv, fuse_update) == 16) launch_assign_r<16>(f, nframes, rem, stride, Hv, buf, sbuf, fuse_update, st);
    else launch_assign_r<8>(f, nframes, rem, stride, Hv, buf, sbuf, fuse_update, st);
    return pg;
}

// ---------------------------------------------------------------------------------------------
// Generic gather: one thread per visited pixel, candidates read straight from the coarse cells,
// 64-bit key (u16 distance << 32 | phase << 16 | k), spatial term from the (2S+1)^2 patch itself and
// the distance wrapped to 16 bits exactly like src/context.cpp:285-286.  It is the fallback for inputs
// the tiled kernel does not take (candidate-list overflow, S < 8, non-Manhattan patch, no exact
// fixed-point multiplier) and an independent cross-check of the tiled kernel in the tests.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ void generic_test(const ClusterRec& rec, int y, int x, int S, int P, uint32_t pix,
                                                    const uint16_t* __restrict__ patch, unsigned long long& best) {
    const int dy = y - (int)(rec.yx >> 16), dx = x - (int)(rec.yx & 0xFFFFu);
    if (dy < -S || dy > S || dx < -S || dx > S) return;
    const uint32_t sp = patch[(dy + S) * P + (dx + S)];
    const uint32_t d = (sp + __builtin_amdgcn_sad_u8(pix, rec.lab, 0u)) & 0xFFFFu;
    const unsigned long long key = ((unsigned long long)d << 32) | rec.tag;
    best = key < best ? key : best;
}

template <bool FUSE>
__global__ __launch_bounds__(256) void k_assign_generic(FrameDev f, int rem, int stride, int Hv, int buf, int sbuf) {
    f.select(blockIdx.y);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Hv * f.W) return;
    const int v = (int)(i / f.W), x = (int)(i - (size_t)v * f.W);
    const int y = rem + v * stride;
    const size_t p = (size_t)y * f.W + x;
    const uint32_t pix = f.lab[p];
    const int S = f.S, P = 2 * S + 1, sh = f.cs_shift;
    unsigned long long best = ~0ull;
    const int cxa = max(x - S, 0) >> sh, cxb = min(x + S, f.W - 1) >> sh;
    const int cya = max(y - S, 0) >> sh, cyb = min(y + S, f.H - 1) >> sh;
    for (int cyi = cya; cyi <= cyb; ++cyi)
        for (int cxi = cxa; cxi <= cxb; ++cxi) {
            const int cell = cyi * f.ncx + cxi;
            const uint32_t cnt = min(f.cell_cnt[0][cell], (uint32_t)kCellCap);
            for (uint32_t s = 0; s < cnt; ++s)
                generic_test(f.cell_items[0][(size_t)cell * kCellCap + s], y, x, S, P, pix, f.patch2d, best);
        }
    const uint32_t novf = *f.ovf_cnt[0];
    for (uint32_t s = 0; s < novf; ++s) generic_test(f.ovf_items[0][s], y, x, S, P, pix, f.patch2d, best);
    uint32_t lbl;
    if (best != ~0ull) { lbl = (uint32_t)(best & 0xFFFFull); f.labels[p] = (uint16_t)lbl; }
    else lbl = f.labels[p];
    if (FUSE && lbl != 0xFFFFu) global_accumulate(f.sums[0], lbl, (uint32_t)y, (uint32_t)x, pix);
}

void launch_assign_generic(const FrameDev& f_, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const size_t n = (size_t)Hv * f.W;
    const int blocks = (int)((n + 255) / 256);
    if (fuse_update) hipLaunchKernelGGL(k_assign_generic<true>, dim3(blocks, nframes), dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
    else hipLaunchKernelGGL(k_assign_generic<false>, dim3(blocks, nframes), dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
}

// =============================================================================================
// Connectivity / min-size pass (src/cca.cpp).  GPU union-find whose roots are the minimum raster
// index of each component, exactly the "leader" the reference ends up with (src/cca.h:36-57).
//   k_cca_local    64x64 tiles resolved entirely in LDS (row runs by ballot, vertical unions by LDS
//                  atomicMin), writes parent[p] = global index of the tile-local root, tile-local
//                  areas, and appends tile-local roots to a list
//   k_cca_border   unions across tile borders (global atomicMin)
//   k_cca_compress tile-local roots -> global roots, areas summed into the global root
//   k_cca_collect  global roots with area >= threshold -> candidate list for the top-K step
//   (host)         std::partial_sort / ranking of <= few thousand (leader, area) pairs
//   k_cca_chain    every other component takes the final label of the component left of (or
//                  above) its leader, src/cca.cpp:240-254, resolved by chasing toward smaller leaders
//   k_cca_relabel  out[p] = final[root(p)]
// =============================================================================================
// LDS accessors.  They are relaxed workgroup-scope atomics rather than `volatile` accesses on purpose:
// volatile blocks the address-space inference and the accesses become FLAT instructions, which reach the LDS
// through the vector-memory path at global-memory latency (measured: 1.2 ms instead of tens of us for the
// tile kernel); atomics still lower to ds_read_b32 / ds_write_b32 and are re-read on every loop trip.
static __device__ __forceinline__ uint32_t lds_ld(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// find with path halving.  The halving store races with concurrent atomicMin hooks, which is safe:
// it only ever replaces a parent by one of its ancestors (cf. ECL-CC style concurrent union-find).
static __device__ __forceinline__ uint32_t lds_find(uint32_t* par, uint32_t x) {
    for (;;) {
        const uint32_t p = lds_ld(par + x);
        if (p == x) return x;
        const uint32_t gp = lds_ld(par + p);
        if (gp == p) return p;
        lds_st(par + x, gp);
        x = gp;
    }
}
static __device__ __forceinline__ void lds_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(par + a, b);
        if (old == a) return;
        a = old;
    }
}

// read-only find (no halving stores), unrolled for the flatten pass
static __device__ __forceinline__ uint32_t lds_find_ro(uint32_t* par, uint32_t x) {
    uint32_t p = lds_ld(par + x);
    while (p != x) { x = p; p = lds_ld(par + x); }
    return x;
}

// One block = one 64x64 tile = 4 wavefronts x 16 rows.  Lane = column.
//  A  each wavefront resolves its own 16-row strip top-down: horizontal runs by ballot, one LDS union per
//     horizontal overlap with the row above; the (possibly stale) root of the pixel above travels in a VGPR,
//     so a row costs a handful of dependent LDS operations instead of chain walks
//  B  the three strip seams inside the tile are merged in LDS
//  C  flatten: every pixel -> tile-local root (min raster index), parent plane written, run lengths added
//     to the root's area
//  D  tile-local roots appended to the root list (one global atomic per wavefront row)
__global__ __launch_bounds__(256) void k_cca_local(CcaDev c) {
    c.select(blockIdx.z);
    constexpr int T = kCcaTile;
    constexpr int TH = kCcaTileH;
    constexpr int RW = TH / 4;
    __shared__ uint16_t s_lbl[T * TH];
    __shared__ uint32_t s_par[T * TH];
    __shared__ uint32_t s_area[T * TH];
    const int lane = LANE();
    const int wave = (int)rfl((uint32_t)(threadIdx.x >> 6));
    const int tx0 = blockIdx.x * T, ty0 = blockIdx.y * TH;
    const int x = tx0 + lane;
    const bool okx = x < c.W;
    const int ly0 = wave * RW;
    const unsigned long long lowmask = (2ull << lane) - 1ull;

    uint32_t lbl[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int y = ty0 + ly0 + i;
        lbl[i] = (okx && y < c.H) ? (uint32_t)c.in[(size_t)y * c.W + x] : 0x10000u;
    }
    // ---- A ----
    // Per row: every pixel whose upper neighbour has the same label folds the (possibly stale) root carried in
    // its VGPR into its run's node with a non-returning ds_min; one dependent ds_read then yields the run's
    // root.  Only where a run joins two so-far distinct components (U shapes) is a real union needed.
    uint32_t up_root = 0;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int ly = ly0 + i;
        const bool ok = okx && (ty0 + ly) < c.H;
        const int p = ly * T + lane;
        const uint32_t l = lbl[i];
        const uint32_t left = __shfl_up(l, 1);
        const bool start = (lane == 0) || (l != left) || !ok;
        const unsigned long long m = __ballot(start);
        const uint32_t node = (uint32_t)(ly * T + (63 - __clzll(m & lowmask)));
        s_lbl[p] = (uint16_t)l;
        s_area[p] = 0u;
        s_par[p] = node;
        uint32_t cur_root = node;
        if (i > 0) {
            const bool up_same = ok && (lbl[i - 1] == l);
            if (up_same) __hip_atomic_fetch_min(s_par + node, up_root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ok) cur_root = lds_ld(s_par + node);
            if (up_same && up_root != cur_root) lds_union(s_par, up_root, cur_root);
        }
        up_root = cur_root;
    }
    __syncthreads();
    // ---- B ----
    if (wave > 0) {
        const int p = ly0 * T + lane;
        const bool ok = okx && (ty0 + ly0) < c.H;
        const uint32_t l = lbl[0];
        const uint32_t left = __shfl_up(l, 1);
        if (ok && (uint32_t)s_lbl[p - T] == l) {
            const bool first = (lane == 0) || (left != l) || ((uint32_t)s_lbl[p - T - 1] != l);
            if (first) lds_union(s_par, (uint32_t)p, (uint32_t)(p - T));
        }
    }
    __syncthreads();
    // ---- C ----
    uint32_t root[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) root[i] = s_par[(ly0 + i) * T + lane];
#pragma unroll
    for (int i = 0; i < RW; ++i) root[i] = s_par[root[i]];
#pragma unroll
    for (int i = 0; i < RW; ++i) root[i] = lds_find_ro(s_par, root[i]);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int ly = ly0 + i;
        const int y = ty0 + ly;
        const bool ok = okx && y < c.H;
        const uint32_t r = ok ? root[i] : 0xFFFFFFFFu;
        if (ok) c.parent[(size_t)y * c.W + x] = (int32_t)((size_t)(ty0 + (int)(r / T)) * c.W + tx0 + (int)(r % T));
        const uint32_t rl = __shfl_up(r, 1);
        const bool start = (lane == 0) || (r != rl) || !ok;
        const unsigned long long m = __ballot(start);
        if (ok && start) {
            const unsigned long long higher = (lane == 63) ? 0ull : (m >> (lane + 1));
            const int len = higher ? __ffsll((long long)higher) : (64 - lane);
            atomicAdd(&s_area[r], (uint32_t)len);
        }
    }
    __syncthreads();
    // ---- D ---- tile-local roots -> this tile's slice of the root list.  No global atomic: a single
    // returning atomic word saturates near 88 ops/us on this chip, which at one append per wavefront-row
    // (130k at 4K) cost 1.2 ms in the first version of this kernel.
    __shared__ uint32_t s_nroots;
    if (threadIdx.x == 0) s_nroots = 0;
    __syncthreads();
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    int32_t* const my_roots = c.roots + (size_t)tile * (T * TH);
#pragma unroll 4
    for (int i = 0; i < RW; ++i) {
        const int ly = ly0 + i;
        const int y = ty0 + ly;
        const bool ok = okx && y < c.H;
        const int p = ly * T + lane;
        const bool isroot = ok && root[i] == (uint32_t)p;
        const unsigned long long m = __ballot(isroot);
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&s_nroots, (uint32_t)__popcll(m));
            base = __shfl(base, leader);
            if (isroot) {
                const int32_t g = (int32_t)((size_t)y * c.W + x);
                c.area[g] = s_area[p];
                my_roots[base + __popcll(m & ((1ull << lane) - 1ull))] = g;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) c.tile_cnt[tile] = s_nroots;
}

static __device__ __forceinline__ int32_t g_load(int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ int32_t g_find(int32_t* parent, int32_t x) {
    for (;;) {
        const int32_t p = g_load(parent + x);
        if (p == x) return x;
        const int32_t gp = g_load(parent + p);
        if (gp == p) return p;
        __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = gp;
    }
}
static __device__ __forceinline__ void g_union(int32_t* parent, int32_t a, int32_t b) {
    for (;;) {
        a = g_find(parent, a);
        b = g_find(parent, b);
        if (a == b) return;
        if (a < b) { const int32_t t = a; a = b; b = t; }
        const int32_t old = atomicMin(parent + a, b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_cca_border(CcaDev c, int nh, int nv) {
    c.select(blockIdx.y);
    constexpr int T = kCcaTile;
    constexpr int TH = kCcaTileH;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = c.W, H = c.H;
    if (i < nh) {            // pixel on the first row of a tile row, merge with the pixel above
        const int by = i / W + 1, x = i - (by - 1) * W;
        const int p = by * TH * W + x;
        const uint16_t lbl = c.in[p];
        if (c.in[p - W] == lbl) {
            const bool first = (x % T == 0) || c.in[p - 1] != lbl || c.in[p - W - 1] != lbl;
            if (first) g_union(c.parent, p, p - W);
        }
    } else if (i < nh + nv) { // pixel on the first column of a tile column, merge with the pixel to the left
        const int j = i - nh;
        const int bx = j / H + 1, y = j - (bx - 1) * H;
        const int p = y * W + bx * T;
        const uint16_t lbl = c.in[p];
        if (c.in[p - 1] == lbl) {
            const bool first = (y % TH == 0) || c.in[p - W] != lbl || c.in[p - W - 1] != lbl;
            if (first) g_union(c.parent, p, p - 1);
        }
    }
}

// One block per tile for the passes over tile-local roots.
__global__ __launch_bounds__(256) void k_cca_compress(CcaDev c) {
    c.select(blockIdx.y);
    constexpr int TT = kCcaTilePx;
    const int tile = blockIdx.x;
    const uint32_t n = c.tile_cnt[tile];
    const int32_t* __restrict__ my_roots = c.roots + (size_t)tile * TT;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t r = my_roots[i];
        // READ-ONLY walk: a path-halving store of another thread (parent[r] = some ancestor) could land after this
        // thread's parent[r] = g and leave r one level short of flat, and chain / relabel rely on parent[parent[p]]
        // being the global root (an intermittent wrong label at 4K, about one run in six, was exactly that)
        int32_t g = r;
        for (;;) {
            const int32_t p = g_load(c.parent + g);
            if (p == g) break;
            g = p;
        }
        if (g != r) {
            __hip_atomic_store(c.parent + r, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&c.area[g], c.area[r]);
        }
    }
}

// Global roots: reset their final label; those with area >= threshold (src/cca.cpp:213-217) become candidates,
// appended to the frame's dense candidate arrays (one returning atomic per tile on counters[1], which ends up as M).
__global__ __launch_bounds__(256) void k_cca_collect(CcaDev c) {
    c.select(blockIdx.y);
    constexpr int TT = kCcaTilePx;
    __shared__ uint32_t s_n, s_base;
    __shared__ int32_t s_cl[256];
    __shared__ uint32_t s_ca[256];
    const int tile = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t n = c.tile_cnt[tile];
    const int32_t* __restrict__ my_roots = c.roots + (size_t)tile * TT;
    for (uint32_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        bool cand = false;
        int32_t r = 0;
        uint32_t a = 0;
        if (i < n) {
            r = my_roots[i];
            if (c.parent[r] == r) {
                c.final_lab[r] = 0xFFFFu;
                a = c.area[r];
                cand = (int64_t)a >= (int64_t)c.min_threshold;
            }
        }
        if (cand) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            s_cl[pos] = r; s_ca[pos] = a;
        }
        __syncthreads();
        const uint32_t m = s_n;
        if (m) {
            if (threadIdx.x == 0) s_base = atomicAdd(&c.counters[1], m);
            __syncthreads();
            if (threadIdx.x < m) {
                c.cand_leader[s_base + threadIdx.x] = s_cl[threadIdx.x];
                c.cand_area[s_base + threadIdx.x] = s_ca[threadIdx.x];
            }
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Device top-K + ranking (src/cca.cpp:205-237) in ONE block, for the unambiguous case:
//   M <= K            : every candidate is kept
//   M >  K            : the K largest areas are kept; the K-th largest area a* is found by an 8-bit radix
//                       select.  If only part of the components with area == a* fit, which ones
//                       std::partial_sort keeps depends on libstdc++'s heap order -> status = 1 and the host
//                       redoes the step with std::partial_sort itself.
// Kept leaders are sorted (bitonic, LDS) and final_lab[leader] = rank, i.e. labels 0,1,2.. in raster order
// of each component's first pixel (src/cca.cpp:229-237).
// status word (counters[3]): 0 = done here; 1 = area tie at the cut: the candidates have been written to
// dense_leader / dense_area SORTED BY LEADER (counters[1] = M), which is the order the reference feeds to
// std::partial_sort, and the host only has to run that one call; 2 = too many candidates for this block, the
// host gathers and sorts them itself.
// ---------------------------------------------------------------------------------------------
constexpr int kSelCap = 10240;     // candidates the device path holds in LDS (3 x 40 KB)
constexpr int kSelSortCap = 8192;  // candidates the tie path sorts in LDS (power of two <= kSelCap)
constexpr int kSelBuckets = 2048;  // ranking buckets (consecutive raster-index ranges)
constexpr int kSelBins = 4096;     // histogram bins of the selection

// block-wide exclusive scan of one value per thread (1024 threads): wave scans by shuffles, 16 wave totals in LDS
static __device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    __syncthreads();                 // s_w may still be read from a previous call
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t t = s_w[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

// K-th largest of s_area[0..M) restricted to values v with (v >> (sh + 12)) == prefix (all values when first):
// histogram of (v >> sh) & 4095 (areas spread over many bins: few same-address LDS atomics), descending scan.
// Returns through s_v[1] the bin holding the kk-th largest, through s_v[2] the rank left inside that bin.
static __device__ __forceinline__ void select_pass(const uint32_t* s_area, uint32_t M, int sh, bool first, uint32_t prefix,
                                                   uint32_t kk, uint32_t* s_hist, uint32_t* s_w, uint32_t* s_v) {
    const int tid = threadIdx.x;
    for (int i = tid; i < kSelBins; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint32_t a = s_area[i];
        if (first || (a >> (sh + 12)) == prefix) atomicAdd(&s_hist[(a >> sh) & (kSelBins - 1)], 1u);
    }
    __syncthreads();
    // thread t owns bins 4095-4t .. 4092-4t (descending order)
    uint32_t h[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = s_hist[kSelBins - 1 - 4 * tid - q]; tot += h[q]; }
    uint32_t total;
    uint32_t run = block_excl_scan_1024(tot, s_w, total);    // count in bins above this thread's
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (run < kk && run + h[q] >= kk) {
            s_v[1] = (uint32_t)(kSelBins - 1 - 4 * tid - q);
            s_v[2] = kk - run;
        }
        run += h[q];
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_cca_select(CcaDev c, int cap, unsigned long long* dbg) {
#define SEL_STAMP(i_) do { if (dbg && threadIdx.x == 0 && blockIdx.x == 0) dbg[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)   // 100 MHz
    SEL_STAMP(0);
    c.select(blockIdx.x);
    extern __shared__ uint32_t s_dyn[];
    uint32_t* const s_area = s_dyn;
    uint32_t* const s_lead = s_dyn + cap;
    uint32_t* const s_key = s_dyn + 2 * cap;
    uint32_t* const s_bcnt = s_dyn + 3 * cap;           // ranking buckets; the selection's histogram lives here first
    uint32_t* const s_bstart = s_bcnt + kSelBuckets;
    uint32_t* const s_bfill = s_bstart + kSelBuckets;
    uint32_t* const s_hist = s_bcnt;                    // kSelBins <= 3 * kSelBuckets words
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_v[8];     // [0] max area, [1] bin, [2] kk, [4] ties
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int K = c.K;
    // 1. the frame's candidates -> LDS
    const uint32_t M = c.counters[1];
    if (tid < 8) s_v[tid] = 0;
    if (tid == 0) c.counters[2] = M;
    if (M > (uint32_t)cap) {
        if (tid == 0) c.counters[3] = 2u;
        return;
    }
    uint32_t amax = 0;
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint32_t a = c.cand_area[i];
        s_area[i] = a;
        s_lead[i] = (uint32_t)c.cand_leader[i];
        amax = max(amax, a);
    }
    __syncthreads();
    if (M > (uint32_t)K) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = max(amax, (uint32_t)__shfl_xor(amax, off));
        if (lane == 0 && amax) atomicMax(&s_v[0], amax);
        __syncthreads();
    }
    SEL_STAMP(1);
    // 2. K-th largest area: 12 bits per pass, starting just above the top bit of the largest area
    uint32_t astar = 0;           // keep iff area >= astar
    if (M > (uint32_t)K) {
        const uint32_t mx = s_v[0];
        int sh = 0;
        while ((mx >> sh) >= (uint32_t)kSelBins) sh += 12;
        uint32_t prefix = 0, kk = (uint32_t)K;
        bool first = true;
        for (; sh >= 0; sh -= 12) {
            select_pass(s_area, M, sh, first, prefix, kk, s_hist, s_w, s_v);
            prefix = (prefix << 12) | s_v[1];
            kk = s_v[2];
            first = false;
            __syncthreads();
        }
        astar = prefix;
        // kk of the components with area == a* are needed; unless that is all of them the choice is libstdc++'s
        uint32_t eq = 0;
        for (uint32_t i = tid; i < M; i += 1024) eq += (uint32_t)(s_area[i] == astar);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) eq += (uint32_t)__shfl_xor(eq, off);
        if (lane == 0 && eq) atomicAdd(&s_v[4], eq);
        __syncthreads();
        if (s_v[4] != kk) {
            // tie at the cut: hand the host the candidates in ascending-leader order.  Same bucket ranking as step 3
            // below, applied to every candidate: rank = position in the leader-sorted list (a bitonic sort of the
            // pairs took 30 us here).
            if (M > (uint32_t)kSelSortCap) {
                if (tid == 0) c.counters[3] = 2u;
                return;
            }
            int bsh = 0;
            while ((((uint32_t)c.N - 1u) >> bsh) >= (uint32_t)kSelBuckets) bsh++;
            for (int i = tid; i < 3 * kSelBuckets; i += 1024) s_bcnt[i] = 0;
            __syncthreads();
            for (uint32_t i = tid; i < M; i += 1024) atomicAdd(&s_bcnt[s_lead[i] >> bsh], 1u);
            __syncthreads();
            {
                const uint32_t a = s_bcnt[2 * tid], b = s_bcnt[2 * tid + 1];
                uint32_t tot;
                const uint32_t ex = block_excl_scan_1024(a + b, s_w, tot);
                s_bstart[2 * tid] = ex;
                s_bstart[2 * tid + 1] = ex + a;
            }
            __syncthreads();
            for (uint32_t i = tid; i < M; i += 1024) {
                const uint32_t b = s_lead[i] >> bsh;
                s_key[s_bstart[b] + atomicAdd(&s_bfill[b], 1u)] = s_lead[i];
            }
            __syncthreads();
            for (uint32_t i = tid; i < M; i += 1024) {
                const uint32_t me = s_lead[i];
                const uint32_t b = me >> bsh;
                const uint32_t lo = s_bstart[b], hi = lo + s_bcnt[b];
                uint32_t r = lo;
                for (uint32_t j = lo; j < hi; ++j) r += (uint32_t)(s_key[j] < me);
                c.cand_leader[r] = (int32_t)me;
                c.cand_area[r] = s_area[i];
            }
            if (tid == 0) c.counters[3] = 1u;
            return;
        }
    }
    SEL_STAMP(2);
    // 3. rank of every kept leader among the kept leaders (labels 0,1,2.. in raster order of the first pixel,
    //    src/cca.cpp:229-237): bucket by raster-index range, exclusive scan of the bucket sizes, rank inside the bucket
    int bsh = 0;
    while ((((uint32_t)c.N - 1u) >> bsh) >= (uint32_t)kSelBuckets) bsh++;
    for (int i = tid; i < 3 * kSelBuckets; i += 1024) s_bcnt[i] = 0;     // bcnt, bstart, bfill (the histogram is dead)
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024)
        if (s_area[i] >= astar) atomicAdd(&s_bcnt[s_lead[i] >> bsh], 1u);
    __syncthreads();
    {
        const uint32_t a = s_bcnt[2 * tid], b = s_bcnt[2 * tid + 1];
        uint32_t n_keep;
        const uint32_t ex = block_excl_scan_1024(a + b, s_w, n_keep);
        s_bstart[2 * tid] = ex;
        s_bstart[2 * tid + 1] = ex + a;
    }
    __syncthreads();
    SEL_STAMP(3);
    for (uint32_t i = tid; i < M; i += 1024)
        if (s_area[i] >= astar) {
            const uint32_t b = s_lead[i] >> bsh;
            s_key[s_bstart[b] + atomicAdd(&s_bfill[b], 1u)] = s_lead[i];
        }
    __syncthreads();
    SEL_STAMP(4);
    for (uint32_t i = tid; i < M; i += 1024)
        if (s_area[i] >= astar) {
            const uint32_t me = s_lead[i];
            const uint32_t b = me >> bsh;
            const uint32_t lo = s_bstart[b], hi = lo + s_bcnt[b];
            uint32_t r = lo;
            for (uint32_t j = lo; j < hi; ++j) r += (uint32_t)(s_key[j] < me);
            c.final_lab[me] = (uint16_t)r;
        }
    if (tid == 0) c.counters[3] = 0u;
    SEL_STAMP(5);
#undef SEL_STAMP
}

void launch_cca_phase1(const CcaDev& c, int nframes, hipStream_t st) {
    constexpr int T = kCcaTile;
    const int tx = (c.W + T - 1) / T, ty = (c.H + kCcaTileH - 1) / kCcaTileH;
    hipLaunchKernelGGL(k_cca_local, dim3(tx, ty, nframes), dim3(256), 0, st, c);
    const int nh = (ty - 1) * c.W, nv = (tx - 1) * c.H;
    if (nh + nv > 0)
        hipLaunchKernelGGL(k_cca_border, dim3((nh + nv + 255) / 256, nframes), dim3(256), 0, st, c, nh, nv);
    hipLaunchKernelGGL(k_cca_compress, dim3(tx * ty, nframes), dim3(256), 0, st, c);
    hipLaunchKernelGGL(k_cca_collect, dim3(tx * ty, nframes), dim3(256), 0, st, c);
}

void launch_cca_select(const CcaDev& c, int nframes, hipStream_t st, unsigned long long* dbg) {
    // LDS capacity: no frame has more candidates than N / max(threshold, 1)
    long long need = (long long)c.N / (c.min_threshold > 1 ? c.min_threshold : 1) + 1;
    int cap = 1024;
    while (cap < need && cap < kSelCap) cap <<= 1;
    if (cap > kSelCap) cap = kSelCap;
    const size_t lds = (3 * (size_t)cap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t);
    static_assert(kSelBins <= 3 * kSelBuckets, "the selection histogram reuses the ranking buckets");
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cca_select), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                           (int)((3 * (size_t)kSelCap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t)));
    (void)attr_once;
    hipLaunchKernelGGL(k_cca_select, dim3(nframes), dim3(1024), lds, st, c, cap, dbg);
}

// Host path only: forget what the device select / chain wrote, then set the host's choice.
__global__ __launch_bounds__(256) void k_cca_reset_final(CcaDev c) {
    constexpr int TT = kCcaTilePx;
    const int tile = blockIdx.x;
    const uint32_t n = c.tile_cnt[tile];
    const int32_t* __restrict__ my_roots = c.roots + (size_t)tile * TT;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t r = my_roots[i];
        if (c.parent[r] == r) c.final_lab[r] = 0xFFFFu;
    }
}
__global__ void k_cca_set_kept(CcaDev c, const int32_t* __restrict__ leaders, const uint16_t* __restrict__ labels, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c.final_lab[leaders[i]] = labels[i];
}
void launch_cca_set_kept(const CcaDev& c, const int32_t* d_leaders, const uint16_t* d_labels, int n, hipStream_t st) {
    constexpr int T = kCcaTile;
    const int ntiles = ((c.W + T - 1) / T) * ((c.H + kCcaTileH - 1) / kCcaTileH);
    hipLaunchKernelGGL(k_cca_reset_final, dim3(ntiles), dim3(256), 0, st, c);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cca_set_kept, dim3((n + 255) / 256), dim3(256), 0, st, c, d_leaders, d_labels, n);
}

// Every component that was not kept takes the final label of the component left of (or, in column 0,
// above) its leader (src/cca.cpp:240-254).  Leaders strictly decrease along the chase, so it terminates.
__global__ __launch_bounds__(256) void k_cca_chain(CcaDev c) {
    c.select(blockIdx.y);
    constexpr int TT = kCcaTilePx;
    const int tile = blockIdx.x;
    const uint32_t n = c.tile_cnt[tile];
    const int32_t* __restrict__ my_roots = c.roots + (size_t)tile * TT;
    const int W = c.W;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int32_t r = my_roots[i];
        if (c.parent[r] != r) continue;
        if (__hip_atomic_load(c.final_lab + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0xFFFFu) continue;   // kept
        int32_t cur = r;
        uint32_t f;
        for (;;) {
            if (cur == 0) { f = 0; break; }        // src/cca.cpp:238
            const int32_t t = (cur % W > 0) ? cur - 1 : cur - W;   // src/cca.cpp:243-248
            const int32_t g = c.parent[c.parent[t]];
            f = __hip_atomic_load(c.final_lab + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f != 0xFFFFu) break;
            cur = g;
        }
        __hip_atomic_store(c.final_lab + r, (uint16_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(256) void k_cca_relabel(CcaDev c) {
    c.select(blockIdx.y);
    const int gstride = gridDim.x * blockDim.x;
    const int nquad = c.N >> 2;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += gstride) {
        const int4 lr = reinterpret_cast<const int4*>(c.parent)[q];
        const uint32_t a = c.final_lab[c.parent[lr.x]], b = c.final_lab[c.parent[lr.y]];
        const uint32_t d = c.final_lab[c.parent[lr.z]], e = c.final_lab[c.parent[lr.w]];
        uint2 o;
        o.x = a | (b << 16);
        o.y = d | (e << 16);
        reinterpret_cast<uint2*>(c.out)[q] = o;
    }
    for (int p = (nquad << 2) + blockIdx.x * blockDim.x + threadIdx.x; p < c.N; p += gstride)
        c.out[p] = c.final_lab[c.parent[c.parent[p]]];
}

void launch_cca_phase2(const CcaDev& c, int nframes, hipStream_t st) {
    constexpr int T = kCcaTile;
    const int ntiles = ((c.W + T - 1) / T) * ((c.H + kCcaTileH - 1) / kCcaTileH);
    hipLaunchKernelGGL(k_cca_chain, dim3(ntiles, nframes), dim3(256), 0, st, c);
    int blocks = (c.N / 4 + 255) / 256;
    const int cap = (4096 + nframes - 1) / nframes;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_cca_relabel, dim3(blocks, nframes), dim3(256), 0, st, c);
}

}  // namespace fslic
