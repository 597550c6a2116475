// lsc.hip -- the LSC variant (src/lsc.cpp): 10-D feature map, feature-space assignment, weighted centroid update.
// Hand-written gfx950 (CDNA4, wave64) kernels; see kernels.h for the launch interface and DESIGN.md for the design.
//
// Unlike the integer SLIC path this one is f32 and cannot be bit-identical to the reference: the reference's feature
// means and centroid sums are sequential f32 accumulations in raster order (src/lsc.cpp:143-149, 256-269) whose
// rounding depends on that order.  What is mirrored exactly: the tables (built on the host with the reference's own
// expressions), the FMA contraction of the reference build in the weight chain, the visit order (phase, k) with a strict
// comparison, NaN centroids of memberless clusters, and the integer centroid update.  Evaluated differently, inside the
// same tolerance (the fixture agreement did not move, profiles/r03_lsc_recompute.txt): features times the reciprocal of the
// weight instead of divided by it, the distance around a block origin instead of subtract-and-square, sum F for sum w * (F / w).
// Everything here is deterministic (fixed reduction trees, order-independent fixed point for every sum).
#include "device_common.h"
#include <cfloat>

namespace fslic {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kLscRows = 4;                               // visited rows per wavefront of the assign kernel
constexpr int kLscBH = kWavesPerBlock * kLscRows;         // visited rows per block

// ---- feature map ----------------------------------------------------------------------------
// Histograms of the three colour planes.  The colour features are table look-ups of one byte, so their means
// follow from 3 x 256 counts (exactly, in double) instead of a pass over 6 float planes.
__global__ __launch_bounds__(256) void k_lsc_hist(FrameDev f, LscDev l) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    // eight lane-striped copies of every bin: neighbouring pixels mostly share their values, and same-address LDS
    // atomics of one wavefront serialise
    __shared__ uint32_t h[768 * 8];
    for (int i = threadIdx.x; i < 768 * 8; i += 256) h[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 4096;
    const uint32_t cp = threadIdx.x & 7u;
    uint32_t px[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {                       // all loads in flight before the first atomic
        const size_t i = base + (size_t)u * 256 + threadIdx.x;
        px[u] = f.lab[i < (size_t)f.N ? i : (size_t)f.N - 1];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const size_t i = base + (size_t)u * 256 + threadIdx.x;
        if (i < (size_t)f.N) {
            const uint32_t p = px[u];
            atomicAdd(&h[(p & 255u) * 8 + cp], 1u);
            atomicAdd(&h[(256 + ((p >> 8) & 255u)) * 8 + cp], 1u);
            atomicAdd(&h[(512 + ((p >> 16) & 255u)) * 8 + cp], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256) {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) v += h[i * 8 + ((c + threadIdx.x) & 7)];
        if (v) atomicAdd(&l.hist[i], v);
    }
}

// The ten feature means (src/lsc.cpp:143-149; the reference sums sequentially in f32, this is the exact mean).
__global__ __launch_bounds__(256) void k_lsc_means(FrameDev f, LscDev l) {
    f.select(blockIdx.x);
    l.select(blockIdx.x);
    __shared__ double red[kLscFeat][256];
    const int t = threadIdx.x;
    const double hL = l.hist[t], ha = l.hist[256 + t], hb = l.hist[512 + t];
    double a[kLscFeat];
    a[0] = hL * (double)l.lut_color[t];
    a[1] = hL * (double)l.lut_color[256 + t];
    a[2] = ha * (double)l.lut_color[512 + t];
    a[3] = ha * (double)l.lut_color[768 + t];
    a[4] = hb * (double)l.lut_color[512 + t];
    a[5] = hb * (double)l.lut_color[768 + t];
    a[6] = a[7] = a[8] = a[9] = 0.0;
    for (int x = t; x < f.W; x += 256) { a[6] += (double)l.lut_x[x]; a[7] += (double)l.lut_x[f.W + x]; }
    for (int y = t; y < f.H; y += 256) { a[8] += (double)l.lut_y[y]; a[9] += (double)l.lut_y[f.H + y]; }
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) red[q][t] = a[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int q = 0; q < kLscFeat; ++q) red[q][t] += red[q][t + s];
        }
        __syncthreads();
    }
    if (t < kLscFeat) {
        double v = red[t][0];
        if (t == 6 || t == 7) v *= (double)f.H;      // every row repeats the x table, every column the y table
        if (t == 8 || t == 9) v *= (double)f.W;
        l.means[t] = (float)(v / (double)f.N);
    }
}

// ---- features on the fly ----------------------------------------------------------------------
// The reference materialises the ten features of every pixel, divided by the pixel's weight (src/lsc.cpp:103-162: 40 B of
// feature planes + 4 B of weight per pixel).  They are table look-ups of the pixel's L, a, b bytes, its column and its row,
// so the kernels below rebuild them from the 4-byte LAB word where they need them: weight w = sum_f mean_f * F_f (FMA chain
// from zero, as the reference build contracts src/lsc.cpp:156-158) and the features times 1 / w (v_rcp_f32 + one Newton
// step; the reference divides -- the quotients differ by at most an ulp, far below what the order of the reference's own f32
// centroid sums moves them).  Measured against the materialised planes this path started with (8 x 1280x720 K=1600 per
// launch, profiles/r03_lsc_recompute.txt): both numbers are in DESIGN.md.
// Colour tables in LDS as (cos, sin) pairs: word pair v = L table, 256 + v = a/b table -- one ds_read_b64 per byte.
static __device__ __forceinline__ void lsc_stage_lut(const LscDev& l, float* s_lut) {       // 256 threads
    const int t = threadIdx.x;
    reinterpret_cast<float2*>(s_lut)[t] = make_float2(l.lut_color[t], l.lut_color[256 + t]);
    reinterpret_cast<float2*>(s_lut)[256 + t] = make_float2(l.lut_color[512 + t], l.lut_color[768 + t]);
}
// minimum over the 64 lanes (DPP steps inside the 16-lane rows, four readlanes across them)
static __device__ __forceinline__ uint32_t lsc_wave_min(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));     // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));     // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}
static __device__ __forceinline__ float lsc_uniform(float v) { return __uint_as_float(rfl(__float_as_uint(v))); }
// two pixels of one column (LAB words p0, p1; rows with the y features y1, y2): features / weight - org as packed pairs, the weights
static __device__ __forceinline__ void lsc_pair_features(const float* s_lut, uint32_t p0, uint32_t p1, float x1, float x2, f2 y1, f2 y2,
                                                         const float* mean, const float* org, f2* G, f2& w) {
    const float2* tl = reinterpret_cast<const float2*>(s_lut);
    const float2 L0 = tl[p0 & 255u], L1 = tl[p1 & 255u];
    const float2 A0 = tl[256u + ((p0 >> 8) & 255u)], A1 = tl[256u + ((p1 >> 8) & 255u)];
    const float2 B0 = tl[256u + ((p0 >> 16) & 255u)], B1 = tl[256u + ((p1 >> 16) & 255u)];
    const f2 F[kLscFeat] = {{L0.x, L1.x}, {L0.y, L1.y}, {A0.x, A1.x}, {A0.y, A1.y}, {B0.x, B1.x}, {B0.y, B1.y}, {x1, x1}, {x2, x2}, y1, y2};
    f2 ww = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) ww = __builtin_elementwise_fma((f2){mean[q], mean[q]}, F[q], ww);
    f2 r = {__builtin_amdgcn_rcpf(ww[0]), __builtin_amdgcn_rcpf(ww[1])};
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(-ww, r, (f2){1.0f, 1.0f}), r, r);
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) G[q] = __builtin_elementwise_fma(F[q], r, (f2){-org[q], -org[q]});      // F / w - origin
    w = ww;
}
// one pixel, tables read from global memory (the seed kernel: 169 pixels per cluster)
static __device__ __forceinline__ void lsc_pixel_features(const LscDev& l, uint32_t p, int x, int y, int W, int H, const float* mean, float* G) {
    const uint32_t L = p & 255u, A = (p >> 8) & 255u, B = (p >> 16) & 255u;
    const float F[kLscFeat] = {l.lut_color[L], l.lut_color[256 + L], l.lut_color[512 + A], l.lut_color[768 + A], l.lut_color[512 + B],
                               l.lut_color[768 + B], l.lut_x[x], l.lut_x[W + x], l.lut_y[y], l.lut_y[H + y]};
    float w = 0.0f;
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) w = __builtin_fmaf(mean[q], F[q], w);
    float r = __builtin_amdgcn_rcpf(w);
    r = __builtin_fmaf(__builtin_fmaf(-w, r, 1.0f), r, r);
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) G[q] = F[q] * r;
}

static __device__ __forceinline__ float wave_sum(float v) { return wave_reduce_add(v); }
// 4 privatised copies of every accumulator (by quad): lanes of a run of columns with one label meet on the same words, and
// same-address LDS atomics are served one lane after the other (2 copies 45.5, 4 copies 44.1 us per launch of 8 x 1280x720; 8 copies
// for lists of up to 32 candidates, chosen at run time: 45.4)
constexpr int kLscCopyShift = 2;
constexpr int kLscAccEntries = 64 << kLscCopyShift;
// A block's accumulators (LDS), per (candidate, copy) seven 64-bit words: the eleven float sums in a coarser fixed point that keeps a
// block's total (1024 pixels) inside 32 bits (LscDev::blk_fix_*: 2^-15 and 2^-8 for the default constants), two to a word -- the low
// half always one of the colour features or the BIASED weight (LscDev::blk_bias_w), which are never negative, so no borrow crosses into the high half, where
// the spatial features (cos / sin of any angle) wrap in two's complement -- and the six integer sums in the bits that leaves:
//   word 0..3: F0..F3 | F6..F9 << 32      word 4: F4 | F5 << 32
//   word 5: sum L | sum a << 18 | sum b << 36 | count << 54      (255 * 1024 < 2^18; the count has the ten bits that are left: a block
//           of 1024 pixels that is ONE cluster's wraps to 0 and is told from an empty one by its column sum, 16 * (0 + ... + 63))
//   word 6: w | sum (x - x0) << 32 | sum (row inside the block) << 48      (63 * 1024 < 2^16, 15 * 1024 < 2^14)
// 7 LDS atomics per flush instead of 13 (8 until the weight moved in with the positions: an LDS atomic costs its issue whatever the
// number of lanes that take part, profiles/r05_lsc_knockout.txt).
// Word-major: word q of record r = (candidate, copy) lies at q * kLscAccEntries + r.  One atomic instruction adds word q of every
// lane's record, so the lanes of an instruction hit neighbouring 8-byte words; record-major (records 64 bytes apart) put word q of
// ANY record on one of two bank pairs and queued a wavefront's 16 quad leaders 8 deep (profiles/r05_lsc_knockout.txt).
constexpr int kLscAccW = 7;
static_assert(kLscBH == 16 && kTileW == 64, "the integer fields of words 5 and 6 are sized for blocks of 64 x 16 pixels: count <= 1024 (ten bits + the wrap rule), "
                                            "column sum <= 16 * 2016 < 2^16, row sum <= 64 * 120 < 2^14, colour sums <= 255 * 1024 < 2^18");
static __device__ __forceinline__ int lsc_acc_index(int word, uint32_t record) { return word * kLscAccEntries + (int)record; }
constexpr uint32_t kLscNoSlot = 0xFFFFFFFFu;
constexpr int kDppQuadSwap1 = 0xB1, kDppQuadSwap2 = 0x4E;     // quad_perm [1,0,3,2] and [2,3,0,1]

// Seed centroids: unweighted mean of the features over the (2*(S/4)+1)^2 window around the centre, clipped to the
// image (src/lsc.cpp:165-195).  One wavefront per cluster.
__global__ __launch_bounds__(256) void k_lsc_seed(FrameDev f, LscDev l) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = LANE();
    if (k >= f.K) return;
    const uint32_t yx = f.cl_yx[k];
    const int cy = (int)(yx >> 16), cx = (int)(yx & 0xFFFFu), q4 = f.S / 4;
    const int y_lo = max(cy - q4, 0), y_hi = min(cy + q4 + 1, f.H);
    const int x_lo = max(cx - q4, 0), x_hi = min(cx + q4 + 1, f.W);
    const int nw = x_hi - x_lo, n = nw * (y_hi - y_lo);
    float mean[kLscFeat];
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) mean[q] = lsc_uniform(l.means[q]);
    float acc[kLscFeat];
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) acc[q] = 0.0f;
    for (int t = lane; t < n; t += 64) {
        const int yy = y_lo + t / nw, xx = x_lo + t % nw;
        float G[kLscFeat];
        lsc_pixel_features(l, f.lab[(size_t)yy * f.W + xx], xx, yy, f.W, f.H, mean, G);
#pragma unroll
        for (int q = 0; q < kLscFeat; ++q) acc[q] += G[q];
    }
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) {
        float* c = l.cfeat + (size_t)k * kLscCfPitch;
        const float cnt = (float)n;
#pragma unroll
        for (int q = 0; q < kLscFeat; ++q) c[q] = acc[q] / cnt;
    }
}

// ---- assignment -----------------------------------------------------------------------------
// Block = 64 columns x (4 wavefronts x kLscRows visited rows).  Wavefront 0 collects the clusters whose window can
// reach the block from the coarse bins, orders them by the reference's visit order (phase, k) and stages their
// feature-space centroids in LDS; every wavefront then walks the list with its pixels' features in registers.
// A pixel takes cluster k iff it lies in k's window and dist < best so far (strict, src/lsc.cpp:217; a NaN
// centroid therefore never wins and +inf never beats the initial FLT_MAX).  More than 64 candidates (piled-up
// clusters): the block walks all K clusters in visit order instead.
#if defined(FSLIC_LAB) && FSLIC_LAB == 4      // lab build 4: phase time stamps (100 MHz ticks since the block's start), 16 words per (pass, block)
#define LSC_STAMP(i_, cond_) do { if ((cond_) && lane == 0) { asm volatile("" ::: "memory"); lsc_w[(i_)] = (uint32_t)__builtin_amdgcn_s_memrealtime() - lsc_t0; asm volatile("" ::: "memory"); } } while (0)
#else
#define LSC_STAMP(i_, cond_) do { } while (0)
#endif
template <bool stale_sums>
static __device__ __forceinline__ void lsc_assign_body(FrameDev& f, LscDev& l, int rem, int stride, int Hv) {
    f.select(blockIdx.z);
    l.select(blockIdx.z);
    __shared__ uint32_t s_yx[64], s_k[64];
    __shared__ __attribute__((aligned(16))) float s_cf[64 * 12];      // per candidate: centroid - origin (10), its squared norm, pad
    __shared__ __attribute__((aligned(16))) float s_org[12];          // the origin
    __shared__ __attribute__((aligned(8))) float s_lut[1024];
    __shared__ int s_cnt;
    // block-level accumulators of the fused update: per candidate of the block's list (and privatised copy) 11 feature sums
    // in fixed point and the 6 integer sums packed beside them (kLscAccW above: 1024 pixels per block bound every field);
    // LDS atomics, flushed with ONE set of global atomics per (block, cluster)
    __shared__ unsigned long long s_accx[stale_sums ? kLscAccEntries * kLscAccW : 1];
    const int wave = threadIdx.x >> 6, lane = LANE();
    const int S = f.S, W = f.W;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kLscBH;
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
    const uint32_t lsc_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    volatile uint32_t* const lsc_w = f.lab_stamps + ((size_t)(f.gen_off - 1u) * 2048u + (size_t)(blockIdx.y * gridDim.x + blockIdx.x)) * 16u;
    if (wave == 0 && lane == 0) lsc_w[0] = lsc_t0;
#endif
    if (stale_sums)
        for (int i = threadIdx.x; i < kLscAccEntries * kLscAccW; i += 256) s_accx[i] = 0ull;
    lsc_stage_lut(l, s_lut);
    const int x = x0 + lane;
    const bool xok = x < W;
    const int vw = (int)rfl((uint32_t)(v0 + wave * kLscRows));
    int yr[kLscRows];
    uint32_t pix[kLscRows];
    const int xc = min(x, W - 1);
    float xcos, xsin, yf_mine = 0.0f, mean_raw[kLscFeat];
    // ---- the block's candidate list, second form (round 6).  Until then wavefront 0 built it alone -- six bin slots per lane and trip, the
    // (phase, k) ranks, then the centroid records of the listed clusters in a dependent round trip -- while the other three waited at
    // the barrier: 6.2 of a block's 22 us (profiles/r03_lab_lsc_phases.txt).  Now, as in the SLIC block kernel (assign.hip, second form):
    //   A  every thread takes ONE 16-byte slot of the cells the block's window can see (all of them in one trip at S >= 16), keeps the
    //      live records inside the window, and REQUESTS THE CENTROID RECORD of the first cluster it keeps at once (three 16-byte loads that
    //      stay in flight across the first barrier; further records of the same thread: s_ext, below); positions inside the wavefront by
    //      ballot, the counts meet in LDS
    //   B  kept records go to their arrival index (wavefront 0's first, ...): {yx, tag} and the ten centroid features
    //   C  wavefront 0, lane j = arrival j: ranks by (phase, k), the origin (first finite centroid in VISIT order, so the rounding does
    //      not depend on the arrival order), centroid - origin and its squared norm staged by rank -- from LDS, no memory round trip
    __shared__ __attribute__((aligned(16))) float s_cfraw[64 * 12];      // centroid records by arrival index
    __shared__ uint2 s_arr[64];                                           // {yx, tag} by arrival index
    __shared__ uint32_t s_wc[kWavesPerBlock];
    // {yx, tag | position << 20} of the records a thread kept beyond its first, 64 per wavefront: in the bytes of s_cf (3 KB), which is
    // written only behind the second barrier, when these have been consumed -- an array of its own (2 KB) took the block's LDS from
    // 25.8 to 27.8 KB, i.e. from six resident blocks per CU to five
    uint2 (* const s_ext)[64] = reinterpret_cast<uint2 (*)[64]>(s_cf);
    static_assert(sizeof(s_cf) >= kWavesPerBlock * 64 * sizeof(uint2), "s_ext lives in s_cf");
    {
        const int v_last = min(v0 + kLscBH, Hv) - 1;
        TileWindow win;
        win.wy_lo = rem + stride * v0 - S; win.wy_hi = rem + stride * v_last + S;
        win.wx_lo = x0 - S; win.wx_hi = min(x0 + 63, W - 1) + S;
        const uint32_t gen = f.generation();
        const int sh = f.cs_shift;
        const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, W - 1) >> sh;
        const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
        const uint32_t ncxr = (uint32_t)(cxb - cxa + 1);
        const uint32_t nslots = ncxr * (uint32_t)(cyb - cya + 1) * (uint32_t)kCellCap;
        const uint32_t novf = *f.ovf_cnt[0];
        uint32_t my_yx = 0, my_tag = 0, my_pos = 0;
        bool mine_kept = false;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        int cntw = 0, cnte = 0;
        // A thread holds ONE centroid record in registers: that of the first record it keeps.  A further kept record of the same thread
        // (more than 256 slots, i.e. S < 16, or the spill list) goes to the wavefront's s_ext list with its position, and its centroid
        // record is fetched after the first barrier, the dependent round trip the first form paid for every record.  Which thread meets
        // which record is the bin pass's insertion order, i.e. it varies from run to run: the LIST must not (until this was added a second
        // record made the block walk over every cluster -- exact, but with the roundings of the other distance form, so that ties of
        // a constant frame fell differently from run to run: tests/fuzz_parity.py, lsc mode, seed 1 case 146).
        auto visit = [&](const ClusterRec& rec, bool keep) {
            const unsigned long long m = ballot(keep);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)cntw));
            const bool extra = keep && mine_kept;
            const unsigned long long me = ballot(extra);
            if (me) {                                  // (uniform)
                const uint32_t epos = __builtin_amdgcn_mbcnt_hi((uint32_t)(me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me, (uint32_t)cnte));
                if (extra && epos < 64u && pos < 64u) s_ext[wave][epos] = make_uint2(rec.yx, rec.tag | (pos << 20));
                cnte += __popcll(me);
            }
            if (keep && !mine_kept) {
                mine_kept = true; my_yx = rec.yx; my_tag = rec.tag; my_pos = pos;
                const float4* src = reinterpret_cast<const float4*>(l.cfeat + (size_t)(rec.tag & 0xFFFFu) * kLscCfPitch);
                q0 = src[0]; q1 = src[1]; q2 = src[2];
            }
            cntw += __popcll(m);
        };
        auto slot_rec = [&](uint32_t e) -> ClusterRec {
            const uint32_t cell = e < nslots ? e >> 3 : 0u;
            const uint32_t cyi = cell / ncxr, cxi = cell - cyi * ncxr;
            return f.cell_items[0][((size_t)(cya + (int)cyi) * f.ncx + cxa + (int)cxi) * kCellCap + (e & 7u)];
        };
        const ClusterRec rec_first = slot_rec((uint32_t)threadIdx.x);
        // the wavefront's pixel loads, issued BEHIND the slot load the list waits for (the vector-memory counter is in order: a wait for
        // the slot then leaves these in flight, and they stay in flight across the list's barriers; nothing needs them until the walk)
#pragma unroll
        for (int r = 0; r < kLscRows; ++r) {
            yr[r] = rem + stride * min(vw + r, Hv - 1);
            pix[r] = f.lab[(size_t)yr[r] * W + xc];
        }
        // ... and with them everything else the walk needs from memory (the column's x features, the feature means, this lane's y feature):
        // requested after the list's barriers (until this round) they were what the first row pair waited for -- its walk took 4.5 us,
        // the second pair's 2.4 (lab build 4 stamps)
        xcos = l.lut_x[xc]; xsin = l.lut_x[W + xc];
#pragma unroll
        for (int q = 0; q < kLscFeat; ++q) mean_raw[q] = l.means[q];
        if (lane < 2 * kLscRows) {
            const int r = (lane >> 2) * 2 + (lane & 1);
            yf_mine = l.lut_y[((lane >> 1) & 1) * f.H + rem + stride * min(vw + r, Hv - 1)];
        }
        visit(rec_first, (uint32_t)threadIdx.x < nslots && rec_first.pad == gen && in_window(rec_first, win));
        for (uint32_t e0 = 256u; e0 < nslots; e0 += 256u) {
            const uint32_t e = e0 + (uint32_t)threadIdx.x;
            const ClusterRec rec = slot_rec(e);
            visit(rec, e < nslots && rec.pad == gen && in_window(rec, win));
        }
        if (wave == kWavesPerBlock - 1) {
            for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
                const uint32_t i = i0 + (uint32_t)lane;
                ClusterRec rec;
                rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
                if (i < novf) rec = f.ovf_items[0][i];
                visit(rec, i < novf && in_window(rec, win));
            }
        }
        if (lane == 0) s_wc[wave] = (uint32_t)min(cntw, 0xFFFF);
        LSC_STAMP(1, wave == 0);                // wave 0: its slots visited, centroid records requested
        lds_barrier();
        const uint32_t n0 = s_wc[0], n1 = s_wc[1], n2 = s_wc[2], n3 = s_wc[3];
        const uint32_t bn = n0 + n1 + n2 + n3;                       // (the number of live records inside the window: the same in every run)
        const uint32_t off = wave == 0 ? 0u : wave == 1 ? n0 : wave == 2 ? n0 + n1 : n0 + n1 + n2;
        if (bn <= 64u) {
            if (mine_kept) {
                const uint32_t idx = off + my_pos;
                s_arr[idx] = make_uint2(my_yx, my_tag);
                float4* dst = reinterpret_cast<float4*>(s_cfraw + idx * 12);
                dst[0] = q0; dst[1] = q1; dst[2] = q2;
            }
            if (cnte > 0 && lane < cnte) {             // this wavefront's further records (cnte <= bn <= 64): fetched now
                const uint2 ex = s_ext[wave][lane];
                const uint32_t tag = ex.y & 0xFFFFFu, idx = off + (ex.y >> 20);
                const float4* src = reinterpret_cast<const float4*>(l.cfeat + (size_t)(tag & 0xFFFFu) * kLscCfPitch);
                const float4 e0 = src[0], e1 = src[1], e2 = src[2];
                s_arr[idx] = make_uint2(ex.x, tag);
                float4* dst = reinterpret_cast<float4*>(s_cfraw + idx * 12);
                dst[0] = e0; dst[1] = e1; dst[2] = e2;
            }
        }
        lds_barrier();
        if (wave == 0) {
            const int cnt = (int)bn;
            if (cnt <= 64) {
                const bool act = lane < cnt;
                const uint2 at = act ? s_arr[lane] : make_uint2(0u, 0xFFFFFFFFu);
                const uint32_t mytag = at.y;
                int rank = 0;
                for (int jn = 0; jn < cnt; ++jn) rank += (uint32_t)__builtin_amdgcn_readlane((int)mytag, jn) < mytag ? 1 : 0;
                // The distance is evaluated as |C - O|^2 - 2 (G - O) . (C - O) (+ |G - O|^2, the same for every candidate of a pixel)
                // around an origin O of the block's own neighbourhood -- the first finite centroid of the list -- instead of
                // sum (G - C)^2 (src/lsc.cpp:212-215): 11 FMAs per candidate instead of 10 subtractions + 10 FMAs.  Around O every
                // term is of the size of the distances themselves, so the rounding stays at a few f32 ulps of them (measured on the
                // reference fixtures: the same agreement as the subtract-and-square form, profiles/r03_lsc_recompute.txt).
                const uint32_t k = act ? (mytag & 0xFFFFu) : 0u;
                const float4* src = reinterpret_cast<const float4*>(s_cfraw + (act ? lane : 0) * 12);
                const float4 r0 = src[0], r1 = src[1], r2 = src[2];
                float cf[kLscFeat] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
                float tot = 0.0f;
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) tot += cf[q];
                // (the first in VISIT order: the lanes hold the list in the order the bin pass happened to insert it, which varies
                // from run to run -- an origin picked by lane would move the rounding with it)
                const bool finite = act && (tot - tot) == 0.0f;                                 // (NaN centroid of a memberless cluster: never the origin)
                const uint32_t first = lsc_wave_min(finite ? (uint32_t)rank : 64u);
                const unsigned long long fin = __ballot(finite && (uint32_t)rank == first);
                const int o = fin ? __builtin_ctzll(fin) : 0;
                float nrm = 0.0f;
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) {
                    const float org = fin ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cf[q]), o)) : 0.0f;
                    if (lane == 0) s_org[q] = org;
                    cf[q] -= org;
                    nrm = __builtin_fmaf(cf[q], cf[q], nrm);
                }
                if (act) {
                    s_yx[rank] = at.x;
                    s_k[rank] = k;
                    float4* dst = reinterpret_cast<float4*>(s_cf + rank * 12);
                    dst[0] = make_float4(cf[0], cf[1], cf[2], cf[3]); dst[1] = make_float4(cf[4], cf[5], cf[6], cf[7]);
                    dst[2] = make_float4(cf[8], cf[9], nrm, 0.0f);
                }
            } else if (lane < 12) {
                s_org[lane] = 0.0f;        // the walk over every cluster subtracts and squares, around no origin
            }
            if (lane == 0) s_cnt = cnt;
        }
    }
    // this wavefront's pixels: column x0 + lane of kLscRows visited rows (clamped addresses, masked results), taken two rows
    // (r, r+1) at a time: the feature and distance chains run on packed f32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, two
    // rows per instruction; each half is the same IEEE operation as the scalar form), and a pair's twenty quotients are dead
    // before the next pair's are formed -- all four rows at once cost 104 VGPRs (4 wavefronts per SIMD) with the sums below.
    // the rows' y features, all of them now: fetched pair by pair inside the loop below, the second pair's loads sat behind the first
    // pair's label stores in the vector-memory queue and their s_waitcnt exposed the stores' (written-through) acknowledgements
    // (through LDS: eight more scalar registers spill, eight vector registers cost a wavefront per SIMD)
    float mean[kLscFeat];
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) mean[q] = lsc_uniform(mean_raw[q]);
    __shared__ __attribute__((aligned(16))) float s_yf[4][2 * kLscRows];      // per wavefront and row pair: cos r0, cos r1, sin r0, sin r1
    static_assert(kLscRows == 4, "two row pairs per wavefront");
    if (lane < 2 * kLscRows) s_yf[wave][lane] = yf_mine;      // (requested with the pixel loads, above)
    LSC_STAMP(2, wave == 1);                    // wave 1: its pixel loads issued
    lds_barrier();      // LDS contents only pass between the wavefronts here: pixel loads / label stores stay in flight (device_common.h)
    LSC_STAMP(3, wave == 1);                    // barrier passed
    const int cnt = s_cnt;
    const bool listed = cnt <= 64;                 // a pixel's winner is kept as its position in the block's list (else: the cluster itself)
    const int y_top = yr[0], y_bot = yr[kLscRows - 1];
    // the block's list across the lanes (lane j = candidate j): which candidates a pair of rows has to look at is one ballot,
    // and the walk below meets only those -- no LDS round trip and scalar window tests per listed candidate
    const uint32_t yx_j = (listed && lane < cnt) ? s_yx[lane] : 0u;
    const int cy_j = (int)(yx_j >> 16), cx_j = (int)(yx_j & 0xFFFFu);
    const bool col_j = listed && lane < cnt && !(cx_j + S < x0 || cx_j - S > x0 + 63);
    const float2* const tl = reinterpret_cast<const float2*>(s_lut);
    constexpr int cshift = kLscCopyShift;
    const uint32_t copy = ((uint32_t)lane >> 2) & ((1u << cshift) - 1u);  // by quad: mostly a quad's first lane adds for the four (below)
#pragma unroll
    for (int h = 0; h < kLscRows / 2; ++h) {
        const int r0 = 2 * h, r1 = 2 * h + 1;
        const float4 yv = *reinterpret_cast<const float4*>(&s_yf[wave][4 * h]);
        const f2 ycos = {lsc_uniform(yv.x), lsc_uniform(yv.y)};
        const f2 ysin = {lsc_uniform(yv.z), lsc_uniform(yv.w)};
        f2 G[kLscFeat], wv;
        {
            const float4* po = reinterpret_cast<const float4*>(s_org);
            const float4 o0 = po[0], o1 = po[1], o2 = po[2];
            const float org[kLscFeat] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w, o2.x, o2.y};
#if !(FSLIC_EXP & 32)     // (knock-out build 32: no feature rebuild -- six table reads, the weight chain and the twenty quotients of a row pair)
            lsc_pair_features(s_lut, pix[r0], pix[r1], xcos, xsin, ycos, ysin, mean, org, G, wv);
#else
#pragma unroll
            for (int q = 0; q < kLscFeat; ++q) G[q] = (f2){__uint_as_float(pix[r0] + q) - org[q], __uint_as_float(pix[r1] + q) - org[q]};
            wv = (f2){1.0f, 1.0f};
#endif
        }
        f2 best = {FLT_MAX, FLT_MAX};
        uint32_t bk0 = 0xFFFFu, bk1 = 0xFFFFu;
        // cf: centroid - origin; nrm: its squared norm (listed candidates), or < 0: subtract and square
        auto consider = [&](bool in0, bool in1, int cx, uint32_t id, const float* cf, float nrm, bool dot_form) {      // in0, in1: uniform
            const bool inx = abs(x - cx) <= S;
            f2 d = {0.0f, 0.0f};
            if (dot_form) {
#if !(FSLIC_EXP & 16)     // (knock-out build 16, scripts/gpu_knock.sh: the walk without its ten packed FMAs per candidate; results wrong on purpose)
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) d = __builtin_elementwise_fma(G[q], (f2){cf[q], cf[q]}, d);
#else
                d = G[0] + (f2){cf[0], cf[9]};
#endif
                d = __builtin_elementwise_fma(d, (f2){-2.0f, -2.0f}, (f2){nrm, nrm});
            } else {
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) {
                    const f2 c2 = {cf[q], cf[q]};
                    const f2 diff = G[q] - c2;
                    d = __builtin_elementwise_fma(diff, diff, d);
                }
            }
            if (in0 && inx && d[0] < best[0]) { best[0] = d[0]; bk0 = id; }
            if (in1 && inx && d[1] < best[1]) { best[1] = d[1]; bk1 = id; }
        };
        if (listed) {
            const unsigned long long m0 = __ballot(col_j && abs(yr[r0] - cy_j) <= S), m1 = __ballot(col_j && abs(yr[r1] - cy_j) <= S);
            unsigned long long m = m0 | m1;
            while (m) {                                  // ascending list position = the reference's visit order
                const int c = __builtin_ctzll(m);
                m &= m - 1;
                const int cx = __builtin_amdgcn_readlane(cx_j, c);
                const float4* src = reinterpret_cast<const float4*>(s_cf + c * 12);
                const float4 c0 = src[0], c1 = src[1], c2 = src[2];
                const float cf[kLscFeat] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
                consider((m0 >> c) & 1ull, (m1 >> c) & 1ull, cx, (uint32_t)c, cf, c2.z, true);
            }
        } else {
            // visit order of the reference (phase, then k) over every cluster
            for (uint32_t ph = 0; ph < 4; ++ph) {
                for (int k0 = 0; k0 < f.K; k0 += 64) {
                    const int k = k0 + lane;
                    const bool live = k < f.K;
                    const uint32_t yxl = live ? f.cl_yx[k] : 0u;
                    const uint32_t tg = live ? f.cl_tag[k] : 0xFFFFFFFFu;
                    const int cyl = (int)(yxl >> 16), cxl = (int)(yxl & 0xFFFFu);
                    const bool use = live && (tg >> 16) == ph && !(cxl + S < x0 || cxl - S > x0 + 63 || cyl + S < y_top || cyl - S > y_bot);
                    unsigned long long m = __ballot(use);
                    while (m) {
                        const int c = __builtin_ctzll(m);
                        m &= m - 1;
                        const uint32_t yx = (uint32_t)__builtin_amdgcn_readlane((int)yxl, c);
                        const float* src = l.cfeat + (size_t)(k0 + c) * kLscCfPitch;
                        float cf[kLscFeat];
#pragma unroll
                        for (int q = 0; q < kLscFeat; ++q) cf[q] = lsc_uniform(src[q]);      // (scalar registers: this rare path sets the kernel's VGPR peak)
                        const int cy = (int)(yx >> 16);
                        const bool in0 = abs(yr[r0] - cy) <= S, in1 = abs(yr[r1] - cy) <= S;
                        if (in0 || in1) consider(in0, in1, (int)(yx & 0xFFFFu), (uint32_t)(k0 + c), cf, 0.0f, false);
                    }
                }
            }
        }
        LSC_STAMP(4 + 2 * h, wave == 1);        // pair h: candidates walked
        // labels out; with an update following (stale_sums), also the label each visited pixel carries into update(): the new
        // one, or the one it keeps when nothing accepted it (src/lsc.cpp:217-220 never touches such a pixel; 0xFFFF: none yet)
        uint32_t lbl[2], slot[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = 2 * h + u;
            lbl[u] = 0xFFFFu; slot[u] = kLscNoSlot;
            if (!xok || vw + r >= Hv) continue;
            const uint32_t bk = u ? bk1 : bk0;
            const size_t p = (size_t)yr[r] * W + x;
            if (best[u] < FLT_MAX) {
                lbl[u] = listed ? s_k[bk] : bk;
                st_stream(f.labels + p, (uint16_t)lbl[u]);
                if (listed) slot[u] = bk;
            } else if (stale_sums) {
                lbl[u] = f.labels[p];
                // waited for HERE, inside the rare branch: left to the compiler the wait lands behind the merge of the two branches,
                // where it also holds every wavefront that took the other one until its label store (written through) is
                // acknowledged -- four store round trips per wavefront and pass
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            }
        }
        if (stale_sums) {
            // ---- update(), fused: the sums of src/context.cpp:301-354 (count, y, x, L, a, b) and of src/lsc.cpp:256-269 (sum
            // w*F per feature, sum w) over this block's pixels.  Every lane adds its two pixels of the pair to the block's
            // accumulators with 8 LDS atomics (kLscAccW) -- together when they carry the same label (a superpixel is taller than two
            // rows: nearly always), one after the other otherwise: the float sums in fixed point, so the totals do not depend
            // on the order in which lanes and wavefronts arrive (run-to-run identical), the six integer sums packed into two
            // words.  Lanes of a run of columns with one label meet on the same words: privatised copies (kLscAccEntries).
            // The reference adds w * (F / w) per pixel and feature (src/lsc.cpp:262-264), which is the table value F up to an
            // ulp: the sums take F itself.  Nothing of the sums is live while the candidates are walked (keeping a lane's runs
            // in registers across the pairs saved a third of the atomics and held 16 registers: 5 wavefronts per SIMD instead of 6).
            // (Measured per launch of 8 x 1280x720 K=1600 on the materialised planes, the forms this replaces: a loop over the
            // wavefront's labels with a halving butterfly over the lanes per label 61 us -- 8 labels x (44 masked FMAs + 43
            // cross-lane operations + 17 LDS atomics) per wavefront --, lanes folding their own label first 75 us, full wave sums
            // 87 us, 16-lane row sums 172 us; the kernel that re-scanned every cluster's window for its members 64 of 100 us per
            // iteration.  Column runs + LDS atomics: 49 us, profiles/r03_lsc_sums_ablation.txt.)
            float Fr[2][kLscFeat + 1];
            uint32_t ia[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * h + u;
                const uint32_t px = pix[r];
                const float2 Lr = tl[px & 255u], Ar = tl[256u + ((px >> 8) & 255u)], Br = tl[256u + ((px >> 16) & 255u)];
                Fr[u][0] = Lr.x; Fr[u][1] = Lr.y; Fr[u][2] = Ar.x; Fr[u][3] = Ar.y; Fr[u][4] = Br.x; Fr[u][5] = Br.y;
                Fr[u][6] = xcos; Fr[u][7] = xsin; Fr[u][8] = ycos[u]; Fr[u][9] = ysin[u]; Fr[u][kLscFeat] = wv[u] + l.blk_bias_w;      // (biased: never negative, LscDev::blk_bias_w)
                ia[u][0] = (px & 255u) | ((px & 0xFF00u) << 10);                 // L | a << 18
                ia[u][1] = ((px >> 12) & 0xFF0u) | (1u << 22);                  // (b << 36 | count << 54) >> 32
                ia[u][2] = (uint32_t)lane | ((uint32_t)(wave * kLscRows + r) << 16);      // (x - x0 | the row's number inside the block << 16), the high half of the weight's word
                if (lbl[u] != 0xFFFFu && slot[u] == kLscNoSlot) {
                    // its label is not on the block's list (it kept a label whose window has left it, or the block walks
                    // every cluster): straight to the global accumulators.  Rare.
                    unsigned long long* e = reinterpret_cast<unsigned long long*>(l.extra + (size_t)lbl[u] * kLscExPitch);
#pragma unroll
                    for (int q = 0; q < kLscFeat; ++q) atomicAdd(e + q, (unsigned long long)(long long)(int)__builtin_rintf(Fr[u][q] * kLscFixF));
                    atomicAdd(e + kLscFeat, (unsigned long long)(long long)(int)__builtin_rintf(wv[u] * kLscFixW));
                    global_accumulate(f.sums[0], lbl[u], (uint32_t)yr[r], (uint32_t)x, px);
                }
            }
            const bool both = slot[0] == slot[1];
            if (both) {
#pragma unroll
                for (int q = 0; q <= kLscFeat; ++q) Fr[0][q] += Fr[1][q];
#pragma unroll
                for (int q = 0; q < 3; ++q) ia[0][q] += ia[1][q];
            }
            // Lanes of one label hit the same words, and same-address LDS atomics are served one lane after the other (a
            // 24-column superpixel: a dozen lanes per copy).  Four neighbouring columns mostly agree, so a quad whose lanes all
            // hold one label for both rows adds ONE set of values (two DPP additions per value, fixed order) from its first
            // lane; the other quads -- a label boundary runs through them -- add lane by lane.
            const uint32_t sa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)slot[0], kDppQuadSwap1, 0xF, 0xF, true);
            const uint32_t sb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)slot[0], kDppQuadSwap2, 0xF, 0xF, true);
            const uint32_t sc = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sa, kDppQuadSwap2, 0xF, 0xF, true);
            const unsigned long long bq = __ballot(both);
            const bool quad = ((bq >> (lane & ~3)) & 0xFull) == 0xFull && slot[0] == sa && slot[0] == sb && slot[0] == sc && slot[0] != kLscNoSlot;
            if (__ballot(quad)) {                     // uniform
                float t[kLscFeat + 1];
                uint32_t it[3];
#pragma unroll
                for (int q = 0; q <= kLscFeat; ++q) {
                    t[q] = Fr[0][q] + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(Fr[0][q]), kDppQuadSwap1, 0xF, 0xF, true));
                    t[q] = t[q] + __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(t[q]), kDppQuadSwap2, 0xF, 0xF, true));
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    it[q] = ia[0][q] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ia[0][q], kDppQuadSwap1, 0xF, 0xF, true);
                    it[q] = it[q] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)it[q], kDppQuadSwap2, 0xF, 0xF, true);
                }
                if (quad) {
#pragma unroll
                    for (int q = 0; q <= kLscFeat; ++q) Fr[0][q] = t[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) ia[0][q] = it[q];
                    if (lane & 3) slot[0] = kLscNoSlot;          // the quad's first lane speaks for it
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (slot[u] == kLscNoSlot || (u == 1 && both)) continue;
                                // the copy: by quad for a quad's first lane that speaks for the four, by lane inside a quad a label boundary runs
                // through (its lanes of one label would otherwise meet on one address: 37.7 -> 36.1 us per launch of 8 x 1280x720)
                unsigned long long* a = s_accx + ((slot[u] << cshift) + (quad ? copy : ((uint32_t)lane & ((1u << cshift) - 1u))));
                auto fixed = [&](int q) { return (uint32_t)(int)__builtin_rintf(Fr[u][q] * l.blk_fix_f); };
#if !(FSLIC_EXP & 8)      // (knock-out build 8, scripts/gpu_knock.sh: what the block accumulators' LDS atomics cost; results wrong on purpose)
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicAdd(a + lsc_acc_index(q, 0), (unsigned long long)fixed(q) | ((unsigned long long)fixed(6 + q) << 32));
                atomicAdd(a + lsc_acc_index(4, 0), (unsigned long long)fixed(4) | ((unsigned long long)fixed(5) << 32));
                atomicAdd(a + lsc_acc_index(5, 0), (unsigned long long)ia[u][0] | ((unsigned long long)ia[u][1] << 32));
                atomicAdd(a + lsc_acc_index(6, 0), (unsigned long long)(uint32_t)(int)__builtin_rintf(Fr[u][kLscFeat] * l.blk_fix_w) | ((unsigned long long)ia[u][2] << 32));
#else
                asm volatile("" :: "v"(a), "v"(fixed(0)), "v"(fixed(1)), "v"(fixed(2)), "v"(fixed(3)), "v"(fixed(4)), "v"(fixed(5)), "v"(fixed(6)), "v"(fixed(7)), "v"(fixed(8)), "v"(fixed(9)),
                             "v"(ia[u][0]), "v"(ia[u][1]), "v"(ia[u][2]));
#endif
            }
        }
        LSC_STAMP(5 + 2 * h, wave == 1);        // pair h: labels stored, runs summed
    }
    if (!stale_sums) return;
    LSC_STAMP(8, wave == 1);
    lds_barrier();      // LDS contents only pass between the wavefronts here: pixel loads / label stores stay in flight (device_common.h)
    LSC_STAMP(9, wave == 1);
    // one set of global atomics per (block, cluster that received pixels here).  (Plain stores into per-(cluster, block) entries
    // summed by the cluster pass, as the SLIC block kernel delivers its sums, were built and measured: 48.6 against 47.1 us per launch
    // of 8 x 1280x720 -- the window arithmetic in the prologue costs what the atomics cost at the end.)
    const uint32_t yb0 = (uint32_t)(rem + stride * v0);
    for (int i = threadIdx.x; i < cnt * (kLscFeat + 1 + 6) && listed; i += 256) {
        const int c = i / (kLscFeat + 1 + 6), q = i - c * (kLscFeat + 1 + 6);
        const unsigned long long* a = s_accx + ((size_t)c << cshift);
        unsigned long long A = 0, B = 0;                                      // words 5 and 6 (the copies' words add up field by field)
        for (int cpy = 0; cpy < (1 << cshift); ++cpy) { A += a[lsc_acc_index(5, cpy)]; B += a[lsc_acc_index(6, cpy)]; }
        const uint32_t xs = (uint32_t)(B >> 32) & 0xFFFFu, n10 = (uint32_t)(A >> 54);
        const uint32_t n = n10 ? n10 : (xs ? 1024u : 0u);                     // (ten bits of count: see kLscAccW)
        if (n == 0u) continue;                                              // nothing for this cluster
        const uint32_t k = s_k[c];
        if (q <= kLscFeat) {
            // the value's half of its word (the copies' words add up without a carry between the halves: a block's total fits its half)
            const int word = q < 4 ? q : q < 6 ? 4 : q < kLscFeat ? q - 6 : 6;
            const bool high = q == 5 || (q >= 6 && q < kLscFeat);
            unsigned long long w = 0;
            for (int cpy = 0; cpy < (1 << cshift); ++cpy) w += a[lsc_acc_index(word, cpy)];
            long long field = high ? (long long)(int)(uint32_t)(w >> 32) : (long long)(uint32_t)w;
            if (q == kLscFeat) field -= (long long)n * (long long)(l.blk_bias_w * l.blk_fix_w);      // the bias of n pixels' weights (an integer in this fixed point)
            const long long v = field << (q < kLscFeat ? l.blk_shift_f : l.blk_shift_w);      // to the fixed point of the global accumulators
            atomicAdd(reinterpret_cast<unsigned long long*>(l.extra + (size_t)k * kLscExPitch) + q, (unsigned long long)v);
        } else {
            const int u = q - kLscFeat - 1;
            uint32_t v;
            if (u == 0) v = n;
            else if (u == 1) v = yb0 * n + (uint32_t)stride * ((uint32_t)(B >> 48) & 0x3FFFu);
            else if (u == 2) v = (uint32_t)x0 * n + xs;
            else if (u == 3) v = (uint32_t)A & 0x3FFFFu;
            else if (u == 4) v = (uint32_t)(A >> 18) & 0x3FFFFu;
            else v = (uint32_t)(A >> 36) & 0x3FFFFu;
            atomicAdd(f.sums[0] + 6 * (size_t)k + u, v);
        }
    }
    LSC_STAMP(10, wave == 1);
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
    if (wave == 1 && lane == 0) lsc_w[11] = (uint32_t)cnt;
#endif
}

// Two kernels over one body: the pass that is followed by an update carries the fused sums and takes 81 VGPRs left to itself (5
// wavefronts per SIMD); held to 80 it spills nothing and runs 6 (50.0 against 51.9 us per launch of 8 x 1280x720; 7 at 72 VGPRs: 49.9).
// The final pass needs 53 (8 wavefronts per SIMD).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_lsc_assign(FrameDev f, LscDev l, int rem, int stride, int Hv) {
    lsc_assign_body<false>(f, l, rem, stride, Hv);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_lsc_assign_sums(FrameDev f, LscDev l, int rem, int stride, int Hv) {
    lsc_assign_body<true>(f, l, rem, stride, Hv);
}

// ---- preemptive mode (src/preemptive.h with ContextLSC's hooks) ------------------------------------------------------------------
// The reference's update() sums the visited pixels of the ACTIVE cells as they stood before the pass (src/context.cpp:304-343), then
// PreemptiveGrid::set_new_clusters recomputes the updatable counters, the active clusters and the active cells (src/preemptive.h:114-178),
// and only then ContextLSC::after_update accumulates w * F -- over every visited pixel when all clusters are active, otherwise over the
// pixels of the NEW active cells whose cluster is still updatable (src/lsc.cpp:226-307).  Two different masks around one state change:
// the fused sums of k_lsc_assign_sums cannot serve, so this mode runs the assignment alone (k_lsc_assign) and three small kernels, one
// thread per visited pixel / 16 threads per cluster.  The mode is an approximation meant to save CPU time; here it is served for
// interface completeness, not accelerated.
__global__ __launch_bounds__(256) void k_lsc_pre_sums(FrameDev f, int rem, int stride, int Hv) {
    f.select(blockIdx.y);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Hv * f.W) return;
    const int v = (int)(i / f.W), x = (int)(i - (size_t)v * f.W), y = rem + v * stride;
    const size_t p = (size_t)y * f.W + x;
    const uint32_t lbl = f.labels[p];
    if (lbl != 0xFFFFu && in_update(f, y, x)) global_accumulate(f.sums[0], lbl, (uint32_t)y, (uint32_t)x, f.lab[p]);
}
__global__ __launch_bounds__(256) void k_lsc_pre_feats(FrameDev f, LscDev l, int rem, int stride, int Hv) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Hv * f.W) return;
    const int v = (int)(i / f.W), x = (int)(i - (size_t)v * f.W), y = rem + v * stride;
    const size_t p = (size_t)y * f.W + x;
    const uint32_t lbl = f.labels[p];
    if (lbl == 0xFFFFu) return;
    // (after k_preempt_update: pre_flags / cell_act / cl_upd are the NEW state)
    if (!f.pre_flags[0] && !(f.cell_act[(y / (2 * f.S)) * f.pre_cw + x / (2 * f.S)] && f.cl_upd[lbl])) return;
    const uint32_t px = f.lab[p];
    const uint32_t L = px & 255u, A = (px >> 8) & 255u, B = (px >> 16) & 255u;
    const float F[kLscFeat] = {l.lut_color[L], l.lut_color[256 + L], l.lut_color[512 + A], l.lut_color[768 + A], l.lut_color[512 + B],
                               l.lut_color[768 + B], l.lut_x[x], l.lut_x[f.W + x], l.lut_y[y], l.lut_y[f.H + y]};
    float w = 0.0f;
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) w = __builtin_fmaf(l.means[q], F[q], w);
    // w * (F / w) = F up to an ulp (as the fused sums take it, see the header of this file)
    unsigned long long* e = reinterpret_cast<unsigned long long*>(l.extra + (size_t)lbl * kLscExPitch);
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) atomicAdd(e + q, (unsigned long long)(long long)__float2ll_rn(F[q] * kLscFixF));
    atomicAdd(e + kLscFeat, (unsigned long long)(long long)__float2ll_rn(w * kLscFixW));
}
// C_f of the clusters that are still updatable: sum w F / sum w (0 / 0 = NaN without members, as the reference); of the others: their
// old value with weight 1 plus whatever the pass added -- nothing unless all clusters are active, src/lsc.cpp:234-246, :302-306
__global__ __launch_bounds__(256) void k_lsc_pre_update(FrameDev f, LscDev l) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 4, q = t & 15;
    if (k >= f.K) return;
    long long* e = l.extra + (size_t)k * kLscExPitch;
    const long long mine = q <= kLscFeat ? e[q] : 0ll;
    const long long wsum = e[kLscFeat];
    const float acc = (float)((double)mine / (double)kLscFixF), tot_w = (float)((double)wsum / (double)kLscFixW);
    if (q < kLscFeat) {
        float* c = l.cfeat + (size_t)k * kLscCfPitch + q;
        *c = f.cl_upd[k] ? acc / tot_w : (*c + acc) / (1.0f + tot_w);
    }
    __builtin_amdgcn_wave_barrier();
    if (q <= kLscFeat) e[q] = 0;
}
void launch_lsc_pre_sums(const FrameDev& f_, int nframes, int rem, int stride, int sbuf, hipStream_t st) {
    const FrameDev f = rotated(f_, 0, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    launch(k_lsc_pre_sums, dim3((unsigned)(((size_t)Hv * f.W + 255) / 256), nframes), dim3(256), 0, st, f, rem, stride, Hv);
}
void launch_lsc_pre_feats(const FrameDev& f, const LscDev& l, int nframes, int rem, int stride, hipStream_t st) {
    const int Hv = (f.H - rem + stride - 1) / stride;
    // a pass without visited rows (H <= rem) still recomputes the centroids of the updatable clusters: 0 / 0 = NaN, as
    // ContextLSC::after_update leaves them (src/lsc.cpp:256-269)
    if (Hv > 0) launch(k_lsc_pre_feats, dim3((unsigned)(((size_t)Hv * f.W + 255) / 256), nframes), dim3(256), 0, st, f, l, rem, stride, Hv);
    launch(k_lsc_pre_update, dim3((f.K * 16 + 255) / 256, nframes), dim3(256), 0, st, f, l);
}

// ---- launches -------------------------------------------------------------------------------
void launch_lsc_prepare(const FrameDev& f, const LscDev& l, int nframes, hipStream_t st) {
    const unsigned N = (unsigned)f.N;
    launch(k_lsc_hist, dim3((N + 4095) / 4096, nframes), dim3(256), 0, st, f, l);
    launch(k_lsc_means, dim3(nframes), dim3(256), 0, st, f, l);
    launch(k_lsc_seed, dim3((f.K + 3) / 4, nframes), dim3(256), 0, st, f, l);
}

void launch_lsc_assign(const FrameDev& f_, const LscDev& l, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const dim3 grid((f.W + 63) / 64, (Hv + kLscBH - 1) / kLscBH, nframes);
    if (stale_sums) launch(k_lsc_assign_sums, grid, dim3(256), 0, st, f, l, rem, stride, Hv);
    else launch(k_lsc_assign, grid, dim3(256), 0, st, f, l, rem, stride, Hv);
}

}  // namespace fslic
