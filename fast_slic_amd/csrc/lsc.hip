// lsc.hip -- the LSC variant (src/lsc.cpp): 10-D feature map, feature-space assignment, weighted centroid update.
// Hand-written gfx950 (CDNA4, wave64) kernels; see kernels.h for the launch interface and DESIGN.md for the design.
//
// Unlike the integer SLIC path this one is f32 and cannot be bit-identical to the reference: the reference's feature
// means and centroid sums are sequential f32 accumulations in raster order (src/lsc.cpp:143-149, 256-269) whose
// rounding depends on that order.  What is mirrored exactly: the tables (built on the host with the reference's own
// expressions), the FMA contraction of the reference build in the weight and distance chains, IEEE division, the
// visit order (phase, k) with a strict comparison, NaN centroids of memberless clusters, and the integer centroid
// update.  Everything here is deterministic (fixed reduction trees, order-independent fixed point on the rare path).
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

// Per pixel: ten features from the tables, weight w = sum_f mean_f * F_f (FMA chain from zero, as the reference build
// contracts src/lsc.cpp:156-158), features stored divided by w (src/lsc.cpp:309-316).
__global__ __launch_bounds__(256) void k_lsc_features(FrameDev f, LscDev l) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    __shared__ float lc[1024];
    __shared__ float mean[kLscFeat];
    for (int i = threadIdx.x; i < 1024; i += 256) lc[i] = l.lut_color[i];
    if (threadIdx.x < kLscFeat) mean[threadIdx.x] = l.means[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)f.N) return;
    const int y = (int)(i / (size_t)f.W), x = (int)(i - (size_t)y * f.W);
    const uint32_t p = f.lab[i];
    const uint32_t L = p & 255u, A = (p >> 8) & 255u, B = (p >> 16) & 255u;
    float F[kLscFeat];
    F[0] = lc[L]; F[1] = lc[256 + L];
    F[2] = lc[512 + A]; F[3] = lc[768 + A];
    F[4] = lc[512 + B]; F[5] = lc[768 + B];
    F[6] = l.lut_x[x]; F[7] = l.lut_x[f.W + x];
    F[8] = l.lut_y[y]; F[9] = l.lut_y[f.H + y];
    float w = 0.0f;
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) w = __builtin_fmaf(mean[q], F[q], w);
    st_stream(l.weight + i, w);
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) st_stream(l.feat + (size_t)q * f.N + i, F[q] / w);
}

static __device__ __forceinline__ float wave_sum(float v) { return wave_reduce_add(v); }
// fixed point of the fused update's accumulators: a wavefront's partial sum (256 pixels) of w*F stays below 256 * 51 and
// one of w below 256 * 23000, so these scales keep the rounded partials inside 32 bits; the accumulators themselves are 64-bit
constexpr float kLscFixF = 65536.0f, kLscFixW = 128.0f;

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
    float acc[kLscFeat];
#pragma unroll
    for (int q = 0; q < kLscFeat; ++q) acc[q] = 0.0f;
    for (int t0 = lane; t0 - lane < n; t0 += 192) {
        float v[3][kLscFeat];
#pragma unroll
        for (int u = 0; u < 3; ++u) {                   // loads of three steps in flight (clamped index, masked below)
            const int t = min(t0 + 64 * u, n - 1);
            const int yy = y_lo + t / nw, xx = x_lo + t % nw;
            const size_t i = (size_t)yy * f.W + xx;
#pragma unroll
            for (int q = 0; q < kLscFeat; ++q) v[u][q] = l.feat[(size_t)q * f.N + i];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (t0 + 64 * u < n) {
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) acc[q] += v[u][q];
            }
        }
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

// ---- cross-lane fold of the fused update ----------------------------------------------------
// A halving butterfly: at each level two values are folded across one lane bit into ONE register, the lanes with the bit
// clear keeping the first value's partial sum and the lanes with the bit set the second's (v_permlane32_swap / v_permlane16_swap
// across the halves and the 16-lane rows, DPP moves under bank masks below that).  The order of the additions is fixed.
typedef unsigned int lsc_u2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ float lsc_add(float a, float b) { return a + b; }
static __device__ __forceinline__ uint32_t lsc_add(uint32_t a, uint32_t b) { return a + b; }
static __device__ __forceinline__ uint32_t lsc_bits(float a) { return __float_as_uint(a); }
static __device__ __forceinline__ uint32_t lsc_bits(uint32_t a) { return a; }
template <typename T> static __device__ __forceinline__ T lsc_from(uint32_t a);
template <> __device__ __forceinline__ float lsc_from<float>(uint32_t a) { return __uint_as_float(a); }
template <> __device__ __forceinline__ uint32_t lsc_from<uint32_t>(uint32_t a) { return a; }
template <typename T> static __device__ __forceinline__ T fold32(T a, T b) {     // lanes 0..31: a[l] + a[l+32]; lanes 32..63: b[l-32] + b[l]
    const lsc_u2 r = __builtin_amdgcn_permlane32_swap(lsc_bits(a), lsc_bits(b), false, false);
    return lsc_add(lsc_from<T>(r[0]), lsc_from<T>(r[1]));
}
template <typename T> static __device__ __forceinline__ T fold16(T a, T b) {     // even rows: a[row] + a[row+1]; odd rows: b[row-1] + b[row]
    const lsc_u2 r = __builtin_amdgcn_permlane16_swap(lsc_bits(a), lsc_bits(b), false, false);
    return lsc_add(lsc_from<T>(r[0]), lsc_from<T>(r[1]));
}
template <int CTRL, int BANKS_A, typename T> static __device__ __forceinline__ T fold_dpp(T a, T b) {   // banks A: a + a', the others: b + b'
    const uint32_t m1 = __builtin_amdgcn_update_dpp(lsc_bits(b), lsc_bits(a), CTRL, 0xF, BANKS_A, false);
    const uint32_t m2 = __builtin_amdgcn_update_dpp(lsc_bits(a), lsc_bits(b), CTRL, 0xF, 0xF ^ BANKS_A, false);
    return lsc_add(lsc_from<T>(m1), lsc_from<T>(m2));
}
template <int CTRL, typename T> static __device__ __forceinline__ T sum_dpp(T a) {
    return lsc_add(a, lsc_from<T>(__builtin_amdgcn_update_dpp(0u, lsc_bits(a), CTRL, 0xF, 0xF, true)));
}
constexpr int kDppRor8 = 0x128, kDppHalfMirror = 0x141, kDppQuadSwap1 = 0xB1, kDppQuadSwap2 = 0x4E;
// 12 floats -> the total of v[q] in every lane of the quads with slot_f == q (see the slot arithmetic in the assign kernel)
static __device__ __forceinline__ float lsc_fold12(const float* v) {
    float g[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = fold32(v[2 * i], v[2 * i + 1]);
    float h[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) h[i] = fold16(g[2 * i], g[2 * i + 1]);
    const float m0 = fold_dpp<kDppRor8, 0x3>(h[0], h[1]);
    const float m1 = sum_dpp<kDppRor8>(h[2]);
    float t = fold_dpp<kDppHalfMirror, 0x5>(m0, m1);
    t = sum_dpp<kDppQuadSwap1>(t);
    return sum_dpp<kDppQuadSwap2>(t);
}
// 6 integers -> the total of v[q] in every lane of the octets with slot_u == q
static __device__ __forceinline__ uint32_t lsc_fold6(const uint32_t* v) {
    uint32_t g[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = fold32(v[2 * i], v[2 * i + 1]);
    const uint32_t h0 = fold16(g[0], g[1]);
    const uint32_t h1 = fold16(g[2], 0u);
    uint32_t t = fold_dpp<kDppRor8, 0x3>(h0, h1);
    t = sum_dpp<kDppHalfMirror>(t);
    t = sum_dpp<kDppQuadSwap1>(t);
    return sum_dpp<kDppQuadSwap2>(t);
}

// ---- assignment -----------------------------------------------------------------------------
// Block = 64 columns x (4 wavefronts x kLscRows visited rows).  Wavefront 0 collects the clusters whose window can
// reach the block from the coarse bins, orders them by the reference's visit order (phase, k) and stages their
// feature-space centroids in LDS; every wavefront then walks the list with its pixels' features in registers.
// A pixel takes cluster k iff it lies in k's window and dist < best so far (strict, src/lsc.cpp:217; a NaN
// centroid therefore never wins and +inf never beats the initial FLT_MAX).  More than 64 candidates (piled-up
// clusters): the block walks all K clusters in visit order instead.
template <bool stale_sums>
static __device__ __forceinline__ void lsc_assign_body(FrameDev& f, LscDev& l, int rem, int stride, int Hv) {
    f.select(blockIdx.z);
    l.select(blockIdx.z);
    __shared__ uint32_t s_raw[192];
    __shared__ uint32_t s_yx[64], s_k[64];
    __shared__ __attribute__((aligned(16))) float s_cf[64 * 12];
    __shared__ int s_cnt;
    // block-level accumulators of the fused update: per candidate of the block's list 11 feature sums (2^-20 fixed point)
    // and 6 integer sums, 64-bit words (LDS atomics; flushed with ONE set of global atomics per (block, cluster))
    constexpr int kAccW = kLscFeat + 1 + 6;
    __shared__ unsigned long long s_accx[64 * kAccW];
    const int wave = threadIdx.x >> 6, lane = LANE();
    const int S = f.S, W = f.W;
    const size_t N = (size_t)f.N;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kLscBH;
    if (stale_sums)
        for (int i = threadIdx.x; i < 64 * kAccW; i += 256) s_accx[i] = 0ull;
    if (wave == 0) {
        const int v_last = min(v0 + kLscBH, Hv) - 1;
        TileWindow win;
        win.wy_lo = rem + stride * v0 - S; win.wy_hi = rem + stride * v_last + S;
        win.wx_lo = x0 - S; win.wx_hi = min(x0 + 63, W - 1) + S;
        const int cnt = gather_candidates(f, win, s_raw, lane);
        wave_lds_sync();
        if (cnt <= 64) {
            const uint32_t mytag = lane < cnt ? s_raw[128 + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += s_raw[128 + j] < mytag ? 1 : 0;
            if (lane < cnt) {
                const uint32_t k = mytag & 0xFFFFu;
                s_yx[rank] = s_raw[lane];
                s_k[rank] = k;
                const float4* src = reinterpret_cast<const float4*>(l.cfeat + (size_t)k * kLscCfPitch);
                float4* dst = reinterpret_cast<float4*>(s_cf + rank * 12);
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            }
        }
        if (lane == 0) s_cnt = cnt;
    }
    // this wavefront's pixels: column x0 + lane of kLscRows visited rows (clamped addresses, masked results).  Rows are
    // held in pairs (r, r+1) per feature: the distance chain then runs on packed f32 (v_pk_add_f32 / v_pk_fma_f32,
    // two rows per instruction; each half is the same IEEE operation as the scalar form)
    const int x = x0 + lane;
    const bool xok = x < W;
    const int vw = (int)rfl((uint32_t)(v0 + wave * kLscRows));
    f2 F[kLscRows / 2][kLscFeat];
    int yr[kLscRows];
    const int xc = min(x, W - 1);
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        yr[r] = rem + stride * min(vw + r, Hv - 1);
        const size_t p = (size_t)yr[r] * W + xc;
#pragma unroll
        for (int q = 0; q < kLscFeat; ++q) F[r >> 1][q][r & 1] = l.feat[(size_t)q * N + p];
    }
    __syncthreads();
    const int cnt = s_cnt;
    f2 best[kLscRows / 2];
    uint32_t bk[kLscRows];
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) { best[r >> 1][r & 1] = FLT_MAX; bk[r] = 0xFFFFu; }
    const int y_top = yr[0], y_bot = yr[kLscRows - 1];
    auto consider = [&](int cy, int cx, uint32_t k, const float* cf) {
        const bool inx = abs(x - cx) <= S;
#pragma unroll
        for (int h = 0; h < kLscRows / 2; ++h) {
            const bool in0 = abs(yr[2 * h] - cy) <= S, in1 = abs(yr[2 * h + 1] - cy) <= S;   // uniform
            if (!(in0 || in1)) continue;
            f2 d = {0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < kLscFeat; ++q) {
                const f2 c2 = {cf[q], cf[q]};
                const f2 diff = F[h][q] - c2;
                d = __builtin_elementwise_fma(diff, diff, d);
            }
            if (in0 && inx && d[0] < best[h][0]) { best[h][0] = d[0]; bk[2 * h] = k; }
            if (in1 && inx && d[1] < best[h][1]) { best[h][1] = d[1]; bk[2 * h + 1] = k; }
        }
    };
    if (cnt <= 64) {
        for (int c = 0; c < cnt; ++c) {
            const uint32_t yx = rfl(s_yx[c]);
            const int cy = (int)(yx >> 16), cx = (int)(yx & 0xFFFFu);
            if (cx + S < x0 || cx - S > x0 + 63 || cy + S < y_top || cy - S > y_bot) continue;   // uniform
            const float4* src = reinterpret_cast<const float4*>(s_cf + c * 12);
            const float4 c0 = src[0], c1 = src[1], c2 = src[2];
            const float cf[kLscFeat] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
            consider(cy, cx, rfl(s_k[c]), cf);
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
                    for (int q = 0; q < kLscFeat; ++q) cf[q] = src[q];
                    consider((int)(yx >> 16), (int)(yx & 0xFFFFu), (uint32_t)(k0 + c), cf);
                }
            }
        }
    }
    // labels out; with an update following (stale_sums), also the label each visited pixel carries into update(): the new one,
    // or the one it keeps when nothing accepted it (src/lsc.cpp:217-220 never touches such a pixel; 0xFFFF: none yet)
    uint32_t lbl[kLscRows];
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        lbl[r] = 0xFFFFu;
        if (!xok || vw + r >= Hv) continue;
        const size_t p = (size_t)yr[r] * W + x;
        if (best[r >> 1][r & 1] < FLT_MAX) {
            st_stream(f.labels + p, (uint16_t)bk[r]);
            lbl[r] = bk[r];
        } else if (stale_sums) {
            lbl[r] = f.labels[p];
        }
    }
    if (!stale_sums) return;
    // ---- update(), fused: the sums of src/context.cpp:301-354 (count, y, x, L, a, b) and of src/lsc.cpp:256-269 (sum w*F per
    // feature, sum w) over this wavefront's pixels, label by label: every lane folds its rows that carry the label (f32 FMA
    // chain like the reference build; lanes without such a row contribute zeros), a halving butterfly (lsc_fold12 / lsc_fold6:
    // 43 cross-lane operations for the 17 values, against 102 for 17 full wave sums) leaves value q's total in the lanes of
    // slot q, and ONE 64-bit LDS atomic instruction adds all 17 to the block's accumulators (the float sums in fixed point, so
    // the totals do not depend on the order in which wavefronts arrive: run-to-run identical).  A wavefront of the 1280x720
    // K=1600 fixture meets 8 labels on average.  (Measured alternatives per launch of 8 x 720p: lanes first folding their OWN
    // label and the labels then served from those partial sums 75 us -- 17 butterflies per wavefront instead of 8 --, full wave
    // sums instead of the butterfly 87 us, 16-lane row sums and LDS atomics from every row's first lane 172 us; this form 61 us.)
    // This replaces a kernel that re-scanned every cluster's window for its members (64 of 100 us per iteration).
    float wgt[kLscRows];
    uint32_t pix[kLscRows];
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        const size_t p = (size_t)yr[r] * W + xc;
        wgt[r] = l.weight[p];
        pix[r] = f.lab[p];
    }
    uint32_t todo_rows = 0;                       // bit r: row r still has to be summed
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) todo_rows |= (lbl[r] != 0xFFFFu ? 1u : 0u) << r;
    const uint32_t my_k = lane < min(cnt, 64) && cnt <= 64 ? s_k[lane] : 0xFFFFFFFFu;      // the block's list: lane c = candidate c
    // slot of this lane after the fold: the float sums end in the quads (lane bit 2 clear: slots 0..7, set: 8..10), the
    // integer sums in the octets; one lane of each adds
    const int b5 = (lane >> 5) & 1, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
    const int slot_f = b2 ? 8 + 2 * b4 + b5 : 4 * b3 + 2 * b4 + b5;
    const int slot_u = b3 ? 4 + b5 : 2 * b4 + b5;
    const bool adds_f = (lane & 3) == 0 && !(b2 && b3) && slot_f <= kLscFeat;
    const bool adds_u = (lane & 7) == 1 && !(b3 && b4);
    const int slot = adds_f ? slot_f : kLscFeat + 1 + slot_u;
    const float fix = slot_f == kLscFeat ? kLscFixW : kLscFixF;
    for (;;) {
        uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < kLscRows; ++r) mine = ((todo_rows >> r) & 1u) ? lbl[r] : mine;
        const unsigned long long pend = __ballot(mine != 0xFFFFFFFFu);
        if (!pend) break;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(pend));
        float fv[kLscFeat + 2];
        uint32_t uv[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < kLscFeat + 2; ++q) fv[q] = 0.0f;
#pragma unroll
        for (int r = 0; r < kLscRows; ++r) {
            const bool m = ((todo_rows >> r) & 1u) && lbl[r] == k0;
            if (m) {
                uv[0] += 1u; uv[1] += (uint32_t)yr[r]; uv[2] += (uint32_t)x;
                uv[3] += pix[r] & 255u; uv[4] += (pix[r] >> 8) & 255u; uv[5] += (pix[r] >> 16) & 255u;
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) fv[q] = __builtin_fmaf(wgt[r], F[r >> 1][q][r & 1], fv[q]);
                fv[kLscFeat] += wgt[r];
                todo_rows &= ~(1u << r);
            }
        }
        const float tf = lsc_fold12(fv);
        const uint32_t tu = lsc_fold6(uv);
        const unsigned long long v = adds_f ? (unsigned long long)(long long)(int)__builtin_rintf(tf * fix) : (unsigned long long)tu;
        const unsigned long long hit = __ballot(my_k == k0);
        if (adds_f || adds_u) {
            if (hit) {
                atomicAdd(&s_accx[__builtin_ctzll(hit) * kAccW + slot], v);
            } else if (adds_f) {
                atomicAdd(reinterpret_cast<unsigned long long*>(l.extra + (size_t)k0 * kLscExPitch) + slot_f, v);
            } else {
                atomicAdd(f.sums[0] + 6 * (size_t)k0 + slot_u, tu);
            }
        }
    }
    __syncthreads();
    // one set of global atomics per (block, cluster that received pixels here)
    for (int i = threadIdx.x; i < min(cnt, 64) * kAccW && cnt <= 64; i += 256) {
        const int c = i / kAccW, q = i - c * kAccW;
        if (s_accx[c * kAccW + kLscFeat + 1] == 0ull) continue;          // count == 0: nothing for this cluster
        const unsigned long long v = s_accx[i];
        const uint32_t k = s_k[c];
        if (q <= kLscFeat) atomicAdd(reinterpret_cast<unsigned long long*>(l.extra + (size_t)k * kLscExPitch) + q, v);
        else atomicAdd(f.sums[0] + 6 * (size_t)k + (q - kLscFeat - 1), (uint32_t)v);
    }
}

// Two kernels over one body: the pass that is followed by an update carries the fused sums and needs 90 VGPRs (5 wavefronts per SIMD); the
// final pass does not and fits 7 wavefronts per SIMD.  (Forcing the first to 80 VGPRs spills 7 registers: 66 instead of 60 us.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_lsc_assign(FrameDev f, LscDev l, int rem, int stride, int Hv) {
    lsc_assign_body<false>(f, l, rem, stride, Hv);
}
__global__ __launch_bounds__(256) void k_lsc_assign_sums(FrameDev f, LscDev l, int rem, int stride, int Hv) {
    lsc_assign_body<true>(f, l, rem, stride, Hv);
}

// ---- update, second half ----------------------------------------------------------------------
// C_f = sum w*F / sum w per cluster (src/lsc.cpp:256-269 then :309-316; 0/0 = NaN for a memberless cluster exactly as the
// reference produces it) from the fixed-point accumulators the assign pass filled, which are cleared for the next pass.
// The integer sums of the shared centroid update are in FrameDev::sums, where the cluster pass reads them.
__global__ __launch_bounds__(256) void k_lsc_update(FrameDev f, LscDev l) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 4, q = t & 15;                       // 16 threads per cluster: 10 features, the weight, 5 idle
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

// ---- launches -------------------------------------------------------------------------------
void launch_lsc_prepare(const FrameDev& f, const LscDev& l, int nframes, hipStream_t st) {
    const unsigned N = (unsigned)f.N;
    launch(k_lsc_hist, dim3((N + 4095) / 4096, nframes), dim3(256), 0, st, f, l);
    launch(k_lsc_means, dim3(nframes), dim3(256), 0, st, f, l);
    launch(k_lsc_features, dim3((N + 255) / 256, nframes), dim3(256), 0, st, f, l);
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

void launch_lsc_update(const FrameDev& f_, const LscDev& l, int nframes, int rem, int stride, int sbuf, hipStream_t st) {
    (void)rem; (void)stride;
    const FrameDev f = rotated(f_, 0, sbuf);
    launch(k_lsc_update, dim3((f.K * 16 + 255) / 256, nframes), dim3(256), 0, st, f, l);
}

}  // namespace fslic
