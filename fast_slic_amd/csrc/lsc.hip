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
// fixed point of the fused update's accumulators: a lane's partial sum (its column of kLscRows pixels) of w*F stays below
// 4 * 51 and one of w below 4 * 23000, so these scales keep the rounded partials inside 32 bits; the accumulators themselves
// are 64-bit (a cluster that owned a whole 4K frame would reach 4.4e14 and 7.8e14)
constexpr float kLscFixF = 1048576.0f, kLscFixW = 4096.0f;
constexpr int kLscCopies = 4;                             // privatised copies of a block's accumulators (lane & 3)
constexpr int kLscAccW = kLscFeat + 1 + 2;                // 64-bit words per (candidate, copy): 10 x sum w*F, sum w, two packed integer words
constexpr uint32_t kLscNoSlot = 0xFFFFFFFFu;

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
    // block-level accumulators of the fused update: per candidate of the block's list (and privatised copy) 11 feature sums
    // in fixed point and the 6 integer sums packed into two words (sum L | sum a << 20 | sum b << 40 and sum (x - x0) |
    // sum (y - y0) << 20 | count << 40, as the SLIC block kernel packs them: 1024 pixels per block, no field reaches 2^20);
    // LDS atomics, flushed with ONE set of global atomics per (block, cluster)
    __shared__ unsigned long long s_accx[stale_sums ? 64 * kLscCopies * kLscAccW : 1];
    const int wave = threadIdx.x >> 6, lane = LANE();
    const int S = f.S, W = f.W;
    const size_t N = (size_t)f.N;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kLscBH;
    if (stale_sums)
        for (int i = threadIdx.x; i < 64 * kLscCopies * kLscAccW; i += 256) s_accx[i] = 0ull;
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
            consider(cy, cx, (uint32_t)c, cf);
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
    const bool listed = cnt <= 64;                 // bk = position in the block's list (else the cluster itself)
    uint32_t lbl[kLscRows], slot[kLscRows];
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        lbl[r] = 0xFFFFu;
        slot[r] = kLscNoSlot;
        if (!xok || vw + r >= Hv) continue;
        const size_t p = (size_t)yr[r] * W + x;
        if (best[r >> 1][r & 1] < FLT_MAX) {
            lbl[r] = listed ? s_k[bk[r]] : bk[r];
            st_stream(f.labels + p, (uint16_t)lbl[r]);
            if (listed) slot[r] = bk[r];
        } else if (stale_sums) {
            lbl[r] = f.labels[p];
        }
    }
    if (!stale_sums) return;
    // ---- update(), fused: the sums of src/context.cpp:301-354 (count, y, x, L, a, b) and of src/lsc.cpp:256-269 (sum w*F per
    // feature, sum w) over this block's pixels.  A column's rows fall into one or two runs of equal label (a superpixel is
    // taller than a wavefront's rows), so every lane sums its column run by run in registers (f32 FMA chain like the reference
    // build) and adds a finished run to the block's accumulators with 13 LDS atomics: the float sums in fixed point, so the
    // totals do not depend on the order in which lanes and wavefronts arrive (run-to-run identical), the six integer sums
    // packed into two words.  Lanes of a run of columns with one label meet on the same words: kLscCopies privatised copies.
    // (Measured per launch of 8 x 1280x720 K=1600, the forms this replaces: a loop over the wavefront's labels with a
    // halving butterfly over the lanes per label 61 us -- 8 labels x (44 masked FMAs + 43 cross-lane operations + 17 LDS
    // atomics) per wavefront --, lanes folding their own label first 75 us, full wave sums 87 us, 16-lane row sums 172 us;
    // the kernel that re-scanned every cluster's window for its members 64 of 100 us per iteration.)
    float wgt[kLscRows];
    uint32_t pix[kLscRows];
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        const size_t p = (size_t)yr[r] * W + xc;
        wgt[r] = l.weight[p];
        pix[r] = f.lab[p];
    }
    // a pixel whose label is not on the block's list (it kept a label whose window has left it, or the block walks every
    // cluster): straight to the global accumulators.  Rare.
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        if (lbl[r] == 0xFFFFu || slot[r] != kLscNoSlot) continue;
        unsigned long long* e = reinterpret_cast<unsigned long long*>(l.extra + (size_t)lbl[r] * kLscExPitch);
#pragma unroll
        for (int q = 0; q < kLscFeat; ++q)
            atomicAdd(e + q, (unsigned long long)(long long)(int)__builtin_rintf(__builtin_fmaf(wgt[r], F[r >> 1][q][r & 1], 0.0f) * kLscFixF));
        atomicAdd(e + kLscFeat, (unsigned long long)(long long)(int)__builtin_rintf(wgt[r] * kLscFixW));
        global_accumulate(f.sums[0], lbl[r], (uint32_t)yr[r], (uint32_t)x, pix[r]);
    }
    {
        const uint32_t copy = (uint32_t)lane & (kLscCopies - 1);
        uint32_t cur = slot[0], a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
        float fv[kLscFeat + 1];
#pragma unroll
        for (int q = 0; q <= kLscFeat; ++q) fv[q] = 0.0f;
        auto flush = [&]() {
            if (cur != kLscNoSlot) {
                unsigned long long* a = s_accx + (cur * kLscCopies + copy) * kLscAccW;
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) atomicAdd(a + q, (unsigned long long)(long long)(int)__builtin_rintf(fv[q] * kLscFixF));
                atomicAdd(a + kLscFeat, (unsigned long long)(long long)(int)__builtin_rintf(fv[kLscFeat] * kLscFixW));
                atomicAdd(a + kLscFeat + 1, (unsigned long long)a_lo | ((unsigned long long)a_hi << 32));
                atomicAdd(a + kLscFeat + 2, (unsigned long long)b_lo | ((unsigned long long)b_hi << 32));
            }
        };
#pragma unroll
        for (int r = 0; r < kLscRows; ++r) {
            if (r > 0 && slot[r] != cur) {
                flush();
                cur = slot[r]; a_lo = 0; a_hi = 0; b_lo = 0; b_hi = 0;
#pragma unroll
                for (int q = 0; q <= kLscFeat; ++q) fv[q] = 0.0f;
            }
#pragma unroll
            for (int q = 0; q < kLscFeat; ++q) fv[q] = __builtin_fmaf(wgt[r], F[r >> 1][q][r & 1], fv[q]);
            fv[kLscFeat] += wgt[r];
            const uint32_t p = pix[r];
            a_lo += (p & 255u) | ((p & 0xFF00u) << 12);                      // sum L | sum a << 20
            a_hi += (p >> 8) & 0xFF00u;                                     // sum b << 40
            b_lo += (uint32_t)lane | (1u << 20);                            // sum (x - x0) | count << 20
            b_hi += (uint32_t)(wave * kLscRows + r);                        // sum of the row's number inside the block << 32
        }
        flush();
    }
    __syncthreads();
    // one set of global atomics per (block, cluster that received pixels here)
    const uint32_t yb0 = (uint32_t)(rem + stride * v0);
    for (int i = threadIdx.x; i < cnt * (kLscFeat + 1 + 6) && listed; i += 256) {
        const int c = i / (kLscFeat + 1 + 6), q = i - c * (kLscFeat + 1 + 6);
        const unsigned long long* a = s_accx + (size_t)c * kLscCopies * kLscAccW;
        unsigned long long B = 0;
#pragma unroll
        for (int cpy = 0; cpy < kLscCopies; ++cpy) B += a[cpy * kLscAccW + kLscFeat + 2];
        const uint32_t n = (uint32_t)(B >> 20) & 0xFFFu;
        if (n == 0u) continue;                                              // nothing for this cluster
        const uint32_t k = s_k[c];
        if (q <= kLscFeat) {
            unsigned long long v = 0;
#pragma unroll
            for (int cpy = 0; cpy < kLscCopies; ++cpy) v += a[cpy * kLscAccW + q];
            atomicAdd(reinterpret_cast<unsigned long long*>(l.extra + (size_t)k * kLscExPitch) + q, v);
        } else {
            unsigned long long A = 0;
#pragma unroll
            for (int cpy = 0; cpy < kLscCopies; ++cpy) A += a[cpy * kLscAccW + kLscFeat + 1];
            const int u = q - kLscFeat - 1;
            uint32_t v;
            if (u == 0) v = n;
            else if (u == 1) v = yb0 * n + (uint32_t)stride * (uint32_t)(B >> 32);
            else if (u == 2) v = (uint32_t)x0 * n + ((uint32_t)B & 0xFFFFFu);
            else if (u == 3) v = (uint32_t)A & 0xFFFFFu;
            else if (u == 4) v = (uint32_t)(A >> 20) & 0xFFFFFu;
            else v = (uint32_t)(A >> 40);
            atomicAdd(f.sums[0] + 6 * (size_t)k + u, v);
        }
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
