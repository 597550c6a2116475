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
static __device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return wave_reduce_add(v); }

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
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_lsc_assign(FrameDev f, LscDev l, int rem, int stride, int Hv, int stale_sums) {
    f.select(blockIdx.z);
    l.select(blockIdx.z);
    __shared__ uint32_t s_raw[192];
    __shared__ uint32_t s_yx[64], s_k[64];
    __shared__ __attribute__((aligned(16))) float s_cf[64 * 12];
    __shared__ int s_cnt;
    const int wave = threadIdx.x >> 6, lane = LANE();
    const int S = f.S, W = f.W;
    const size_t N = (size_t)f.N;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kLscBH;
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
#pragma unroll
    for (int r = 0; r < kLscRows; ++r) {
        if (!xok || vw + r >= Hv) continue;
        const size_t p = (size_t)yr[r] * W + x;
        if (best[r >> 1][r & 1] < FLT_MAX) {
            st_stream(f.labels + p, (uint16_t)bk[r]);
        } else if (stale_sums) {
            // Nothing accepted this pixel: it keeps its label (src/lsc.cpp:217-220 never touches it) and the update sums
            // it under that label (src/context.cpp:301-354, src/lsc.cpp:256-269).  The update kernel gathers over
            // cluster windows, so a pixel outside the window of its own (stale) cluster is added here instead.
            const uint32_t old = f.labels[p];
            if (old != 0xFFFFu) {
                const uint32_t oyx = f.cl_yx[old];
                if (abs(yr[r] - (int)(oyx >> 16)) > S || abs(x - (int)(oyx & 0xFFFFu)) > S) {
                    global_accumulate(f.sums[0], old, (uint32_t)yr[r], (uint32_t)x, f.lab[p]);
                    const float w = l.weight[p];
                    unsigned long long* e = reinterpret_cast<unsigned long long*>(l.extra + (size_t)old * kLscExPitch);
#pragma unroll
                    for (int q = 0; q < kLscFeat; ++q)
                        atomicAdd(e + q, (unsigned long long)(long long)__builtin_rintf(w * F[r >> 1][q][r & 1] * kLscFixScale));
                    atomicAdd(e + kLscFeat, (unsigned long long)(long long)__builtin_rintf(w * kLscFixScale));
                }
            }
        }
    }
}

// ---- update ---------------------------------------------------------------------------------
// One block per cluster: gather over the visited rows of the cluster's window the pixels labelled k: integer sums
// (count, y, x, L, a, b) for the shared centroid update (src/context.cpp:301-373) and the weighted feature sums
// sum w*F, sum w (src/lsc.cpp:256-269; FMA as the reference build contracts it).  C_f = sum w*F / sum w, 0/0 = NaN
// for a memberless cluster exactly as src/lsc.cpp:309-316 produces it.  No atomics, fixed reduction order.
__global__ __launch_bounds__(256) void k_lsc_update(FrameDev f, LscDev l, int rem, int stride) {
    f.select(blockIdx.y);
    l.select(blockIdx.y);
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = LANE();      // one wavefront per cluster, four clusters per block
    if (k >= f.K) return;
    const uint32_t yx = f.cl_yx[k];
    const int cy = (int)(yx >> 16), cx = (int)(yx & 0xFFFFu), S = f.S, W = f.W;
    const size_t N = (size_t)f.N;
    const int y_lo = max(cy - S, 0), y_hi = min(cy + S, f.H - 1);
    const int x_lo = max(cx - S, 0), x_hi = min(cx + S, W - 1);
    const int t = y_lo - rem;
    const int v_first = t <= 0 ? 0 : (t + stride - 1) / stride;
    // the stale-pixel contributions of the assign pass, fetched now so that the tail does not wait for them
    uint32_t pre_sum = 0;
    long long pre_ex = 0;
    if (lane < 6) pre_sum = f.sums[0][6 * (size_t)k + lane];
    if (lane <= kLscFeat) pre_ex = l.extra[(size_t)k * kLscExPitch + lane];
    uint32_t in[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    float acc[kLscFeat + 1];
#pragma unroll
    for (int q = 0; q <= kLscFeat; ++q) acc[q] = 0.0f;
    // kUpdRows visited rows of the window per trip, lanes over the columns.  Two memory round trips per trip: the labels
    // of all rows, then the features of the matching pixels -- the loads of a non-matching lane are redirected to the
    // window's first pixel (one cached line for all of them) so that they can be issued unconditionally and all at once.
    constexpr int kUpdRows = 4;
    const size_t p_first = (size_t)min(rem + stride * v_first, f.H - 1) * W + x_lo;
    for (int y0 = rem + stride * v_first; y0 <= y_hi; y0 += kUpdRows * stride) {
        for (int x = x_lo + lane; x - lane <= x_hi; x += 64) {
            size_t p[kUpdRows];
            uint32_t lb[kUpdRows];
#pragma unroll
            for (int u = 0; u < kUpdRows; ++u) {
                const int y = y0 + u * stride;
                p[u] = (y <= y_hi && x <= x_hi) ? (size_t)y * W + x : p_first;
                lb[u] = f.labels[p[u]];
            }
            uint32_t pix[kUpdRows];
            float w[kUpdRows], Fv[kUpdRows][kLscFeat];
            bool m[kUpdRows];
#pragma unroll
            for (int u = 0; u < kUpdRows; ++u) {
                const int y = y0 + u * stride;
                m[u] = y <= y_hi && x <= x_hi && lb[u] == (uint32_t)k;
                const size_t pp = m[u] ? p[u] : p_first;
                pix[u] = f.lab[pp];
                w[u] = l.weight[pp];
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) Fv[u][q] = l.feat[(size_t)q * N + pp];
            }
#pragma unroll
            for (int u = 0; u < kUpdRows; ++u) {
                if (!m[u]) continue;
                const int y = y0 + u * stride;
                in[0] += 1u; in[1] += (uint32_t)y; in[2] += (uint32_t)x;
                in[3] += pix[u] & 255u; in[4] += (pix[u] >> 8) & 255u; in[5] += (pix[u] >> 16) & 255u;
#pragma unroll
                for (int q = 0; q < kLscFeat; ++q) acc[q] = __builtin_fmaf(w[u], Fv[u][q], acc[q]);
                acc[kLscFeat] += w[u];
            }
        }
    }
    // fixed reduction order (DPP tree, then the four 16-lane rows): run-to-run identical
#pragma unroll
    for (int q = 0; q < 6; ++q) in[q] = wave_sum(in[q]);
#pragma unroll
    for (int q = 0; q <= kLscFeat; ++q) acc[q] = wave_sum(acc[q]);
    uint32_t my_in = in[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) my_in = lane == q ? in[q] : my_in;
    if (lane < 6) f.sums[0][6 * (size_t)k + lane] = pre_sum + my_in;
    float my_acc = acc[0];
#pragma unroll
    for (int q = 1; q <= kLscFeat; ++q) my_acc = lane == q ? acc[q] : my_acc;
    if (lane <= kLscFeat && pre_ex != 0) {
        my_acc += (float)((double)pre_ex / (double)kLscFixScale);
        l.extra[(size_t)k * kLscExPitch + lane] = 0;
    }
    const float tot_w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_acc), kLscFeat));
    if (lane < kLscFeat) l.cfeat[(size_t)k * kLscCfPitch + lane] = my_acc / tot_w;
}

// ---- launches -------------------------------------------------------------------------------
void launch_lsc_prepare(const FrameDev& f, const LscDev& l, int nframes, hipStream_t st) {
    const unsigned N = (unsigned)f.N;
    hipLaunchKernelGGL(k_lsc_hist, dim3((N + 4095) / 4096, nframes), dim3(256), 0, st, f, l);
    hipLaunchKernelGGL(k_lsc_means, dim3(nframes), dim3(256), 0, st, f, l);
    hipLaunchKernelGGL(k_lsc_features, dim3((N + 255) / 256, nframes), dim3(256), 0, st, f, l);
    hipLaunchKernelGGL(k_lsc_seed, dim3((f.K + 3) / 4, nframes), dim3(256), 0, st, f, l);
}

void launch_lsc_assign(const FrameDev& f_, const LscDev& l, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    hipLaunchKernelGGL(k_lsc_assign, dim3((f.W + 63) / 64, (Hv + kLscBH - 1) / kLscBH, nframes), dim3(256), 0, st,
                       f, l, rem, stride, Hv, stale_sums ? 1 : 0);
}

void launch_lsc_update(const FrameDev& f_, const LscDev& l, int nframes, int rem, int stride, int sbuf, hipStream_t st) {
    const FrameDev f = rotated(f_, 0, sbuf);
    hipLaunchKernelGGL(k_lsc_update, dim3(f.K, nframes), dim3(256), 0, st, f, l, rem, stride);
}

}  // namespace fslic
