// realdist.hip -- the float-distance variants ContextRealDist ('standard') and ContextRealDistL2 ('l2')
// (src/context.h:100-111, src/context.cpp:258-298 instantiated for float, :394-447).
// Hand-written gfx950 (CDNA4, wave64) kernels; see kernels.h for the launch interface and DESIGN.md for the design.
//
// distance = spatial_dist_patch[dy + S][dx + S] (f32, built on the host with the reference's expressions) + colour term:
//   standard: |dL| + |da| + |db| (integers, converted once)        l2: dL^2 + da^2 + db^2 (exact in f32)
// One f32 addition per pixel-candidate in the reference's operand order, strict '<' against the running minimum in visit
// order (phase, k): bit-identical labels.  Centroids stay quantised (src/context.cpp:356-373), so the update is the integer
// one of the SLIC path: a per-cluster gather of the member sums (no atomics), then the shared k_bin_clusters<1>.
#include "device_common.h"
#include <cfloat>

// Every rounding of this file is stated: a product that is added is fused only where __builtin_fmaf says so.  hipcc's default for device
// code (-ffp-contract=fast) fuses a * b + c wherever it finds one -- __fmul_rn / __fadd_rn are inline functions of the HIP headers, compiled
// under that default whatever the pragma below says, so the operators are written out here -- and whether it
// found the last addition of the 'noq' Euclidean distance depended on the unrolling around it: t + dy * dy became fma(dy, dy, t), one
// rounding fewer than the reference makes, and on a constant frame (every pixel between two grid centres is a tie decided by the last
// bit) 26 of 647 878 pixels of a 1078x601 K=1124 frame went to the other neighbour (tests/fuzz_parity.py, variants mode, seed 1 case 140).
#pragma clang fp contract(off)

namespace fslic {

constexpr int kRdRows = 4;                              // visited rows per wavefront
constexpr int kRdBH = kWavesPerBlock * kRdRows;

template <bool L2NORM>
__global__ __launch_bounds__(256) void k_rd_assign(FrameDev f, const float* __restrict__ patchf, int rem, int stride, int Hv, int stale_sums) {
    f.select(blockIdx.z);
    __shared__ uint32_t s_raw[192];
    __shared__ uint32_t s_yx[64], s_k[64], s_lab[64];
    __shared__ int s_cnt;
    const int wave = (int)rfl((uint32_t)(threadIdx.x >> 6)), lane = LANE();
    const int S = f.S, W = f.W, P = 2 * S + 1;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kRdBH;
    if (wave == (int)((blockIdx.x + blockIdx.y) & 3u)) {
        const int v_last = min(v0 + kRdBH, Hv) - 1;
        TileWindow win;
        win.wy_lo = rem + stride * v0 - S; win.wy_hi = rem + stride * v_last + S;
        win.wx_lo = x0 - S; win.wx_hi = min(x0 + 63, W - 1) + S;
        const int cnt = gather_candidates(f, win, s_raw, lane);
        wave_lds_sync();
        if (cnt <= 64) {
            const uint32_t mytag = lane < cnt ? s_raw[128 + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += s_raw[128 + j] < mytag ? 1 : 0;
            if (lane < cnt) { s_yx[rank] = s_raw[lane]; s_lab[rank] = s_raw[64 + lane]; s_k[rank] = mytag & 0xFFFFu; }
        }
        if (lane == 0) s_cnt = cnt;
    }
    const int x = x0 + lane;
    const bool xok = x < W;
    const int xc = min(x, W - 1);
    const int vw = v0 + wave * kRdRows;
    uint32_t pix[kRdRows];
    int yr[kRdRows];
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) {
        yr[r] = rem + stride * min(vw + r, Hv - 1);
        pix[r] = f.lab[(size_t)yr[r] * W + xc];
    }
    __syncthreads();
    const int cnt = s_cnt;
    float best[kRdRows];
    uint32_t bk[kRdRows];
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) { best[r] = FLT_MAX; bk[r] = 0xFFFFu; }
    const int y_top = yr[0], y_bot = yr[kRdRows - 1];
    auto consider = [&](int cy, int cx, uint32_t k, uint32_t clab) {
        const int dx = x - cx;
        const bool inx = abs(dx) <= S;
        const float* prow = patchf + (inx ? dx + S : 0);
        float sp[kRdRows];
#pragma unroll
        for (int r = 0; r < kRdRows; ++r) {              // table values of all rows in flight (clamped row index, masked below)
            const int dy = min(max(yr[r] - cy, -S), S);
            sp[r] = prow[(dy + S) * P];
        }
#pragma unroll
        for (int r = 0; r < kRdRows; ++r) {
            if (abs(yr[r] - cy) > S) continue;            // uniform
            float col;
            if (L2NORM) {
                const int dL = (int)(pix[r] & 255u) - (int)(clab & 255u);
                const int da = (int)((pix[r] >> 8) & 255u) - (int)((clab >> 8) & 255u);
                const int db = (int)((pix[r] >> 16) & 255u) - (int)((clab >> 16) & 255u);
                col = (float)(dL * dL + da * da + db * db);     // < 2^24: the reference's f32 products and sums are exact too
            } else {
                col = (float)__builtin_amdgcn_sad_u8(pix[r], clab, 0u);
            }
            const float d = sp[r] + col;                  // dist_row = patch_row; dist_row += color_dist (src/context.cpp:275-287)
            if (inx && d < best[r]) { best[r] = d; bk[r] = k; }
        }
    };
    if (cnt <= 64) {
        for (int c = 0; c < cnt; ++c) {
            const uint32_t yx = rfl(s_yx[c]);
            const int cy = (int)(yx >> 16), cx = (int)(yx & 0xFFFFu);
            if (cx + S < x0 || cx - S > x0 + 63 || cy + S < y_top || cy - S > y_bot) continue;   // uniform
            consider(cy, cx, rfl(s_k[c]), rfl(s_lab[c]));
        }
    } else {
        // more than 64 clusters reach the block (piled-up centres): every cluster, in the reference's visit order
        for (uint32_t ph = 0; ph < 4; ++ph) {
            for (int k0 = 0; k0 < f.K; k0 += 64) {
                const int k = k0 + lane;
                const bool live = k < f.K;
                const uint32_t yxl = live ? f.cl_yx[k] : 0u;
                const uint32_t labl = live ? f.cl_lab[k] : 0u;
                const uint32_t tg = live ? f.cl_tag[k] : 0xFFFFFFFFu;
                const int cyl = (int)(yxl >> 16), cxl = (int)(yxl & 0xFFFFu);
                const bool use = live && (tg >> 16) == ph && !(cxl + S < x0 || cxl - S > x0 + 63 || cyl + S < y_top || cyl - S > y_bot);
                unsigned long long m = __ballot(use);
                while (m) {
                    const int c = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t yx = (uint32_t)__builtin_amdgcn_readlane((int)yxl, c);
                    const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)labl, c);
                    consider((int)(yx >> 16), (int)(yx & 0xFFFFu), (uint32_t)(k0 + c), lb);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) {
        if (!xok || vw + r >= Hv) continue;
        const size_t p = (size_t)yr[r] * W + x;
        if (best[r] < FLT_MAX) {
            st_stream(f.labels + p, (uint16_t)bk[r]);
        } else if (stale_sums) {
            // unassigned this pass: the pixel keeps its label and the update sums it under that label; the member gather
            // below scans cluster windows, so a pixel outside the window of its own (stale) cluster is added here
            const uint32_t old = f.labels[p];
            if (old != 0xFFFFu) {
                const uint32_t oyx = f.cl_yx[old];
                if ((abs(yr[r] - (int)(oyx >> 16)) > S || abs(x - (int)(oyx & 0xFFFFu)) > S) && in_update(f, yr[r], x))
                    global_accumulate(f.sums[0], old, (uint32_t)yr[r], (uint32_t)x, pix[r]);
            }
        }
    }
}

// Integer member sums of every cluster over the visited rows of its window (count, y, x, L, a, b), added to what the
// assign pass left in FrameDev::sums for out-of-window members.  One wavefront per cluster (four clusters per block),
// no atomics, no LDS: lanes over the window's columns, kRows rows in flight per trip.
__global__ __launch_bounds__(256) void k_member_sums(FrameDev f, int rem, int stride, int margin) {
    f.select(blockIdx.y);
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = LANE();
    if (k >= f.K) return;
    const uint32_t yx = f.cl_yx[k];
    const int cy = (int)(yx >> 16), cx = (int)(yx & 0xFFFFu), S = f.S + margin, W = f.W;
    const int y_lo = max(cy - S, 0), y_hi = min(cy + S, f.H - 1);
    const int x_lo = max(cx - S, 0), x_hi = min(cx + S, W - 1);
    const int t = y_lo - rem;
    const int v_first = t <= 0 ? 0 : (t + stride - 1) / stride;
    uint32_t pre_sum = 0;
    if (lane < 6) pre_sum = f.sums[0][6 * (size_t)k + lane];
    const bool restricted = f.cell_act && !f.pre_flags[0];      // preemptive mode with inactive cells (in_update)
    uint32_t in[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    constexpr int kRows = 6;
    const size_t p_first = (size_t)min(rem + stride * v_first, f.H - 1) * W + x_lo;
    for (int y0 = rem + stride * v_first; y0 <= y_hi; y0 += kRows * stride) {
        for (int x = x_lo + lane; x - lane <= x_hi; x += 64) {
            size_t p[kRows];
            uint32_t lb[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int y = y0 + u * stride;
                p[u] = (y <= y_hi && x <= x_hi) ? (size_t)y * W + x : p_first;
                lb[u] = f.labels[p[u]];
            }
            uint32_t pix[kRows];
            bool m[kRows];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int y = y0 + u * stride;
                m[u] = y <= y_hi && x <= x_hi && lb[u] == (uint32_t)k && (!restricted || f.cell_act[(y / (2 * f.S)) * f.pre_cw + x / (2 * f.S)]);
                pix[u] = f.lab[m[u] ? p[u] : p_first];
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                if (!m[u]) continue;
                in[0] += 1u; in[1] += (uint32_t)(y0 + u * stride); in[2] += (uint32_t)x;
                in[3] += pix[u] & 255u; in[4] += (pix[u] >> 8) & 255u; in[5] += (pix[u] >> 16) & 255u;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) in[q] = wave_reduce_add(in[q]);
    uint32_t mine = in[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) mine = lane == q ? in[q] : mine;
    if (lane < 6) f.sums[0][6 * (size_t)k + lane] = pre_sum + mine;
}

void launch_rd_assign(const FrameDev& f_, const float* patchf, bool l2, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const dim3 grid((f.W + 63) / 64, (Hv + kRdBH - 1) / kRdBH, nframes);
    if (l2) launch(k_rd_assign<true>, grid, dim3(256), 0, st, f, patchf, rem, stride, Hv, stale_sums ? 1 : 0);
    else launch(k_rd_assign<false>, grid, dim3(256), 0, st, f, patchf, rem, stride, Hv, stale_sums ? 1 : 0);
}

void launch_member_sums(const FrameDev& f_, int nframes, int rem, int stride, int sbuf, int margin, hipStream_t st) {
    const FrameDev f = rotated(f_, 0, sbuf);
    launch(k_member_sums, dim3((f.K + 3) / 4, nframes), dim3(256), 0, st, f, rem, stride, margin);
}

// ---- 'noq': ContextRealDistNoQ::assign_clusters_proto (src/context.cpp:461-499) ------------------------------
// Float centroids; per pixel-candidate the reference evaluates, in this order and with these roundings (checked against
// the disassembly of the reference built with its own flags):
//   manhattan:  ((|dr| + |dg|) + |db|) + |coef*(j - cx)|, then + |coef*(i - cy)|
//   otherwise:  t = dg*dg; t = fma(dr,dr,t); t = fma(db,db,t); t = fma(dx,dx,t); t + (dy*dy)
//               (vmulss dg,dg; vfmadd132ss dr,t,dr; .. db ..; .. dx ..; vaddss dy2 -- the square that is rounded on its own is dg's, not dr's:
//               until round 6 this kernel had the two the other way round, one pixel in ~10^5 of a noise frame on the wrong side of a tie)
// and the window [ (int)(cy - S), (int)((cy + S) + 1) ) x [ (int)(cx - S), (int)((cx + S) + 1) ) clipped to the image.
// The candidate list is gathered with one pixel of slack around the integer window (the float expression can round up
// across an integer); the exact test decides.
template <bool MANHATTAN>
__global__ __launch_bounds__(256) void k_noq_assign(FrameDev f, float coef, int rem, int stride, int Hv, int stale_sums) {
    f.select(blockIdx.z);
    __shared__ uint32_t s_raw[192];
    __shared__ uint32_t s_k[64];
    __shared__ __attribute__((aligned(16))) float s_c[64 * 8];
    __shared__ int s_cnt;
    const int wave = (int)rfl((uint32_t)(threadIdx.x >> 6)), lane = LANE();
    const int S = f.S, W = f.W, H = f.H;
    const float fS = (float)S;
    const int x0 = blockIdx.x * 64;
    const int v0 = blockIdx.y * kRdBH;
    if (wave == (int)((blockIdx.x + blockIdx.y) & 3u)) {
        const int v_last = min(v0 + kRdBH, Hv) - 1;
        TileWindow win;
        win.wy_lo = rem + stride * v0 - S - 1; win.wy_hi = rem + stride * v_last + S + 1;
        win.wx_lo = x0 - S - 1; win.wx_hi = min(x0 + 63, W - 1) + S + 1;
        const int cnt = gather_candidates(f, win, s_raw, lane);
        wave_lds_sync();
        if (cnt <= 64) {
            const uint32_t mytag = lane < cnt ? s_raw[128 + lane] : 0xFFFFFFFFu;
            int rank = 0;
            for (int j = 0; j < cnt; ++j) rank += s_raw[128 + j] < mytag ? 1 : 0;
            if (lane < cnt) {
                const uint32_t k = mytag & 0xFFFFu;
                s_k[rank] = k;
                const float4* src = reinterpret_cast<const float4*>(f.cl_f + 8 * (size_t)k);
                float4* dst = reinterpret_cast<float4*>(s_c + rank * 8);
                dst[0] = src[0]; dst[1] = src[1];
            }
        }
        if (lane == 0) s_cnt = cnt;
    }
    const int x = x0 + lane;
    const bool xok = x < W;
    const int xc = min(x, W - 1);
    const int vw = v0 + wave * kRdRows;
    uint32_t pix[kRdRows];
    int yr[kRdRows];
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) {
        yr[r] = rem + stride * min(vw + r, Hv - 1);
        pix[r] = f.lab[(size_t)yr[r] * W + xc];
    }
    __syncthreads();
    const int cnt = s_cnt;
    float best[kRdRows];
    uint32_t bk[kRdRows];
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) { best[r] = FLT_MAX; bk[r] = 0xFFFFu; }
    const float fx = (float)x;
    auto consider = [&](uint32_t k, float cy, float cx, float cr, float cg, float cb) {
        const int y_lo = max((int)(cy - fS), 0), y_hi = min((int)((cy + fS) + 1.0f), H);
        const int x_lo = max((int)(cx - fS), 0), x_hi = min((int)((cx + fS) + 1.0f), W);
        if (x_hi <= x0 || x_lo > x0 + 63) return;            // uniform
        const bool inx = x >= x_lo && x < x_hi;
        const float dx = coef * (fx - cx);
#pragma unroll
        for (int r = 0; r < kRdRows; ++r) {
            if (yr[r] < y_lo || yr[r] >= y_hi) continue;       // uniform
            const float dr = (float)(pix[r] & 255u) - cr;
            const float dg = (float)((pix[r] >> 8) & 255u) - cg;
            const float db = (float)((pix[r] >> 16) & 255u) - cb;
            const float dy = coef * ((float)yr[r] - cy);
            float d;
            if (MANHATTAN) {
                d = ((__builtin_fabsf(dr) + __builtin_fabsf(dg)) + __builtin_fabsf(db)) + __builtin_fabsf(dx);
                d = d + __builtin_fabsf(dy);
            } else {
                float t = dg * dg;
                t = __builtin_fmaf(dr, dr, t);
                t = __builtin_fmaf(db, db, t);
                t = __builtin_fmaf(dx, dx, t);
                const float dy2 = dy * dy;         // (its own rounding: plain operators under `fp contract(off)`, see the top of the file)
                d = t + dy2;
            }
            if (inx && d < best[r]) { best[r] = d; bk[r] = k; }
        }
    };
    if (cnt <= 64) {
        for (int c = 0; c < cnt; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(s_c + c * 8);
            const float cb = s_c[c * 8 + 4];
            consider(rfl(s_k[c]), __uint_as_float(rfl(__float_as_uint(a.x))), __uint_as_float(rfl(__float_as_uint(a.y))),
                     __uint_as_float(rfl(__float_as_uint(a.z))), __uint_as_float(rfl(__float_as_uint(a.w))), __uint_as_float(rfl(__float_as_uint(cb))));
        }
    } else {
        for (uint32_t ph = 0; ph < 4; ++ph) {
            for (int k0 = 0; k0 < f.K; k0 += 64) {
                const int k = k0 + lane;
                const bool live = k < f.K;
                const uint32_t tg = live ? f.cl_tag[k] : 0xFFFFFFFFu;
                unsigned long long m = __ballot(live && (tg >> 16) == ph);
                while (m) {
                    const int c = __builtin_ctzll(m);
                    m &= m - 1;
                    const float* cc = f.cl_f + 8 * (size_t)(k0 + c);
                    consider((uint32_t)(k0 + c), cc[0], cc[1], cc[2], cc[3], cc[4]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRdRows; ++r) {
        if (!xok || vw + r >= Hv) continue;
        const size_t p = (size_t)yr[r] * W + x;
        if (best[r] < FLT_MAX) {
            st_stream(f.labels + p, (uint16_t)bk[r]);
        } else if (stale_sums) {
            const uint32_t old = f.labels[p];
            if (old != 0xFFFFu) {
                const uint32_t oyx = f.cl_yx[old];
                if ((abs(yr[r] - (int)(oyx >> 16)) > S + 1 || abs(x - (int)(oyx & 0xFFFFu)) > S + 1) && in_update(f, yr[r], x))      // same margin as k_member_sums
                    global_accumulate(f.sums[0], old, (uint32_t)yr[r], (uint32_t)x, pix[r]);
            }
        }
    }
}

void launch_noq_assign(const FrameDev& f_, float coef, bool manhattan, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const dim3 grid((f.W + 63) / 64, (Hv + kRdBH - 1) / kRdBH, nframes);
    if (manhattan) launch(k_noq_assign<true>, grid, dim3(256), 0, st, f, coef, rem, stride, Hv, stale_sums ? 1 : 0);
    else launch(k_noq_assign<false>, grid, dim3(256), 0, st, f, coef, rem, stride, Hv, stale_sums ? 1 : 0);
}

}  // namespace fslic
