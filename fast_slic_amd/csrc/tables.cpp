// tables.cpp -- host-built tables of the engine: the RGB->LAB look-up tables (src/cielab.h:296-305) and the spatial
// patch of set_spatial_patch (src/context.cpp:22-40) with its device encodings.  Part of the host engine, see
// engine_internal.h.
#include "engine_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace fslic {

namespace {
// ---- RGB->LAB tables, built exactly like FastCIELabCvt's constructor (src/cielab.h:296-305) ----
// The gamma table is the sRGB transfer function given in the header comment (src/cielab.h:11-20),
// evaluated in double and stored as float like _srgb_gamma_tbl.  lab_tbl depends on the host libm's
// powf (src/cielab.h:327-332), so it is built on the host and uploaded, never recomputed on device.
HostTables g_tables;
std::once_flag g_tables_once;

void build_host_tables() {
    static const float C[9] = {0.43395633f, 0.37621531f, 0.18984309f, 0.2126729f, 0.7151522f,
                               0.072175f,   0.01775782f, 0.1094756f,  0.87283638f};
    for (int i = 0; i < 256; i++) {
        const double v = i / 255.0;
        const double X = (v <= 0.04045) ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4);
        const float Xf = (float)X;
        g_tables.gamma[i] = (uint16_t)(int)(Xf * 8192);
    }
    for (int i = 0; i < 9; i++) g_tables.cb[i] = (int)roundf(C[i] * 65536);
    for (int i = 0; i <= 8192; i++) {
        const float v = (float)i / 8192;
        const float lo = 7.787f * v + 0.137931f;
        const float hi = powf(v, 0.333333f);
        g_tables.lab[i] = (uint16_t)(int)roundf(((v > 0.008856f) ? hi : lo) * 8192);
    }
    g_tables.lab[8193] = 0;
}

// src/context.cpp:22-40 set_spatial_patch (host, float arithmetic identical to the reference)
void build_patch(int S, float compactness, int color_shift, bool manhattan, std::vector<uint16_t>& patch) {
    const int P = 2 * S + 1;
    patch.resize((size_t)P * P);
    float coef = 1.0f / ((float)S / compactness);
    coef *= (1 << color_shift);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
            if (manhattan)
                patch[(size_t)i * P + j] = (uint16_t)(coef * (std::abs(i - S) + std::abs(j - S)));
            else
                patch[(size_t)i * P + j] = (uint16_t)(coef * hypotf((float)(i - S), (float)(j - S)));
        }
}

// The Manhattan patch is a function of d = |di| + |dj| only.  Find (M, C) with (d*M + C) >> 16 == lut[d]
// for every d in [0, 2S] so the kernel can form the spatial term with one 24-bit multiply-add and fold it
// into the packed key.  The offset C absorbs the float rounding of coef*d in set_spatial_patch (e.g. S=22,
// compactness=10: lut[11] == 10 although coef < 10/11).  Returns false when no pair exists (the generic
// kernel then reads the patch itself).
bool find_multiplier(int S, float coef, const std::vector<uint16_t>& patch, uint32_t& M, uint32_t& C, uint32_t& max_spatial) {
    const int P = 2 * S + 1;
    std::vector<int64_t> L(2 * S + 1);
    max_spatial = 0;
    for (int d = 0; d <= 2 * S; d++) {
        // element of the patch with |di|+|dj| == d: (S - min(d,S), S + (d - min(d,S)))
        const int di = std::min(d, S), dj = d - di;
        L[d] = patch[(size_t)(S - di) * P + (S + dj)];
        max_spatial = std::max<uint32_t>(max_spatial, (uint32_t)L[d]);
    }
    if (L[0] != 0) return false;
    const int64_t M0 = (int64_t)llround((double)coef * 65536.0);
    for (int64_t delta = 0; delta <= 512; delta++) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const int64_t m = M0 + (sgn ? -delta : delta);
            if (m < 0 || m >= (1 << 24) || (sgn && delta == 0)) continue;
            int64_t lo = 0, hi = 65535;
            for (int d = 1; d <= 2 * S && lo <= hi; d++) {
                lo = std::max<int64_t>(lo, L[d] * 65536 - d * m);
                hi = std::min<int64_t>(hi, (L[d] + 1) * 65536 - 1 - d * m);
            }
            if (lo <= hi) {
                if ((int64_t)(2 * S) * m + lo >= (1ll << 30)) return false;
                M = (uint32_t)m;
                C = (uint32_t)lo;
                return true;
            }
        }
    }
    return false;
}
}  // namespace

const HostTables& host_tables() {
    std::call_once(g_tables_once, build_host_tables);
    return g_tables;
}

// f32 spatial patch of the float-distance variants, expression by expression as the reference builds it:
//   'standard' BaseContext<float>::set_spatial_patch, src/context.cpp:22-40 (the cast to DistType is a no-op for float)
//   'l2'       ContextRealDistL2::set_spatial_patch, src/context.cpp:436-447; the reference build contracts
//              di*di + dj*dj into fma(dj, dj, di*di) (disassembly of the reference built with its own flags), spelled out here
int configure_patchf(Slot& s, int S, const fslic_params* p) {
    const int color_shift = p->convert_to_lab ? 1 : 0;
    const bool manhattan = p->manhattan_spatial_dist != 0;
    if (s.pf_variant == p->variant && s.pf_S == S && s.pf_compactness == p->compactness && s.pf_shift == color_shift && s.pf_manhattan == manhattan)
        return FSLIC_OK;
    const int P = 2 * S + 1;
    float coef = 1.0f / ((float)S / p->compactness);
    coef *= (1 << color_shift);
    for (int i = 0; i < P; i++)
        for (int j = 0; j < P; j++) {
            float v;
            if (p->variant == FSLIC_VARIANT_REALDIST_L2) {
                const float di = coef * (i - S), dj = coef * (j - S);
                const float di2 = di * di;
                v = std::fmaf(dj, dj, di2);
            } else if (manhattan) {
                v = (float)(coef * (std::abs(i - S) + std::abs(j - S)));
            } else {
                v = (float)(coef * hypotf((float)(i - S), (float)(j - S)));
            }
            s.h_patchf[(size_t)i * P + j] = v;
        }
    HIPCHK(hipMemcpyAsync(s.d_patchf, s.h_patchf, sizeof(float) * (size_t)P * P, hipMemcpyHostToDevice, s.st));
    s.pf_variant = p->variant; s.pf_S = S; s.pf_compactness = p->compactness; s.pf_shift = color_shift; s.pf_manhattan = manhattan;
    return FSLIC_OK;
}

// Spatial-distance configuration of a frame: the reference's patch (src/context.cpp:22-40) and the three device
// encodings of it.  Cached per slot: a video stream calls iterate() with the same (S, compactness, options)
// for every frame, so the tables are built and uploaded once.
//   packed kernel : u16 tables of the patch values (V-shaped row vectors or pair-packed columns), kTabOut outside the window
//   32-bit kernel : exact fixed-point multiplier (M, C), or an LDS table when none exists / non-Manhattan
//   generic kernel: the patch itself
int configure_spatial(Slot& s, int S, const fslic_params* p) {
    const int color_shift = p->convert_to_lab ? 1 : 0;   // src/cielab.h:284,352 ; src/context.cpp:126
    const bool manhattan = p->manhattan_spatial_dist != 0;
    const int stride_it = p->subsample_stride;
    if (s.sp_valid && s.sp_S == S && s.sp_compactness == p->compactness && s.sp_shift == color_shift &&
        s.sp_manhattan == manhattan && s.sp_stride == stride_it)
        return FSLIC_OK;
    FrameDev& f = s.f;
    std::vector<uint16_t> patch;
    build_patch(S, p->compactness, color_shift, manhattan, patch);
    std::memcpy(s.h_patch, patch.data(), patch.size() * sizeof(uint16_t));
    s.sp_patch_uploaded = false;
    uint32_t M = 0, Coff = 0, max_sp = 0;
    float coef = 1.0f / ((float)S / p->compactness);
    coef *= (1 << color_shift);
    const bool have_mc = manhattan && find_multiplier(S, coef, patch, M, Coff, max_sp);
    for (uint16_t v : patch) max_sp = std::max<uint32_t>(max_sp, v);
    bool tiled_ok = (max_sp + 765u < 16384u) && S >= 8;
    f.lut_words = 0; f.lut_row = 0; f.lut_oow = 0; f.lut = s.d_lut;
    const size_t Pp = (size_t)(2 * S + 1);
    if (tiled_ok && !have_mc) {
        // LDS table mode.  Manhattan: 1-D in d = ux + uy (out-of-window -> 2S+1, so any index > 2S is huge).
        // Otherwise 2-D in (uy, ux) with one extra out-of-window row and column.
        if (manhattan) {
            const int oow = 2 * S + 1, words = 4 * S + 3;
            if ((size_t)words <= kLutMaxWords) {
                for (int d = 0; d < words; d++) {
                    if (d <= 2 * S) { const int di = std::min(d, S), dj = d - di; s.h_lut[d] = (uint32_t)patch[(size_t)(S - di) * Pp + (S + dj)] << 16; }
                    else s.h_lut[d] = kHuge;
                }
                f.lut_words = words; f.lut_row = 1; f.lut_oow = oow;
            }
        } else {
            const int row = S + 2, words = row * row;
            if ((size_t)words <= kLutMaxWords) {
                for (int uy = 0; uy < row; uy++)
                    for (int ux = 0; ux < row; ux++)
                        s.h_lut[uy * row + ux] = (uy > S || ux > S) ? kHuge : ((uint32_t)patch[(size_t)(S + uy) * Pp + (S + ux)] << 16);
                f.lut_words = words; f.lut_row = row; f.lut_oow = S + 1;
            }
        }
    }
    if (!have_mc && f.lut_words == 0) tiled_ok = false;        // the 32-bit kernel has neither multiplier nor table
    f.tab = s.d_tab; f.tab_words = 0; f.tab_nrpad = 0; f.tab_dyoff = 0; f.tab_vmode = 0; f.tab_xout = 0;
    // The 2-D table costs (S+2) * (2(S+pad)+1) * 2 bytes of L2 -> LDS traffic per block (26 KB at S = 72); the V table
    // a few hundred bytes, at the price of walking a candidate twice when its centre row lies inside the
    // wavefront's rows (frequent when S is small).  Measured: V wins at S = 72 (4K, K = 1600), loses at S = 24.
    const size_t tab2d_bytes = (size_t)(S + 2) * (2 * (S + std::max(7 * std::min(stride_it, 3), 15)) + 1) * 2;
    constexpr size_t vtab_min = 12288;
    f.tabs = s.d_tab + kTabMaxBytes / 2; f.tabs_words = 0; f.tabs_dyoff = 0;
    f.tabs16 = s.d_tab + kTabMaxBytes; f.tabs16_words = 0; f.tabs16_dyoff = 0; f.tabs16_nrpad = 0;
    f.tabv16 = s.d_tab + kTabMaxBytes; f.tabv16_words = 0; f.tabv16_dyoff = 0; f.tab_rows32 = 0;      // (the third slot serves one table mode or the other)
    if (manhattan && tab2d_bytes > vtab_min && S >= 8 && max_sp + 765u < 1024u) {
        // Row-vector V tables (the Manhattan patch depends on |di| + |dj| only): one for the full pass (stride 1, up to
        // 32 rows per wavefront) and one for the subsampled passes (stride_it, up to 16 rows per wavefront).
        const int xout = 2 * S + 1;
        auto build = [&](int stride, int rows, uint16_t* dst, int& words, int& cc) -> bool {
            const int pad = (rows - 1) * stride;
            const int Cc = xout + S + pad;
            const int tlen = 2 * Cc + 1 + (rows - 8) * stride;   // entries that can be addressed: t0 + 8*stride for every further eight rows
            if ((size_t)tlen * 16 > kTabMaxBytes) return false;
            auto V = [&](int t) -> uint16_t {
                const int d = std::abs(t - Cc);
                if (d > 2 * S) return kTabOut;
                const int di = std::min(d, S), dj = d - di;
                return patch[(size_t)(S - di) * Pp + (S + dj)];
            };
            for (int t = 0; t < tlen; t++)
                for (int k = 0; k < 8; k++) dst[(size_t)t * 8 + k] = V(t + k * stride);
            words = tlen * 4; cc = Cc;
            return true;
        };
        int w1 = 0, c1 = 0, ws = 0, cs = 0, w16 = 0, c16 = 0;
        const int st3 = std::min(stride_it, 3);
        // the full pass's table for 32 rows per wavefront where it fits the LDS budget (S <= 411), else for 16 (S <= 419); next to a
        // 32-row table the 16-row one is kept too: launches of at most 16 rows per wavefront reserve 768 bytes of LDS per block less
        const bool rows32 = build(1, 32, s.h_tab, w1, c1);
        if ((rows32 || build(1, 16, s.h_tab, w1, c1)) && build(st3, 16, s.h_tab + kTabMaxBytes / 2, ws, cs)) {
            f.tab_words = w1; f.tab_dyoff = c1; f.tab_vmode = 1; f.tab_xout = xout; f.tab_rows32 = rows32 ? 1 : 0;
            f.tabs_words = ws; f.tabs_dyoff = cs;
            if (rows32 && build(1, 16, s.h_tab + kTabMaxBytes, w16, c16)) { f.tabv16_words = w16; f.tabv16_dyoff = c16; }
            tiled_ok = true;
        }
    }
    if (!f.tab_vmode && S >= 8 && max_sp + 765u < 1024u) {
        // Pair-packed column-major copy of the patch: dword [ux][j] = (e(j), e(j + stride)), e(j) = patch[dy = j - dyoff][ux]
        // or kTabOut outside the window.  A pair of rows is then ONE 4-byte-aligned ds_read_b32 with an immediate offset
        // (the u16 layout needed either a v_perm per pair or, at stride 1, 2-byte-aligned b128 reads, which cost 64 clk of
        // the CU's LDS pipe each).  One table per stride: `tab` for the full pass, `tabs` for the subsampled passes.
        auto build2d = [&](int stride, int rows, uint32_t* dst, int& words, int& nrpad_out, int& dyoff_out) -> bool {
            const int pad = (rows - 1) * stride;
            const int dyoff = S + pad, NR = 2 * dyoff + 1;
            int nrpad = NR | 1;                                       // odd dword pitch: conflict-free columns
            const size_t bytes = ((size_t)(S + 2) * nrpad * 4 + 15) / 16 * 16;
            if (bytes > kTabMaxBytes) return false;
            auto e = [&](int ux, int j) -> uint32_t {
                const int dy = j - dyoff;
                if (j < 0 || j >= NR || dy < -S || dy > S || ux > S) return kTabOut;
                return (uint32_t)patch[(size_t)(S + dy) * Pp + (S + ux)];
            };
            for (int ux = 0; ux <= S + 1; ux++)
                for (int j = 0; j < nrpad; j++) dst[(size_t)ux * nrpad + j] = e(ux, j) | (e(ux, j + stride) << 16);
            for (size_t i = (size_t)(S + 2) * nrpad; i < bytes / 4; i++) dst[i] = kTabOut | ((uint32_t)kTabOut << 16);
            words = (int)(bytes / 4); nrpad_out = nrpad; dyoff_out = dyoff;
            return true;
        };
        int w1 = 0, n1 = 0, d1 = 0, ws = 0, ns = 0, ds = 0;
        const int st3 = std::min(stride_it, 3);
        if (build2d(1, 16, reinterpret_cast<uint32_t*>(s.h_tab), w1, n1, d1) &&
            build2d(st3, 8, reinterpret_cast<uint32_t*>(s.h_tab + kTabMaxBytes / 2), ws, ns, ds)) {
            f.tab_words = w1; f.tab_nrpad = n1; f.tab_dyoff = d1;
            f.tabs_words = ws; f.tabs_nrpad = ns; f.tabs_dyoff = ds;
            tiled_ok = true;
            // the same for 16 rows per wavefront, kept only while a block (10 KB of lists and accumulators + the table) leaves six
            // blocks per CU resident
            int w16 = 0, n16 = 0, d16 = 0;
            if (build2d(st3, 16, reinterpret_cast<uint32_t*>(s.h_tab + kTabMaxBytes), w16, n16, d16) && (size_t)w16 * 4 <= 16384) {
                f.tabs16_words = w16; f.tabs16_nrpad = n16; f.tabs16_dyoff = d16;
            }
        }
    }
    if (stride_it > 3 && !have_mc && f.lut_words == 0) tiled_ok = false;   // strides the packed kernel does not take need the 32-bit kernel
    f.M = M;
    f.C = Coff;
    if (f.tab_words > 0)
        HIPCHK(hipMemcpyAsync(s.d_tab, s.h_tab, (size_t)f.tab_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.tabs_words > 0)
        HIPCHK(hipMemcpyAsync(const_cast<uint16_t*>(f.tabs), s.h_tab + kTabMaxBytes / 2, (size_t)f.tabs_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.tabs16_words > 0)
        HIPCHK(hipMemcpyAsync(const_cast<uint16_t*>(f.tabs16), s.h_tab + kTabMaxBytes, (size_t)f.tabs16_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.tabv16_words > 0)
        HIPCHK(hipMemcpyAsync(const_cast<uint16_t*>(f.tabv16), s.h_tab + kTabMaxBytes, (size_t)f.tabv16_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.lut_words > 0)
        HIPCHK(hipMemcpyAsync(s.d_lut, s.h_lut, (size_t)f.lut_words * 4, hipMemcpyHostToDevice, s.st));
    s.sp_tiled_ok = tiled_ok;
    s.sp_valid = true; s.sp_S = S; s.sp_compactness = p->compactness; s.sp_shift = color_shift;
    s.sp_manhattan = manhattan; s.sp_stride = stride_it;
    return FSLIC_OK;
}

}  // namespace fslic
