// arena.cpp -- per-slot device and pinned-host buffers of the engine: allocation, carving, release.  The reference
// allocates its planes per call (BaseContext ctor, src/context.h:59-66; AlignedArray, src/simd-helper.hpp:106-118);
// here they are cached across calls keyed by (H, W, K).  Part of the host engine, see engine_internal.h.
#include "engine_internal.h"

#include <algorithm>
#include <cmath>

namespace fslic {

void free_slot(Slot& s) {
    // the slot's host thread has been stopped by the caller (stop_slot_thread)
    for (auto& g : s.graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
    s.graphs.clear();
    if (s.arena) hipFree(s.arena);
    if (s.lsc_arena) hipFree(s.lsc_arena);
    if (s.h_lsc_lut) hipHostFree(s.h_lsc_lut);
    if (s.h_patchf) hipHostFree(s.h_patchf);
    if (s.h_clf) hipHostFree(s.h_clf);
    if (s.h_upd) hipHostFree(s.h_upd);
    if (s.d_ptrs) hipFree(s.d_ptrs);
    if (s.d_gen) hipFree(s.d_gen);
    if (s.h_ptrs) hipHostFree(s.h_ptrs);
    if (s.h_cl) hipHostFree(s.h_cl);
    if (s.h_misc) hipHostFree(s.h_misc);
    if (s.h_lut) hipHostFree(s.h_lut);
    if (s.h_tab) hipHostFree(s.h_tab);
    if (s.h_patch) hipHostFree(s.h_patch);
    if (s.h_cand_leader) hipHostFree(s.h_cand_leader);
    if (s.h_cand_area) hipHostFree(s.h_cand_area);
    for (auto& e : s.ev) if (e) hipEventDestroy(e);
    for (auto& e : s.ev_it) if (e) hipEventDestroy(e);
    if (s.st) hipStreamDestroy(s.st);
    s = Slot();
}

// (Re)allocate the slot's buffers for groups of up to G frames of H x W with K clusters and carve the arena.
int prepare(fslic_engine* e, Slot& s, int H, int W, int K, int S, int G) {
    const size_t N = (size_t)H * W;
    const int sh = [&] { int v = 2; while ((1 << v) < S) v++; return v; }();   // bin side = 2^ceil(log2 S): ~(side/S)^2 <= 4 clusters per bin
    const int ncx = (W + (1 << sh) - 1) >> sh, ncy = (H + (1 << sh) - 1) >> sh;
    const size_t ncell = (size_t)ncx * ncy;
    const size_t P = (size_t)(2 * S + 1);
    if (G < s.cap_frames && s.keyH == H && s.keyW == W && s.keyK == K) G = s.cap_frames;

    if (s.h_cl_words < (size_t)G * 4 * K) {
        if (s.h_cl) hipHostFree(s.h_cl);
        s.h_cl = nullptr; s.h_cl_words = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_cl, sizeof(uint32_t) * 4 * (size_t)K * G));
        s.h_cl_words = (size_t)G * 4 * K;
    }
    if (!s.h_misc) HIPCHK(hipHostMalloc((void**)&s.h_misc, 64 * kMaxGroup));
    if (!s.h_lut) HIPCHK(hipHostMalloc((void**)&s.h_lut, kLutMaxWords * 4));
    if (!s.h_tab) HIPCHK(hipHostMalloc((void**)&s.h_tab, 3 * kTabMaxBytes));
    if (s.h_patchf_cap < P * P) {
        if (s.h_patchf) hipHostFree(s.h_patchf);
        s.h_patchf = nullptr; s.h_patchf_cap = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_patchf, sizeof(float) * P * P));
        s.h_patchf_cap = P * P;
    }
    if (s.h_patch_cap < P * P) {
        if (s.h_patch) hipHostFree(s.h_patch);
        s.h_patch = nullptr; s.h_patch_cap = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_patch, sizeof(uint16_t) * P * P));
        s.h_patch_cap = P * P;
    }

    // ---- carve: shared tables, then one frame's sub-arena ----
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_patch = take(P * P * 2);
    const size_t o_lut = take(kLutMaxWords * 4);
    const size_t o_tab = take(3 * kTabMaxBytes);
    const size_t o_patchf = take(P * P * 4);
    const size_t shared_bytes = off;
    off = 0;
    const size_t o_lab = take(N * 4), o_labels = take(N * 2);
    const size_t o_cl = take((size_t)K * 4 * 8);                     // yx, lab, n, moved, tag, shadow, yx', yx'' (contiguous)
    const size_t o_clf = take((size_t)K * 8 * 4);                    // float centroids of the 'noq' variant
    const int pre_cw = (W + 2 * S - 1) / (2 * S), pre_ch = (H + 2 * S - 1) / (2 * S);      // src/preemptive.h:36-37
    const size_t o_pre = take(((size_t)K * 2 + (size_t)pre_cw * pre_ch + 16) * 4);         // is_updatable, is_active, active cells, flags
    // partial centroid sums of the block kernel: one entry per (cluster, assign block its window can overlap);
    // blocks are 64 px wide and at least 32 visited rows (stride >= 1) tall
    // (the smallest assign block: 64 columns x 16 visited rows).  Both counts are rounded up to powers of two: an entry is addressed by
    // the block's coordinates modulo the grid (device_common.h, cpart_slot), so that the cluster pass can load a cluster's entries
    // before it knows where the cluster is
    auto pow2ceil = [](int v) { int p2 = 1; while (p2 < v) p2 <<= 1; return p2; };
    const int nbx = pow2ceil(2 * S / 64 + 2), nby = pow2ceil(2 * S / 16 + 2);
    const size_t o_cpart = take((size_t)K * nbx * nby * 6 * 4);
    const size_t o_zero = off;
    const size_t o_sums = take((size_t)K * 6 * 4), o_sums1 = take((size_t)K * 6 * 4);
    const size_t o_cc0 = take(ncell * 4), o_cc1 = take(ncell * 4);
    const size_t o_misc = take(256);                                 // ovf_cnt[0..1], err, ovf_cnt[2], cca counters
    const size_t o_arrive = take((size_t)K * 4);                     // fused cluster pass: arrival counters
    const size_t zero_bytes = off - o_zero;
    const size_t o_ci0 = take(ncell * kCellCap * sizeof(ClusterRec)), o_ci1 = take(ncell * kCellCap * sizeof(ClusterRec));
    const size_t o_ov0 = take((size_t)K * sizeof(ClusterRec)), o_ov1 = take((size_t)K * sizeof(ClusterRec)), o_ov2 = take((size_t)K * sizeof(ClusterRec));
    const size_t ntiles = (size_t)((W + kCcaTile - 1) / kCcaTile) * ((H + kCcaTileH - 1) / kCcaTileH);
    const size_t TT = (size_t)kCcaTilePx;
    const size_t NP = ntiles * TT;            // node capacity: every tile's full range
    const size_t o_lid = take(N * 2), o_nrec = take(NP * 16), o_narea = take(NP * 4);
    const size_t o_nfinal = take(NP * 2), o_vedge = take(ntiles * 2 * kCcaTileH * 4), o_roots = take(NP * 4);
    const size_t o_candl = take(ntiles * TT * 4), o_canda = take(ntiles * TT * 4);
    const size_t o_keepl = take((size_t)K * 4), o_keepb = take((size_t)K * 2);
    const size_t o_rgb = take(N * 3), o_out = take(N * 2);
    const size_t frame_bytes = off;
    const size_t total = shared_bytes + frame_bytes * (size_t)G;

    if (s.arena_cap < total) {
        if (s.arena) { hipFree(s.arena); s.arena = nullptr; s.arena_cap = 0; }
        HIPCHK(hipMalloc((void**)&s.arena, total));
        s.arena_cap = total;
    }
    s.cap_frames = G;
    s.frame_bytes = frame_bytes;
    char* b = s.arena + shared_bytes;          // frame 0
    FrameDev& f = s.f;
    f.frame_bytes = frame_bytes;
    f.H = H; f.W = W; f.N = (int)N; f.K = K; f.S = S; f.T = 2 * S + 32; f.T_rcp = 0xFFFFFFFFu / (uint32_t)f.T;
    f.cs_shift = sh; f.ncx = ncx; f.ncy = ncy; f.ncell = (int)ncell;
    f.lab = (uint32_t*)(b + o_lab);
    f.labels = (uint16_t*)(b + o_labels);
    uint32_t* cl = (uint32_t*)(b + o_cl);
    f.cl_yx = cl; f.cl_lab = cl + K; f.cl_n = cl + 2 * (size_t)K; f.cl_moved = cl + 3 * (size_t)K; f.cl_tag = cl + 4 * (size_t)K;
    f.cl_shadow = cl + 5 * (size_t)K;
    f.cl_f = nullptr;                                                // set per call for the 'noq' variant only
    f.cl_upd = f.cl_act = f.cell_act = f.pre_flags = nullptr;        // set per call in preemptive mode only
    f.pre_cw = pre_cw; f.pre_ch = pre_ch;
    s.d_pre = (uint32_t*)(b + o_pre);
    s.d_clf = (float*)(b + o_clf);
    s.d_yx_alt[0] = cl + 6 * (size_t)K; s.d_yx_alt[1] = cl + 7 * (size_t)K;
    f.cl_yx_out = s.d_yx_alt[0];
    f.cpart = (uint32_t*)(b + o_cpart); f.cpart_ns = nbx * nby; f.nbx = nbx;
    f.sums[0] = (uint32_t*)(b + o_sums); f.sums[1] = (uint32_t*)(b + o_sums1);
    f.cell_cnt[0] = (uint32_t*)(b + o_cc0); f.cell_cnt[1] = (uint32_t*)(b + o_cc1);
    uint32_t* misc = (uint32_t*)(b + o_misc);
    s.d_misc = misc;
    f.ovf_cnt[0] = misc + 0; f.ovf_cnt[1] = misc + 1; f.err_flag = misc + 2; f.ovf_cnt[2] = misc + 3;
    f.cl_arrive = (uint32_t*)(b + o_arrive);
    f.cell_items[0] = (ClusterRec*)(b + o_ci0); f.cell_items[1] = (ClusterRec*)(b + o_ci1);
    f.ovf_items[0] = (ClusterRec*)(b + o_ov0); f.ovf_items[1] = (ClusterRec*)(b + o_ov1); f.ovf_items[2] = (ClusterRec*)(b + o_ov2);
    s.d_patch = (uint16_t*)(s.arena + o_patch);
    f.patch2d = s.d_patch;
    s.d_lut = (uint32_t*)(s.arena + o_lut);
    s.d_tab = (uint16_t*)(s.arena + o_tab);
    s.d_patchf = (float*)(s.arena + o_patchf);
    s.pf_variant = -1;                                   // the device copy lives in the arena that was just re-carved
    s.zero_block = b + o_zero; s.zero_bytes = zero_bytes;
    CcaDev& c = s.c;
    c.frame_bytes = frame_bytes;
    c.H = H; c.W = W; c.N = (int)N; c.K = K;
    c.tiles_x = (W + kCcaTile - 1) / kCcaTile;
    c.lid = (uint16_t*)(b + o_lid); c.nrec = (uint4*)(b + o_nrec); c.narea = (uint32_t*)(b + o_narea);
    c.nfinal = (uint16_t*)(b + o_nfinal); c.vedge = (uint32_t*)(b + o_vedge); c.roots = (uint32_t*)(b + o_roots);
    c.counters = misc + 4;
    c.cand_leader = (int32_t*)(b + o_candl); c.cand_area = (uint32_t*)(b + o_canda);
    s.cand_capacity = ntiles * TT;
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
    f.lab_stamps = c.cand_area;
#endif
    s.d_keep_leader = (int32_t*)(b + o_keepl); s.d_keep_label = (uint16_t*)(b + o_keepb);
    s.d_rgb_stage = (uint8_t*)(b + o_rgb); s.d_out_stage = (uint16_t*)(b + o_out);
    s.keyH = H; s.keyW = W; s.keyK = K;
    s.sp_valid = false;    // device copies of the spatial tables live in the arena that was just re-carved
    (void)e;
    // testing aid: a freshly carved arena normally holds whatever the previous geometry left behind; FSLIC_POISON
    // makes that worst case deterministic (every byte 0xA5) so that reads of never-written memory show up in tests
    if (knobs().poison) HIPCHK(hipMemsetAsync(s.arena, 0xA5, total, s.st));
    // Bin slots are validated by a generation stamp (ClusterRec::pad == FrameDev::gen, gen >= 1 and strictly
    // increasing per slot), so a freshly carved region must not contain bytes that could pass for a stamp.
    s.stamp_block = b + o_ci0; s.stamp_bytes = o_ov0 - o_ci0;
    HIPCHK(hipMemset2DAsync(s.stamp_block, frame_bytes, 0, s.stamp_bytes, (size_t)G, s.st));
    return FSLIC_OK;
}

int ensure_prepared(fslic_engine* e, Slot& s, int H, int W, int K, int S, int G) {
    // room for the groups batching will form later (fslic_hip_pipeline_batching): applied HERE, so that every caller --
    // including the host-memory entry points, which stage frames into the arena before group_begin runs -- carves once
    G = std::max(G, std::min(e->reserve_frames.load(), (int)kMaxGroup));
    if (s.keyH == H && s.keyW == W && s.keyK == K && s.cap_frames >= G) return FSLIC_OK;
    HIPCHK(hipStreamSynchronize(s.st));
    return prepare(e, s, H, W, K, S, G);
}

// LSC variant: carve the arena (means, centroids, accumulators: no feature planes) for groups of up to G frames and (re)build the tables.  The tables follow
// src/lsc.cpp:24-27,76-101 expression by expression (float vs double evaluation included) and use the host libm,
// like the reference.
int prepare_lsc(Slot& s, int H, int W, int K, int S, int G, float compactness) {
    const size_t lut_floats = 1024 + 2 * (size_t)W + 2 * (size_t)H;
    if (s.lsc_H != H || s.lsc_W != W || s.lsc_K != K || s.lsc_G < G) {
        G = std::max(G, (s.lsc_H == H && s.lsc_W == W && s.lsc_K == K) ? s.lsc_G : 0);
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
        const size_t o_lut = take(lut_floats * 4);
        const size_t shared_bytes = off;
        off = 0;
        const size_t o_hist = take(768 * 4), o_extra = take((size_t)K * kLscExPitch * 8);
        const size_t zero_bytes = off - o_hist;
        const size_t o_means = take(64), o_cfeat = take((size_t)K * kLscCfPitch * 4);
        const size_t frame_bytes = off, total = shared_bytes + frame_bytes * (size_t)G;
        if (s.lsc_cap < total) {
            if (s.lsc_arena) { hipFree(s.lsc_arena); s.lsc_arena = nullptr; s.lsc_cap = 0; }
            HIPCHK(hipMalloc((void**)&s.lsc_arena, total));
            s.lsc_cap = total;
        }
        if (s.h_lsc_lut_cap < lut_floats) {
            if (s.h_lsc_lut) hipHostFree(s.h_lsc_lut);
            s.h_lsc_lut = nullptr; s.h_lsc_lut_cap = 0;
            HIPCHK(hipHostMalloc((void**)&s.h_lsc_lut, lut_floats * 4));
            s.h_lsc_lut_cap = lut_floats;
        }
        char* b = s.lsc_arena + shared_bytes;
        LscDev& l = s.l;
        l.frame_bytes = frame_bytes;
        l.hist = (uint32_t*)(b + o_hist); l.extra = (long long*)(b + o_extra);
        l.means = (float*)(b + o_means); l.cfeat = (float*)(b + o_cfeat);
        const float* lut = (const float*)(s.lsc_arena + o_lut);
        l.lut_color = lut; l.lut_x = lut + 1024; l.lut_y = lut + 1024 + 2 * (size_t)W;
        s.lsc_zero = b + o_hist; s.lsc_zero_bytes = zero_bytes; s.lsc_frame_bytes = frame_bytes;
        s.lsc_H = H; s.lsc_W = W; s.lsc_K = K; s.lsc_G = G;
        s.lsc_S = 0;                                   // tables live in the arena that was just carved
        if (knobs().poison) HIPCHK(hipMemsetAsync(s.lsc_arena, 0xA5, total, s.st));
    }
    if (s.lsc_S != S || s.lsc_compactness != compactness) {
        const float C_color = 20.0f;                   // src/lsc.h:8
        const float PI = 3.1415926f, halfPI = PI / 2;
        const float ratio = compactness / 100.0f;
        const float C_spatial = C_color * ratio;
        float* t = s.h_lsc_lut;
        for (int X = 0; X < 256; X++) {
            const float theta = halfPI * (X / 255.0f);
            const float cosine = (float)cos((double)theta), sine = (float)sin((double)theta);
            t[512 + X] = C_color * cosine * 2.55f;                       // a, b planes
            t[768 + X] = C_color * sine * 2.55f;
            t[X] = (float)((double)C_color * cos((double)theta));       // L plane: the product is formed in double
            t[256 + X] = (float)((double)C_color * sin((double)theta));
        }
        float* tx = t + 1024;
        float* ty = tx + 2 * (size_t)W;
        const float step = halfPI / S;
        for (int i = 0; i < W; i++) {
            const float theta = i * step;
            tx[i] = (float)((double)C_spatial * cos((double)theta));
            tx[W + i] = (float)((double)C_spatial * sin((double)theta));
        }
        for (int i = 0; i < H; i++) {
            const float theta = i * step;
            ty[i] = (float)((double)C_spatial * cos((double)theta));
            ty[H + i] = (float)((double)C_spatial * sin((double)theta));
        }
        HIPCHK(hipMemcpyAsync(const_cast<float*>(s.l.lut_color), t, lut_floats * 4, hipMemcpyHostToDevice, s.st));
        // block-level fixed point of the fused update's sums (LscDev::blk_fix_*): bounds of one pixel's feature and weight
        const double f_max = std::max({(double)C_color, (double)C_color * 2.55, (double)fabsf(C_spatial)});
        const double w_max = 2.0 * C_color * C_color + 4.0 * (C_color * 2.55) * (C_color * 2.55) + 4.0 * (double)C_spatial * C_spatial;
        int bits_f = 20, bits_w = 12;                  // never finer than the global accumulators (2^-20, 2^-12: lsc.hip kLscFixF / kLscFixW)
        while (bits_f > 0 && 1024.0 * f_max * ldexp(1.0, bits_f) >= 2147483648.0) bits_f--;
        const double w_bias = ceil(4.0 * (double)C_spatial * C_spatial);      // LscDev::blk_bias_w: the stored weight is never negative
        while (bits_w > 0 && 1024.0 * (w_max + w_bias) * ldexp(1.0, bits_w) >= 4294967296.0) bits_w--;
        s.l.blk_bias_w = (float)w_bias;
        s.l.blk_fix_f = (float)ldexp(1.0, bits_f); s.l.blk_fix_w = (float)ldexp(1.0, bits_w);
        s.l.blk_shift_f = 20 - bits_f; s.l.blk_shift_w = 12 - bits_w;
        s.lsc_S = S; s.lsc_compactness = compactness;
    }
    return FSLIC_OK;
}

int ensure_cand_capacity(Slot& s, size_t M) {
    if (s.h_cand_cap >= M) return FSLIC_OK;
    if (s.h_cand_leader) { hipHostFree(s.h_cand_leader); hipHostFree(s.h_cand_area); s.h_cand_leader = nullptr; s.h_cand_area = nullptr; }
    const size_t cap = std::max<size_t>(M, 16384);
    HIPCHK(hipHostMalloc((void**)&s.h_cand_leader, sizeof(int32_t) * cap));
    HIPCHK(hipHostMalloc((void**)&s.h_cand_area, sizeof(uint32_t) * cap));
    s.h_cand_cap = cap;
    return FSLIC_OK;
}

}  // namespace fslic
