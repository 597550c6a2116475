// engine.cpp -- host side of the MI355X SLIC engine and its C ABI (include/fslic_hip.h).
//
// The engine replaces, for the arch "hip/gfx950", what SlicModel.iterate builds per call in the
// reference: a Context (src/context.h:59-66), initialize_state(), iterate() (src/context.cpp:108-197)
// and its teardown (cfast_slic.pyx:171-197).  Device and pinned buffers are owned here and cached
// across calls; the only state that carries over between calls is the caller's Cluster[K].
#include "kernels.h"
#include "../../include/fslic_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <numeric>
#include <string>
#include <vector>

using namespace fslic;

static_assert(sizeof(fslic_cluster) == 32, "Cluster ABI (src/fast-slic-common.h:10-23)");

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(FSLIC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));       \
    } while (0)

// ---- RGB->LAB tables, built exactly like FastCIELabCvt's constructor (src/cielab.h:296-305) ----
// The gamma table is the sRGB transfer function given in the header comment (src/cielab.h:11-20),
// evaluated in double and stored as float like _srgb_gamma_tbl.  lab_tbl depends on the host libm's
// powf (src/cielab.h:327-332), so it is built on the host and uploaded, never recomputed on device.
struct HostTables {
    uint16_t gamma[256];
    uint16_t lab[8194];
    int cb[9];
};
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

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

constexpr int kDenseCap = 8192;          // leader-sorted candidates the select kernel hands over on a tie (== its LDS sort capacity)
constexpr size_t kTabMaxBytes = 40960;   // LDS budget of the packed kernel's spatial table (row-vector mode: per table)
constexpr size_t kLutMaxWords = 12288;   // 48 KB of LDS for the spatial table at most

struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // HIP events bracketing every subsampled assign launch of the loop (the roofline figure of bench.py sums them)
    static constexpr int kMaxTimedIters = 16;
    hipEvent_t ev_it[2 * kMaxTimedIters] = {};
    int n_timed_iters = 0;
    float assign_loop_ms = 0;        // sum of the fused assign launches' durations of the last group
    double assign_loop_px = 0;       // pixels those launches visited (all frames of the group)
    // device: one arena = [shared spatial tables][frame 0][frame 1]...; f / c hold frame 0's pointers
    char* arena = nullptr;
    size_t arena_cap = 0;
    int cap_frames = 0;              // frames the arena is carved for
    size_t frame_bytes = 0;
    FrameDev f{};
    CcaDev c{};
    // per-frame regions, given as frame 0's pointers (frame i: + i * frame_bytes)
    char* zero_block = nullptr;
    size_t zero_bytes = 0;
    uint32_t* d_misc = nullptr;      // 64 B: ovf_cnt[2], err flag, pad, cca counters[4]
    uint32_t* d_yx_alt[2] = {nullptr, nullptr};   // the cluster pass's alternating position buffers (frame 0)
    uint8_t* d_rgb_stage = nullptr;
    uint16_t* d_out_stage = nullptr;
    int32_t* d_keep_leader = nullptr;
    uint16_t* d_keep_label = nullptr;
    // per-frame caller pointers of the current group: [0, kMaxGroup) inputs, [kMaxGroup, 2*kMaxGroup) outputs
    void** d_ptrs = nullptr;         // device copy (allocated with the slot)
    void** h_ptrs = nullptr;         // pinned staging
    // shared tables
    uint16_t* d_patch = nullptr;
    uint32_t* d_lut = nullptr;
    uint32_t* h_lut = nullptr;
    uint16_t* d_tab = nullptr;
    uint16_t* h_tab = nullptr;
    // pinned host
    uint32_t* h_cl = nullptr;        // per frame 4K words: in [0,K) yx; out (yx, lab, n, moved)
    size_t h_cl_words = 0;
    uint32_t* h_misc = nullptr;      // per frame 16 words (copy of d_misc)
    int32_t* h_keep_leader = nullptr;   // per frame hKeep entries
    uint16_t* h_keep_label = nullptr;
    int hKeepFrames = 0;
    int32_t* h_dense_leader = nullptr;  // per frame kDenseCap entries: leader-sorted candidates of a tie frame
    uint32_t* h_dense_area = nullptr;
    int hDenseFrames = 0;
    uint16_t* h_patch = nullptr;
    size_t h_patch_cap = 0;
    int32_t* h_cand_leader = nullptr;
    uint32_t* h_cand_area = nullptr;
    size_t h_cand_cap = 0;
    int hKeep = 0;
    uint32_t* d_gen = nullptr;       // device word: base of the bin generation stamps (FrameDev::gen_base)
    unsigned long long* d_dbg = nullptr;
    int keyH = 0, keyW = 0, keyK = 0;
    size_t cand_capacity = 0;        // entries of the per-frame candidate arrays (>= N)
    // Captured launch sequences (hipGraph), one per distinct (geometry, options, group size, arena carving): a group
    // start is ~45 stream operations at ~3 us of host time each when enqueued one by one, ~8 us as one graph launch
    // (scripts/microbench/graph_launch.hip).  A key is captured the second time it is seen.
    struct GraphEntry {
        std::vector<unsigned char> key;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int seen = 0;
        bool failed = false;
    };
    std::vector<GraphEntry> graphs;
    int last_launch_mode = 0;        // 0 direct, 1 captured this call, 2 replayed
    bool launch_timing = false;      // the engine's flag as it stood when this slot's current group was submitted
    // per-call state
    int H = 0, W = 0, K = 0, S = 0;
    int nframes = 0;
    bool generic = false;
    fslic_params p{};
    fslic_cluster* clusters[kMaxGroup] = {};
    const uint8_t* d_rgb[kMaxGroup] = {};
    uint16_t* d_out[kMaxGroup] = {};
    float total_ms = 0, fa_ms = 0, lab_ms = 0, loop_ms = 0, cca_ms = 0;
    int last_path = 0;
    bool have_pre = false;
    bool pending = false;            // a group submitted with fslic_hip_submit_group has not been waited for yet
    // asynchronous groups run on the slot's own host thread (launches, stream sync, std::partial_sort on ties,
    // cluster write-back), so the host work of one slot overlaps that of the others and the caller's
    struct Async {
        std::thread worker;
        std::mutex m;
        std::condition_variable cv;
        bool has_job = false, done = false, quit = false;
        fslic_params p{};
        int H = 0, W = 0, K = 0, n = 0;
        const uint8_t* d_rgb[kMaxGroup] = {};
        fslic_cluster* clusters[kMaxGroup] = {};
        uint16_t* d_out[kMaxGroup] = {};
        int rc = 0;
        std::string err;
    };
    std::unique_ptr<Async> async;
    uint32_t* d_pre = nullptr;       // frame 0's preemptive state: is_updatable[K], is_active[K], cells, flags
    uint32_t* h_upd = nullptr;       // pinned: is_updatable counters back from the device, per frame K words
    size_t h_upd_words = 0;
    float* d_clf = nullptr;          // frame 0's float centroids ('noq')
    float* h_clf = nullptr;          // pinned, per frame K * 8 floats: upload (y, x) / download (y, x, r, g, b)
    size_t h_clf_floats = 0;
    // float-distance variants: f32 spatial patch (shared region of the arena) and its pinned staging
    float* d_patchf = nullptr;
    float* h_patchf = nullptr;
    size_t h_patchf_cap = 0;
    int pf_S = 0, pf_shift = -1, pf_variant = -1;
    float pf_compactness = -1.0f;
    bool pf_manhattan = true;
    // LSC variant: own arena = [shared tables][frame 0][frame 1]...; l holds frame 0's pointers
    char* lsc_arena = nullptr;
    size_t lsc_cap = 0, lsc_frame_bytes = 0, lsc_zero_bytes = 0;
    char* lsc_zero = nullptr;
    LscDev l{};
    float* h_lsc_lut = nullptr;      // pinned staging of the tables: [4][256] colour, [2][W], [2][H]
    size_t h_lsc_lut_cap = 0;
    int lsc_H = 0, lsc_W = 0, lsc_K = 0, lsc_G = 0, lsc_S = 0;
    float lsc_compactness = -1.0f;
    // cached spatial configuration (configure_spatial)
    bool sp_valid = false, sp_tiled_ok = false, sp_manhattan = true, sp_patch_uploaded = false;
    int sp_S = 0, sp_shift = 0, sp_stride = 0;
    float sp_compactness = 0.0f;

    template <class T> T* at(T* p, int frame) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(p) + (size_t)frame * frame_bytes); }
};

}  // namespace

struct fslic_engine {
    int device = 0;
    int group_size = 8;              // frames per launch group of iterate_batch (FSLIC_GROUP)
    bool launch_timing = false;      // bracket every subsampled assign launch with HIP events (fslic_hip_set_launch_timing)
    std::vector<Slot> slots;
    uint16_t* d_gamma = nullptr;
    uint16_t* d_labtbl = nullptr;
    LabTables tables{};
    std::string timing;
};

namespace {

void free_slot(Slot& s) {
    if (s.async) {
        {
            std::lock_guard<std::mutex> lk(s.async->m);
            s.async->quit = true;
        }
        s.async->cv.notify_all();
        if (s.async->worker.joinable()) s.async->worker.join();
        s.async.reset();
    }
    for (auto& g : s.graphs) { if (g.exec) hipGraphExecDestroy(g.exec); if (g.graph) hipGraphDestroy(g.graph); }
    s.graphs.clear();
    if (s.arena) hipFree(s.arena);
    if (s.lsc_arena) hipFree(s.lsc_arena);
    if (s.h_lsc_lut) hipHostFree(s.h_lsc_lut);
    if (s.h_patchf) hipHostFree(s.h_patchf);
    if (s.h_clf) hipHostFree(s.h_clf);
    if (s.h_upd) hipHostFree(s.h_upd);
    if (s.d_dbg) hipFree(s.d_dbg);
    if (s.d_ptrs) hipFree(s.d_ptrs);
    if (s.d_gen) hipFree(s.d_gen);
    if (s.h_ptrs) hipHostFree(s.h_ptrs);
    if (s.h_cl) hipHostFree(s.h_cl);
    if (s.h_misc) hipHostFree(s.h_misc);
    if (s.h_lut) hipHostFree(s.h_lut);
    if (s.h_tab) hipHostFree(s.h_tab);
    if (s.h_keep_leader) hipHostFree(s.h_keep_leader);
    if (s.h_keep_label) hipHostFree(s.h_keep_label);
    if (s.h_dense_leader) hipHostFree(s.h_dense_leader);
    if (s.h_dense_area) hipHostFree(s.h_dense_area);
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
    if (s.hKeep < K || s.hKeepFrames < G) {
        if (s.h_keep_leader) { hipHostFree(s.h_keep_leader); hipHostFree(s.h_keep_label); }
        s.h_keep_leader = nullptr; s.h_keep_label = nullptr; s.hKeep = 0; s.hKeepFrames = 0;
        const int kk = std::max(K, s.hKeep), gg = std::max(G, s.hKeepFrames);
        HIPCHK(hipHostMalloc((void**)&s.h_keep_leader, sizeof(int32_t) * (size_t)kk * gg));
        HIPCHK(hipHostMalloc((void**)&s.h_keep_label, sizeof(uint16_t) * (size_t)kk * gg));
        s.hKeep = kk; s.hKeepFrames = gg;
    }
    if (s.hDenseFrames < G) {
        if (s.h_dense_leader) { hipHostFree(s.h_dense_leader); hipHostFree(s.h_dense_area); }
        s.h_dense_leader = nullptr; s.h_dense_area = nullptr; s.hDenseFrames = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_dense_leader, sizeof(int32_t) * (size_t)kDenseCap * G));
        HIPCHK(hipHostMalloc((void**)&s.h_dense_area, sizeof(uint32_t) * (size_t)kDenseCap * G));
        s.hDenseFrames = G;
    }
    if (!s.h_misc) HIPCHK(hipHostMalloc((void**)&s.h_misc, 64 * kMaxGroup));
    if (!s.h_lut) HIPCHK(hipHostMalloc((void**)&s.h_lut, kLutMaxWords * 4));
    if (!s.h_tab) HIPCHK(hipHostMalloc((void**)&s.h_tab, 2 * kTabMaxBytes));
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
    const size_t o_tab = take(2 * kTabMaxBytes);
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
    const int nbx = 2 * S / 64 + 2, nby = 2 * S / 32 + 2;
    const size_t o_cpart = take((size_t)K * nbx * nby * 6 * 4);
    const size_t o_zero = off;
    const size_t o_sums = take((size_t)K * 6 * 4), o_sums1 = take((size_t)K * 6 * 4);
    const size_t o_cc0 = take(ncell * 4), o_cc1 = take(ncell * 4);
    const size_t o_misc = take(256);                                 // ovf_cnt[2], err, cca counters
    const size_t zero_bytes = off - o_zero;
    const size_t o_ci0 = take(ncell * kCellCap * sizeof(ClusterRec)), o_ci1 = take(ncell * kCellCap * sizeof(ClusterRec));
    const size_t o_ov0 = take((size_t)K * sizeof(ClusterRec)), o_ov1 = take((size_t)K * sizeof(ClusterRec));
    const size_t ntiles = (size_t)((W + kCcaTile - 1) / kCcaTile) * ((H + kCcaTileH - 1) / kCcaTileH);
    const size_t TT = (size_t)kCcaTilePx;
    const size_t o_parent = take(N * 4), o_area = take(N * 4), o_final = take(N * 2), o_roots = take(ntiles * TT * 4);
    const size_t o_candl = take(ntiles * TT * 4), o_canda = take(ntiles * TT * 4);
    const size_t o_tcnt = take(ntiles * 4);
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
    f.H = H; f.W = W; f.N = (int)N; f.K = K; f.S = S; f.T = 2 * S + 32;
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
    f.ovf_cnt[0] = misc + 0; f.ovf_cnt[1] = misc + 1; f.err_flag = misc + 2;
    f.cell_items[0] = (ClusterRec*)(b + o_ci0); f.cell_items[1] = (ClusterRec*)(b + o_ci1);
    f.ovf_items[0] = (ClusterRec*)(b + o_ov0); f.ovf_items[1] = (ClusterRec*)(b + o_ov1);
    f.dbg = nullptr;
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
    c.parent = (int32_t*)(b + o_parent); c.area = (uint32_t*)(b + o_area);
    c.final_lab = (uint16_t*)(b + o_final); c.roots = (int32_t*)(b + o_roots);
    c.counters = misc + 4;
    c.cand_leader = (int32_t*)(b + o_candl); c.cand_area = (uint32_t*)(b + o_canda);
    s.cand_capacity = ntiles * TT;
    c.tile_cnt = (uint32_t*)(b + o_tcnt);
    s.d_keep_leader = (int32_t*)(b + o_keepl); s.d_keep_label = (uint16_t*)(b + o_keepb);
    s.d_rgb_stage = (uint8_t*)(b + o_rgb); s.d_out_stage = (uint16_t*)(b + o_out);
    s.keyH = H; s.keyW = W; s.keyK = K;
    s.sp_valid = false;    // device copies of the spatial tables live in the arena that was just re-carved
    (void)e;
    // testing aid: a freshly carved arena normally holds whatever the previous geometry left behind; FSLIC_POISON
    // makes that worst case deterministic (every byte 0xA5) so that reads of never-written memory show up in tests
    if (getenv("FSLIC_POISON")) HIPCHK(hipMemsetAsync(s.arena, 0xA5, total, s.st));
    // Bin slots are validated by a generation stamp (ClusterRec::pad == FrameDev::gen, gen >= 1 and strictly
    // increasing per slot), so a freshly carved region must not contain bytes that could pass for a stamp.
    HIPCHK(hipMemset2DAsync(b + o_ci0, frame_bytes, 0, (o_ov0 - o_ci0), (size_t)G, s.st));
    return FSLIC_OK;
}

int ensure_prepared(fslic_engine* e, Slot& s, int H, int W, int K, int S, int G) {
    if (s.keyH == H && s.keyW == W && s.keyK == K && s.cap_frames >= G) return FSLIC_OK;
    HIPCHK(hipStreamSynchronize(s.st));
    return prepare(e, s, H, W, K, S, G);
}

// LSC variant: carve the feature arena for groups of up to G frames and (re)build the tables.  The tables follow
// src/lsc.cpp:24-27,76-101 expression by expression (float vs double evaluation included) and use the host libm,
// like the reference.
int prepare_lsc(Slot& s, int H, int W, int K, int S, int G, float compactness) {
    const size_t N = (size_t)H * W;
    const size_t lut_floats = 1024 + 2 * (size_t)W + 2 * (size_t)H;
    if (s.lsc_H != H || s.lsc_W != W || s.lsc_K != K || s.lsc_G < G) {
        G = std::max(G, (s.lsc_H == H && s.lsc_W == W && s.lsc_K == K) ? s.lsc_G : 0);
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
        const size_t o_lut = take(lut_floats * 4);
        const size_t shared_bytes = off;
        off = 0;
        const size_t o_feat = take(N * 4 * kLscFeat), o_w = take(N * 4);
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
        l.feat = (float*)(b + o_feat); l.weight = (float*)(b + o_w);
        l.hist = (uint32_t*)(b + o_hist); l.extra = (long long*)(b + o_extra);
        l.means = (float*)(b + o_means); l.cfeat = (float*)(b + o_cfeat);
        const float* lut = (const float*)(s.lsc_arena + o_lut);
        l.lut_color = lut; l.lut_x = lut + 1024; l.lut_y = lut + 1024 + 2 * (size_t)W;
        s.lsc_zero = b + o_hist; s.lsc_zero_bytes = zero_bytes; s.lsc_frame_bytes = frame_bytes;
        s.lsc_H = H; s.lsc_W = W; s.lsc_K = K; s.lsc_G = G;
        s.lsc_S = 0;                                   // tables live in the arena that was just carved
        if (getenv("FSLIC_POISON")) HIPCHK(hipMemsetAsync(s.lsc_arena, 0xA5, total, s.st));
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
        s.lsc_S = S; s.lsc_compactness = compactness;
    }
    return FSLIC_OK;
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

int validate(const fslic_params* p, int H, int W, int K, int& S) {
    if (!p) return fail(FSLIC_E_INVALID, "params is NULL");
    if (H <= 0 || W <= 0) return fail(FSLIC_E_INVALID, "H and W must be positive");
    if (H > 32767 || W > 32767) return fail(FSLIC_E_INVALID, "H and W must fit int16 (src/context.cpp:262)");
    if ((long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "H*W must fit int32");
    if (K <= 0 || K >= 65534) return fail(FSLIC_E_INVALID, "num_components must be in [1, 65533] (cfast_slic.pyx:24-27)");
    if (p->subsample_stride < 1 || p->subsample_stride > 255) return fail(FSLIC_E_INVALID, "subsample_stride must be in [1, 255]");
    if (p->max_iter < 0) return fail(FSLIC_E_INVALID, "max_iter must be >= 0");
    if (p->preemptive && p->variant != FSLIC_VARIANT_SLIC) return fail(FSLIC_E_UNSUPPORTED, "preemptive=True is implemented for the Slic variant only on hip/gfx950");
    if (p->preemptive && !std::isfinite(p->preemptive_thres)) return fail(FSLIC_E_INVALID, "preemptive_thres must be finite");
    if (p->variant < FSLIC_VARIANT_SLIC || p->variant > FSLIC_VARIANT_REALDIST_NOQ) return fail(FSLIC_E_UNSUPPORTED, "unknown variant");
    if (!(p->compactness > 0.0f) || !std::isfinite(p->compactness)) return fail(FSLIC_E_INVALID, "compactness must be a positive finite number");
    S = (int)(int16_t)sqrt(H * W / K);   // src/context.h:60
    if (S < 1) return fail(FSLIC_E_INVALID, "num_components exceeds the number of pixels (S == 0)");
    return FSLIC_OK;
}

// ---- the top-K / ranking step of ConnectivityEnforcer::execute (src/cca.cpp:205-237) on the host ----
// Input: candidates (components with area >= threshold) in ASCENDING LEADER order, which is ascending component
// number, the order the reference hands to std::partial_sort.  libstdc++'s std::partial_sort decides ties at
// the cut exactly as it does for the reference (same library, same input order).  Output: kept leaders in
// ascending order; label = position (src/cca.cpp:229-237).
void select_kept_sorted(const int32_t* leader, const uint32_t* area, uint32_t M, int K,
                        int32_t* keep_leader, uint16_t* keep_label, uint32_t& n_keep) {
    if ((size_t)K >= (size_t)M) {
        for (uint32_t i = 0; i < M; i++) { keep_leader[i] = leader[i]; keep_label[i] = (uint16_t)i; }
        n_keep = M;
        return;
    }
    std::vector<uint32_t> order(M);
    std::iota(order.begin(), order.end(), 0u);
    // Which K components survive is decided by the heap-select half of std::partial_sort (src/cca.cpp:225-228 calls
    // std::partial_sort; on an area tie at the cut the survivors depend on libstdc++'s heap mechanics).  The second half,
    // sort_heap, only orders the survivors, and they are re-sorted by leader anyway: with libstdc++ the first half is
    // called directly (one third less host time per tie frame), elsewhere the whole std::partial_sort.
    auto by_area = [&](uint32_t a, uint32_t b) { return area[a] > area[b]; };
#if defined(__GLIBCXX__)
    std::__heap_select(order.begin(), order.begin() + K, order.end(), __gnu_cxx::__ops::__iter_comp_iter(by_area));
#else
    std::partial_sort(order.begin(), order.begin() + K, order.end(), by_area);
#endif
    std::vector<uint8_t> kept(M, 0);
    for (int i = 0; i < K; i++) kept[order[i]] = 1;
    uint32_t n = 0;
    for (uint32_t i = 0; i < M; i++)
        if (kept[i]) { keep_leader[n] = leader[i]; keep_label[n] = (uint16_t)n; n++; }
    n_keep = n;
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

// The per-frame output pointers must already be in s.d_ptrs[kMaxGroup + i] (upload_ptrs).
CcaDev cca_view(const Slot& s, int i0, const uint16_t* d_in0, size_t in_stride, int K, int min_threshold) {
    CcaDev c = s.c;
    c.in = d_in0; c.in_stride = in_stride; c.K = K; c.min_threshold = min_threshold;
    c.outs = nullptr;
    c.out = nullptr;
    // host-side advance to frame i0 (CcaDev::select dereferences the device array, so spell it out here)
    FrameDev::adv(c.in, (size_t)i0 * in_stride);
    const size_t d = (size_t)i0 * c.frame_bytes;
    FrameDev::adv(c.parent, d); FrameDev::adv(c.area, d); FrameDev::adv(c.final_lab, d); FrameDev::adv(c.roots, d);
    FrameDev::adv(c.tile_cnt, d); FrameDev::adv(c.counters, d);
    FrameDev::adv(c.cand_leader, d); FrameDev::adv(c.cand_area, d);
    c.outs = reinterpret_cast<uint16_t* const*>(s.d_ptrs + kMaxGroup + i0);
    static const bool tie_copy = getenv("FSLIC_TIE_COPY") && atoi(getenv("FSLIC_TIE_COPY"));
    c.host_leader = tie_copy ? nullptr : s.h_dense_leader + (size_t)i0 * kDenseCap;
    c.host_area = tie_copy ? nullptr : s.h_dense_area + (size_t)i0 * kDenseCap;
    c.host_stride = kDenseCap;
    return c;
}

// Caller's per-frame buffers of the group -> device pointer arrays.
int upload_ptrs(Slot& s, int n, const uint8_t* const* d_rgb, uint16_t* const* d_out) {
    for (int i = 0; i < kMaxGroup; i++) {
        s.h_ptrs[i] = (i < n && d_rgb) ? const_cast<uint8_t*>(d_rgb[i]) : nullptr;
        s.h_ptrs[kMaxGroup + i] = (i < n && d_out) ? d_out[i] : nullptr;
    }
    HIPCHK(hipMemcpyAsync(s.d_ptrs, s.h_ptrs, sizeof(void*) * 2 * kMaxGroup, hipMemcpyHostToDevice, s.st));
    return FSLIC_OK;
}

// Connectivity pass, fully asynchronous on the slot's stream: tile CCL, seams, compress, collect, device
// top-K + ranking, chain, relabel, for frames [i0, i0+n) (c = cca_view(s, i0, ...)).  The select status of every
// frame ends up in its misc words; cca_finish_group() serves the frames whose status is not 0.
void cca_enqueue(Slot& s, const CcaDev& c, int i0, int n) {
    launch_cca_phase1(c, n, s.st);
    const bool dbg3 = getenv("FSLIC_DEBUG_TIMING") && atoi(getenv("FSLIC_DEBUG_TIMING")) == 3;
    if (dbg3 && !s.d_dbg) hipMalloc((void**)&s.d_dbg, sizeof(unsigned long long) * 8 * 65536);
    launch_cca_select(c, n, s.st, dbg3 ? s.d_dbg : nullptr);
    if (dbg3) {
        unsigned long long h[6];
        hipStreamSynchronize(s.st);
        hipMemcpy(h, s.d_dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[fslic dbg] select ticks: load %llu  select+ties %llu  buckets+scan %llu  scatter %llu  rank+store %llu\n",
                h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4]);
    }
    launch_cca_phase2(c, n, s.st);
}

// After the stream has been synchronised and h_misc refreshed.  Status 1 (area tie at the top-K cut): the device
// left the candidates sorted by leader; fetch them, run std::partial_sort, upload the kept set, redo chain +
// relabel.  All such frames of the group share two stream synchronisations.  Status 2 (more candidates than
// the device sorts): gather + sort on the host, frame by frame.
int cca_finish_group(Slot& s, int first, int n, const uint16_t* d_in0, size_t in_stride, int K, int thres) {
    int n_tie = 0, n_slow = 0;
    for (int i = first; i < first + n; i++) {
        const uint32_t st = s.h_misc[16 * (size_t)i + 4 + 3];
        n_tie += st == 1; n_slow += st == 2;
    }
    if (n_tie == 0 && n_slow == 0) return FSLIC_OK;
    if (n_tie) {
        static const bool tie_copy = getenv("FSLIC_TIE_COPY") && atoi(getenv("FSLIC_TIE_COPY"));      // A/B: fetch the lists with copy commands instead
        if (tie_copy) {
            for (int i = first; i < first + n; i++) {
                const uint32_t* hm = s.h_misc + 16 * (size_t)i;
                if (hm[4 + 3] != 1) continue;
                const uint32_t M = hm[4 + 1];
                HIPCHK(hipMemcpyAsync(s.h_dense_leader + (size_t)i * kDenseCap, s.at(s.c.cand_leader, i), sizeof(int32_t) * M, hipMemcpyDeviceToHost, s.st));
                HIPCHK(hipMemcpyAsync(s.h_dense_area + (size_t)i * kDenseCap, s.at(s.c.cand_area, i), sizeof(uint32_t) * M, hipMemcpyDeviceToHost, s.st));
            }
            HIPCHK(hipStreamSynchronize(s.st));
        }
        for (int i = first; i < first + n; i++) {      // the select kernel left the candidate lists in pinned host memory
            const uint32_t* hm = s.h_misc + 16 * (size_t)i;
            if (hm[4 + 3] != 1) continue;
            const uint32_t M = hm[4 + 1];
            int32_t* kl = s.h_keep_leader + (size_t)i * s.hKeep;
            uint16_t* kb = s.h_keep_label + (size_t)i * s.hKeep;
            uint32_t n_keep = 0;
            select_kept_sorted(s.h_dense_leader + (size_t)i * kDenseCap, s.h_dense_area + (size_t)i * kDenseCap, M, K, kl, kb, n_keep);
            int32_t* d_kl = s.at(s.d_keep_leader, i);
            uint16_t* d_kb = s.at(s.d_keep_label, i);
            if (n_keep) {
                HIPCHK(hipMemcpyAsync(d_kl, kl, sizeof(int32_t) * n_keep, hipMemcpyHostToDevice, s.st));
                HIPCHK(hipMemcpyAsync(d_kb, kb, sizeof(uint16_t) * n_keep, hipMemcpyHostToDevice, s.st));
            }
            const CcaDev c1 = cca_view(s, i, d_in0, in_stride, K, thres);
            launch_cca_set_kept(c1, d_kl, d_kb, (int)n_keep, s.st);
            launch_cca_phase2(c1, 1, s.st);
        }
    }
    for (int i = first; i < first + n && n_slow; i++) {
        uint32_t* hm = s.h_misc + 16 * (size_t)i;
        if (hm[4 + 3] != 2) continue;
        const CcaDev c1 = cca_view(s, i, d_in0, in_stride, K, thres);
        int32_t* d_dl = s.at(s.c.cand_leader, i);      // dense already (k_cca_collect), unordered
        uint32_t* d_da = s.at(s.c.cand_area, i);
        const uint32_t M = hm[4 + 1];
        int rc = ensure_cand_capacity(s, std::max<size_t>(M, 1) * 2);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(s.h_cand_leader, d_dl, sizeof(int32_t) * M, hipMemcpyDeviceToHost, s.st));
        HIPCHK(hipMemcpyAsync(s.h_cand_area, d_da, sizeof(uint32_t) * M, hipMemcpyDeviceToHost, s.st));
        HIPCHK(hipStreamSynchronize(s.st));
        // ascending-leader order first (the device gathered tile by tile)
        std::vector<uint32_t> order(M);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return s.h_cand_leader[a] < s.h_cand_leader[b]; });
        int32_t* sl = s.h_cand_leader + M;       // second half of the (2M-entry) host buffers
        uint32_t* sa = s.h_cand_area + M;
        for (uint32_t q = 0; q < M; q++) { sl[q] = s.h_cand_leader[order[q]]; sa[q] = s.h_cand_area[order[q]]; }
        std::vector<int32_t> kl(std::min<size_t>(M, (size_t)K) + 1);
        std::vector<uint16_t> kb(kl.size());
        uint32_t n_keep = 0;
        select_kept_sorted(sl, sa, M, K, kl.data(), kb.data(), n_keep);
        int32_t* d_kl = s.at(s.d_keep_leader, i);
        uint16_t* d_kb = s.at(s.d_keep_label, i);
        if (n_keep) {
            HIPCHK(hipMemcpy(d_kl, kl.data(), sizeof(int32_t) * n_keep, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d_kb, kb.data(), sizeof(uint16_t) * n_keep, hipMemcpyHostToDevice));
        }
        launch_cca_set_kept(c1, d_kl, d_kb, (int)n_keep, s.st);
        launch_cca_phase2(c1, 1, s.st);
    }
    HIPCHK(hipStreamSynchronize(s.st));
    return FSLIC_OK;
}

// bring-up aid: average per-phase cycle stamps of one assign launch (FSLIC_DEBUG_TIMING=1|2), to stderr
int print_phase_stamps(Slot& s, const char* what) {
    HIPCHK(hipStreamSynchronize(s.st));
    const int nw = 65536;
    std::vector<unsigned long long> h((size_t)nw * 8);
    HIPCHK(hipMemcpy(h.data(), s.d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double ph[5] = {0, 0, 0, 0, 0}, nr = 0, nn = 0;
    int cnt = 0;
    for (int b = 0; b < 32768; b++) {
        if (h[(size_t)b * 8] == 0 || h[(size_t)b * 8 + 5] == 0) continue;
        cnt++;
        for (int q = 0; q < 5; q++) ph[q] += (double)(h[(size_t)b * 8 + q + 1] - h[(size_t)b * 8 + q]);
        nr += (double)h[(size_t)b * 8 + 6]; nn += (double)h[(size_t)b * 8 + 7];
    }
    {
        std::vector<double> life, pro;
        for (int b = 0; b < 32768; b++) {
            if (h[(size_t)b * 8] == 0 || h[(size_t)b * 8 + 5] == 0) continue;
            life.push_back((double)(h[(size_t)b * 8 + 5] - h[(size_t)b * 8]));
            pro.push_back((double)(h[(size_t)b * 8 + 1] - h[(size_t)b * 8]));
        }
        if (!life.empty()) {
            std::sort(life.begin(), life.end()); std::sort(pro.begin(), pro.end());
            auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
            fprintf(stderr, "[fslic dbg]   wavefront life (cycles): min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f | prologue: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f\n",
                    q(life, 0), q(life, 0.1), q(life, 0.5), q(life, 0.9), q(life, 1), q(pro, 0), q(pro, 0.1), q(pro, 0.5), q(pro, 0.9), q(pro, 1));
        }
    }
    {   // sub-phases of the list-building wavefronts (block kernel): records 32768.. of the buffer
        double g[5] = {0, 0, 0, 0, 0};
        int gc = 0;
        for (int b = 32768; b < nw; b++) {
            if (h[(size_t)b * 8] == 0 || h[(size_t)b * 8 + 5] == 0) continue;
            gc++;
            for (int q = 0; q < 5; q++) g[q] += (double)(h[(size_t)b * 8 + q + 1] - h[(size_t)b * 8 + q]);
        }
        if (gc > 0)
            fprintf(stderr, "[fslic dbg]   list builder (%d blocks) avg ticks: address+issue %.0f  bins arrive %.0f  compact %.0f  overflow list %.0f  rank+entries %.0f\n",
                    gc, g[0] / gc, g[1] / gc, g[2] / gc, g[3] / gc, g[4] / gc);
    }
    if (cnt == 0) cnt = 1;
    const double tot = ph[0] + ph[1] + ph[2] + ph[3] + ph[4];
    fprintf(stderr, "[fslic dbg] %s %dx%d waves=%d avg ticks: loads+gather+barrier %.0f  filter+rank %.0f  argmin %.0f  store %.0f  update %.0f | candidates %.1f | wavefront life %.2f us, shader clock %.2f GHz\n",
            what, s.W, s.H, cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, nr / cnt, nn / cnt * 0.01, nn > 0 ? tot / (nn * 10.0) : 0.0);
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
    static const bool no_vtab = getenv("FSLIC_NO_VTAB") != nullptr;     // A/B aid: force the 2-D table
    // The 2-D table costs (S+2) * (2(S+pad)+1) * 2 bytes of L2 -> LDS traffic per block (26 KB at S = 72); the V table
    // a few hundred bytes, at the price of walking a candidate twice when its centre row lies inside the
    // wavefront's rows (frequent when S is small).  Measured: V wins at S = 72 (4K, K = 1600), loses at S = 24.
    const size_t tab2d_bytes = (size_t)(S + 2) * (2 * (S + std::max(7 * std::min(stride_it, 3), 15)) + 1) * 2;
    static const int vtab_min = getenv("FSLIC_VTAB_MIN_BYTES") ? atoi(getenv("FSLIC_VTAB_MIN_BYTES")) : 12288;
    f.tabs = s.d_tab + kTabMaxBytes / 2; f.tabs_words = 0; f.tabs_dyoff = 0;
    if (manhattan && !no_vtab && tab2d_bytes > (size_t)vtab_min && S >= 8 && max_sp + 765u < 1024u) {
        // Row-vector V tables (the Manhattan patch depends on |di| + |dj| only): one for the full pass (stride 1, up to
        // 16 rows per wavefront) and one for the subsampled passes (stride_it, 8 rows per wavefront).
        const int xout = 2 * S + 1;
        auto build = [&](int stride, int rows, uint16_t* dst, int& words, int& cc) -> bool {
            const int pad = (rows - 1) * stride;
            const int Cc = xout + S + pad;
            const int tlen = 2 * Cc + 1 + 8 * stride;            // entries that can be addressed: t0 + 8*stride for the upper half
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
        int w1 = 0, c1 = 0, ws = 0, cs = 0;
        const int st3 = std::min(stride_it, 3);
        if (build(1, 16, s.h_tab, w1, c1) && build(st3, 8, s.h_tab + kTabMaxBytes / 2, ws, cs)) {
            f.tab_words = w1; f.tab_dyoff = c1; f.tab_vmode = 1; f.tab_xout = xout;
            f.tabs_words = ws; f.tabs_dyoff = cs;
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
        }
    }
    if (stride_it > 3 && !have_mc && f.lut_words == 0) tiled_ok = false;   // strides the packed kernel does not take need the 32-bit kernel
    f.M = M;
    f.C = Coff;
    if (f.tab_words > 0)
        HIPCHK(hipMemcpyAsync(s.d_tab, s.h_tab, (size_t)f.tab_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.tabs_words > 0)
        HIPCHK(hipMemcpyAsync(const_cast<uint16_t*>(f.tabs), s.h_tab + kTabMaxBytes / 2, (size_t)f.tabs_words * 4, hipMemcpyHostToDevice, s.st));
    if (f.lut_words > 0)
        HIPCHK(hipMemcpyAsync(s.d_lut, s.h_lut, (size_t)f.lut_words * 4, hipMemcpyHostToDevice, s.st));
    s.sp_tiled_ok = tiled_ok;
    s.sp_valid = true; s.sp_S = S; s.sp_compactness = p->compactness; s.sp_shift = color_shift;
    s.sp_manhattan = manhattan; s.sp_stride = stride_it;
    return FSLIC_OK;
}


// Host half of a group start: the integer part of the clamped centres (src/context.cpp:129-131,208-211,262) into the
// pinned staging block, and what the launch-timing events of this call will cover.  Nothing here touches the stream, so
// it runs on every call whether the device half below is enqueued directly or replayed as a graph.
void stage_group(fslic_engine* e, Slot& s, int i0, int n, bool timed) {
    const int H = s.H, W = s.W, K = s.K;
    for (int z = 0; z < n; z++) {
        const fslic_cluster* cl = s.clusters[i0 + z];
        uint32_t* h = s.h_cl + (size_t)(i0 + z) * 4 * K;
        for (int k = 0; k < K; k++) {
            const int y = clampi((int)cl[k].y, 0, H - 1), x = clampi((int)cl[k].x, 0, W - 1);
            h[k] = ((uint32_t)y << 16) | (uint32_t)x;
        }
    }
    if (s.p.variant == FSLIC_VARIANT_REALDIST_NOQ) {      // assign()'s safeguard on the float position, src/context.cpp:208-211
        for (int z = 0; z < n; z++) {
            const fslic_cluster* cl = s.clusters[i0 + z];
            float* h = s.h_clf + (size_t)(i0 + z) * 8 * K;
            for (int k = 0; k < K; k++) {
                h[8 * k + 0] = clampf(cl[k].y, 0.0f, (float)(H - 1));
                h[8 * k + 1] = clampf(cl[k].x, 0.0f, (float)(W - 1));
                for (int q = 2; q < 8; q++) h[8 * k + q] = 0.0f;
            }
        }
    }
    if (timed) {
        s.n_timed_iters = 0; s.assign_loop_px = 0;
        const int stride = s.p.subsample_stride;
        int rem = 0;
        for (int it = 0; it < s.p.max_iter && s.launch_timing && it < Slot::kMaxTimedIters; it++) {
            s.n_timed_iters = it + 1;
            s.assign_loop_px += (double)((H - rem + stride - 1) / stride) * W * n;
            rem = (rem + 1) % stride;
        }
    }
}

// Device half: everything of iterate() for frames [i0, i0+n) of the slot's current group, enqueued asynchronously on
// the slot's stream: cluster upload, LAB, the assign/update loop, full assign, connectivity, result download.  Pure
// stream work with no per-call values in any launch argument (caller pointers sit in the device pointer table, bin
// generations come from device memory), so the same sequence can be captured once and replayed (group_begin).
int enqueue_frames(fslic_engine* e, Slot& s, int i0, int n, bool generic, bool timed) {
    const fslic_params* p = &s.p;
    const int K = s.K, S = s.S;
    FrameDev f = s.f;
    f.rgbs = nullptr;
    f.select(i0);                                  // host-side: pointers of frame i0
    f.rgbs = reinterpret_cast<const uint8_t* const*>(s.d_ptrs + i0);
    const size_t fb = s.frame_bytes;

    if (generic && !s.sp_patch_uploaded) {
        HIPCHK(hipMemcpyAsync(s.d_patch, s.h_patch, (size_t)(2 * S + 1) * (2 * S + 1) * sizeof(uint16_t), hipMemcpyHostToDevice, s.st));
        s.sp_patch_uploaded = true;
    }
    // cluster centres: staged by stage_group() into the pinned block this copy reads
    HIPCHK(hipMemcpy2DAsync(f.cl_yx, fb, s.h_cl + (size_t)i0 * 4 * K, sizeof(uint32_t) * 4 * (size_t)K,
                            sizeof(uint32_t) * (size_t)K, (size_t)n, hipMemcpyHostToDevice, s.st));
    const bool pre = p->preemptive != 0;
    if (pre) {
        // PreemptiveGrid::initialize, src/preemptive.h:59-67: every counter = cooldown, everything active
        uint32_t* base = s.at(s.d_pre, i0);
        f.cl_upd = base; f.cl_act = base + K; f.cell_act = base + 2 * (size_t)K; f.pre_flags = f.cell_act + (size_t)f.pre_cw * f.pre_ch;
        for (int z = 0; z < n; z++) {
            uint32_t* bz = s.at(s.d_pre, i0 + z);
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)bz, 2, (size_t)K, s.st));                                   // cooldown
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(bz + 2 * (size_t)K + (size_t)f.pre_cw * f.pre_ch), 1, 1, s.st));   // all_active
        }
    }
    const bool noq = p->variant == FSLIC_VARIANT_REALDIST_NOQ;
    if (noq) {
        f.cl_f = s.at(s.d_clf, i0);
        HIPCHK(hipMemcpy2DAsync(f.cl_f, fb, s.h_clf + (size_t)i0 * 8 * K, sizeof(float) * 8 * (size_t)K,
                                sizeof(float) * 8 * (size_t)K, (size_t)n, hipMemcpyHostToDevice, s.st));
    }
    // sums, bin counters, flags and cl_n / cl_moved start from zero: cleared by the LAB kernel below (two memset launches less)

    if (timed) HIPCHK(hipEventRecord(s.ev[0], s.st));
    // + labels <- 0xFFFF (src/context.cpp:138-145); + a fresh range of bin generation stamps for this group's passes
    f.gen_base = s.d_gen;
    launch_rgb_to_lab(f, n, p->convert_to_lab, e->tables, true, s.st, (uint32_t)p->max_iter + 3u,
                      s.at(s.zero_block, i0), s.zero_bytes, f.cl_n, 2 * (size_t)K);
    if (timed) HIPCHK(hipEventRecord(s.ev[1], s.st));
    f.gen_off = 1;
    PassGeom pg{};
    // positions: the uploaded centres are in s.f.cl_yx; every cluster pass reads one buffer and writes the other
    uint32_t* const yx_up = f.cl_yx;
    uint32_t* const yx_a = yx_up + 6 * (size_t)K;       // this frame's d_yx_alt[0]
    uint32_t* const yx_b = yx_up + 7 * (size_t)K;       // this frame's d_yx_alt[1]
    f.cl_yx = yx_up; f.cl_yx_out = yx_a;
    launch_bin_clusters(f, n, 0, 0, 0, pg, s.st);
    uint32_t* yx_cur = yx_a;
    const int stride = p->subsample_stride;
    int rem = 0;
    f.dbg = nullptr;
    const bool lsc = p->variant == FSLIC_VARIANT_LSC;
    LscDev l = s.l;
    if (lsc) {
        l.select(i0);
        HIPCHK(hipMemset2DAsync(s.lsc_zero + (size_t)i0 * s.lsc_frame_bytes, s.lsc_frame_bytes, 0, s.lsc_zero_bytes, (size_t)n, s.st));
        launch_lsc_prepare(f, l, n, s.st);                        // before_iteration, src/lsc.cpp:12-15
    }
    const bool rd = p->variant == FSLIC_VARIANT_REALDIST || p->variant == FSLIC_VARIANT_REALDIST_L2 || noq;
    const bool rd_l2 = p->variant == FSLIC_VARIANT_REALDIST_L2;
    float noq_coef = 1.0f / ((float)S / p->compactness);        // src/context.cpp:463-464
    noq_coef *= (1 << (p->convert_to_lab ? 1 : 0));
    const bool noq_manhattan = p->manhattan_spatial_dist != 0;
    for (int it = 0; rd && it < p->max_iter; it++) {            // src/context.cpp:158-175, BaseContext<float>
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        f.cl_yx = yx_cur; f.cl_yx_out = (yx_cur == yx_a) ? yx_b : yx_a;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        if (noq) launch_noq_assign(f, noq_coef, noq_manhattan, n, rem, stride, it & 1, it & 1, true, s.st);
        else launch_rd_assign(f, s.d_patchf, rd_l2, n, rem, stride, it & 1, it & 1, true, s.st);
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
        launch_member_sums(f, n, rem, stride, it & 1, noq ? 1 : 0, s.st);
        f.gen_off++;
        pg = PassGeom{};
        launch_bin_clusters(f, n, 1, (it + 1) & 1, it & 1, pg, s.st);
        yx_cur = f.cl_yx_out;
        rem = (rem + 1) % stride;
    }
    const float pre_l1_thres = std::max(roundf(2 * S * p->preemptive_thres), 1.0f);      // src/preemptive.h:129
    const int dbg_mode = getenv("FSLIC_DEBUG_TIMING") ? atoi(getenv("FSLIC_DEBUG_TIMING")) : 0;
    for (int it = 0; lsc && it < p->max_iter; it++) {           // src/context.cpp:158-175 with ContextLSC's hooks
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        f.cl_yx = yx_cur; f.cl_yx_out = (yx_cur == yx_a) ? yx_b : yx_a;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        launch_lsc_assign(f, l, n, rem, stride, it & 1, it & 1, true, s.st);
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
        launch_lsc_update(f, l, n, rem, stride, it & 1, s.st);   // update() sums + after_update()
        f.gen_off++;
        pg = PassGeom{};                                          // use_slots = 0: everything is in FrameDev::sums
        launch_bin_clusters(f, n, 1, (it + 1) & 1, it & 1, pg, s.st);
        yx_cur = f.cl_yx_out;
        rem = (rem + 1) % stride;
    }
    for (int it = 0; !lsc && !rd && it < p->max_iter; it++) {   // src/context.cpp:158-175
        if (it == p->max_iter - 1 && dbg_mode == 2) {
            if (!s.d_dbg) HIPCHK(hipMalloc((void**)&s.d_dbg, sizeof(unsigned long long) * 8 * 65536));
            HIPCHK(hipMemsetAsync(s.d_dbg, 0, sizeof(unsigned long long) * 8 * 65536, s.st));
            f.dbg = s.d_dbg;
        }
        pg = PassGeom{};
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        if (generic) launch_assign_generic(f, n, rem, stride, it & 1, it & 1, true, s.st);
        else pg = launch_assign(f, n, rem, stride, it & 1, it & 1, true, s.st);
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
        if (f.dbg && dbg_mode == 2) {
            int rc2 = print_phase_stamps(s, "last fused assign");
            if (rc2) return rc2;
            f.dbg = nullptr;
        }
        f.gen_off++;
        f.cl_yx = yx_cur; f.cl_yx_out = (yx_cur == yx_a) ? yx_b : yx_a;
        if (pre) launch_preempt_update(f, n, (it + 1) & 1, it & 1, false, pre_l1_thres, s.st);     // update() + set_new_clusters()
        else launch_bin_clusters(f, n, 1, (it + 1) & 1, it & 1, pg, s.st);
        yx_cur = f.cl_yx_out;
        rem = (rem + 1) % stride;
    }
    int full_buf = p->max_iter & 1;
    if (pre) {       // PreemptiveGrid::finalize (src/preemptive.h:69-74): every cluster takes part in the full assignment
        f.gen_off++;
        f.cl_yx = yx_cur;
        launch_preempt_update(f, n, (p->max_iter + 1) & 1, 0, true, pre_l1_thres, s.st);
        full_buf = (p->max_iter + 1) & 1;
    }
    if (dbg_mode == 1) {
        if (!s.d_dbg) HIPCHK(hipMalloc((void**)&s.d_dbg, sizeof(unsigned long long) * 8 * 65536));
        HIPCHK(hipMemsetAsync(s.d_dbg, 0, sizeof(unsigned long long) * 8 * 65536, s.st));
        f.dbg = s.d_dbg;
    }
    if (timed) HIPCHK(hipEventRecord(s.ev[2], s.st));
    // full_assign: stride 1, rem 0, no update afterwards (src/context.cpp:246-256)
    if (lsc) { f.cl_yx = yx_cur; launch_lsc_assign(f, l, n, 0, 1, p->max_iter & 1, 0, false, s.st); }
    else if (noq) { f.cl_yx = yx_cur; launch_noq_assign(f, noq_coef, noq_manhattan, n, 0, 1, p->max_iter & 1, 0, false, s.st); }
    else if (rd) { f.cl_yx = yx_cur; launch_rd_assign(f, s.d_patchf, rd_l2, n, 0, 1, p->max_iter & 1, 0, false, s.st); }
    else if (generic) launch_assign_generic(f, n, 0, 1, full_buf, 0, false, s.st);
    else (void)launch_assign(f, n, 0, 1, p->max_iter & 1, 0, false, s.st);
    if (timed) HIPCHK(hipEventRecord(s.ev[3], s.st));
    if (f.dbg) {
        int rc2 = print_phase_stamps(s, "full_assign");
        if (rc2) return rc2;
        f.dbg = nullptr;
    }
    // enforce_connectivity, src/context.cpp:14-20
    const int thres = (int)round((double)(S * S) * (double)p->min_size_factor);
    const CcaDev c = cca_view(s, i0, s.f.labels, fb, K, thres);
    cca_enqueue(s, c, i0, n);
    if (timed) HIPCHK(hipEventRecord(s.ev[4], s.st));
    // cluster state, overflow flag and connectivity status back to the host
    HIPCHK(hipMemcpy2DAsync(s.h_cl + (size_t)i0 * 4 * K, sizeof(uint32_t) * 4 * (size_t)K, yx_cur, fb,
                            sizeof(uint32_t) * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    HIPCHK(hipMemcpy2DAsync(s.h_cl + (size_t)i0 * 4 * K + K, sizeof(uint32_t) * 4 * (size_t)K, yx_up + K, fb,
                            sizeof(uint32_t) * 3 * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    HIPCHK(hipMemcpy2DAsync(s.h_misc + 16 * (size_t)i0, 64, s.at(s.d_misc, i0), fb, 64, (size_t)n, hipMemcpyDeviceToHost, s.st));
    if (pre) HIPCHK(hipMemcpy2DAsync(s.h_upd + (size_t)i0 * K, sizeof(uint32_t) * (size_t)K, f.cl_upd, fb,
                                     sizeof(uint32_t) * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    if (noq) HIPCHK(hipMemcpy2DAsync(s.h_clf + (size_t)i0 * 8 * K, sizeof(float) * 8 * (size_t)K, f.cl_f, fb,
                                     sizeof(float) * 8 * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    return FSLIC_OK;
}

// Start a group of n frames (identical geometry) on the slot.  d_rgb / d_out are device pointers.
double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
const bool g_host_timing = getenv("FSLIC_HOST_TIMING") != nullptr;

// The device half of a group start, replayed from a captured graph when this exact sequence has been seen before.
const bool g_use_graphs = !(getenv("FSLIC_GRAPH") && atoi(getenv("FSLIC_GRAPH")) == 0) && !getenv("FSLIC_DEBUG_TIMING");
int launch_group(fslic_engine* e, Slot& s, int n) {
    s.last_launch_mode = 0;
    // Per-launch timing wants real event records between the launches; events replayed inside a graph keep the time
    // stamps of an earlier execution.  Timed groups therefore take the direct path.
    if (!g_use_graphs || s.launch_timing) return enqueue_frames(e, s, 0, n, s.generic, true);
    // everything a launch argument can depend on
    std::vector<unsigned char> key;
    auto put = [&](const void* p, size_t bytes) { const unsigned char* b = (const unsigned char*)p; key.insert(key.end(), b, b + bytes); };
    const int hdr[8] = {s.H, s.W, s.K, s.S, n, s.generic ? 1 : 0, 0, s.sp_patch_uploaded ? 1 : 0};
    const void* const ptrs[11] = {s.h_cl, s.h_misc, s.h_dense_leader, s.h_dense_area, s.d_ptrs, s.h_patch, s.d_gen, s.lsc_zero, s.d_patchf, s.h_clf, s.h_upd};   // baked into copy nodes / arguments
    put(hdr, sizeof hdr); put(ptrs, sizeof ptrs); put(&s.p, sizeof s.p); put(&s.f, sizeof s.f); put(&s.c, sizeof s.c);
    if (s.p.variant == FSLIC_VARIANT_LSC) put(&s.l, sizeof s.l);
    Slot::GraphEntry* ge = nullptr;
    for (auto& g : s.graphs) if (g.key == key) { ge = &g; break; }
    if (!ge) {
        if (s.graphs.size() >= 8) {                  // keep the cache small: drop the oldest entry
            if (s.graphs.front().exec) hipGraphExecDestroy(s.graphs.front().exec);
            if (s.graphs.front().graph) hipGraphDestroy(s.graphs.front().graph);
            s.graphs.erase(s.graphs.begin());
        }
        s.graphs.emplace_back();
        ge = &s.graphs.back();
        ge->key = key;
    }
    if (!ge->exec) {
        if (ge->failed || ge->seen++ == 0) return enqueue_frames(e, s, 0, n, s.generic, true);   // first sighting: direct (also warms one-time setup)
        if (hipStreamBeginCapture(s.st, hipStreamCaptureModeThreadLocal) != hipSuccess) { ge->failed = true; (void)hipGetLastError(); return enqueue_frames(e, s, 0, n, s.generic, true); }
        const int rc = enqueue_frames(e, s, 0, n, s.generic, false);     // no event records inside the graph
        hipGraph_t g = nullptr;
        const hipError_t ec = hipStreamEndCapture(s.st, &g);
        if (rc != FSLIC_OK || ec != hipSuccess || !g || hipGraphInstantiate(&ge->exec, g, nullptr, nullptr, 0) != hipSuccess) {
            if (g) hipGraphDestroy(g);
            ge->exec = nullptr; ge->failed = true;
            (void)hipGetLastError();
            if (rc != FSLIC_OK) return rc;
            return enqueue_frames(e, s, 0, n, s.generic, true);
        }
        ge->graph = g;
        s.last_launch_mode = 1;
    } else {
        s.last_launch_mode = 2;
    }
    // the whole group between two real event records: total device time stays available, the per-phase split does not
    HIPCHK(hipEventRecord(s.ev[0], s.st));
    HIPCHK(hipGraphLaunch(ge->exec, s.st));
    HIPCHK(hipEventRecord(s.ev[4], s.st));
    return FSLIC_OK;
}

int group_begin(fslic_engine* e, Slot& s, const fslic_params* p, int H, int W, int K, int n,
                const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_out) {
    const double t_begin = now_us();
    int S = 0;
    int rc = validate(p, H, W, K, S);
    if (rc) return rc;
    if (n < 1 || n > kMaxGroup) return fail(FSLIC_E_INTERNAL, "bad group size");
    for (int i = 0; i < n; i++)
        if (!clusters[i] || !d_rgb[i] || !d_out[i]) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    rc = ensure_prepared(e, s, H, W, K, S, n);
    if (rc) return rc;
    s.H = H; s.W = W; s.K = K; s.S = S; s.p = *p; s.nframes = n;
    for (int i = 0; i < n; i++) { s.clusters[i] = clusters[i]; s.d_rgb[i] = d_rgb[i]; s.d_out[i] = d_out[i]; }
    rc = configure_spatial(s, S, p);
    if (rc) return rc;
    if (p->variant == FSLIC_VARIANT_LSC) {
        rc = prepare_lsc(s, H, W, K, S, n, p->compactness);
        if (rc) return rc;
    }
    if (p->variant == FSLIC_VARIANT_REALDIST || p->variant == FSLIC_VARIANT_REALDIST_L2) {
        rc = configure_patchf(s, S, p);
        if (rc) return rc;
    }
    if (p->variant == FSLIC_VARIANT_REALDIST_NOQ && s.h_clf_floats < (size_t)s.cap_frames * K * 8) {
        if (s.h_clf) hipHostFree(s.h_clf);
        s.h_clf = nullptr; s.h_clf_floats = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_clf, sizeof(float) * 8 * (size_t)K * s.cap_frames));
        s.h_clf_floats = (size_t)s.cap_frames * K * 8;
    }
    rc = upload_ptrs(s, n, d_rgb, d_out);
    if (rc) return rc;
    s.generic = p->variant == FSLIC_VARIANT_SLIC && (p->force_generic || !s.sp_tiled_ok || p->preemptive);
    if (p->preemptive && s.h_upd_words < (size_t)s.cap_frames * K) {
        if (s.h_upd) hipHostFree(s.h_upd);
        s.h_upd = nullptr; s.h_upd_words = 0;
        HIPCHK(hipHostMalloc((void**)&s.h_upd, sizeof(uint32_t) * (size_t)K * s.cap_frames));
        s.h_upd_words = (size_t)s.cap_frames * K;
    }
    stage_group(e, s, 0, n, true);
    rc = launch_group(e, s, n);
    if (rc) return rc;
    s.have_pre = true;
    s.last_path = s.generic ? 1 : 0;
    if (g_host_timing) fprintf(stderr, "[fslic host] group_begin n=%d: %.1f us | slot %d begin %.1f .. %.1f\n", n, now_us() - t_begin, (int)(&s - e->slots.data()), t_begin, now_us());
    return FSLIC_OK;
}

// Wait for the group, serve the rare host-side steps frame by frame (candidate-list overflow -> generic redo,
// area tie at the top-K cut -> std::partial_sort on the host), then write the cluster state back.
int group_finish(fslic_engine* e, Slot& s) {
    const double t_begin = now_us();
    HIPCHK(hipStreamSynchronize(s.st));
    const double t_sync = now_us();
    const int n = s.nframes, K = s.K, H = s.H, W = s.W;
    auto elapsed = [](float& ms, hipEvent_t a, hipEvent_t b) { if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { ms = 0; (void)hipGetLastError(); } };
    elapsed(s.total_ms, s.ev[0], s.ev[4]);
    if (s.last_launch_mode == 0) {
        elapsed(s.lab_ms, s.ev[0], s.ev[1]);
        elapsed(s.loop_ms, s.ev[1], s.ev[2]);
        elapsed(s.fa_ms, s.ev[2], s.ev[3]);
        elapsed(s.cca_ms, s.ev[3], s.ev[4]);
    } else {
        s.lab_ms = s.loop_ms = s.fa_ms = s.cca_ms = 0;       // graph replay: only the group total is event-timed
    }
    s.assign_loop_ms = 0;
    for (int it = 0; it < s.n_timed_iters; it++) {
        float ms = 0;
        elapsed(ms, s.ev_it[2 * it], s.ev_it[2 * it + 1]);
        s.assign_loop_ms += ms;
    }
    const int thres = (int)round((double)(s.S * s.S) * (double)s.p.min_size_factor);
    const double t_ev = now_us();
    int n_host_topk = 0;
    for (int i = 0; i < n; i++) {
        n_host_topk += s.h_misc[16 * (size_t)i + 4 + 3] != 0;
        if (!s.generic && s.h_misc[16 * (size_t)i + 2] != 0) {     // the tiled kernel overflowed an LDS candidate list
            if (g_host_timing) fprintf(stderr, "[fslic host] candidate-list overflow: frame %d redone with the generic kernel\n", i);
            stage_group(e, s, i, 1, false);
            int rc = enqueue_frames(e, s, i, 1, true, false);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(s.st));
            if (s.h_misc[16 * (size_t)i + 2] != 0) return fail(FSLIC_E_INTERNAL, "generic path reported a candidate overflow");
            s.last_path = 1;
        }
    }
    {
        int rc = cca_finish_group(s, 0, n, s.f.labels, s.frame_bytes, K, thres);
        if (rc) return rc;
    }
    const double t_cca = now_us();
    // write the cluster state back the way the reference leaves it
    for (int i = 0; i < n; i++) {
        const uint32_t* yx = s.h_cl + (size_t)i * 4 * K;
        const uint32_t* lab = yx + K;
        const uint32_t* nm = yx + 2 * (size_t)K;
        const uint32_t* moved = yx + 3 * (size_t)K;
        fslic_cluster* cl = s.clusters[i];
        for (int k = 0; k < K; k++) {
            cl[k].r = (float)(lab[k] & 255u);                 // src/context.cpp:132-134 / :370-372
            cl[k].g = (float)((lab[k] >> 8) & 255u);
            cl[k].b = (float)((lab[k] >> 16) & 255u);
            if (moved[k] && s.p.variant == FSLIC_VARIANT_REALDIST_NOQ) {      // src/context.cpp:374-380
                const float* c = s.h_clf + ((size_t)i * K + k) * 8;
                cl[k].y = c[0]; cl[k].x = c[1]; cl[k].r = c[2]; cl[k].g = c[3]; cl[k].b = c[4];
            } else if (moved[k]) {                             // src/context.cpp:368-369
                cl[k].y = (float)(yx[k] >> 16);
                cl[k].x = (float)(yx[k] & 0xFFFFu);
            } else {                                           // assign() safeguard, src/context.cpp:208-211
                cl[k].x = clampf(cl[k].x, 0.0f, (float)(W - 1));
                cl[k].y = clampf(cl[k].y, 0.0f, (float)(H - 1));
            }
            if (s.p.max_iter > 0) cl[k].num_members = nm[k];  // src/context.cpp:362
            cl[k].is_updatable = s.p.preemptive ? (uint8_t)s.h_upd[(size_t)i * K + k] : 2;   // src/preemptive.h:59-67, :131-139
            cl[k].is_active = 1;                               // src/preemptive.h:69-74
        }
    }
    if (g_host_timing)
        fprintf(stderr, "[fslic host] group_finish n=%d slot %d at %.1f .. %.1f: sync wait %.1f us, events %.1f us, host top-K (%d frames) %.1f us, write-back %.1f us (device %.1f us)\n",
                n, (int)(&s - e->slots.data()), t_begin, now_us(), t_sync - t_begin, t_ev - t_sync, n_host_topk, t_cca - t_ev, now_us() - t_cca, s.total_ms * 1e3);
    return FSLIC_OK;
}

void make_timing_report(fslic_engine* e, const Slot& s) {
    char buf[1024];
    auto us = [](float ms) { return (long long)(ms * 1000.0f); };
    snprintf(buf, sizeof buf,
             "{\"name\": \"iterate\", \"duration\": %lld, \"children\": ["
             "{\"name\": \"cielab_conversion\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"assign_update_loop\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"full_assign\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"enforce_connectivity\", \"duration\": %lld, \"children\": []}]}",
             us(s.total_ms), us(s.lab_ms), us(s.loop_ms), us(s.fa_ms), us(s.cca_ms));
    e->timing = buf;
}

}  // namespace

// ================================= C ABI ==========================================================
extern "C" {

const char* fslic_hip_last_error(void) { return g_err.c_str(); }
const char* fslic_hip_version(void) { return "fast_slic_amd 0.1 (gfx950)"; }

int fslic_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fslic_hip_create(int device, int n_slots, fslic_engine** out) {
    if (!out) return fail(FSLIC_E_INVALID, "out is NULL");
    *out = nullptr;
    if (n_slots < 1 || n_slots > 64) return fail(FSLIC_E_INVALID, "n_slots must be in [1, 64]");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(FSLIC_E_HIP, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    std::call_once(g_tables_once, build_host_tables);
    fslic_engine* e = new fslic_engine();
    e->device = device;
    if (const char* g = getenv("FSLIC_GROUP")) e->group_size = std::min(std::max(atoi(g), 1), (int)kMaxGroup);
    e->slots.resize(n_slots);
    for (auto& s : e->slots) {
        if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "hipStreamCreate failed"); }
        for (auto& ev : s.ev)
            if (hipEventCreate(&ev) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "hipEventCreate failed"); }
        for (auto& ev : s.ev_it)
            if (hipEventCreate(&ev) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "hipEventCreate failed"); }
        if (hipMalloc((void**)&s.d_ptrs, sizeof(void*) * 2 * kMaxGroup) != hipSuccess ||
            hipHostMalloc((void**)&s.h_ptrs, sizeof(void*) * 2 * kMaxGroup) != hipSuccess ||
            hipMalloc((void**)&s.d_gen, 256) != hipSuccess || hipMemset(s.d_gen, 0, 256) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "pointer table allocation failed"); }
    }
    if (hipMalloc((void**)&e->d_gamma, sizeof g_tables.gamma) != hipSuccess ||
        hipMalloc((void**)&e->d_labtbl, sizeof g_tables.lab) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "hipMalloc(tables) failed"); }
    if (hipMemcpy(e->d_gamma, g_tables.gamma, sizeof g_tables.gamma, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_labtbl, g_tables.lab, sizeof g_tables.lab, hipMemcpyHostToDevice) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "table upload failed"); }
    e->tables.gamma = e->d_gamma;
    e->tables.labtbl = e->d_labtbl;
    for (int i = 0; i < 9; i++) e->tables.cb[i] = g_tables.cb[i];
    *out = e;
    return FSLIC_OK;
}

void fslic_hip_destroy(fslic_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    for (auto& s : e->slots) {
        if (s.st) hipStreamSynchronize(s.st);
        free_slot(s);
    }
    if (e->d_gamma) hipFree(e->d_gamma);
    if (e->d_labtbl) hipFree(e->d_labtbl);
    delete e;
}

// BaseContext::initialize_clusters, src/context.cpp:42-97 (host, serial, O(K)).
int fslic_hip_initialize_clusters(int H, int W, int K, const uint8_t* rgb, fslic_cluster* cl) {
    if (!rgb || !cl) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (H <= 0 || W <= 0 || K <= 0) return FSLIC_OK;   // the reference returns silently (:43)
    const int n_y = (int)sqrt((double)K);
    std::vector<int> n_xs(n_y, K / n_y);
    int remainder = K % n_y, row = 0;
    while (remainder-- > 0) {
        n_xs[row]++;
        row += 2;
        if (row >= n_y) row = 1 % n_y;
    }
    const int h = (H + n_y - 1) / n_y;
    int acc = 0;
    for (int i = 0; i < H; i += h) {
        const int w = (W + n_xs[std::min(i / h, n_y - 1)] - 1) / n_xs[std::min(i / h, n_y - 1)];
        for (int j = 0; j < W; j += w) {
            if (acc >= K) break;
            cl[acc].y = (float)clampi(i + h / 2, 0, H - 1);
            cl[acc].x = (float)clampi(j + w / 2, 0, W - 1);
            cl[acc].is_active = 1;
            cl[acc].is_updatable = 1;
            acc++;
        }
    }
    for (; acc < K; acc++) {            // leftovers pile up on the image centre (:80-86)
        cl[acc].is_active = 1;
        cl[acc].is_updatable = 1;
        cl[acc].y = (float)(H / 2);
        cl[acc].x = (float)(W / 2);
    }
    for (int k = 0; k < K; k++) {
        const size_t base = (size_t)W * (int)cl[k].y + (int)cl[k].x;
        cl[k].r = rgb[3 * base];
        cl[k].g = rgb[3 * base + 1];
        cl[k].b = rgb[3 * base + 2];
        cl[k].number = (uint16_t)k;
        cl[k].num_members = 0;
    }
    return FSLIC_OK;
}

int fslic_hip_iterate_device(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K,
                             const uint8_t* d_rgb, fslic_cluster* clusters, uint16_t* d_labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    HIPCHK(hipSetDevice(e->device));
    Slot& s = e->slots[slot];
    if (s.pending) return fail(FSLIC_E_INVALID, "the slot still owns an unfinished group");
    s.launch_timing = e->launch_timing;
    int rc = group_begin(e, s, p, H, W, K, 1, &d_rgb, &clusters, &d_labels);
    if (rc) return rc;
    rc = group_finish(e, s);
    if (rc == FSLIC_OK && slot == 0) make_timing_report(e, s);
    return rc;
}

// The slot's host thread: one group at a time, begin + finish.
static void slot_worker(fslic_engine* e, Slot* s) {
    Slot::Async& a = *s->async;
    (void)hipSetDevice(e->device);
    for (;;) {
        std::unique_lock<std::mutex> lk(a.m);
        a.cv.wait(lk, [&] { return a.has_job || a.quit; });
        if (a.quit) return;
        lk.unlock();
        int rc = group_begin(e, *s, &a.p, a.H, a.W, a.K, a.n, a.d_rgb, a.clusters, a.d_out);
        if (rc == FSLIC_OK) rc = group_finish(e, *s);
        lk.lock();
        a.rc = rc;
        a.err = rc ? g_err : std::string();
        a.has_job = false;
        a.done = true;
        lk.unlock();
        a.cv.notify_all();
    }
}

int fslic_hip_submit_group(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K, int n_frames,
                           const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    if (n_frames < 1 || n_frames > (int)kMaxGroup || !d_rgb || !clusters || !d_labels) return fail(FSLIC_E_INVALID, "bad group arguments");
    Slot& s = e->slots[slot];
    if (s.pending) return fail(FSLIC_E_INVALID, "the slot still owns an unfinished group (call fslic_hip_wait_group first)");
    int S = 0;
    int rc = validate(p, H, W, K, S);          // argument errors surface here, on the caller's thread
    if (rc) return rc;
    for (int i = 0; i < n_frames; i++)
        if (!clusters[i] || !d_rgb[i] || !d_labels[i]) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (!s.async) {
        s.async.reset(new Slot::Async());
        s.async->worker = std::thread(slot_worker, e, &s);
    }
    Slot::Async& a = *s.async;
    {
        std::lock_guard<std::mutex> lk(a.m);
        a.p = *p; a.H = H; a.W = W; a.K = K; a.n = n_frames;
        s.launch_timing = e->launch_timing;      // sampled on the caller's thread: the worker may start later
        for (int i = 0; i < n_frames; i++) { a.d_rgb[i] = d_rgb[i]; a.clusters[i] = clusters[i]; a.d_out[i] = d_labels[i]; }
        a.done = false;
        a.has_job = true;
    }
    a.cv.notify_all();
    s.pending = true;
    return FSLIC_OK;
}

int fslic_hip_wait_group(fslic_engine* e, int slot) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    Slot& s = e->slots[slot];
    if (!s.pending) return FSLIC_OK;
    Slot::Async& a = *s.async;
    int rc;
    {
        std::unique_lock<std::mutex> lk(a.m);
        a.cv.wait(lk, [&] { return a.done; });
        rc = a.rc;
        if (rc) g_err = a.err;
    }
    s.pending = false;
    if (rc == FSLIC_OK && slot == 0) make_timing_report(e, s);
    return rc;
}

int fslic_hip_group_done(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    Slot& s = e->slots[slot];
    if (!s.pending) return 1;
    std::lock_guard<std::mutex> lk(s.async->m);
    return s.async->done ? 1 : 0;
}

int fslic_hip_iterate(fslic_engine* e, const fslic_params* p, int H, int W, int K, const uint8_t* rgb,
                      fslic_cluster* clusters, uint16_t* labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (!rgb || !labels || !clusters) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    const uint8_t* rgbs[1] = {rgb};
    fslic_cluster* cls[1] = {clusters};
    uint16_t* labs[1] = {labels};
    int rc = fslic_hip_iterate_batch(e, p, H, W, K, 1, rgbs, cls, labs, 0);
    if (rc == FSLIC_OK) make_timing_report(e, e->slots[0]);
    return rc;
}

int fslic_hip_iterate_batch(fslic_engine* e, const fslic_params* p, int H, int W, int K, int n_frames,
                            const uint8_t* const* rgb, fslic_cluster* const* clusters,
                            uint16_t* const* labels, int device_ptrs) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (n_frames < 0 || (n_frames > 0 && (!rgb || !clusters || !labels))) return fail(FSLIC_E_INVALID, "bad batch arguments");
    HIPCHK(hipSetDevice(e->device));
    const int ns = (int)e->slots.size();
    const size_t N = (size_t)H * W;
    int S = 0;
    int rc = validate(p, H, W, K, S);
    if (rc) return rc;
    for (int i = 0; i < n_frames; i++)
        if (!rgb[i] || !clusters[i] || !labels[i]) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    for (const Slot& sl : e->slots)
        if (sl.pending) return fail(FSLIC_E_INVALID, "a slot still owns an unfinished group");
    // Frames are cut into groups of up to group_size; every launch of a group covers all of its frames (frame =
    // last grid dimension).  Groups rotate over the slots (one stream each), so the host-side tail of one group
    // (cluster write-back, rare host top-K) overlaps the kernels of the next.
    int G = std::min(std::max(e->group_size, 1), (int)kMaxGroup);
    if (n_frames < G * ns) G = std::max(1, (n_frames + ns - 1) / ns);       // spread a small batch over the slots
    struct Pending { int slot, first, n; };
    std::vector<Pending> inflight;
    auto finish_one = [&](const Pending& pd) -> int {
        Slot& s = e->slots[pd.slot];
        int r = group_finish(e, s);
        if (r) return r;
        if (!device_ptrs) {
            for (int i = 0; i < pd.n; i++)
                HIPCHK(hipMemcpyAsync(labels[pd.first + i], s.at(s.d_out_stage, i), N * 2, hipMemcpyDeviceToHost, s.st));
            HIPCHK(hipStreamSynchronize(s.st));
        }
        return FSLIC_OK;
    };
    int next_slot = 0;
    for (int first = 0; first < n_frames; first += G) {
        const int n = std::min(G, n_frames - first);
        const int si = next_slot;
        next_slot = (next_slot + 1) % ns;
        // the slot may still own an unfinished group
        for (size_t q = 0; q < inflight.size(); q++)
            if (inflight[q].slot == si) {
                rc = finish_one(inflight[q]);
                if (rc) return rc;
                inflight.erase(inflight.begin() + q);
                break;
            }
        Slot& s = e->slots[si];
        const uint8_t* d_rgb[kMaxGroup];
        uint16_t* d_out[kMaxGroup];
        if (!device_ptrs) {
            rc = ensure_prepared(e, s, H, W, K, S, n);
            if (rc) return rc;
            for (int i = 0; i < n; i++) {
                HIPCHK(hipMemcpyAsync(s.at(s.d_rgb_stage, i), rgb[first + i], N * 3, hipMemcpyHostToDevice, s.st));
                d_rgb[i] = s.at(s.d_rgb_stage, i);
                d_out[i] = s.at(s.d_out_stage, i);
            }
        } else {
            for (int i = 0; i < n; i++) { d_rgb[i] = rgb[first + i]; d_out[i] = labels[first + i]; }
        }
        s.launch_timing = e->launch_timing;
        rc = group_begin(e, s, p, H, W, K, n, d_rgb, clusters + first, d_out);
        if (rc) return rc;
        inflight.push_back({si, first, n});
    }
    for (const Pending& pd : inflight) {
        rc = finish_one(pd);
        if (rc) return rc;
    }
    return FSLIC_OK;
}

int fslic_hip_rgb_to_lab(fslic_engine* e, int H, int W, const uint8_t* rgb, int convert, uint8_t* lab4) {
    if (!e || !rgb || !lab4) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "bad image size");
    HIPCHK(hipSetDevice(e->device));
    const size_t N = (size_t)H * W;
    uint8_t* d_rgb = nullptr;
    uint32_t* d_lab = nullptr;
    HIPCHK(hipMalloc((void**)&d_rgb, N * 3));
    if (hipMalloc((void**)&d_lab, N * 4) != hipSuccess) { hipFree(d_rgb); return fail(FSLIC_E_HIP, "hipMalloc failed"); }
    hipStream_t st = e->slots[0].st;
    hipMemcpyAsync(d_rgb, rgb, N * 3, hipMemcpyHostToDevice, st);
    FrameDev f{};
    f.N = (int)N; f.H = H; f.W = W;
    f.lab = d_lab;
    const uint8_t* in1[1] = {d_rgb};
    if (upload_ptrs(e->slots[0], 1, in1, nullptr) != FSLIC_OK) { hipFree(d_rgb); hipFree(d_lab); return FSLIC_E_HIP; }
    f.rgbs = reinterpret_cast<const uint8_t* const*>(e->slots[0].d_ptrs);
    launch_rgb_to_lab(f, 1, convert, e->tables, false, st);
    hipMemcpyAsync(lab4, d_lab, N * 4, hipMemcpyDeviceToHost, st);
    hipError_t err = hipStreamSynchronize(st);
    hipFree(d_rgb);
    hipFree(d_lab);
    if (err != hipSuccess) return fail(FSLIC_E_HIP, hipGetErrorString(err));
    return FSLIC_OK;
}

int fslic_hip_enforce_connectivity(fslic_engine* e, uint16_t* labels, int H, int W, int K, int min_threshold) {
    if (!e || !labels) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "bad image size");
    if (K <= 0) return FSLIC_OK;   // src/context.cpp:16
    HIPCHK(hipSetDevice(e->device));
    Slot& s = e->slots[0];
    const int Kb = std::min(std::max(K, 1), 65536);
    // any S works for the connectivity buffers; reuse the frame carve
    int S = (int)sqrt((double)H * W / Kb);
    if (S < 1) S = 1;
    HIPCHK(hipStreamSynchronize(s.st));
    int rc = prepare(e, s, H, W, Kb, S, 1);
    if (rc) return rc;
    s.keyH = s.keyW = s.keyK = 0;   // the carve above is not a frame configuration
    s.have_pre = false;
    const size_t N = (size_t)H * W;
    HIPCHK(hipMemsetAsync(s.zero_block, 0, s.zero_bytes, s.st));
    HIPCHK(hipMemcpyAsync(s.d_out_stage, labels, N * 2, hipMemcpyHostToDevice, s.st));
    uint16_t* outs[1] = {s.d_out_stage};
    rc = upload_ptrs(s, 1, nullptr, outs);
    if (rc) return rc;
    const CcaDev c = cca_view(s, 0, s.d_out_stage, 0, K, min_threshold);
    cca_enqueue(s, c, 0, 1);
    HIPCHK(hipMemcpyAsync(s.h_misc, s.d_misc, 64, hipMemcpyDeviceToHost, s.st));
    HIPCHK(hipStreamSynchronize(s.st));
    rc = cca_finish_group(s, 0, 1, s.d_out_stage, 0, K, min_threshold);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(labels, s.d_out_stage, N * 2, hipMemcpyDeviceToHost, s.st));
    HIPCHK(hipStreamSynchronize(s.st));
    return FSLIC_OK;
}

// ---- superpixel-graph utilities on a finished label map (src/fast-slic.cpp) ----
namespace {
// A caller's plane: used in place when it already lives in device memory, staged through a temporary otherwise.
struct Staged {
    void* p = nullptr;
    bool owned = false;
    ~Staged() { if (owned && p) hipFree(p); }
};
bool is_device_pointer(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain host memory
    return a.type == hipMemoryTypeDevice;
}
int stage_in(const void* src, size_t bytes, hipStream_t st, Staged& out) {
    if (is_device_pointer(src)) { out.p = const_cast<void*>(src); return FSLIC_OK; }
    HIPCHK(hipMalloc(&out.p, std::max<size_t>(bytes, 16)));
    out.owned = true;
    HIPCHK(hipMemcpyAsync(out.p, src, bytes, hipMemcpyHostToDevice, st));
    return FSLIC_OK;
}
int device_scratch(size_t bytes, Staged& out) {
    HIPCHK(hipMalloc(&out.p, std::max<size_t>(bytes, 16)));
    out.owned = true;
    return FSLIC_OK;
}
int check_map(const fslic_engine* e, int H, int W, int K) {
    if (!e) return fail(FSLIC_E_INVALID, "NULL engine");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "bad image size");
    if (K <= 0 || K > 65536) return fail(FSLIC_E_INVALID, "K must be in [1, 65536]");
    return FSLIC_OK;
}
}  // namespace

int fslic_hip_get_mask_density(fslic_engine* e, int H, int W, int K, const fslic_cluster* clusters, const uint16_t* labels,
                               const uint8_t* mask, uint8_t* densities) {
    int rc = check_map(e, H, W, K);
    if (rc) return rc;
    if (!clusters || !labels || !mask || !densities) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->slots[0].st;
    const size_t N = (size_t)H * W;
    Staged d_lab, d_mask, d_sums;
    if ((rc = stage_in(labels, N * 2, st, d_lab)) || (rc = stage_in(mask, N, st, d_mask)) || (rc = device_scratch((size_t)K * 4, d_sums))) return rc;
    HIPCHK(hipMemsetAsync(d_sums.p, 0, (size_t)K * 4, st));
    launch_mask_sums((const uint16_t*)d_lab.p, (const uint8_t*)d_mask.p, (uint32_t*)d_sums.p, H, W, K, st);
    std::vector<uint32_t> sums((size_t)K);
    HIPCHK(hipMemcpyAsync(sums.data(), d_sums.p, (size_t)K * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // src/fast-slic.cpp:152-154: int / unsigned -> unsigned division, then min<int>(255, .)
    for (int k = 0; k < K; k++) {
        const uint32_t q = sums[(size_t)k] / std::max<uint32_t>(clusters[k].num_members, 1u);
        densities[k] = (uint8_t)std::min<int>(255, (int)q);
    }
    return FSLIC_OK;
}

int fslic_hip_cluster_density_to_mask(fslic_engine* e, int H, int W, int K, const uint16_t* labels, const uint8_t* densities,
                                      uint8_t* result) {
    int rc = check_map(e, H, W, K);
    if (rc) return rc;
    if (!labels || !densities || !result) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->slots[0].st;
    const size_t N = (size_t)H * W;
    Staged d_lab, d_dens, d_out;
    if ((rc = stage_in(labels, N * 2, st, d_lab)) || (rc = stage_in(densities, (size_t)K, st, d_dens))) return rc;
    const bool out_on_device = is_device_pointer(result);
    if (out_on_device) d_out.p = result;
    else if ((rc = device_scratch(N, d_out))) return rc;
    launch_density_to_mask((const uint16_t*)d_lab.p, (const uint8_t*)d_dens.p, (uint8_t*)d_out.p, N, K, st);
    if (!out_on_device) HIPCHK(hipMemcpyAsync(result, d_out.p, N, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return FSLIC_OK;
}

int fslic_hip_get_connectivity(fslic_engine* e, int H, int W, int K, const uint16_t* labels, int* num_neighbors, uint32_t* neighbors) {
    constexpr int kMaxConn = 12;                       // src/fast-slic.cpp:17
    int rc = check_map(e, H, W, K);
    if (rc) return rc;
    if (!labels || !num_neighbors || !neighbors) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    std::fill_n(num_neighbors, K, 0);
    std::fill_n(neighbors, (size_t)K * kMaxConn, 0u);
    if (H < 2 || W < 2) return FSLIC_OK;               // the scan covers i < H-1, j < W-1 (src/fast-slic.cpp:31-32)
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->slots[0].st;
    const size_t N = (size_t)H * W;
    Staged d_lab;
    if ((rc = stage_in(labels, N * 2, st, d_lab))) return rc;
    // distinct adjacent pairs with the scan position of their first occurrence; the table doubles until it is at most
    // half full (a planar label map has about 3K pairs; a noisy one may have many more)
    uint32_t cap = 4096;
    while (cap < 32u * (uint32_t)K) cap <<= 1;
    std::vector<uint32_t> h_pair;
    std::vector<unsigned long long> h_key;
    for (;; cap <<= 1) {
        if (cap > (1u << 26)) return fail(FSLIC_E_INVALID, "label map has more than 2^25 distinct adjacent label pairs");
        Staged d_pair, d_key, d_cnt;
        if ((rc = device_scratch((size_t)cap * 4, d_pair)) || (rc = device_scratch((size_t)cap * 8, d_key)) || (rc = device_scratch(8, d_cnt))) return rc;
        HIPCHK(hipMemsetAsync(d_pair.p, 0xFF, (size_t)cap * 4, st));
        HIPCHK(hipMemsetAsync(d_key.p, 0xFF, (size_t)cap * 8, st));
        HIPCHK(hipMemsetAsync(d_cnt.p, 0, 8, st));
        launch_adjacent_pairs((const uint16_t*)d_lab.p, H, W, K, (uint32_t*)d_pair.p, (unsigned long long*)d_key.p, cap, (uint32_t*)d_cnt.p, st);
        uint32_t cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (cnt[1] != 0 || cnt[0] > cap / 2) continue;
        h_pair.resize(cap); h_key.resize(cap);
        HIPCHK(hipMemcpyAsync(h_pair.data(), d_pair.p, (size_t)cap * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(h_key.data(), d_key.p, (size_t)cap * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        break;
    }
    // Scan order = ascending event key.  A pair is accepted at its first occurrence iff both ends still have fewer than
    // 12 neighbours (src/fast-slic.cpp:41-42; the counts never decrease, so a pair refused once is refused for good),
    // and then appended to both lists (:62-63).
    std::vector<std::pair<unsigned long long, uint32_t>> ev;
    for (uint32_t i = 0; i < cap; i++)
        if (h_pair[i] != 0xFFFFFFFFu) ev.emplace_back(h_key[i], h_pair[i]);
    std::sort(ev.begin(), ev.end());
    for (const auto& x : ev) {
        const uint32_t a = x.second >> 16, b = x.second & 0xFFFFu;
        if (num_neighbors[a] >= kMaxConn || num_neighbors[b] >= kMaxConn) continue;
        neighbors[(size_t)a * kMaxConn + num_neighbors[a]++] = b;
        neighbors[(size_t)b * kMaxConn + num_neighbors[b]++] = a;
    }
    return FSLIC_OK;
}

// Host only (K clusters, no pixel data): the reference's grid search with libstdc++ heap operations, whose order the
// result exposes (src/fast-slic.cpp:80-130).  Float expressions as the reference build evaluates them: the cell index
// is trunc(fma(y / S, nw, x / S)) -- one float expression, contracted, NOT floor(y/S)*nw + floor(x/S).
int fslic_hip_knn_connectivity(int H, int W, int K, const fslic_cluster* clusters, size_t n_neighbors, int* num_neighbors, uint32_t* neighbors) {
    if (!clusters || !num_neighbors || (!neighbors && n_neighbors)) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (H <= 0 || W <= 0 || K <= 0) return fail(FSLIC_E_INVALID, "H, W and K must be positive");
    const int S = std::max((int)std::sqrt((double)(H * W / K)), 1);              // :82
    const int nh = (H + S - 1) / S, nw = (W + S - 1) / S;
    std::vector<std::vector<int>> cells((size_t)nh * nw);
    for (int i = 0; i < K; i++) {
        const float v = std::fmaf(clusters[i].y / (float)S, (float)nw, clusters[i].x / (float)S);     // :88
        if (!(v >= 0.0f) || (size_t)v >= cells.size()) return fail(FSLIC_E_INVALID, "cluster centre outside the image");
        cells[(size_t)v].push_back(i);
    }
    std::vector<std::pair<int, int>> heap;            // (distance, cluster index): index order == the reference's pointer order
    for (int i = 0; i < K; i++) {
        const fslic_cluster& c = clusters[i];
        const int ccx = (int)(c.x / (float)S), ccy = (int)(c.y / (float)S);        // :100
        heap.clear();
        for (int cy = std::max(ccy - 3, 0); cy < std::min(nh, ccy + 3); cy++) {
            for (int cx = std::max(ccx - 3, 0); cx < std::min(nw, ccx + 3); cx++) {
                for (int j : cells[(size_t)cy * nw + cx]) {
                    if (j == i) continue;
                    const float dx = clusters[j].x - c.x, dy = clusters[j].y - c.y;
                    const int distance = (int)((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy));     // :107
                    if (!heap.empty() && heap.front().first <= distance) continue;
                    heap.push_back(std::pair<int, int>(distance, j));
                    std::push_heap(heap.begin(), heap.end());
                    while (heap.size() > n_neighbors) {
                        std::pop_heap(heap.begin(), heap.end());
                        heap.pop_back();
                    }
                }
            }
        }
        num_neighbors[i] = (int)heap.size();
        for (size_t j = 0; j < n_neighbors; j++) neighbors[(size_t)i * n_neighbors + j] = j < heap.size() ? clusters[heap[j].second].number : 0u;
    }
    return FSLIC_OK;
}

int fslic_hip_last_prelabels(fslic_engine* e, int slot, uint16_t* prelabels) {
    if (!e || !prelabels) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    Slot& s = e->slots[slot];
    if (!s.have_pre || s.keyH == 0) return fail(FSLIC_E_INVALID, "no frame has been processed on this slot");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(prelabels, s.f.labels, (size_t)s.f.N * 2, hipMemcpyDeviceToHost));
    return FSLIC_OK;
}

const char* fslic_hip_last_timing_report(fslic_engine* e) { return e ? e->timing.c_str() : ""; }

int fslic_hip_last_device_times(fslic_engine* e, int slot, float* total_ms, float* full_assign_ms) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "bad engine/slot");
    if (total_ms) *total_ms = e->slots[slot].total_ms;
    if (full_assign_ms) *full_assign_ms = e->slots[slot].fa_ms;
    return FSLIC_OK;
}

int fslic_hip_set_launch_timing(fslic_engine* e, int on) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    e->launch_timing = on != 0;
    return FSLIC_OK;
}

int fslic_hip_last_assign_loop(fslic_engine* e, int slot, float* sum_ms, double* visited_px, int* launches) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "bad engine/slot");
    const Slot& s = e->slots[slot];
    if (sum_ms) *sum_ms = s.assign_loop_ms;
    if (visited_px) *visited_px = s.assign_loop_px;
    if (launches) *launches = s.n_timed_iters;
    return FSLIC_OK;
}

int fslic_hip_last_group_frames(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    return e->slots[slot].nframes;
}

int fslic_hip_last_launch_mode(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    return e->slots[slot].last_launch_mode;
}

int fslic_hip_last_path(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    return e->slots[slot].last_path;
}

}  // extern "C"
