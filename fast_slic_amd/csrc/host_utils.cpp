// host_utils.cpp -- entry points of the C ABI that are not iterate(): the O(K) host functions (grid seeding, knn
// graph), the stage entry points used by the parity tests (LAB conversion, connectivity on a caller's label map) and the
// superpixel-graph utilities on a finished label map (src/fast-slic.cpp).  Part of the host engine, see engine_internal.h.
#include "engine_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace fslic;

extern "C" {

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

int fslic_hip_rgb_to_lab(fslic_engine* e, int H, int W, const uint8_t* rgb, int convert, uint8_t* lab4) {
    if (!e || !rgb || !lab4) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "bad image size");
    HIPCHK(hipSetDevice(e->device));
    const size_t N = (size_t)H * W;
    uint8_t* d_rgb = nullptr;
    uint32_t* d_lab = nullptr;
    HIPCHK(hipMalloc((void**)&d_rgb, N * 3));
    if (hipMalloc((void**)&d_lab, N * 4) != hipSuccess) { hipFree(d_rgb); return fail(FSLIC_E_HIP, "hipMalloc failed"); }
    SlotLease lease(e);                       // the slot's stream and pointer table for the length of the call
    if (lease.take() != FSLIC_OK) { hipFree(d_rgb); hipFree(d_lab); return FSLIC_E_INVALID; }
    Slot& sl = e->slots[lease.slot];
    hipStream_t st = sl.st;
    hipMemcpyAsync(d_rgb, rgb, N * 3, hipMemcpyHostToDevice, st);
    FrameDev f{};
    f.N = (int)N; f.H = H; f.W = W;
    f.lab = d_lab;
    const uint8_t* in1[1] = {d_rgb};
    if (upload_ptrs(sl, 1, in1, nullptr) != FSLIC_OK) { hipFree(d_rgb); hipFree(d_lab); return FSLIC_E_HIP; }
    f.rgbs = reinterpret_cast<const uint8_t* const*>(sl.d_ptrs);
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
    // K caps the number of kept components, which become the labels 0 .. n-1 of a uint16 map
    if (K > 65536) return fail(FSLIC_E_INVALID, "K must be at most 65536 (labels are uint16)");
    HIPCHK(hipSetDevice(e->device));
    SlotLease lease(e);
    int rc = lease.take();
    if (rc) return rc;
    Slot& s = e->slots[lease.slot];
    // any S works for the connectivity buffers; reuse the frame carve
    int S = (int)sqrt((double)H * W / K);
    if (S < 1) S = 1;
    HIPCHK(hipStreamSynchronize(s.st));
    rc = prepare(e, s, H, W, K, S, 1);
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
    SlotLease lease(e);
    if ((rc = lease.take())) return rc;
    hipStream_t st = e->slots[lease.slot].st;
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
    SlotLease lease(e);
    if ((rc = lease.take())) return rc;
    hipStream_t st = e->slots[lease.slot].st;
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
    SlotLease lease(e);
    if ((rc = lease.take())) return rc;
    hipStream_t st = e->slots[lease.slot].st;
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

// Measurement aid of bench.py: the rate of a plain 16-bytes-per-lane streaming copy on this GPU (bytes read + bytes written per
// second), best of `reps` launches after one warm-up, HIP events on a slot's stream.  Buffers are allocated for the call.
int fslic_hip_copy_bandwidth(fslic_engine* e, size_t bytes, int reps, double* gb_per_s) {
    if (!e || !gb_per_s) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (bytes < 16 || reps < 1) return fail(FSLIC_E_INVALID, "bytes and reps must be positive");
    bytes &= ~(size_t)15;
    SlotLease lease(e);
    int rc = lease.take();
    if (rc) return rc;
    Slot& s = e->slots[lease.slot];
    HIPCHK(hipSetDevice(e->device));
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) {
        if (a) hipFree(a);
        (void)hipGetLastError();
        return fail(FSLIC_E_HIP, "hipMalloc failed for the copy probe");
    }
    hipError_t err = hipMemsetAsync(a, 1, bytes, s.st);
    double best = 0.0;
    for (int r = 0; r <= reps && err == hipSuccess; r++) {
        err = hipEventRecord(s.ev[0], s.st);
        launch_copy16(a, b, bytes, s.st);
        if (err == hipSuccess) err = hipEventRecord(s.ev[1], s.st);
        if (err == hipSuccess) err = hipStreamSynchronize(s.st);
        float ms = 0.0f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, s.ev[0], s.ev[1]);
        if (r > 0 && ms > 0.0f) best = std::max(best, 2.0 * (double)bytes / ((double)ms * 1e-3) / 1e9);
    }
    hipFree(a); hipFree(b);
    if (err != hipSuccess) return fail(FSLIC_E_HIP, std::string("copy probe: ") + hipGetErrorString(err));
    *gb_per_s = best;
    return FSLIC_OK;
}

}  // extern "C"
