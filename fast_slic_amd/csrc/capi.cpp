// capi.cpp -- the C ABI of include/fslic_hip.h: engine lifetime, iterate() in its synchronous forms, accessors.
// (The asynchronous forms live in pipeline.cpp, the O(K) host functions and stage entry points in host_utils.cpp.)
// Part of the host engine, see engine_internal.h.
#include "engine_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace fslic;

static_assert(sizeof(fslic_cluster) == 32, "Cluster ABI (src/fast-slic-common.h:10-23)");

extern "C" {

const char* fslic_hip_last_error(void) { return last_error().c_str(); }
const char* fslic_hip_version(void) { return "fast_slic_amd 0.3 (gfx950)"; }

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
    const HostTables& ht = host_tables();
    fslic_engine* e = new fslic_engine();
    e->device = device;
    e->group_size = knobs().group_size;
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
    if (hipMalloc((void**)&e->d_gamma, sizeof ht.gamma) != hipSuccess ||
        hipMalloc((void**)&e->d_labtbl, sizeof ht.lab) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "hipMalloc(tables) failed"); }
    if (hipMemcpy(e->d_gamma, ht.gamma, sizeof ht.gamma, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->d_labtbl, ht.lab, sizeof ht.lab, hipMemcpyHostToDevice) != hipSuccess) { fslic_hip_destroy(e); return fail(FSLIC_E_HIP, "table upload failed"); }
    e->tables.gamma = e->d_gamma;
    e->tables.labtbl = e->d_labtbl;
    for (int i = 0; i < 9; i++) e->tables.cb[i] = ht.cb[i];
    *out = e;
    return FSLIC_OK;
}

void fslic_hip_destroy(fslic_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    for (auto& s : e->slots) stop_slot_thread(e, s);     // a group still in flight is completed by its thread first
    for (auto& s : e->slots) {
        if (s.st) hipStreamSynchronize(s.st);
        free_slot(s);
    }
    if (e->d_gamma) hipFree(e->d_gamma);
    if (e->d_labtbl) hipFree(e->d_labtbl);
    delete e;
}

int fslic_hip_iterate_device(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K,
                             const uint8_t* d_rgb, fslic_cluster* clusters, uint16_t* d_labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    HIPCHK(hipSetDevice(e->device));
    SlotLease lease(e);
    int rc = lease.take(slot);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    s.launch_timing = e->launch_timing;
    rc = group_begin(e, s, p, H, W, K, 1, &d_rgb, &clusters, &d_labels);
    if (rc == FSLIC_OK) rc = group_finish(e, s);
    if (rc != FSLIC_OK) {            // a failure after a partial enqueue: nothing may still touch the caller's buffers on return
        const std::string msg = last_error();
        (void)hipStreamSynchronize(s.st);
        (void)hipGetLastError();
        set_last_error(msg);
        return rc;
    }
    set_thread_timing_report(make_timing_report(s));
    return rc;
}

// Host frame in, host label map out, on whichever slot is free: concurrent calls from different threads (different
// SlicModels sharing the process-wide engine) proceed in parallel on different slots.
int fslic_hip_iterate(fslic_engine* e, const fslic_params* p, int H, int W, int K, const uint8_t* rgb,
                      fslic_cluster* clusters, uint16_t* labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (!rgb || !labels || !clusters) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    HIPCHK(hipSetDevice(e->device));
    int S = 0;
    int rc = validate(p, H, W, K, S);
    if (rc) return rc;
    SlotLease lease(e);
    rc = lease.take();
    if (rc) return rc;
    Slot& s = e->slots[lease.slot];
    const size_t N = (size_t)H * W;
    rc = ensure_prepared(e, s, H, W, K, S, 1);
    if (rc) return rc;
    const uint8_t* d_rgb = s.d_rgb_stage;
    uint16_t* d_out = s.d_out_stage;
    // Pageable hipMemcpyAsync in and out on the slot's stream (the runtime's own staging).  Measured against it and not kept
    // (profiles/r03_e2e_probe.txt, 1280x720 K=1600, one caller thread: 432 us per call like this): pinned staging read and written
    // in place by the LAB / relabel kernels (490 us: kernels working over PCIe take 84 us longer than on HBM) and pinned staging
    // filled by a pool of copy threads plus copy commands (543 us).
    const double t0 = knobs().host_timing ? now_us() : 0.0;
    HIPCHK(hipMemcpyAsync(s.d_rgb_stage, rgb, N * 3, hipMemcpyHostToDevice, s.st));
    const double t1 = knobs().host_timing ? now_us() : 0.0;
    s.launch_timing = e->launch_timing;
    rc = group_begin(e, s, p, H, W, K, 1, &d_rgb, &clusters, &d_out);
    const double t2 = knobs().host_timing ? now_us() : 0.0;
    if (rc == FSLIC_OK) rc = group_finish(e, s);
    if (rc != FSLIC_OK) {            // a failure after a partial enqueue: nothing may still run on the slot's stage buffers when the slot is released
        const std::string msg = last_error();
        (void)hipStreamSynchronize(s.st);
        (void)hipGetLastError();
        set_last_error(msg);
        return rc;
    }
    const double t3 = knobs().host_timing ? now_us() : 0.0;
    HIPCHK(hipMemcpyAsync(labels, s.d_out_stage, N * 2, hipMemcpyDeviceToHost, s.st));
    HIPCHK(hipStreamSynchronize(s.st));
    if (knobs().host_timing)
        fprintf(stderr, "[fslic host] iterate: frame in %.1f us, group begin %.1f, group finish (wait + write-back) %.1f, labels out %.1f (device %.1f us)\n",
                t1 - t0, t2 - t1, t3 - t2, now_us() - t3, s.total_ms * 1e3);
    set_thread_timing_report(make_timing_report(s));
    return FSLIC_OK;
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
    rc = acquire_all_slots(e);
    if (rc) return rc;
    // Frames are cut into groups of up to group_size; every launch of a group covers all of its frames (frame =
    // last grid dimension).  Groups rotate over the slots (one stream each), so the host-side tail of one group
    // (cluster write-back) overlaps the kernels of the next.
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
    // On an error nothing may still be running against the caller's buffers when control returns: every group in
    // flight is waited for first (their results are dropped), then the first error is reported.
    auto bail = [&](int code) -> int {
        const std::string msg = last_error();
        for (const Pending& pd : inflight) (void)hipStreamSynchronize(e->slots[pd.slot].st);
        (void)hipGetLastError();
        release_all_slots(e);
        set_last_error(msg);
        return code;
    };
    int next_slot = 0;
    for (int first = 0; first < n_frames; first += G) {
        const int n = std::min(G, n_frames - first);
        const int si = next_slot;
        next_slot = (next_slot + 1) % ns;
        // the slot may still own an unfinished group
        for (size_t q = 0; q < inflight.size(); q++)
            if (inflight[q].slot == si) {
                const Pending pd = inflight[q];
                inflight.erase(inflight.begin() + q);
                rc = finish_one(pd);
                if (rc) return bail(rc);
                break;
            }
        Slot& s = e->slots[si];
        const uint8_t* d_rgb[kMaxGroup];
        uint16_t* d_out[kMaxGroup];
        if (!device_ptrs) {
            rc = ensure_prepared(e, s, H, W, K, S, n);
            if (rc) return bail(rc);
            for (int i = 0; i < n; i++) {
                if (hipMemcpyAsync(s.at(s.d_rgb_stage, i), rgb[first + i], N * 3, hipMemcpyHostToDevice, s.st) != hipSuccess)
                    return bail(fail(FSLIC_E_HIP, "frame upload failed"));
                d_rgb[i] = s.at(s.d_rgb_stage, i);
                d_out[i] = s.at(s.d_out_stage, i);
            }
        } else {
            for (int i = 0; i < n; i++) { d_rgb[i] = rgb[first + i]; d_out[i] = labels[first + i]; }
        }
        s.launch_timing = e->launch_timing;
        inflight.push_back({si, first, n});                  // from here on the slot's stream may hold work of this group
        rc = group_begin(e, s, p, H, W, K, n, d_rgb, clusters + first, d_out);
        if (rc) return bail(rc);
    }
    while (!inflight.empty()) {
        const Pending pd = inflight.front();
        inflight.erase(inflight.begin());
        rc = finish_one(pd);
        if (rc) return bail(rc);
    }
    if (n_frames > 0) set_thread_timing_report(make_timing_report(e->slots[0]));
    release_all_slots(e);
    return FSLIC_OK;
}

int fslic_hip_last_prelabels(fslic_engine* e, int slot, uint16_t* prelabels) {
    if (!e || !prelabels) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    HIPCHK(hipSetDevice(e->device));
    SlotLease lease(e);
    int rc = lease.take(slot);
    if (rc) return rc;
    Slot& s = e->slots[slot];
    if (!s.have_pre || s.keyH == 0) return fail(FSLIC_E_INVALID, "no frame has been processed on this slot");
    HIPCHK(hipMemcpy(prelabels, s.f.labels, (size_t)s.f.N * 2, hipMemcpyDeviceToHost));
    return FSLIC_OK;
}

// The report of the calling thread's last synchronous call or fslic_hip_wait_group (thread-local, like the reference's
// timer, src/timer.cpp:45); the engine argument is kept for the shape of the reference interface.
const char* fslic_hip_last_timing_report(fslic_engine* e) { return e ? thread_timing_report().c_str() : ""; }

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

int fslic_hip_lab_force_generic(fslic_engine* e, int on) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    e->lab_force_generic.store(on != 0);
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

int fslic_hip_last_host_topk_frames(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    return e->slots[slot].n_host_topk;
}

int fslic_hip_separate_pass_redos(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    return __atomic_load_n(&e->slots[slot].n_separate_redo, __ATOMIC_RELAXED);
}

#ifdef FSLIC_LAB
// lab build only: the 16 status words of frame `frame` of the last group on `slot` as the export left them in pinned memory
int fslic_hip_debug_status_words(fslic_engine* e, int slot, int frame, uint32_t* out16) {
    if (!e || slot < 0 || slot >= (int)e->slots.size() || frame < 0 || frame >= (int)kMaxGroup || !out16) return FSLIC_E_INVALID;
    std::memcpy(out16, e->slots[slot].h_misc + 16 * (size_t)frame, 64);
    return FSLIC_OK;
}
// lab build only: the last `nwords` words of frame `frame`'s candidate-leader array (where lab build 2 leaves the tile
// kernel's time stamps)
int fslic_hip_debug_cand_tail(fslic_engine* e, int slot, int frame, int nwords, int32_t* out) {
    if (!e || slot < 0 || slot >= (int)e->slots.size() || !out) return FSLIC_E_INVALID;
    Slot& s = e->slots[slot];
    const int32_t* base = s.at(s.c.cand_leader, frame) + (size_t)s.c.N - nwords;
    return hipMemcpy(out, base, sizeof(int32_t) * (size_t)nwords, hipMemcpyDeviceToHost) == hipSuccess ? FSLIC_OK : FSLIC_E_HIP;
}
// lab build 4: `nwords` words from the start of frame `frame`'s candidate-area array (phase stamps of the block assign kernel)
int fslic_hip_debug_cand_area_head(fslic_engine* e, int slot, int frame, int nwords, uint32_t* out) {
    if (!e || slot < 0 || slot >= (int)e->slots.size() || !out) return FSLIC_E_INVALID;
    Slot& s = e->slots[slot];
    const uint32_t* base = s.at(s.c.cand_area, frame);
    return hipMemcpy(out, base, sizeof(uint32_t) * (size_t)nwords, hipMemcpyDeviceToHost) == hipSuccess ? FSLIC_OK : FSLIC_E_HIP;
}
#endif

}  // extern "C"
