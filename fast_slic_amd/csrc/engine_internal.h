// engine_internal.h -- shared declarations of the host engine's translation units (internal to the library).
//
//   tables.cpp      RGB->LAB tables, spatial patch and its device encodings (src/cielab.h, src/context.cpp:22-40)
//   arena.cpp       per-slot device / pinned buffers: allocation, carving, release
//   group.cpp       one launch group: staging, the launch sequence (direct or as a hipGraph), completion, write-back
//   pipeline.cpp    slot ownership (synchronous calls, asynchronous groups, the submit/drain pipeline), slot threads
//   capi.cpp        the C ABI of include/fslic_hip.h for iterate() and friends
//   host_utils.cpp  O(K) host functions and the stage / superpixel-graph entry points
//
// The engine replaces, for the arch "hip/gfx950", what SlicModel.iterate builds per call in the reference: a Context
// (src/context.h:59-66), initialize_state(), iterate() (src/context.cpp:108-197) and its teardown
// (cfast_slic.pyx:171-197).  Device and pinned buffers are owned here and cached across calls; the only state that
// carries over between calls is the caller's Cluster[K].
#pragma once
#include "kernels.h"
#include "launch.h"
#include "../../include/fslic_hip.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace fslic {

// ---- errors: thread-local message, integer status (no exception crosses the ABI) ----
int fail(int code, const std::string& msg);
const std::string& last_error();
void set_last_error(const std::string& msg);

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::fslic::fail(FSLIC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
double now_us();

// Environment switches, read ONCE when the library is loaded (never on a call path).
//   FSLIC_GROUP        frames per launch group of fslic_hip_iterate_batch (default 8, at most 16)
//   FSLIC_GRAPH=0      enqueue every group operation by operation instead of replaying a recorded hipGraph
//   FSLIC_POISON       testing aid: fill a freshly carved arena with 0xA5 (reads of never-written memory show up)
//   FSLIC_HOST_TIMING  one stderr line per group start / completion with host-side durations
//   FSLIC_FUSEBIN      the cluster pass between two assign passes: 1 (default) fused into the assign kernel for launches that do
//                      not fill the chip and a separate launch (k_bin_clusters<1>) otherwise; 0 always separate (round 2; also what
//                      a frame with a stale pixel is redone with); 2 always fused (A/B measurements)
struct Knobs {
    int group_size;
    bool use_graphs, poison, host_timing;
    int fuse_bin;
};
const Knobs& knobs();

constexpr int kDenseCap = 8192;          // candidates the select kernel resolves on the device (== its LDS sort capacity)
constexpr size_t kTabMaxBytes = 40960;   // LDS budget of the packed kernel's spatial table (row-vector mode: per table)
constexpr size_t kLutMaxWords = 12288;   // 48 KB of LDS for the spatial table at most

// Host copies of the RGB->LAB tables (src/cielab.h:296-305), built once per process.
struct HostTables {
    uint16_t gamma[256];
    uint16_t lab[8194];
    int cb[9];
};
const HostTables& host_tables();

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
    char* stamp_block = nullptr;     // the bin slots' generation stamps (cleared when carved and when the stamps wrap)
    size_t stamp_bytes = 0;
    uint32_t* d_misc = nullptr;      // 64 B: ovf_cnt[0..1], err flags (bit 0: candidate-list overflow, bit 1: stale pixel under the fused cluster pass), ovf_cnt[2], cca counters[4]
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
    // pinned host (device-accessible: the first cluster pass reads the centres from it, the export kernel writes the
    // results into it -- no copy commands in a group's launch sequence)
    uint32_t* h_cl = nullptr;        // per frame 4K words: in [0,K) yx; out (yx, lab, n, moved)
    size_t h_cl_words = 0;
    uint32_t* h_misc = nullptr;      // per frame 16 words (copy of d_misc)
    uint16_t* h_patch = nullptr;
    size_t h_patch_cap = 0;
    int32_t* h_cand_leader = nullptr;
    uint32_t* h_cand_area = nullptr;
    size_t h_cand_cap = 0;
    uint32_t* d_gen = nullptr;       // device word: base of the bin generation stamps (FrameDev::gen_base)
    uint64_t gen_host = 0;           // host mirror of *d_gen
    uint32_t gen_span_prev = 0;      // stamps the previous group used above its base (max_iter + 3): what the next group's first kernel adds to *d_gen
    uint32_t gen_step = 0;           // ... for the group being enqueued
    int keyH = 0, keyW = 0, keyK = 0;
    size_t cand_capacity = 0;        // entries of the per-frame candidate arrays (>= N)
    // Recorded launch sequences (hipGraph, launch.h), one per distinct (geometry, options, group size, arena carving): a group
    // start is ~45 stream operations at ~3 us of host time each when enqueued one by one, ~8 us as one graph launch
    // (scripts/microbench/graph_launch.hip).  A key is recorded the second time it is seen.
    struct GraphEntry {
        std::vector<unsigned char> key;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int seen = 0;
        bool failed = false;
    };
    std::vector<GraphEntry> graphs;
    int last_launch_mode = 0;        // 0 direct, 1 recorded this call, 2 replayed
    bool launch_timing = false;      // the engine's flag as it stood when this slot's current group was submitted
    // the slot's own host thread waits for a group by sleeping and polling instead of spinning (group.cpp, nap_wait)
    bool nap_wait = false;           // set for groups served by the slot thread
    double recent_wait_us[4] = {1e30, 1e30, 1e30, 1e30};   // its last four waits (1e30: none yet)
    unsigned recent_wait_i = 0;
    unsigned long long recent_wait_key = 0;   // the workload those waits belong to (geometry, group size, iterations, variant): another one starts a new history
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
    int n_host_topk = 0;             // frames of the last group whose top-K step ran on the host
    int n_separate_redo = 0;         // (accessed with __atomic builtins) frames (since the slot was created) redone with the separate cluster pass because of a stale pixel
    bool have_pre = false;
    // Ownership (guarded by fslic_engine::mu): `busy` while a synchronous call or a stage utility runs on the slot,
    // `pending` from the submission of an asynchronous group until it has been collected.
    bool busy = false;
    bool pending = false;
    int wanted = 0;                  // synchronous callers waiting for THIS slot (fslic_hip_iterate_device with a slot number): its thread leaves the submit queue alone meanwhile
    // asynchronous groups run on the slot's own host thread (launches, stream sync, cluster write-back), so the host
    // work of one slot overlaps that of the others
    struct Async {
        std::thread worker;
        bool has_job = false, done = false, quit = false;      // guarded by fslic_engine::mu
        fslic_params p{};
        int H = 0, W = 0, K = 0, n = 0;
        const uint8_t* d_rgb[kMaxGroup] = {};
        fslic_cluster* clusters[kMaxGroup] = {};
        uint16_t* d_out[kMaxGroup] = {};
        int rc = 0;
        std::string err;
        bool from_queue = false;      // the job was taken from the engine's submit queue (the thread collects it itself)
        int jobs = 0;                 // ... and is this many submissions in one group
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

}  // namespace fslic

struct fslic_engine {
    int device = 0;
    int group_size = 8;              // frames per launch group of iterate_batch (FSLIC_GROUP)
    bool launch_timing = false;      // bracket every subsampled assign launch with HIP events (fslic_hip_set_launch_timing)
    std::vector<fslic::Slot> slots;
    uint16_t* d_gamma = nullptr;
    uint16_t* d_labtbl = nullptr;
    fslic::LabTables tables{};
    // slot ownership and the completion of asynchronous groups
    std::mutex mu;
    std::condition_variable cv;        // callers: a slot has become free, a group is complete, the queue has room, everything has drained
    std::condition_variable cv_work;   // the slot threads: a job has been handed over, the queue has work, a slot may serve the queue again, quit
                                       // (two variables so that a submission does not wake the callers and a completion does not wake the idle slot threads)
    // the submit / drain pipeline (fslic_hip_pipeline_*): first error of a collected group, totals since the last drain
    int pipe_rc = 0;
    std::string pipe_err;
    double pipe_device_ms = 0;
    long long pipe_groups = 0, pipe_frames = 0, pipe_host_topk = 0;
    // Submissions wait here for a slot thread (guarded by mu).  With batching on (fslic_hip_pipeline_batching), a thread
    // that finds several compatible submissions waiting serves them as ONE group: fewer, fuller launches when the
    // caller submits faster than the device finishes.
    struct PipeJob {
        fslic_params p{};
        int H = 0, W = 0, K = 0, n = 0;
        const uint8_t* d_rgb[fslic::kMaxGroup] = {};
        fslic_cluster* clusters[fslic::kMaxGroup] = {};
        uint16_t* d_out[fslic::kMaxGroup] = {};
    };
    std::deque<PipeJob> pipe_q;
    int pipe_inflight = 0;           // groups taken from the queue and not yet collected
    int pipe_batch_frames = 0;       // 0: one submission per group; otherwise the most frames a group may gather
    bool pipe_gathering = false;     // a slot thread is waiting briefly for a companion of the submission it took
    double pipe_last_submit_us = 0;  // (mu) when the last submission arrived: a companion is only waited for while submissions are arriving
    std::atomic<int> reserve_frames{0};   // arenas are carved for at least this many frames per group
    int sync_waiters = 0;                    // (mu) synchronous callers waiting for ANY slot (or all of them): the slot threads leave the submit queue alone meanwhile
    std::atomic<int> lab_force_generic{0};   // testing aid (fslic_hip_lab_force_generic): the brute-force gather kernels instead of the tiled ones
};

// Testing aid, exported but deliberately NOT part of include/fslic_hip.h: every following group of the Slic variant on this engine
// runs the brute-force gather kernel (k_assign_generic) instead of the tiled one -- the tests' independent cross-check.
extern "C" int fslic_hip_lab_force_generic(fslic_engine* e, int on);

namespace fslic {

// ---- tables.cpp ----
int configure_spatial(Slot& s, int S, const fslic_params* p);
int configure_patchf(Slot& s, int S, const fslic_params* p);

// ---- arena.cpp ----
void free_slot(Slot& s);
int prepare(fslic_engine* e, Slot& s, int H, int W, int K, int S, int G);
int ensure_prepared(fslic_engine* e, Slot& s, int H, int W, int K, int S, int G);
int prepare_lsc(Slot& s, int H, int W, int K, int S, int G, float compactness);
int ensure_cand_capacity(Slot& s, size_t M);

// ---- group.cpp ----
int validate(const fslic_params* p, int H, int W, int K, int& S);
int upload_ptrs(Slot& s, int n, const uint8_t* const* d_rgb, uint16_t* const* d_out);
CcaDev cca_view(const Slot& s, int i0, const uint16_t* d_in0, size_t in_stride, int K, int min_threshold);
void cca_enqueue(Slot& s, const CcaDev& c, int i0, int n, const ExportDev* ex = nullptr);
int cca_finish_group(Slot& s, int first, int n, const uint16_t* d_in0, size_t in_stride, int K, int thres);
int group_begin(fslic_engine* e, Slot& s, const fslic_params* p, int H, int W, int K, int n,
                const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_out);
int group_finish(fslic_engine* e, Slot& s);
std::string make_timing_report(const Slot& s);
void set_thread_timing_report(const std::string& json);
const std::string& thread_timing_report();

// ---- pipeline.cpp ----
// Synchronous calls take a slot for their duration (they wait while every slot is taken; a slot with an uncollected
// asynchronous group is never handed out).  `want` >= 0 asks for that slot.
int acquire_slot(fslic_engine* e, int want, int& slot);
int acquire_all_slots(fslic_engine* e);
void release_slot(fslic_engine* e, int slot);
void release_all_slots(fslic_engine* e);
void stop_slot_thread(fslic_engine* e, Slot& s);
struct SlotLease {               // RAII: one slot for the length of a scope
    fslic_engine* e;
    int slot = -1;
    explicit SlotLease(fslic_engine* e_) : e(e_) {}
    int take(int want = -1) { return acquire_slot(e, want, slot); }
    ~SlotLease() { if (slot >= 0) release_slot(e, slot); }
};

}  // namespace fslic
