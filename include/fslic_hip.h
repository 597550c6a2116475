/* fslic_hip.h -- C ABI of the MI355X (gfx950) SLIC superpixel engine.
 *
 * This is the drop-in boundary for ONE hot path of Algy/fast-slic: fast_slic.Slic.iterate()
 * (RGB->CIELAB prepass, subsampled assign/update loop, full assign, connectivity/min-size pass).
 * Every entry point names the reference interface it replaces; paths are relative to the
 * reference repository root.  Plain C types only: no C++ or torch types cross this boundary.
 *
 * Threading (cfast_slic.pyx:188-193 releases the GIL around the calls replaced here, and the reference's
 * per-call Context makes concurrent iterate() calls on different models legal): every entry point may be
 * called from any thread at any time.  A call needs one of the engine's n_slots slots (a HIP stream and
 * its buffers); synchronous calls take a free slot for their duration and wait while all are taken, so
 * up to n_slots calls run concurrently on one engine and further ones queue.  A slot that owns a group
 * submitted with fslic_hip_submit_group is handed to nobody until fslic_hip_wait_group has collected it.
 * The timing report and the error message are per calling thread.  The library never calls into Python.
 *
 * All functions returning int return 0 on success and a non-zero FSLIC_E_* code on failure;
 * fslic_hip_last_error() then returns a thread-local human-readable message.  No exception
 * crosses the ABI (the reference maps C++ exceptions through Cython `except +`, cfast_slic.pxd:47,51).
 */
#ifndef FSLIC_HIP_H
#define FSLIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSLIC_HIP_ARCH_NAME "hip/gfx950"   /* new entry for archtbl[], src/context-impl.cpp:15-24 */

enum {
    FSLIC_OK = 0,
    FSLIC_E_INVALID = 1,       /* bad argument (maps to ValueError in the binding)             */
    FSLIC_E_UNSUPPORTED = 2,   /* option outside the implemented surface (NotImplementedError) */
    FSLIC_E_HIP = 3,           /* HIP runtime failure: no device, OOM, launch error            */
    FSLIC_E_INTERNAL = 4
};

/* Bit-compatible with `Cluster`, src/fast-slic-common.h:10-23 (32 bytes; Cython mirror
 * cfast_slic.pxd:8-18).  r,g,b hold L,a,b when convert_to_lab is set.  `a` is never written. */
typedef struct fslic_cluster {
    float y, x, r, g, b, a;
    uint16_t number;
    uint8_t is_active;
    uint8_t is_updatable;
    uint32_t num_members;
} fslic_cluster;

/* The public configuration fields of BaseContext (src/context.h:26-36) as set by
 * SlicModel.iterate (cfast_slic.pyx:179-187), plus max_iter (context.iterate's argument). */
typedef struct fslic_params {
    int32_t max_iter;              /* BaseContext::iterate(assignment, max_iter), src/context.cpp:108 */
    float compactness;             /* src/context.h:28 */
    float min_size_factor;         /* src/context.h:29 */
    int32_t subsample_stride;      /* subsample_stride_config, src/context.h:26 (>= 1) */
    int32_t convert_to_lab;        /* src/context.h:30 */
    int32_t manhattan_spatial_dist;/* src/context.h:35; 0 (hypotf patch) takes the generic kernel */
    int32_t preemptive;            /* src/context.h:32, src/preemptive.h; every variant (BaseContext<DistType>::iterate, src/context.cpp:152-181) */
    float preemptive_thres;        /* src/context.h:33; ignored while preemptive == 0 */
    int32_t num_threads;           /* src/context.h:27; ignored on the GPU */
    int32_t debug_mode;            /* src/context.h:36; ignored (no recorder report) */
    int32_t abi;                   /* must be FSLIC_PARAMS_ABI: the layout of this struct has changed between library versions (a testing
                                    * flag lived in this slot in 0.2 and `variant` in the header of 0.2+); a caller built against
                                    * another layout is refused with FSLIC_E_INVALID instead of being misread */
    int32_t variant;               /* FSLIC_VARIANT_SLIC: Context (src/context.h:127); FSLIC_VARIANT_LSC: ContextLSC
                                    * (src/lsc.h:6-26), picked by cfast_slic.pyx:199-216 from real_dist_type */
    int32_t reserved[4];           /* must be 0 */
} fslic_params;
#define FSLIC_PARAMS_ABI 0x46533033  /* "FS03" */

enum {
    FSLIC_VARIANT_SLIC = 0,          /* Context, uint16 distances */
    FSLIC_VARIANT_LSC = 1,           /* ContextLSC, real_dist_type 'lsc' */
    FSLIC_VARIANT_REALDIST = 2,      /* ContextRealDist, real_dist_type 'standard' (src/context.h:100-103) */
    FSLIC_VARIANT_REALDIST_L2 = 3,   /* ContextRealDistL2, real_dist_type 'l2' (src/context.h:105-111) */
    FSLIC_VARIANT_REALDIST_NOQ = 4   /* ContextRealDistNoQ, real_dist_type 'noq' (src/context.h:113-125): float centroids */
};

typedef struct fslic_engine fslic_engine;

/* Number of visible HIP devices (0 when none / runtime unusable). */
int fslic_hip_device_count(void);

/* Create / destroy an engine bound to one GPU.  The engine owns every device and pinned-host
 * buffer and caches them across calls keyed by (H, W, K); the reference re-allocates per call
 * (BaseContext ctor, src/context.h:59-66, invoked from cfast_slic.pyx:171-177).
 * n_slots >= 1 is the number of frames that may be in flight at once (one HIP stream each). */
int fslic_hip_create(int device, int n_slots, fslic_engine** out);
void fslic_hip_destroy(fslic_engine* e);

/* Replaces BaseContext::initialize_clusters (src/context.cpp:42-97) as called by
 * SlicModel.initialize (cfast_slic.pyx:124-147).  Pure host code, O(K); `rgb` is a host pointer
 * to C-contiguous uint8[H][W][3]. */
int fslic_hip_initialize_clusters(int H, int W, int K, const uint8_t* rgb, fslic_cluster* clusters);

/* Replaces Context construction + initialize_state() + iterate() + delete for the integer SLIC path
 * (cfast_slic.pyx:171-197 -> src/context.cpp:108-197).
 *   rgb      : HOST pointer, borrowed, C-contiguous uint8[H][W][3]
 *   clusters : HOST pointer to K clusters, updated in place exactly like the reference does
 *   labels   : HOST pointer to uint16[H][W]; all H*W entries are written (0xFFFF possible);
 *              the caller applies astype(int16) / 0xFFFF -> -1 (cfast_slic.pyx:258-260). */
int fslic_hip_iterate(fslic_engine* e, const fslic_params* p, int H, int W, int K,
                      const uint8_t* rgb, fslic_cluster* clusters, uint16_t* labels);

/* Same computation with the frame already resident in HBM (NEW surface, absent in the reference;
 * SURVEY.md section 8f-4).  d_rgb / d_labels are DEVICE pointers on the engine's GPU; clusters stays a
 * host pointer (K*32 bytes).  `slot` selects the in-flight slot / stream (0 <= slot < n_slots).
 * The call returns after the slot's stream has been synchronised. */
int fslic_hip_iterate_device(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K,
                             const uint8_t* d_rgb, fslic_cluster* clusters, uint16_t* d_labels);

/* Independent frames of identical geometry (NEW surface; BASELINE.json config 4).  The frames are cut into
 * groups of up to 8 (FSLIC_GROUP, at most 16); every kernel launch of a group covers all of its frames (frame =
 * last grid dimension), groups rotate over the engine's slots (one stream each).
 * rgb[i] / labels[i] are host pointers when device_ptrs == 0 and device pointers otherwise;
 * clusters[i] are host pointers, each K clusters. */
int fslic_hip_iterate_batch(fslic_engine* e, const fslic_params* p, int H, int W, int K, int n_frames,
                            const uint8_t* const* rgb, fslic_cluster* const* clusters,
                            uint16_t* const* labels, int device_ptrs);

/* Asynchronous form for pipelining groups across calls (NEW surface): submit enqueues ONE group of n_frames
 * (1..16) device-resident frames on `slot` and returns without waiting; wait blocks until that group is complete,
 * serves its host-side steps and writes the clusters back.  The pointer arrays are copied; clusters[i] and the
 * device buffers must stay valid until the wait.  With two slots the host work of one group (cluster upload and
 * write-back) overlaps the kernels of the other. */
int fslic_hip_submit_group(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K, int n_frames,
                           const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels);
int fslic_hip_wait_group(fslic_engine* e, int slot);
/* Non-blocking: 1 when the group submitted on `slot` is complete (fslic_hip_wait_group will not block) or the slot is
 * idle, 0 while it is in flight, -1 on a bad argument.  Lets a caller with several slots collect groups in completion
 * order instead of submission order (groups with a top-K tie take longer than the others). */
int fslic_hip_group_done(fslic_engine* e, int slot);

/* The same, with the engine choosing the slot (NEW surface): submit puts one submission of n_frames (1..16)
 * device-resident frames into the engine's queue and returns; it blocks only while the queue is full (two submissions
 * per slot).  The slot threads serve the queue; the clusters of a submission are written back when its group is complete.
 * drain waits for everything submitted so far and returns the first error of any group since the previous drain (a
 * submit after a failed group returns that error as well); the optional outputs receive the totals over those groups:
 * device time (ms, HIP events per group), groups (launch groups: with batching on, fewer than submissions), frames, and
 * the number of frames whose top-K step fell back to the host.  A video pipeline calls submit once per batch of frames
 * and drain at the end.  (A slot that is serving the queue counts as owning an unfinished group: drain before addressing
 * slots by number with fslic_hip_submit_group.) */
int fslic_hip_pipeline_submit(fslic_engine* e, const fslic_params* p, int H, int W, int K, int n_frames,
                              const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels);
int fslic_hip_pipeline_drain(fslic_engine* e, double* device_ms, long long* groups, long long* frames,
                             long long* host_topk_frames);
/* Dynamic batching of the pipeline (off by default: max_frames_per_group = 0): a slot thread that finds several
 * submissions of identical geometry and options waiting serves them as ONE group of up to max_frames_per_group (<= 16)
 * frames -- every launch of the group then covers all of them.  Frames are independent, so results do not change; a
 * caller that submits faster than the device finishes gets fewer, fuller launches (1280x720: +10 %).  Arenas are carved
 * for max_frames_per_group frames from then on.  Call it before the first submit. */
int fslic_hip_pipeline_batching(fslic_engine* e, int max_frames_per_group);

/* Stage entry points (used by the parity tests; each mirrors one stage of iterate()). */

/* rgb_to_cielab, src/cielab.h:337-353 (convert != 0) or the raw copy of src/context.cpp:117-127
 * (convert == 0).  Host pointers; lab4 is uint8[H][W][4] = (L, a, b, 0). */
int fslic_hip_rgb_to_lab(fslic_engine* e, int H, int W, const uint8_t* rgb, int convert, uint8_t* lab4);

/* cca::ConnectivityEnforcer(labels,H,W,K,min_threshold).execute(labels), src/cca.cpp:178-265, as
 * exposed by cfast_slic.enforce_connectivity (cfast_slic.pyx:371-396).  In place, host pointer. */
int fslic_hip_enforce_connectivity(fslic_engine* e, uint16_t* labels, int H, int W, int K, int min_threshold);

/* ---- Superpixel-graph utilities on a finished label map (src/fast-slic.h:13-17, src/fast-slic.cpp; reached from
 * SlicModel.get_connectivity / get_knn_connectivity / get_mask_density / broadcast_density_to_mask,
 * cfast_slic.pyx:262-324).  `labels` (H*W uint16), `mask` (H*W uint8) and `result` (H*W uint8) may be host pointers or
 * device pointers (a label map left in HBM by fslic_hip_iterate_device needs no copy); the per-cluster arrays are host
 * arrays.  The reference returns a heap-allocated Connectivity (src/fast-slic-common.h:25-29); here the same content is
 * written into caller arrays: num_neighbors[K] and neighbors[K][stride] (unused entries 0). */

/* fast_slic_get_connectivity(H, W, K, assignment), src/fast-slic.cpp:16-78: for every cluster its adjacent clusters
 * (right / down / down-right scan) in the order the raster scan meets them, at most 12 per cluster with the
 * reference's order-dependent cut-off.  stride = 12. */
int fslic_hip_get_connectivity(fslic_engine* e, int H, int W, int K, const uint16_t* labels,
                               int* num_neighbors, uint32_t* neighbors);

/* fast_slic_knn_connectivity(H, W, K, clusters, num_neighbors), src/fast-slic.cpp:80-130.  Host only (O(K)); the
 * neighbour order is the reference's heap order.  stride = n_neighbors. */
int fslic_hip_knn_connectivity(int H, int W, int K, const fslic_cluster* clusters, size_t n_neighbors,
                               int* num_neighbors, uint32_t* neighbors);

/* fast_slic_get_mask_density(H, W, K, clusters, assignment, mask, cluster_densities), src/fast-slic.cpp:141-154:
 * densities[k] = min(255, sum of mask over the pixels labelled k / max(num_members[k], 1)). */
int fslic_hip_get_mask_density(fslic_engine* e, int H, int W, int K, const fslic_cluster* clusters,
                               const uint16_t* labels, const uint8_t* mask, uint8_t* densities);

/* fast_slic_cluster_density_to_mask(H, W, K, clusters, assignment, cluster_densities, result), src/fast-slic.cpp:156-168
 * (the reference's `clusters` argument is unused and not taken): result[p] = densities[labels[p]], 0 for labels >= K. */
int fslic_hip_cluster_density_to_mask(fslic_engine* e, int H, int W, int K, const uint16_t* labels,
                                      const uint8_t* densities, uint8_t* result);

/* Labels after full_assign and before the connectivity pass (BaseContext::assignment,
 * src/context.cpp:182-190) of the last fslic_hip_iterate*() call on `slot`; host pointer, H*W. */
int fslic_hip_last_prelabels(fslic_engine* e, int slot, uint16_t* prelabels);

/* Replaces BaseContext::get_timing_report (src/context.h:74): JSON with the fstimer schema
 * {"name","duration"(us),"children"} (src/timer.cpp:4-18), durations from HIP events of the CALLING THREAD's last
 * fslic_hip_iterate* / fslic_hip_wait_group (thread-local like the reference's timer, src/timer.cpp:45).  Pointer
 * valid until that thread's next call. */
const char* fslic_hip_last_timing_report(fslic_engine* e);

/* Device time (ms, HIP events on the slot's stream) of the last frame GROUP on `slot`: whole pipeline, and
 * the full-assign launch alone (the roofline kernel; it covers every frame of the group).  */
int fslic_hip_last_device_times(fslic_engine* e, int slot, float* total_ms, float* full_assign_ms);

/* Per-launch timing of the subsampled (fused assign + update) launches: when on, groups submitted afterwards bracket
 * each of those launches with HIP events on the slot's stream (about 1 us of stream time per launch, hence opt-in).
 * last_assign_loop: sum of their durations (ms), pixels they visited over all frames, number of launches timed
 * (0 when the group was submitted with timing off). */
int fslic_hip_set_launch_timing(fslic_engine* e, int on);
int fslic_hip_last_assign_loop(fslic_engine* e, int slot, float* sum_ms, double* visited_px, int* launches);

/* Number of frames in the last group on `slot` (1 for iterate / iterate_device). */
int fslic_hip_last_group_frames(fslic_engine* e, int slot);

/* How the last group on `slot` reached the GPU: 0 = operations enqueued one by one, 1 = recorded as a hipGraph during
 * this call and launched, 2 = replay of a recorded graph.  (The engine records a launch sequence the second time it
 * sees the same geometry / options / group size; the environment variable FSLIC_GRAPH=0, read when the library is
 * loaded, disables that.)  Testing / diagnostics aid. */
int fslic_hip_last_launch_mode(fslic_engine* e, int slot);

/* Which kernel family served the last call on `slot`: 0 = tiled (LDS candidate lists), 1 = generic. */
int fslic_hip_last_path(fslic_engine* e, int slot);

/* Frames of the last group on `slot` whose top-K step of the connectivity pass ran on the host (more candidate
 * components than the device sorts in one block; an area tie at the cut is resolved on the device). */
int fslic_hip_last_host_topk_frames(fslic_engine* e, int slot);

/* Frames redone on `slot` since it was created because a visited pixel that no cluster window covered kept its label
 * (src/context.cpp:138-145 resets the assignment plane once per call, not per iteration) while the cluster pass
 * (src/context.cpp:356-373) ran fused into the assign kernel: such a frame is recomputed with the separate cluster pass,
 * the results are those of the reference either way.  Cost: such a frame is computed TWICE (about 2 x the latency of that call);
 * it only arises in groups small enough to take the fused cluster pass (one or two 1280x720 frames per call), with warm starts whose
 * centres have drifted far enough apart to leave visited pixels uncovered.  A caller that sees this counter grow on a stream can set
 * FSLIC_FUSEBIN=0 (the separate cluster pass everywhere: +8 % latency per one-frame call, no redo).  Diagnostics aid. */
int fslic_hip_separate_pass_redos(fslic_engine* e, int slot);

/* Measurement aid (no counterpart in the reference): bytes read + written per second, in GB/s, of a plain streaming copy of
 * `bytes` on the engine's GPU (best of `reps` launches, HIP events) -- the measured HBM rate that bench.py prints next to the 8 TB/s of
 * the specification.  Allocates and frees 2 x `bytes` of device memory. */
int fslic_hip_copy_bandwidth(fslic_engine* e, size_t bytes, int reps, double* gb_per_s);

const char* fslic_hip_last_error(void);
const char* fslic_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FSLIC_HIP_H */
