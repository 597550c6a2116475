// group.cpp -- one launch group of the engine: a group is up to 16 frames of identical geometry that share every
// kernel launch (frame = last grid dimension).  Staging of the caller's clusters, the launch sequence of iterate()
// (src/context.cpp:108-197) enqueued operation by operation or replayed as one hipGraph, completion, the rare host
// steps, and the write-back of the Cluster[K] blocks.  Part of the host engine, see engine_internal.h.
#include "engine_internal.h"

#include <time.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

namespace fslic {

double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int validate(const fslic_params* p, int H, int W, int K, int& S) {
    if (!p) return fail(FSLIC_E_INVALID, "params is NULL");
    static_assert(sizeof(fslic_params) == 64 && offsetof(fslic_params, abi) == 40 && offsetof(fslic_params, variant) == 44, "fslic_params layout (include/fslic_hip.h)");
    if (p->abi != FSLIC_PARAMS_ABI) return fail(FSLIC_E_INVALID, "fslic_params.abi != FSLIC_PARAMS_ABI: the caller was built against another layout of the struct (include/fslic_hip.h)");
    if (p->reserved[0] | p->reserved[1] | p->reserved[2] | p->reserved[3]) return fail(FSLIC_E_INVALID, "fslic_params.reserved must be 0");
    if (H <= 0 || W <= 0) return fail(FSLIC_E_INVALID, "H and W must be positive");
    if (H > 32767 || W > 32767) return fail(FSLIC_E_INVALID, "H and W must fit int16 (src/context.cpp:262)");
    if ((long long)H * W >= (1ll << 31)) return fail(FSLIC_E_INVALID, "H*W must fit int32");
    if (K <= 0 || K >= 65534) return fail(FSLIC_E_INVALID, "num_components must be in [1, 65533] (cfast_slic.pyx:24-27)");
    if (p->subsample_stride < 1 || p->subsample_stride > 255) return fail(FSLIC_E_INVALID, "subsample_stride must be in [1, 255]");
    if (p->max_iter < 0) return fail(FSLIC_E_INVALID, "max_iter must be >= 0");
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
    // src/cca.cpp:225-228 calls std::partial_sort; on an area tie at the cut the survivors depend on the library's heap
    // mechanics, so the same call is made here (the device path replays libstdc++'s __heap_select, see k_cca_select;
    // this host path only serves frames with more candidates than the device sorts).
    auto by_area = [&](uint32_t a, uint32_t b) { return area[a] > area[b]; };
    std::partial_sort(order.begin(), order.begin() + K, order.end(), by_area);
    std::vector<uint8_t> kept(M, 0);
    for (int i = 0; i < K; i++) kept[order[i]] = 1;
    uint32_t n = 0;
    for (uint32_t i = 0; i < M; i++)
        if (kept[i]) { keep_leader[n] = leader[i]; keep_label[n] = (uint16_t)n; n++; }
    n_keep = n;
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
    FrameDev::adv(c.lid, d); FrameDev::adv(c.nrec, d); FrameDev::adv(c.narea, d);
    FrameDev::adv(c.nfinal, d); FrameDev::adv(c.vedge, d); FrameDev::adv(c.roots, d);
    FrameDev::adv(c.counters, d);
    FrameDev::adv(c.cand_leader, d); FrameDev::adv(c.cand_area, d);
    c.outs = reinterpret_cast<uint16_t* const*>(s.d_ptrs + kMaxGroup + i0);
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
void cca_enqueue(Slot& s, const CcaDev& c, int i0, int n, const ExportDev* ex) {
    (void)i0;
    launch_cca_phase1(c, n, s.st);
    launch_cca_select(c, n, s.st);
    launch_cca_phase2(c, n, s.st, ex);
}

// After the stream has been synchronised and h_misc refreshed.  The select kernel resolves the top-K step on the
// device, including an area tie at the cut (it replays libstdc++'s __heap_select).  Status 2 (more candidates than the
// device sorts): gather + sort on the host, frame by frame, then chain + relabel again for that frame.
int cca_finish_group(Slot& s, int first, int n, const uint16_t* d_in0, size_t in_stride, int K, int thres) {
    int n_slow = 0;
    for (int i = first; i < first + n; i++) n_slow += s.h_misc[16 * (size_t)i + 4 + 3] == 2;
    if (n_slow == 0) return FSLIC_OK;
    for (int i = first; i < first + n && n_slow; i++) {
        uint32_t* hm = s.h_misc + 16 * (size_t)i;
        if (hm[4 + 3] != 2) continue;
        const CcaDev c1 = cca_view(s, i, d_in0, in_stride, K, thres);
        int32_t* d_dl = s.at(s.c.cand_leader, i);      // dense (appended by k_cca_compress), unordered
        uint32_t* d_da = s.at(s.c.cand_area, i);
        const uint32_t M = hm[4 + 1];
        int rc = ensure_cand_capacity(s, std::max<size_t>(M, 1) * 2);
        if (rc) return rc;
        launch_cca_gather_area(c1, s.st);
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
        // the kept list goes into the first halves of the pinned buffers (their content was moved to the second halves above)
        // and from there to the device on the slot's OWN stream: a copy on the default stream fails while another slot
        // of the engine was capturing its graph (stream capture is gone since: launch.h)
        int32_t* kl = s.h_cand_leader;
        uint16_t* kb = reinterpret_cast<uint16_t*>(s.h_cand_area);
        uint32_t n_keep = 0;
        select_kept_sorted(sl, sa, M, K, kl, kb, n_keep);
        int32_t* d_kl = s.at(s.d_keep_leader, i);
        uint16_t* d_kb = s.at(s.d_keep_label, i);
        if (n_keep) {
            HIPCHK(hipMemcpyAsync(d_kl, kl, sizeof(int32_t) * n_keep, hipMemcpyHostToDevice, s.st));
            HIPCHK(hipMemcpyAsync(d_kb, kb, sizeof(uint16_t) * n_keep, hipMemcpyHostToDevice, s.st));
        }
        launch_cca_set_kept(c1, d_kl, d_kb, (int)n_keep, s.st);
        launch_cca_phase2(c1, 1, s.st);
    }
    HIPCHK(hipStreamSynchronize(s.st));
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

// Whether a group's cluster passes run inside its assign launches (the per-cluster fused form, assign.hip: k_assign_bin): the integer
// SLIC path on the block kernel, launches that do not fill the chip (FSLIC_FUSEBIN=1, the default) or all of them (=2).
static bool cluster_pass_fused(const FrameDev& f, const fslic_params& p, int n, int W, int H, bool generic, bool separate_cluster_pass) {
    const int stride = p.subsample_stride;
    const int assign_blocks = n * ((W + kTileW - 1) / kTileW) * (((H + stride - 1) / stride + 31) / 32);
#if defined(FSLIC_LAB) && FSLIC_LAB == 3
    // lab build 3 turns st_stream into plain stores; the fused cluster pass hands partial sums from block to block through
    // written-through (sc1) stores and agent-scope loads and would read stale lines of another XCD's L2: never fused there
    const bool fuse_wanted = false;
    (void)assign_blocks;
#else
    const bool fuse_wanted = knobs().fuse_bin == 2 || (knobs().fuse_bin == 1 && assign_blocks <= 640);
#endif
    const bool plain_slic = p.variant == FSLIC_VARIANT_SLIC && p.preemptive == 0;
    // (H < stride: the passes of the residues >= H visit no row at all and launch nothing -- src/context.cpp:158-175 still runs update()
    // for them: every cluster's num_members becomes 0, positions stay; a cluster pass that rides on the assign blocks would not happen.
    // Found on a 71x1 frame, tests/fuzz_parity.py FUZZ_SHAPES=tiny seed 5 case 3100.)
    const bool every_pass_has_rows = H >= stride;
    return fuse_wanted && !separate_cluster_pass && plain_slic && !generic && every_pass_has_rows && assign_fuses_cluster_pass(f, stride);
}

// Device half: everything of iterate() for frames [i0, i0+n) of the slot's current group, enqueued asynchronously on
// the slot's stream: LAB, the first cluster pass (reads the staged centres), the assign/update loop, full assign,
// connectivity, export of the cluster state.  Pure
// stream work with no per-call values in any launch argument (caller pointers sit in the device pointer table, bin
// generations come from device memory), so the same sequence can be recorded once as a graph and replayed (launch_group).
int enqueue_frames(fslic_engine* e, Slot& s, int i0, int n, bool generic, bool timed, bool separate_cluster_pass = false) {
    const fslic_params* p = &s.p;
    const int K = s.K, S = s.S;
    FrameDev f = s.f;
    f.rgbs = nullptr;
    f.select(i0);                                  // host-side: pointers of frame i0
    f.rgbs = reinterpret_cast<const uint8_t* const*>(s.d_ptrs + i0);
    const size_t fb = s.frame_bytes;

    // copies the graph recorder has no node for (launch.h): such a group is always enqueued directly
    // the (2S+1)^2 patch of the generic kernel: also where a preemptive Slic group falls back to that kernel for its subsampled passes
    // (geometries outside the block kernel: launch_assign_pre returns false)
    const bool need_patch = generic || (p->preemptive != 0 && p->variant == FSLIC_VARIANT_SLIC && !assign_fuses_cluster_pass(f, p->subsample_stride));
    if ((p->preemptive != 0 || p->variant == FSLIC_VARIANT_REALDIST_NOQ || (need_patch && !s.sp_patch_uploaded)) && recording_unsupported()) return FSLIC_OK;
    if (need_patch && !s.sp_patch_uploaded) {
        HIPCHK(hipMemcpyAsync(s.d_patch, s.h_patch, (size_t)(2 * S + 1) * (2 * S + 1) * sizeof(uint16_t), hipMemcpyHostToDevice, s.st));
        s.sp_patch_uploaded = true;
    }
    // cluster centres: staged by stage_group() into the pinned block the first cluster pass reads (no copy command)
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
    // (the integer SLIC path on the block / 32-bit kernels resets the assignment plane lazily, FrameDev::fv_mod: the LAB kernel then writes
    // 4 B/px instead of 6; every other path -- variants, generic kernel, preemptive mode -- has the plane filled here)
    // ... except where the cluster pass is fused into the assign kernel (launches that do not fill the chip: one or two frames per group):
    // there every block drains its label stores before it counts itself in, and stores into a plane the LAB kernel has just written
    // find their lines in the memory-side cache -- without the fill each of the ten launches of a one-frame group takes 0.35 us longer
    // (8.54 against 8.18 us, same-box A/B, profiles/r06_experiments.txt) for the 0.6 us the LAB kernel saves once.
    const bool lazy_labels = p->variant == FSLIC_VARIANT_SLIC && !generic && p->preemptive == 0 &&
                             !cluster_pass_fused(f, *p, n, s.W, s.H, generic, separate_cluster_pass);
    launch_rgb_to_lab(f, n, p->convert_to_lab, e->tables, !lazy_labels, s.st, s.gen_step,
                      s.at(s.zero_block, i0), s.zero_bytes, f.cl_n, 2 * (size_t)K, s.h_cl + (size_t)i0 * 4 * K, 4 * (size_t)K);      // (+ the staged centres -> cl_yx)
    if (timed) HIPCHK(hipEventRecord(s.ev[1], s.st));
    f.gen_off = 1;
    PassGeom pg{};
    // positions: the uploaded centres are in s.f.cl_yx; every cluster pass reads one buffer and writes the other
    uint32_t* const yx_up = f.cl_yx;
    uint32_t* const yx_a = yx_up + 6 * (size_t)K;       // this frame's d_yx_alt[0]
    uint32_t* const yx_b = yx_up + 7 * (size_t)K;       // this frame's d_yx_alt[1]
    f.cl_yx = yx_up; f.cl_yx_out = yx_a;
    const int stride = p->subsample_stride;
    const bool lsc = p->variant == FSLIC_VARIANT_LSC;
    launch_bin_clusters(f, n, 0, 0, 0, pg, s.st);      // (the centres are in cl_yx: copied from the host's staging block by the LAB kernel)
    uint32_t* yx_cur = yx_a;
    int rem = 0;
    LscDev l = s.l;
    if (lsc) {
        l.select(i0);
        HIPCHK(clear_rows(s.lsc_zero + (size_t)i0 * s.lsc_frame_bytes, s.lsc_frame_bytes, s.lsc_zero_bytes, (size_t)n, s.st));
        launch_lsc_prepare(f, l, n, s.st);                        // before_iteration, src/lsc.cpp:12-15
    }
    const bool rd = p->variant == FSLIC_VARIANT_REALDIST || p->variant == FSLIC_VARIANT_REALDIST_L2 || noq;
    const bool rd_l2 = p->variant == FSLIC_VARIANT_REALDIST_L2;
    float noq_coef = 1.0f / ((float)S / p->compactness);        // src/context.cpp:463-464
    noq_coef *= (1 << (p->convert_to_lab ? 1 : 0));
    const bool noq_manhattan = p->manhattan_spatial_dist != 0;
    const float pre_l1_thres = std::max(roundf(2 * S * p->preemptive_thres), 1.0f);      // src/preemptive.h:129
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
        if (pre) launch_preempt_update(f, n, (it + 1) & 1, it & 1, false, pre_l1_thres, s.st);      // update() + set_new_clusters(), src/context.cpp:356-387
        else launch_bin_clusters(f, n, 1, (it + 1) & 1, it & 1, pg, s.st);
        yx_cur = f.cl_yx_out;
        rem = (rem + 1) % stride;
    }
    for (int it = 0; lsc && it < p->max_iter; it++) {           // src/context.cpp:158-175 with ContextLSC's hooks
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        f.cl_yx = yx_cur; f.cl_yx_out = (yx_cur == yx_a) ? yx_b : yx_a;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        launch_lsc_assign(f, l, n, rem, stride, it & 1, it & 1, !pre, s.st);      // (preemptive: the assignment alone, the sums follow)
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
        if (pre) {
            // update() under the activity state the pass started with, set_new_clusters(), then after_update() under the new one
            // (src/context.cpp:301-387, src/preemptive.h:114-178, src/lsc.cpp:226-307): lsc.hip, "preemptive mode"
            launch_lsc_pre_sums(f, n, rem, stride, it & 1, s.st);
            f.gen_off++;
            launch_preempt_update(f, n, (it + 1) & 1, it & 1, false, pre_l1_thres, s.st);
            launch_lsc_pre_feats(f, l, n, rem, stride, s.st);
        } else {
            f.gen_off++;
            pg = PassGeom{};                                          // use_slots = 0: everything is in FrameDev::sums
            launch_bin_clusters_lsc(f, l, n, (it + 1) & 1, it & 1, s.st);   // update() sums + after_update(), re-binning: one launch
        }
        yx_cur = f.cl_yx_out;
        rem = (rem + 1) % stride;
    }
    // The cluster pass between two assign passes (centroids from the sums, re-binning) is done by the assign kernel itself where
    // the block kernel applies: its last block to deliver sums of a cluster finalises it (launch_assign_fused_bin), so the loop is
    // max_iter launches instead of 2 * max_iter.  Positions then live in ONE buffer (yx_a, where the first cluster pass put them).
    // Measured (profiles/r03_fused_cluster_pass.txt): one 1280x720 frame per group 168 instead of 183 us (the loop is ten launches
    // of 9.9 us instead of twenty of 6.7 + 4.9), eight frames 315 instead of 333 us -- but the pipelined rate of 16-frame groups
    // drops from 51.8 to 45.3 GP/s: the finaliser is a chain of dependent memory round trips at the end of every block and
    // vector work a 1600-thread cluster pass does once.  So: fused where a launch does not fill the chip (latency-bound: the
    // reference's own one-frame-per-call pattern), separate where it does.  FSLIC_FUSEBIN=0 / 2: never / always.
    const bool fusebin = cluster_pass_fused(f, *p, n, s.W, s.H, generic, separate_cluster_pass);
    int full_obuf = -1;
    for (int it = 0; fusebin && it < p->max_iter; it++) {
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        f.cl_yx = yx_a; f.cl_yx_out = yx_a;
        f.fv_mod = lazy_labels && it < stride ? 1 : 0; f.fv_from = 0;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        launch_assign_fused_bin(f, n, rem, stride, it, s.st);
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
        f.gen_off++;
        yx_cur = yx_a;
        rem = (rem + 1) % stride;
        full_obuf = (it + 1) % 3;
    }
    if (fusebin && full_obuf < 0) full_obuf = 0;
    for (int it = 0; !fusebin && !lsc && !rd && it < p->max_iter; it++) {   // src/context.cpp:158-175
        pg = PassGeom{};
        const bool tme = timed && s.launch_timing && it < Slot::kMaxTimedIters;
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it], s.st));
        // (preemptive mode: the block kernel on the bins of the active clusters where the geometry takes it, round 5; until then the
        // generic kernel served every preemptive frame)
        f.fv_mod = lazy_labels && it < stride ? 1 : 0; f.fv_from = 0;
        if (generic || (pre && !launch_assign_pre(f, n, rem, stride, it & 1, it & 1, s.st))) launch_assign_generic(f, n, rem, stride, it & 1, it & 1, true, s.st);
        else if (!pre) pg = launch_assign(f, n, rem, stride, it & 1, it & 1, true, s.st);
        if (tme) HIPCHK(hipEventRecord(s.ev_it[2 * it + 1], s.st));
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
    if (timed) HIPCHK(hipEventRecord(s.ev[2], s.st));
    // (lazy reset: the rows of the residues no subsampled pass came to are first looked at by the full pass)
    f.fv_mod = lazy_labels && p->max_iter < stride ? stride : 0; f.fv_from = p->max_iter < stride ? std::max(p->max_iter, 0) : 0;
    f.fv_rcp = f.fv_mod > 1 ? 0xFFFFFFFFu / (uint32_t)f.fv_mod : 0u;
    // full_assign: stride 1, rem 0, no update afterwards (src/context.cpp:246-256)
    if (lsc) { f.cl_yx = yx_cur; launch_lsc_assign(f, l, n, 0, 1, full_buf, 0, false, s.st); }
    else if (noq) { f.cl_yx = yx_cur; launch_noq_assign(f, noq_coef, noq_manhattan, n, 0, 1, full_buf, 0, false, s.st); }
    else if (rd) { f.cl_yx = yx_cur; launch_rd_assign(f, s.d_patchf, rd_l2, n, 0, 1, full_buf, 0, false, s.st); }
    else if (generic) launch_assign_generic(f, n, 0, 1, full_buf, 0, false, s.st);
    else (void)launch_assign(f, n, 0, 1, full_buf, 0, false, s.st, full_obuf);      // (full_buf: the bins of all clusters; max_iter & 1 unless preemptive)
    if (timed) HIPCHK(hipEventRecord(s.ev[3], s.st));
    // enforce_connectivity, src/context.cpp:14-20
    const int thres = (int)round((double)(S * S) * (double)p->min_size_factor);
    const CcaDev c = cca_view(s, i0, s.f.labels, fb, K, thres);
    // cluster state, overflow flag and connectivity status back to the host: the last kernel of the connectivity pass
    // writes them into the slot's pinned blocks (three 2-D copy commands per group before: blit kernels with their own
    // barriers).  The status words are final once the select kernel has run, the cluster state since the loop ended.
    ExportDev ex{};
    ex.frame_bytes = fb; ex.K = K; ex.yx_cur = yx_cur; ex.lab_n_moved = yx_up + K; ex.misc0 = s.at(s.d_misc, i0);
    ex.h_cl = s.h_cl + (size_t)i0 * 4 * K; ex.h_stride = 4 * (size_t)K; ex.h_misc = s.h_misc + 16 * (size_t)i0;
    cca_enqueue(s, c, i0, n, &ex);
    if (timed) HIPCHK(hipEventRecord(s.ev[4], s.st));
    if (pre) HIPCHK(hipMemcpy2DAsync(s.h_upd + (size_t)i0 * K, sizeof(uint32_t) * (size_t)K, f.cl_upd, fb,
                                     sizeof(uint32_t) * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    if (noq) HIPCHK(hipMemcpy2DAsync(s.h_clf + (size_t)i0 * 8 * K, sizeof(float) * 8 * (size_t)K, f.cl_f, fb,
                                     sizeof(float) * 8 * (size_t)K, (size_t)n, hipMemcpyDeviceToHost, s.st));
    return FSLIC_OK;
}

GraphRecorder*& current_recorder() {
    static thread_local GraphRecorder* rec = nullptr;
    return rec;
}

// The device half of a group start, replayed from a graph when this exact sequence has been seen before.
int launch_group(fslic_engine* e, Slot& s, int n) {
    s.last_launch_mode = 0;
    {   // Bin slots are validated by a u32 generation stamp; a group uses max_iter + 3 consecutive values above the base
        // (device word *d_gen, mirrored here).  Long before it could wrap, start over: stamps and base back to zero.
        // The base moves past the stamps of the PREVIOUS group at the start of this one (a group of fewer iterations after one
        // of more would otherwise meet the earlier group's stamps inside its own range and take stale bin slots for live).
        const uint64_t span = (uint64_t)s.p.max_iter + 3u;
        if (s.gen_host + 2 * (s.gen_span_prev + span) >= 0xC0000000ull) {
            HIPCHK(hipMemsetAsync(s.d_gen, 0, 256, s.st));
            HIPCHK(hipMemset2DAsync(s.stamp_block, s.frame_bytes, 0, s.stamp_bytes, (size_t)s.cap_frames, s.st));
            s.gen_host = 0;
        }
        s.gen_step = s.gen_span_prev ? s.gen_span_prev : (uint32_t)span;     // (first group of the slot: any step does)
        s.gen_span_prev = (uint32_t)span;
        s.gen_host += s.gen_step;
    }
    // Per-launch timing wants real event records between the launches; events replayed inside a graph keep the time
    // stamps of an earlier execution.  Timed groups therefore take the direct path.
    if (!knobs().use_graphs || s.launch_timing) return enqueue_frames(e, s, 0, n, s.generic, true);
    // everything a launch argument can depend on
    std::vector<unsigned char> key;
    auto put = [&](const void* p, size_t bytes) { const unsigned char* b = (const unsigned char*)p; key.insert(key.end(), b, b + bytes); };
    const int hdr[8] = {s.H, s.W, s.K, s.S, n, s.generic ? 1 : 0, (int)s.gen_step, s.sp_patch_uploaded ? 1 : 0};
    const void* const ptrs[9] = {s.h_cl, s.h_misc, s.d_ptrs, s.h_patch, s.d_gen, s.lsc_zero, s.d_patchf, s.h_clf, s.h_upd};   // baked into copy nodes / arguments
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
        // second sighting: the same calls with a recorder current on this thread build the graph, node by node (launch.h)
        GraphRecorder rec;
        if (hipGraphCreate(&rec.graph, 0) != hipSuccess) { ge->failed = true; (void)hipGetLastError(); return enqueue_frames(e, s, 0, n, s.generic, true); }
        current_recorder() = &rec;
        const int rc = enqueue_frames(e, s, 0, n, s.generic, false);     // no event records inside the graph
        current_recorder() = nullptr;
        if (rc != FSLIC_OK || rec.err != hipSuccess || rec.nodes == 0 || hipGraphInstantiate(&ge->exec, rec.graph, nullptr, nullptr, 0) != hipSuccess) {
            hipGraphDestroy(rec.graph);
            ge->exec = nullptr; ge->failed = true;
            (void)hipGetLastError();
            if (rc != FSLIC_OK) return rc;
            return enqueue_frames(e, s, 0, n, s.generic, true);
        }
        ge->graph = rec.graph;
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
    rc = ensure_prepared(e, s, H, W, K, S, n);           // (carves for at least the engine's batching reserve)
    if (rc) return rc;
    const int cap_frames = s.cap_frames;
    s.H = H; s.W = W; s.K = K; s.S = S; s.p = *p; s.nframes = n;
    for (int i = 0; i < n; i++) { s.clusters[i] = clusters[i]; s.d_rgb[i] = d_rgb[i]; s.d_out[i] = d_out[i]; }
    rc = configure_spatial(s, S, p);
    if (rc) return rc;
    if (p->variant == FSLIC_VARIANT_LSC) {
        rc = prepare_lsc(s, H, W, K, S, cap_frames, p->compactness);
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
    s.generic = p->variant == FSLIC_VARIANT_SLIC && (e->lab_force_generic.load() != 0 || !s.sp_tiled_ok);
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
    if (knobs().host_timing) fprintf(stderr, "[fslic host] group_begin n=%d: %.1f us | slot %d begin %.1f .. %.1f\n", n, now_us() - t_begin, (int)(&s - e->slots.data()), t_begin, now_us());
    return FSLIC_OK;
}

// Wait for the group, serve the rare host-side steps frame by frame (candidate-list overflow -> generic redo, more
// connectivity candidates than the device sorts -> std::partial_sort on the host), then write the cluster state back.
// Completion of everything enqueued on the slot's stream, for a slot's OWN host thread (the submit / drain pipeline): hipStreamSynchronize
// spins, and six slot threads spinning cost a rank 0.77 ms of CPU per 0.153 ms step (five cores; forty on an 8-GPU node).  Here the
// thread sleeps through the first half of the wait -- half of the SHORTEST of its last four waits, so that a group that finishes
// sooner than its predecessors (the pipeline draining) is not slept past -- and then looks at an event behind the stream's work every
// ten microseconds (the slot threads run with a timer slack of 1 us, pipeline.cpp).  A synchronous caller's thread keeps the spin:
// its timer slack is its own, and one frame per call is a latency measurement.
static int nap_wait(Slot& s) {
    const double t0 = now_us();
    HIPCHK(hipEventRecord(s.ev[5], s.st));
    // The history belongs to ONE workload: after a switch from long groups (16 x 4K: ~10 ms) to short ones (0.15 ms) the first short
    // groups would otherwise sleep 5, 2.5, 1.25 .. ms before the minimum converged.
    const unsigned long long key = ((unsigned long long)(uint32_t)s.H << 48) ^ ((unsigned long long)(uint32_t)s.W << 32) ^ ((unsigned long long)(uint32_t)s.K << 16) ^
                                   ((unsigned long long)(uint32_t)s.nframes << 8) ^ ((unsigned long long)(uint32_t)s.p.max_iter << 4) ^ (unsigned long long)(uint32_t)s.p.variant ^
                                   ((unsigned long long)(uint32_t)s.p.preemptive << 60) ^ ((unsigned long long)(uint32_t)s.p.subsample_stride << 56);
    if (key != s.recent_wait_key) {
        s.recent_wait_key = key;
        for (double& w : s.recent_wait_us) w = 1e30;
    }
    double shortest = 1e30;
    for (double w : s.recent_wait_us) shortest = std::min(shortest, w);
    auto nap = [](double us) { struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)(us * 1000.0); nanosleep(&ts, nullptr); };
    if (shortest < 1e29 && shortest * 0.5 > 30.0) nap(std::min(shortest * 0.5, 20000.0));
    for (;;) {
        const hipError_t q = hipEventQuery(s.ev[5]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(FSLIC_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        nap(10.0);
    }
    s.recent_wait_us[s.recent_wait_i++ & 3] = now_us() - t0;
    return FSLIC_OK;
}

int group_finish(fslic_engine* e, Slot& s) {
    const double t_begin = now_us();
    if (s.nap_wait) { const int rc = nap_wait(s); if (rc) return rc; }
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
        n_host_topk += s.h_misc[16 * (size_t)i + 4 + 3] == 2;
        // Flags of the tiled kernels.  Bit 1: a visited pixel that no window covered kept its label while the cluster pass was
        // fused into the assign kernel (its sums reached the owner through global atomics that nothing orders before the owner's
        // finaliser): that frame alone is redone with the separate cluster pass.  Bit 0: a block's candidate list overflowed: that
        // frame alone is redone with the generic kernel.
        for (int attempt = 0; attempt < 2 && !s.generic && s.h_misc[16 * (size_t)i + 2] != 0; attempt++) {
            const bool overflow = (s.h_misc[16 * (size_t)i + 2] & 1u) != 0;
            if (knobs().host_timing) fprintf(stderr, "[fslic host] frame %d redone with %s\n", i, overflow ? "the generic kernel (candidate-list overflow)" : "the separate cluster pass (stale pixel)");
            stage_group(e, s, i, 1, false);
            s.gen_step = s.gen_span_prev;                 // past the stamps of the pass just made (same span)
            s.gen_host += s.gen_step;
            int rc = enqueue_frames(e, s, i, 1, overflow, false, true);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(s.st));
            if (overflow) {
                if (s.h_misc[16 * (size_t)i + 2] != 0) return fail(FSLIC_E_INTERNAL, "generic path reported a candidate overflow");
                s.last_path = 1;
            } else {
                __atomic_fetch_add(&s.n_separate_redo, 1, __ATOMIC_RELAXED);      // (read by fslic_hip_separate_pass_redos without the engine lock)
            }
        }
        if (!s.generic && s.h_misc[16 * (size_t)i + 2] != 0) return fail(FSLIC_E_INTERNAL, "a redone frame still carries a kernel flag");
    }
    {
        int rc = cca_finish_group(s, 0, n, s.f.labels, s.frame_bytes, K, thres);
        if (rc) return rc;
    }
    s.n_host_topk = n_host_topk;
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
    if (knobs().host_timing)
        fprintf(stderr, "[fslic host] group_finish n=%d slot %d at %.1f .. %.1f: sync wait %.1f us, events %.1f us, host top-K (%d frames) %.1f us, write-back %.1f us (device %.1f us)\n",
                n, (int)(&s - e->slots.data()), t_begin, now_us(), t_sync - t_begin, t_ev - t_sync, n_host_topk, t_cca - t_ev, now_us() - t_cca, s.total_ms * 1e3);
    return FSLIC_OK;
}

// BaseContext::get_timing_report (src/context.h:74) in the fstimer schema (src/timer.cpp:4-18), durations from the
// HIP events of the slot's last group.  The report belongs to the calling thread, like the reference's thread_local
// timer (src/timer.cpp:45): concurrent iterate() calls on different models do not overwrite each other's report.
std::string make_timing_report(const Slot& s) {
    char buf[1024];
    auto us = [](float ms) { return (long long)(ms * 1000.0f); };
    snprintf(buf, sizeof buf,
             "{\"name\": \"iterate\", \"duration\": %lld, \"children\": ["
             "{\"name\": \"cielab_conversion\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"assign_update_loop\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"full_assign\", \"duration\": %lld, \"children\": []},"
             "{\"name\": \"enforce_connectivity\", \"duration\": %lld, \"children\": []}]}",
             us(s.total_ms), us(s.lab_ms), us(s.loop_ms), us(s.fa_ms), us(s.cca_ms));
    return buf;
}

namespace {
thread_local std::string t_timing;
}
void set_thread_timing_report(const std::string& json) { t_timing = json; }
const std::string& thread_timing_report() { return t_timing; }

}  // namespace fslic
