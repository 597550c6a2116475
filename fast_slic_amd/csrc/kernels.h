// kernels.h -- launch interface between the host engine (engine.cpp) and the gfx950 kernels
// (lab / clusters / assign / cca / lsc / realdist / preempt .hip).  Internal to the library; the public boundary is
// include/fslic_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef FSLIC_EXP
#define FSLIC_EXP 0      // experiment builds: make VAR=<name> DEFS=-DFSLIC_EXP=<bits> (assign.hip)
#endif

namespace fslic {

// ---- compile-time geometry ---------------------------------------------------------------
constexpr int kTileW = 64;          // assign tile width  = one wavefront of columns
constexpr int kWavesPerBlock = 4;   // assign block = 4 wavefronts stacked vertically
constexpr int kRowsPerWave = 8;     // visited rows per wavefront
constexpr int kCellCap = 8;         // cluster records per coarse cell before spilling to overflow
constexpr int kAccSlots = 16;       // candidates per wavefront with LDS-privatised centroid sums
constexpr int kAccCopies = 8;       // privatised copies per (slot, quantity)
#ifndef FSLIC_BLK_COPIES
#define FSLIC_BLK_COPIES 4
#endif
constexpr int kBlkCopies = FSLIC_BLK_COPIES;       // block-level kernel: privatised copies per (candidate, quantity)
constexpr uint32_t kHuge = 0x40000000u;   // out-of-window marker added to the distance key
constexpr int kCcaTile = 64;        // connectivity tile width  (one wavefront of columns)
constexpr int kCcaTileH = 32;       // connectivity tile height (4 wavefronts x 8 rows), one block of 256 threads
constexpr int kCcaTilePx = kCcaTile * kCcaTileH;
#ifndef FSLIC_MAX_GROUP
#define FSLIC_MAX_GROUP 16
#endif
constexpr int kMaxGroup = FSLIC_MAX_GROUP;       // frames of one group share every launch (frame index = a grid dimension)

// One cluster as the kernels see it (16 B, one ds_read_b128 / global_load_dwordx4).
//   yx  = y << 16 | x          (integer centre, src/context.cpp:262 casts to int16)
//   lab = L | a << 8 | b << 16 (integer colour, always <= 255 per channel)
//   tag = phase << 16 | k      visit order of the reference's scatter (src/context.cpp:214-242)
struct ClusterRec {
    uint32_t yx, lab, tag, pad;
};

// Device state of a GROUP of frames with identical geometry (H, W, K, S).  Every per-frame buffer lives in
// a per-frame sub-arena of identical layout; the pointers below are those of the group's first frame and
// select(z) advances them to frame z (z = the launch's last grid dimension).  Input frames and output label
// maps are the caller's buffers and come as per-frame pointers.
// Out-of-window marker of the packed kernel's spatial tables: any value v with 1024 <= v and v + 765 < 65536; the key
// (v + colour) * 64 + idx then saturates to 0xffff = never wins.
constexpr uint16_t kTabOut = 0x4000;

struct FrameDev {
    size_t frame_bytes;  // distance between the sub-arenas of consecutive frames
    const uint8_t* const* rgbs;       // DEVICE array: the group's input frames (N*3 bytes each), caller's buffers
    // geometry
    int H, W, N, K, S;
    int T;               // 2S + 32, phase cell side (src/context.cpp:213)
    uint32_t T_rcp;      // floor((2^32 - 1) / T): the cluster pass divides by T without a division (device_common.h, div_by)
    int cs_shift;        // coarse bin side = 1 << cs_shift
    int ncx, ncy, ncell;
    // Generation stamp of the bin pass that filled cell_items[buf] (ClusterRec::pad) = *gen_base + gen_off.  The base
    // lives in device memory and is advanced by the first kernel of every group, so that the launch sequence of a
    // group carries no per-call value and can be replayed as a hipGraph.
    uint32_t* gen_base;
    uint32_t gen_off;
    __device__ __forceinline__ uint32_t generation() const { return *gen_base + gen_off; }
    uint32_t M, C;       // spatial fixed point: patch[d] == (d * M + C) >> 16 for d in [0, 2S]
    // image planes
    const uint8_t* rgb;  // N*3 (set by select)
    uint32_t* lab;       // N, L|a<<8|b<<16
    uint16_t* labels;    // N, pre-connectivity assignment (persists across iterations)
    // cluster state (SoA)
    uint32_t* cl_yx;     // K  positions the cluster pass READS (pre-update; also the uploaded centres)
    uint32_t* cl_yx_out; // K  positions the cluster pass WRITES (the two alternate per iteration: a pass may still be
                         //    reading a neighbour's old position while its owner stores the new one)
    uint32_t* cl_lab;    // K
    uint32_t* cl_tag;    // K  phase << 16 | k of the current centres
    uint32_t* cl_n;      // K  num_members of the last update
    uint32_t* cl_moved;  // K  1 once an update with n > 0 has moved the cluster
    uint32_t* cl_shadow; // K  1 while the cluster is kept out of the bins (identical to an earlier-ordered one)
    // preemptive mode (src/preemptive.h); cell_act == nullptr otherwise
    uint32_t* cl_upd;    // K  is_updatable counters (cooldown 2)
    uint32_t* cl_act;    // K  is_active
    uint32_t* cell_act;  // pre_cw * pre_ch  active (2S x 2S) cells
    uint32_t* pre_flags; // [0] all clusters active
    int pre_cw, pre_ch;
    float* cl_f;         // K * 8  non-quantised centroids (y, x, r, g, b, pad) of the 'noq' variant; nullptr otherwise
                         //    (cl_yx then holds their integer parts: bins, phases and windows work on those)
    uint32_t* sums[2];   // K*6 each: count, sum y, sum x, sum L, sum a, sum b; pass `it` accumulates into [it&1]
                         // (global atomics: generic / 32-bit kernels, and stale-label pixels of the block kernel)
    uint32_t* cpart;     // K * cpart_ns * 6: per-(cluster, overlapping assign block) partial sums of the block
                         // kernel, written with plain stores (one writer per entry), gathered by the cluster pass
    int cpart_ns;        // entries per cluster (allocation): nbx x (a power of two >= the block rows a window overlaps in any pass geometry)
    int nbx;             // assign blocks (64 px wide) a cluster window can overlap horizontally, 2S/64 + 2, rounded up to a power of two.
                         // The entry of (cluster, block (by, bx)) is (by mod NY) * nbx + (bx mod nbx), NY = the pass's own power-of-two
                         // row count (device_common.h, cpart_rows_mask): a window overlaps at most NY x nbx CONSECUTIVE blocks, so
                         // the residues are distinct, and which entries are live follows from the cluster's position alone
    // coarse bins, double buffered by iteration parity
    uint32_t* cell_cnt[2];     // ncell
    ClusterRec* cell_items[2]; // ncell * kCellCap
    uint32_t* ovf_cnt[3];      // 1    (the third buffer serves the fused cluster pass only: read / append / clear rotate over three)
    ClusterRec* ovf_items[3];  // K
    uint32_t* cl_arrive;       // K  fused cluster pass: assign blocks that have delivered their partial sums of the cluster in this pass
    uint32_t* err_flag;        // != 0: a block overflowed its LDS candidate list -> host reruns generic
    // LDS spatial table of the tiled kernel (lut_words == 0: fixed-point mode, no table)
    const uint32_t* lut;       // lut_words entries: patch value << 16, or kHuge for out-of-window indices
    int lut_words, lut_row, lut_oow;   // index = ux' + uy' * lut_row, ux'/uy' = |d| if <= S else lut_oow
    // packed 16-bit kernel: spatial table of patch values, kTabOut outside the window (two layouts, see tab_vmode)
    const uint16_t* tab;       // tab_words * 2 entries (tab_words == 0: packed kernel not applicable)
    int tab_words, tab_nrpad, tab_dyoff;
    int tab_vmode, tab_xout;   // tab_vmode: `tab` is the row-vector form of the V-shaped 1-D table V[t] = patch(|t - tab_dyoff|) (kTabOut beyond 2S)
                               // (Manhattan patch): 16-byte entry t = V[t + k*stride], k = 0..7; lanes outside the x window use
                               // |dx| = tab_xout.  `tab` serves the full (stride 1) pass, `tabs` the subsampled passes
    const uint16_t* tabs;
    int tabs_words, tabs_dyoff, tabs_nrpad;
    // tab_vmode == 1 only: `tab` is built for 32 rows per wavefront when that fits (tab_rows32); `tabv16` is the same stride-1 table
    // for at most 16 rows (tabv16_words == 0: `tab` itself is the 16-row table)
    const uint16_t* tabv16;
    int tabv16_words, tabv16_dyoff, tab_rows32;
    // tab_vmode == 0 only: the subsampled stride's 2-D table built for 16 rows per wavefront (tabs16_words == 0: too large to keep six
    // blocks per CU resident, the 16-row fused pass is then not used)
    const uint16_t* tabs16;
    int tabs16_words, tabs16_dyoff, tabs16_nrpad;
    // tab_vmode == 0: pair-packed column-major patch copy, dword [ux'][j] = (e(j), e(j+stride)) with e = patch or kTabOut
    // outside the window, row pitch tab_nrpad dwords, j = dy + tab_dyoff
    // generic-path tables
    const uint16_t* patch2d;   // (2S+1)^2, src/context.cpp:22-40
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
    uint32_t* lab_stamps;      // lab build 4: phase time stamps of the block assign kernel (the idle candidate-area array of the connectivity pass)
#endif
    // (at the END of the struct: the kernels' scalar loads of every older member keep their offsets and their grouping)
    // Lazy reset of the assignment plane (round 6).  The reference fills it with 0xFFFF once per call (src/context.cpp:138-145); a row's
    // labels are first looked at by the first pass that visits the row, so that pass can treat "no window covers this pixel" as
    // "unlabelled" and store the 0xFFFF itself -- the LAB kernel then moves 7 B/px instead of 9.  A launch's visited row y is a FIRST visit
    // iff fv_mod != 0 && y % fv_mod >= fv_from: subsampled pass it < stride: (1, 0), i.e. all of its rows; later subsampled passes:
    // (0, -); the full pass: (stride, min(max_iter, stride)), i.e. the residues no subsampled pass came to.  fv_mod == 0 everywhere
    // (and the plane filled by the LAB kernel) for the kernels that do not implement it: the variants, the generic kernel, preemptive mode.
    int fv_mod, fv_from;
    uint32_t fv_rcp;     // floor((2^32 - 1) / fv_mod) (fv_mod > 1): y % fv_mod without a division (the test sits in the kernels' rare paths, but a 32-bit division is ~40 instructions each time it is inlined)
    __device__ __forceinline__ bool first_visit(int y) const {
        if (fv_mod <= 1) return fv_mod == 1;
        uint32_t q = __umulhi((uint32_t)y, fv_rcp), rest = (uint32_t)y - q * (uint32_t)fv_mod;      // (device_common.h, div_by: the estimate is at most two short)
        if (rest >= (uint32_t)fv_mod) rest -= (uint32_t)fv_mod;
        if (rest >= (uint32_t)fv_mod) rest -= (uint32_t)fv_mod;
        return (int)rest >= fv_from;
    }

    // byte offset by POINTER arithmetic (not through an integer): the compiler then still knows the pointer came from a
    // kernel argument, i.e. points to global memory, and emits global_load / global_store instead of FLAT instructions
    template <class T> static __host__ __device__ __forceinline__ void adv(T*& p, size_t d) {
        p = reinterpret_cast<T*>(const_cast<char*>(reinterpret_cast<const volatile char*>(p)) + d);
    }
    // advance every per-frame pointer to frame z of the group (tables lut / tab / patch2d are shared)
    __host__ __device__ __forceinline__ void select(int z) {
        // NB: no run-time index into a member array here; that would force this by-value kernel argument into
        // scratch memory.  The caller's per-frame pointers therefore live in a small device array.
        rgb = rgbs ? rgbs[z] : nullptr;
        const size_t d = (size_t)z * frame_bytes;
        adv(lab, d); adv(labels, d);
        adv(cl_yx, d); adv(cl_yx_out, d); adv(cl_lab, d); adv(cl_tag, d); adv(cl_n, d); adv(cl_moved, d); adv(cl_shadow, d);
        if (cl_f) adv(cl_f, d);
        if (cell_act) { adv(cl_upd, d); adv(cl_act, d); adv(cell_act, d); adv(pre_flags, d); }
        adv(sums[0], d); adv(sums[1], d); adv(cpart, d);
        adv(cell_cnt[0], d); adv(cell_cnt[1], d); adv(cell_items[0], d); adv(cell_items[1], d);
        adv(ovf_cnt[0], d); adv(ovf_cnt[1], d); adv(ovf_cnt[2], d); adv(ovf_items[0], d); adv(ovf_items[1], d); adv(ovf_items[2], d);
        adv(cl_arrive, d);
        adv(err_flag, d);
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
        adv(lab_stamps, d);
#endif
    }
};

// Geometry of the assign pass whose partial sums a cluster pass gathers (mode 1).
struct PassGeom {
    int rem, stride, Hv;   // visited rows y = rem + stride * v, v in [0, Hv)
    int BH;                // visited rows per assign block
    int use_slots;         // 1: the block kernel wrote FrameDev::cpart; 0: everything is in FrameDev::sums
};

struct CcaDev {
    size_t frame_bytes;
    uint16_t* const* outs;       // DEVICE array: per-frame output label maps (caller's buffers; may alias `in`)
    size_t in_stride;            // bytes between the input label maps of consecutive frames
    int H, W, N, K;
    int min_threshold;
    const uint16_t* in;   // N labels
    uint16_t* out;        // N labels (set by select)
    int tiles_x;          // 64-pixel tile columns per row of tiles
    // A pixel's component: the node tile * kCcaTilePx + lid[p] (tiles in raster order).  Node arrays hold NP = number of
    // tiles * kCcaTilePx entries, used from the front of every tile's range.
    uint16_t* lid;        // N: number of the pixel's component inside its tile
    uint4* nrec;          // NP node records: x parent (union-find over the nodes; after k_cca_compress the global root, directly),
                          //   y raster index of the node's first pixel (a global root's = the component's leader, src/cca.h:36-57),
                          //   z the node's own (tile-local) area, w node of the pixel left of (image column 0: above) the first pixel, or ~0
    uint32_t* narea;      // NP: component areas, valid at global roots once k_cca_compress has run
    uint16_t* nfinal;     // NP: final labels (global roots: select / chain; every node after k_cca_chain)
    uint32_t* vedge;      // tiles * 2 * kCcaTileH: (label | number << 16) of every tile's first and last column, row by row
    uint32_t* roots;      // the frame's nodes, dense: counters[0] of them (capacity NP)
    uint32_t* counters;   // [0] number of nodes (append cursor of k_cca_local), [1] M = number of candidates (append cursor of k_cca_compress), [2] M again, [3] select status
    int32_t* cand_leader; // dense candidate list (components with area >= threshold): their leaders (capacity NP)
    uint32_t* cand_area;  //   same capacity: the host top-K path gathers the candidates' areas into it
    __host__ __device__ __forceinline__ void select(int z) {
        out = outs[z];
        FrameDev::adv(in, (size_t)z * in_stride);
        const size_t d = (size_t)z * frame_bytes;
        FrameDev::adv(lid, d); FrameDev::adv(nrec, d); FrameDev::adv(narea, d);
        FrameDev::adv(nfinal, d); FrameDev::adv(vedge, d); FrameDev::adv(roots, d);
        FrameDev::adv(counters, d);
        FrameDev::adv(cand_leader, d); FrameDev::adv(cand_area, d);
    }
};

// Device state of the LSC variant (src/lsc.cpp): the feature means of every frame of the group, the feature-space centroids and
// the tables the kernels rebuild a pixel's ten features from (no feature planes: lsc.hip).  Same per-frame sub-arena scheme as
// FrameDev (own arena, own frame stride).
constexpr int kLscFeat = 10;        // l1, l2, a1, a2, b1, b2, x1, x2, y1, y2 (src/lsc.h:12)
constexpr int kLscCfPitch = 16;     // floats per centroid record (10 features, padded to 64 bytes)
constexpr int kLscExPitch = 12;     // int64 per cluster of the stale-pixel accumulators (10 features, weight, pad)
// fixed point of the global accumulators of the fused update (LscDev::extra, 64-bit: a cluster that owned a whole 4K frame
// would reach 4.4e14 in a feature sum and 7.8e14 in the weight sum)
constexpr float kLscFixF = 1048576.0f, kLscFixW = 4096.0f;
struct LscDev {
    size_t frame_bytes;
    uint32_t* hist;      // 768: histograms of the L, a, b planes (feature means of the colour half, src/lsc.cpp:143-149)
    float* means;        // 16 (10 used)
    float* cfeat;        // K * kLscCfPitch
    long long* extra;    // K * kLscExPitch: w*F and w of visited pixels that kept a label whose cluster window has left them
    // shared tables (host-built with the reference's expressions, src/lsc.cpp:76-101)
    const float* lut_color;   // [4][256]: L cos, L sin, a/b cos, a/b sin
    const float* lut_x;       // [2][W]: cos, sin
    const float* lut_y;       // [2][H]
    // fixed point of a block's accumulators in LDS (lsc.hip): the largest powers of two that keep 1024 pixels' sum of a feature
    // inside 31 bits and of the weight inside 32; shifts from there to the 2^-20 / 2^-12 of the global accumulators
    float blk_fix_f, blk_fix_w;
    int blk_shift_f, blk_shift_w;
    // The weight w = sum_q mean_q * F_q holds four spatial products of either sign (each >= -C_spatial^2), so with a large compactness
    // it can be negative -- and it shares a 64-bit block accumulator word with the position sums, where a negative addend would borrow
    // from them.  Every pixel's weight is therefore stored with the bias ceil(4 C_spatial^2) (an integer: exact in the block's fixed
    // point), which the block's flush takes out again as count * bias.
    float blk_bias_w;
    __host__ __device__ __forceinline__ void select(int z) {
        const size_t d = (size_t)z * frame_bytes;
        FrameDev::adv(hist, d); FrameDev::adv(means, d);
        FrameDev::adv(cfeat, d); FrameDev::adv(extra, d);
    }
};

struct LabTables {
    const uint16_t* gamma;   // 256   src/cielab.h:298-299
    const uint16_t* labtbl;  // 8193  src/cielab.h:302-304
    int cb[9];               //       src/cielab.h:300-301
};

// ---- launches (all asynchronous on `st`) ----------------------------------------------------
// labels_init != nullptr: also fill the N-entry label plane with 0xFFFF (src/context.cpp:138-145)
// gen_step != 0: first kernel of a group, advances FrameDev::gen_base by gen_step (more than the group's bin passes)
// zero_a / zero_b: optional per-frame regions (frame 0's address; zero_a_bytes a multiple of 16, zero_b_words dwords) the
// kernel clears as well -- the accumulators, bin counters and flags a group starts from (no separate memset launches)
// stage_yx != nullptr: the group's cluster centres as the host staged them (pinned memory, stage_stride words per frame): blocks of their own
// copy them into FrameDev::cl_yx beside the conversion, and the first cluster pass reads device memory
void launch_rgb_to_lab(const FrameDev& f, int nframes, int convert, const LabTables& t, bool init_labels, hipStream_t st, uint32_t gen_step = 0,
                       void* zero_a = nullptr, size_t zero_a_bytes = 0, uint32_t* zero_b = nullptr, size_t zero_b_words = 0,
                       const uint32_t* stage_yx = nullptr, size_t stage_stride = 0);
// mode 0: resample colours at the current centres (src/context.cpp:128-135) and bin;
// mode 1: integer centroid update from `sums` (src/context.cpp:356-373), zero sums, and bin.
// mode 0 with src_yx != nullptr: the centres come from there (the engine's pinned staging block, read in place; frame z at
// src_yx + z * src_stride words) instead of FrameDev::cl_yx.
void launch_bin_clusters(const FrameDev& f, int nframes, int mode, int buf, int sbuf, const PassGeom& pg, hipStream_t st,
                         const uint32_t* src_yx = nullptr, size_t src_stride = 0);
// The cluster pass of the LSC variant: the same launch also turns the fused update's accumulators into the feature-space centroids
// (src/lsc.cpp:256-269, 309-316) in blocks of its own -- the two passes are independent, and as two launches they were two latency-bound
// kernels back to back in every iteration.  Mode 1 with PassGeom{} (everything is in FrameDev::sums).
void launch_bin_clusters_lsc(const FrameDev& f, const LscDev& l, int nframes, int buf, int sbuf, hipStream_t st);

// One assign pass over rows == rem (mod stride); fuse_update also accumulates the centroid sums
// of src/context.cpp:301-354 for the same rows.  Returns the geometry the following cluster pass needs.
// obuf: which of the three spill lists the pass reads (buf for the two-buffer rotation of the separate cluster pass).
PassGeom launch_assign(const FrameDev& f, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st, int obuf = -1);
// The same with the cluster pass fused in (src/context.cpp:356-373 + the binning of src/context.cpp:214-221): the LAST assign block
// that delivers partial sums of a cluster finalises it (integer means, re-binning for the next pass) -- no k_bin_clusters<1> launch
// between two passes.  Pass `it` reads bins [it & 1] / spill list [it % 3], appends to [(it+1) & 1] / [(it+1) % 3].  Only for
// geometries assign_fuses_cluster_pass() accepts; positions live in FrameDev::cl_yx (updated in place).
bool assign_fuses_cluster_pass(const FrameDev& f, int stride);
void launch_assign_fused_bin(const FrameDev& f, int nframes, int rem, int stride, int it, hipStream_t st);
void launch_assign_generic(const FrameDev& f, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st);
// preemptive mode (src/preemptive.h), Slic variant: the fused pass of the block kernel on the bins of the ACTIVE clusters, sums of the
// pixels of active cells into FrameDev::sums (global atomics per block and cluster); false: geometry outside the block kernel
bool launch_assign_pre(const FrameDev& f, int nframes, int rem, int stride, int buf, int sbuf, hipStream_t st);
// LSC variant (lsc.hip).  prepare: feature means + seed centroids (src/lsc.cpp:12-15), after LAB and the mode-0 cluster pass.
void launch_lsc_prepare(const FrameDev& f, const LscDev& l, int nframes, hipStream_t st);
// 10-D assignment of rows == rem (mod stride) (src/lsc.cpp:197-224); stale_sums: an update follows
void launch_lsc_assign(const FrameDev& f, const LscDev& l, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st);
// per-cluster gather of the integer sums (into FrameDev::sums) and the weighted feature means (src/lsc.cpp:226-307)
// LSC in preemptive mode (lsc.hip): the integer sums of update() over the visited pixels of the active cells (before
// launch_preempt_update); the weighted feature sums and the centroids of after_update() under the NEW activity state (after it)
void launch_lsc_pre_sums(const FrameDev& f, int nframes, int rem, int stride, int sbuf, hipStream_t st);
void launch_lsc_pre_feats(const FrameDev& f, const LscDev& l, int nframes, int rem, int stride, hipStream_t st);
// Float-distance variants (realdist.hip): patchf = (2S+1)^2 f32 spatial patch; l2: squared colour / spatial terms
void launch_rd_assign(const FrameDev& f, const float* patchf, bool l2, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st);
// integer member sums per cluster (gather over the cluster window) into FrameDev::sums, for the variants whose assign
// kernel does not accumulate them itself
// margin: the window scanned is (2(S+margin)+1)^2 (the assign kernel's out-of-window rule must use the same margin)
void launch_member_sums(const FrameDev& f, int nframes, int rem, int stride, int sbuf, int margin, hipStream_t st);
// 'noq' variant (ContextRealDistNoQ, src/context.cpp:449-499): distances from float centroids (FrameDev::cl_f)
void launch_noq_assign(const FrameDev& f, float coef, bool manhattan, int nframes, int rem, int stride, int buf, int sbuf, bool stale_sums, hipStream_t st);
// preemptive mode (preempt.hip): centroid update of the updatable clusters, activity marking, bins of the active clusters;
// rebin_all: bins of ALL clusters at their current positions (before the full assignment, src/preemptive.h:69-74)
void launch_preempt_update(const FrameDev& f, int nframes, int buf, int sbuf, bool rebin_all, float l1_thres, hipStream_t st);
// connectivity
void launch_cca_phase1(const CcaDev& c, int nframes, hipStream_t st);   // local CCL, border merge, compress, collect
// device top-K + ranking, including the area tie at the cut; counters[3] = status (see k_cca_select)
void launch_cca_select(const CcaDev& c, int nframes, hipStream_t st);
// End of a group: what the last kernel also writes into the engine's pinned host blocks (device-accessible), so that a
// group's launch sequence holds no copy command.  Pointers are frame 0's (frame z: + z * frame_bytes); frame z lands at
// h_cl + z * h_stride words: [0,K) positions, [K,4K) colour / member count / moved flag; h_misc + 16 z: the status words.
struct ExportDev {
    size_t frame_bytes;
    int K;                          // 0: nothing to export
    const uint32_t* yx_cur;         // the current position buffer
    const uint32_t* lab_n_moved;    // FrameDev::cl_lab (cl_n and cl_moved follow it, K words each)
    const uint32_t* misc0;          // the frame's 16 status words
    uint32_t* h_cl;
    size_t h_stride;
    uint32_t* h_misc;
};
void launch_cca_phase2(const CcaDev& c, int nframes, hipStream_t st, const ExportDev* ex = nullptr);   // chain resolution + relabel (+ export)
// host top-K path, one frame at a time (the CcaDev passed in is already advanced to that frame)
void launch_cca_gather_area(const CcaDev& c, hipStream_t st);   // cand_area[i] = area of candidate i (c already advanced to the frame)
void launch_cca_set_kept(const CcaDev& c, const int32_t* d_leaders, const uint16_t* d_labels, int n, hipStream_t st);

// superpixel-graph utilities on a finished label map (graph.hip; src/fast-slic.cpp)
void launch_mask_sums(const uint16_t* labels, const uint8_t* mask, uint32_t* sums, int H, int W, int K, hipStream_t st);
void launch_density_to_mask(const uint16_t* labels, const uint8_t* dens, uint8_t* out, size_t N, int K, hipStream_t st);
// tab_pair / tab_key: `cap` (a power of two) entries, all bytes 0xFF; counters: 2 words, zero ([0] pairs stored, [1] table full)
void launch_adjacent_pairs(const uint16_t* labels, int H, int W, int K, uint32_t* tab_pair, unsigned long long* tab_key,
                           uint32_t cap, uint32_t* counters, hipStream_t st);

// streaming copy of `bytes` (a multiple of 16) for the measured HBM rate bench.py prints (fslic_hip_copy_bandwidth)
void launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st);

}  // namespace fslic
