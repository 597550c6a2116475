// cca.hip -- connectivity / minimum-size pass (src/cca.cpp)
// Hand-written gfx950 (CDNA4, wave64) kernels of the SLIC hot path; see kernels.h for the launch interface and
// DESIGN.md for the design.  All arithmetic on this path is integer, so results are bit-identical to the reference.
#include "device_common.h"

namespace fslic {

// =============================================================================================
// Connectivity / min-size pass (src/cca.cpp).  GPU union-find whose roots carry the minimum raster
// index of each component, exactly the "leader" the reference ends up with (src/cca.h:36-57).
// A pixel knows its component through a 16-bit number inside its tile (the `lid` plane, 2 B per pixel); everything per
// component lives in NODE arrays indexed by tile * 2048 + that number: a tile's nodes are contiguous, so the passes over
// the components read and write dense memory (until this layout every component was addressed by the raster index of its
// first pixel in N-entry planes: one 64-byte sector fetched or rewritten per component and access).
//   k_cca_local    64x32 tiles resolved entirely in LDS (row runs by ballot, vertical unions by LDS
//                  atomicMin), writes lid[p], the tile's nodes (leader, area, parent = self) and appends them
//                  to the frame's dense node list
//   k_cca_border   unions across tile borders: the root with the larger leader is hooked under the other (CAS)
//   k_cca_compress node -> global root, areas summed into the global root; the components whose
//                  area reaches the threshold are appended to the candidate list as they cross it
//   k_cca_select   top-K by area and ranking by leader in one block, including the area tie at the cut
//   k_cca_chain    every other component takes the final label of the component left of (or
//                  above) its leader, src/cca.cpp:240-254, resolved by chasing toward smaller leaders
//   k_cca_relabel  out[p] = final[node of p] (the chain pass leaves the label at every node), and the group's
//                  results to the host
// =============================================================================================
constexpr uint32_t kNoAdopt = 0xFFFFFFFFu;      // node record, adopt: the neighbour has to be looked up by pixel
// Node record (CcaDev::nrec, 16 bytes, written by k_cca_local with ONE store): x parent, y leader, z tile-local area, w adopt
static __device__ __forceinline__ uint32_t* n_parent(const CcaDev& c, uint32_t n) { return reinterpret_cast<uint32_t*>(c.nrec + n); }
static __device__ __forceinline__ uint32_t n_leader(const CcaDev& c, uint32_t n) { return reinterpret_cast<const uint32_t*>(c.nrec + n)[1]; }
static __device__ __forceinline__ uint32_t n_local(const CcaDev& c, uint32_t n) { return reinterpret_cast<const uint32_t*>(c.nrec + n)[2]; }
static __device__ __forceinline__ uint32_t n_adopt(const CcaDev& c, uint32_t n) { return reinterpret_cast<const uint32_t*>(c.nrec + n)[3]; }

// LDS accessors.  They are relaxed workgroup-scope atomics rather than `volatile` accesses on purpose:
// volatile blocks the address-space inference and the accesses become FLAT instructions, which reach the LDS
// through the vector-memory path at global-memory latency (measured: 1.2 ms instead of tens of us for the
// tile kernel); atomics still lower to ds_read_b32 / ds_write_b32 and are re-read on every loop trip.
static __device__ __forceinline__ uint32_t lds_ld(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// find with path halving.  The halving store races with concurrent atomicMin hooks, which is safe:
// it only ever replaces a parent by one of its ancestors (cf. ECL-CC style concurrent union-find).
static __device__ __forceinline__ uint32_t lds_find(uint32_t* par, uint32_t x) {
    for (;;) {
        const uint32_t p = lds_ld(par + x);
        if (p == x) return x;
        const uint32_t gp = lds_ld(par + p);
        if (gp == p) return p;
        lds_st(par + x, gp);
        x = gp;
    }
}
static __device__ __forceinline__ void lds_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(par + a, b);
        if (old == a) return;
        a = old;
    }
}

// read-only find (no halving stores), unrolled for the flatten pass
static __device__ __forceinline__ uint32_t lds_find_ro(uint32_t* par, uint32_t x) {
    uint32_t p = lds_ld(par + x);
    while (p != x) { x = p; p = lds_ld(par + x); }
    return x;
}

// ---- wavefront-mask helpers of the tile kernel.  A lane predicate that every lane computes from ONE comparison is a 64-bit mask in
// scalar registers (v_cmp writes it), mask logic is scalar-ALU work, and a mask becomes a lane predicate again at no cost
// (inverse ballot: the mask is the condition register).  Written with bools the same logic goes through the vector ALU: a
// v_cndmask + v_cmp pair per ballot, DPP shifts of predicates, 64-bit find-first-set per lane (round 3: 921 vector instructions per
// wavefront of 512 pixels, the second consumer of vector issue slots in a group).
typedef unsigned long long u64;
// value of the lane to the left; lane 0 reads 0 / keeps `old`
static __device__ __forceinline__ uint32_t lane_left0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true); }
static __device__ __forceinline__ uint32_t lane_left_or(uint32_t old, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xF, 0xF, false); }
// number of set bits of m below this lane
static __device__ __forceinline__ uint32_t rank_in(u64 m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
// LDS word at a BYTE offset (the tile kernel's union-find names its nodes by byte offset: no shift per dependent look-up)
static __device__ __forceinline__ uint32_t* lds_at(uint32_t* base, uint32_t off) { return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(base) + off); }
static __device__ __forceinline__ void ldsb_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        // both walks to the roots in LOCKSTEP -- one LDS round trip per step for the two of them (one after the other, with path
        // halving, a union was five dependent round trips; the knock-out study of round 5 prices the unions of this kernel at a
        // fifth of its duration, profiles/r05_cca_local_knockout.txt).  No halving stores: the trees of a strip are a few levels
        // deep, and the flatten of phase C walks them once more anyway.
        for (;;) {
            const uint32_t pa = lds_ld(lds_at(par, a)), pb = lds_ld(lds_at(par, b));
            if (pa == a && pb == b) break;
            a = pa; b = pb;
        }
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(lds_at(par, a), b);
        if (old == a) return;
        a = old;
    }
}

// One block = one 64x32 tile = 4 wavefronts x 8 rows.  Lane = column.  The nodes of the tile's union-find are the horizontal RUNS:
// run j of tile row r is node r * 64 + j (named by its byte offset, (r * 64 + j) * 4), so a smaller name means an earlier first
// pixel in raster order and the root of a component (minimum by atomicMin hooks) is the run that holds its first pixel.
//  A  each wavefront resolves its own strip top-down: runs by ONE comparison per row (a pixel outside the image carries a
//     label no neighbour has, so validity never enters the predicates), the run index by mbcnt, one non-returning ds_min per
//     distinct (run, root above) pair; the (possibly stale) root of the pixel above travels in a VGPR, so a row costs one LDS
//     round trip
//  B  the three strip seams inside the tile are merged in LDS
//  C  flatten: every run -> tile-local root, run lengths added to the root's area (one LDS atomic per run end: + position of the
//     last pixel + 1 at the end, - position of the first at the start), the roots ranked by ballot
//  D  the tile-local roots become the tile's nodes (numbered, appended to the frame's node list: one returning global atomic per
//     tile), every pixel gets its root's number
#if defined(FSLIC_LAB) && FSLIC_LAB == 2      // lab build 2: time stamps of the tile kernel (100 MHz ticks since the block's start), 8 words per block at the end of the (idle) candidate array
#define LOC_STAMP(i_) do { __syncthreads(); if (threadIdx.x == 0) c.cand_leader[(size_t)c.N - 8 * (size_t)(blockIdx.z * gridDim.y + blockIdx.y + 1) + (i_)] = (int32_t)((uint32_t)__builtin_amdgcn_s_memrealtime() - loc_t); } while (0)
#else
#define LOC_STAMP(i_) do { } while (0)
#endif
// (at most 80 SGPRs: with the ~105 the compiler takes by itself only six wavefronts fit a SIMD's scalar register file
// instead of eight -- measured with per-block time stamps: 1500 blocks resident instead of 2048, 39 us instead of 2 rounds;
// 88 and 96 were measured too: 35 instead of 33.7 us per 8 x 1280x720, although they spill fewer scalars)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void k_cca_local(CcaDev c) {
    // Grid (frames, tile columns, tile rows): the frame is the FASTEST-varying part of the dispatch order: blocks that start
    // together then reserve their root ranges on different frames' counters (with the frame slowest, the first 1800 blocks of a 4K
    // group all hit frame 0's counter at once: 7 us instead of 0.8 us per block for that atomic,
    // profiles/r02_lab_tile_ccl_phases_4k.txt)
    const int tile_x = (int)blockIdx.y, tile_y = (int)blockIdx.z;
    c.select((int)blockIdx.x);
#if defined(FSLIC_LAB) && FSLIC_LAB == 2
    const uint32_t loc_t = (uint32_t)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) c.cand_leader[(size_t)c.N - 8 * (size_t)(blockIdx.z * gridDim.y + blockIdx.y + 1) + 7] = (int32_t)loc_t;      // absolute start
#endif
    constexpr int T = kCcaTile;
    constexpr int TH = kCcaTileH;
    constexpr int RW = TH / 4;
    static_assert(T == 64 && RW == 8, "lane = column, eight rows per wavefront");
    __shared__ uint32_t s_last[4][T];       // (label | node << 17) of every strip's last row (the seam merge of phase B reads it); the edge words afterwards
    __shared__ uint32_t s_par[T * TH];
    __shared__ uint32_t s_area[T * TH];
    __shared__ uint32_t s_nroots, s_base;
    if (threadIdx.x == 0) s_nroots = 0;        // (three barriers lie between this and its first use)
    const int lane = LANE();
    const int wave = (int)rfl((uint32_t)(threadIdx.x >> 6));
    const int tx0 = tile_x * T, ty0 = tile_y * TH;
    const int ly0 = wave * RW;
    // the part of the wavefront's 64 x 8 strip inside the image: the first `ncols` lanes of the first `nrows` rows
    const int ncols = min(c.W - tx0, T), nrows = min(max(c.H - (ty0 + ly0), 0), RW);
    const u64 colmask = ncols >= 64 ? ~0ull : ((1ull << ncols) - 1ull);
#define ROW_MASK(i_) ((i_) < nrows ? colmask : 0ull)

    // A pixel outside the image gets a label that differs from its left and upper neighbours' (and from every real label), so it
    // is a run, a node and a component of its own that nothing ever joins; the masks above keep it out of the outputs.
    const uint16_t* const rowp = c.in + ((size_t)(ty0 + ly0) * c.W + tx0);
    uint32_t lbl[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        lbl[i] = 0x10000u | (uint32_t)((i & 1) << 6) | (uint32_t)lane;
        if (lanes(ROW_MASK(i))) lbl[i] = (uint32_t)rowp[(size_t)i * c.W + lane];
    }
    LOC_STAMP(0);
    // ---- A ----
    // Per row: the pixels whose upper neighbour has the same label fold the (possibly stale) root carried in the VGPR into their
    // run's node with a non-returning ds_min; one dependent ds_read then yields the run's root.  One lane speaks for its
    // neighbours: consecutive pixels of a run that carry the same root above would all hit the same LDS word (same-address
    // atomics of a wavefront are served one lane after the other: a 40-pixel run was 40 serial operations on the CU's one LDS
    // pipe, which every resident block shares).
    // Where a run joins two so-far distinct components (U shapes), the union (two finds and a returning atomic: four or more
    // dependent LDS round trips, and one lane that needs it holds up the whole wavefront) is not done on the spot: the lane keeps
    // the pair and the wavefront executes the pending pairs together -- when a lane would need to keep a second one, and after
    // the last row.  The order of the unions does not matter for the result (the next row only needs SOME root of the pixel
    // above, possibly a stale one).  Row loop: one round trip per row instead of five (4.5 -> 2 us of a 13 us block life).
    uint32_t nodev[RW];
    uint32_t up_root = 0, pend_a = 0, pend_b = 0;
    u64 pending = 0;
    auto run_pending = [&]() {
#if !(FSLIC_EXP & 128)     // (knock-out build 128: no unions, neither the pending pairs of the row loop nor the strip seams)
        if (lanes(pending)) ldsb_union(s_par, pend_a, pend_b);
#endif
        pending = 0;
    };
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const uint32_t l = lbl[i];
        const u64 m = ballot(l != lane_left0(l)) | 1ull;                        // first pixels of the row's runs
        const uint32_t node = ((uint32_t)(ly0 + i) << 8) + (rank_in(m >> 1) << 2);    // (starts in lanes 1..lane = the run's index)
        nodev[i] = node;
        if (lanes(m)) { *lds_at(s_par, node) = node; *lds_at(s_area, node) = 0u; }
        if (i == RW - 1) s_last[wave][lane] = l | (node << 17);
        uint32_t cur_root = node;
        if (i > 0) {
            const u64 same = ballot(lbl[i - 1] == l);                           // same label above
            const u64 other = ballot(lane_left0(up_root) != up_root);           // the left neighbour carries another root
            const u64 rep = same & (m | other | ~(same << 1));
#if !(FSLIC_EXP & 64)      // (knock-out build 64, scripts/gpu_knock.sh: the row loop without its fold + dependent read; results wrong on purpose)
            if (lanes(rep)) __hip_atomic_fetch_min(lds_at(s_par, node), up_root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            cur_root = lds_ld(lds_at(s_par, node));
#endif
            const u64 need = rep & ballot(up_root != cur_root);
            if (need & pending) run_pending();
            const bool nd = lanes(need);
            pend_a = nd ? up_root : pend_a;
            pend_b = nd ? cur_root : pend_b;
            pending |= need;
        }
        up_root = cur_root;
    }
    run_pending();
    lds_barrier();
    LOC_STAMP(1);
    // ---- B ----
    if (wave > 0) {
        const uint32_t l = lbl[0];
        const uint32_t prev = s_last[wave - 1][lane];
        const u64 m = ballot(l != lane_left0(l)) | 1ull;
        const u64 same = ballot((prev & 0x1FFFFu) == l);
        const u64 first = same & (m | ~(same << 1));                            // one lane per (run, run above) contact
#if !(FSLIC_EXP & 128)
        if (lanes(first)) ldsb_union(s_par, nodev[0], prev >> 17);
#endif
    }
    lds_barrier();
    LOC_STAMP(2);
    // ---- C ----
    // every pixel's tile-local root: two unconditional steps (their loads are independent across the rows), then steps for
    // all rows together until none moves
    uint32_t root[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) root[i] = lds_ld(lds_at(s_par, nodev[i]));
#pragma unroll
    for (int i = 0; i < RW; ++i) root[i] = lds_ld(lds_at(s_par, root[i]));
    for (;;) {
        uint32_t nx[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) nx[i] = lds_ld(lds_at(s_par, root[i]));
        u64 moved = 0;
#pragma unroll
        for (int i = 0; i < RW; ++i) { moved |= ballot(nx[i] != root[i]); root[i] = nx[i]; }
        if (!moved) break;
    }
    // run lengths -> the root's area; and, in the same sweep, this wavefront's tile-local roots: their positions inside the
    // wavefront's share of the root list come from ballots alone (running count in a scalar register)
    uint32_t pos[RW];
    uint32_t wcount = 0;
    const uint32_t lane_p1 = (uint32_t)lane + 1u;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const u64 ok = ROW_MASK(i);
        const uint32_t l = lbl[i];
        const u64 m = ballot(l != lane_left0(l)) | 1ull;                        // (as in A: two instructions, cheaper than keeping eight masks)
        const u64 e = (m >> 1) | (1ull << 63);                                  // last pixels of the runs
        const uint32_t len_part = (lanes(e) ? lane_p1 : 0u) - (lanes(m) ? (uint32_t)lane : 0u);
#if !(FSLIC_EXP & 512)     // (knock-out build 512: no run-length atomics; every node then reports the area 2048 and no adopted neighbour, which the later passes survive)
        if (lanes((m | e) & ok)) atomicAdd(lds_at(s_area, root[i]), len_part);
#endif
        const u64 mr = m & ok & ballot(root[i] == nodev[i]);                    // the first pixel of a root run
        pos[i] = lanes(mr) ? wcount + rank_in(mr) : 0xFFFFFFFFu;
        wcount += (uint32_t)__popcll(mr);
    }
    LOC_STAMP(3);
    // ---- D ---- tile-local roots -> the frame's dense root list.  One LDS atomic per wavefront places its roots inside
    // the tile's range, ONE returning global atomic per tile reserves that range (a returning atomic per wavefront row
    // saturated a single word near 88 ops/us: 1.2 ms at 4K in the first version of this kernel).  The global stores of the
    // kernel come AFTER it: a wavefront waiting for the atomic's result would also wait for every store it has in flight
    // (the per-pixel plane is written through), and the other three wait for it at the barrier -- that wait was 8 - 14 us of a
    // 25 us block life.  A dense list lets the passes over the roots run as flat, evenly loaded grids.
    // (both returning atomics of this phase are single-lane by construction and written as instructions: for an atomic with a
    // wave-uniform operand the compiler emits its wavefront-reduction prologue -- mbcnt, first-lane test, multiply by the lane count,
    // readfirstlane -- around them: twenty vector instructions for nothing)
    uint32_t wbase = 0;
    if (wcount != 0u && lanes(1ull)) {
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(wbase) : "v"((uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)&s_nroots), "v"(wcount) : "memory");
    }
    wbase = rfl(wbase);
    lds_barrier();                                // every wavefront's share of s_nroots and every run's area are in
    LOC_STAMP(4);
    if (wave == 0 && lanes(1ull)) {
        const uint32_t nr = lds_ld(&s_nroots);
        uint32_t base = 0;
        if (nr) asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(base) : "v"(c.counters), "v"(nr) : "memory");
        lds_st(&s_base, base);
    }
    // the component numbers inside the tile: a root publishes its number where the pixels of its component look for
    // their root (s_par is dead: every wavefront has flattened, barrier above)
#pragma unroll
    for (int i = 0; i < RW; ++i)
        if (pos[i] != 0xFFFFFFFFu) lds_st(lds_at(s_par, root[i]), wbase + pos[i]);
    lds_barrier();
    LOC_STAMP(5);
    const uint32_t nbase = (uint32_t)(tile_y * (int)gridDim.y + tile_x) * (uint32_t)kCcaTilePx;
    const uint32_t gbase = s_base + wbase;
    // The outputs are organised for few instructions of either pipe: a node is ONE 16-byte record in memory; what a root lane knows
    // about its node (row and lane of the first pixel, area, the neighbour to adopt from) is STAGED in LDS as one packed word at the
    // node's position (positions are dense by construction), and whole wavefronts expand the words into records and store them
    // coalesced.  (Staging the finished 16-byte records cost 13.7 clk of the CU's LDS pipe per ds_write_b128 however few lanes
    // write -- scripts/microbench/lds_atomics.out --, a quarter of the kernel's LDS time; a 4-byte write costs a fifth of that.)
    // The sixteen edge words of the wavefront's rows are collected in LDS and leave through sixteen lanes of one store (two-lane
    // stores row by row were measured: 31.2 instead of 30.8 us per 8 x 1280x720).  The staging area is the wavefront's own strip
    // of s_area (8 rows x 64 words: one word per pixel, so every node fits; the areas of its roots are read into registers first;
    // no other wavefront's roots live in the strip).
    //   staged word: bits 0-8 row * 64 + lane of the first pixel, 9-19 area - 1, 20-31 number of the node to adopt from (0xFFF: none)
    uint32_t ar[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) ar[i] = lds_ld(lds_at(s_area, root[i]));     // (a root lane's root is its own run)
    wave_lds_sync();
    uint32_t* const stage = s_area + ly0 * T;
    uint32_t* const s_edge = &s_last[wave][lane ? 1 : 0];     // this wavefront's 2 * RW edge words: (row, side) at 2 * row + side (s_last is dead since phase B)
    const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(c.lid, 0, (int)((uint32_t)c.N * 2u), 0x00020000);
    // adopt: the node of the pixel left of a node's first pixel (in image column 0: above it), where that pixel lies in this tile
    // and this wavefront's rows: what k_cca_chain follows (src/cca.cpp:243-248) without a look-up by pixel.  Lane 0 has no left
    // neighbour in the tile: the DPP shift leaves it `old`, which is the number above (image column 0) or the "none" mark.
    uint32_t up_id = 0xFFFu;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const uint32_t y = (uint32_t)(ty0 + ly0 + i);
        const uint32_t id = lds_ld(lds_at(s_par, root[i]));
        const uint32_t left_id = lane_left_or(tile_x == 0 ? up_id : 0xFFFu, id);
        if (lanes(ROW_MASK(i))) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)id, lr, lane * 2, (int)((y * (uint32_t)c.W + (uint32_t)tx0) * 2u), 16 /* sc1 */);
        // the tile's first and last column, row by row: what the pass over the vertical tile seams reads (a column of the
        // label plane is one 64-byte sector per pixel)
        if (lanes(0x8000000000000001ull)) lds_st(s_edge + 2 * i, lbl[i] | (id << 16));
        if (pos[i] != 0xFFFFFFFFu) lds_st(stage + pos[i], (uint32_t)(i * T + lane) | ((ar[i] - 1u) << 9) | (left_id << 20));
        up_id = id;
    }
    wave_lds_sync();
    const uint32_t n0 = nbase + wbase;
    const uint32_t first_px = (uint32_t)(ty0 + ly0) * (uint32_t)c.W + (uint32_t)tx0;
    for (uint32_t j = (uint32_t)lane; j < wcount; j += 64u) {
        const uint32_t w = lds_ld(stage + j);
        const uint32_t n = n0 + j, a = ((w >> 9) & 0x7FFu) + 1u, ad = w >> 20;
        c.nrec[n] = make_uint4(n, first_px + __umul24((w >> 6) & 7u, (uint32_t)c.W) + (w & 63u), a, ad == 0xFFFu ? kNoAdopt : nbase + ad);
        c.narea[n] = a;                          // where k_cca_compress adds the other nodes' areas when this one is the global root
        c.roots[gbase + j] = n;
    }
    if (lane < 2 * RW) (c.vedge + (size_t)(nbase / (uint32_t)kCcaTilePx) * (2 * TH))[2 * ly0 + lane] = lds_ld(&s_last[wave][lane]);
    LOC_STAMP(6);
#undef ROW_MASK
}

static __device__ __forceinline__ uint32_t g_load(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// node of the pixel (y, x) = p: its tile's first node + its component's number inside the tile
static __device__ __forceinline__ uint32_t node_at(const CcaDev& c, int y, int x, size_t p) {
    return (uint32_t)((y / kCcaTileH) * c.tiles_x + x / kCcaTile) * (uint32_t)kCcaTilePx + (uint32_t)c.lid[p];
}
static __device__ __forceinline__ uint32_t node_of(const CcaDev& c, uint32_t p) {
    const int y = (int)(p / (uint32_t)c.W);
    return node_at(c, y, (int)(p - (uint32_t)y * (uint32_t)c.W), p);
}
static __device__ __forceinline__ uint32_t g_find(const CcaDev& c, uint32_t x) {
    for (;;) {
        const uint32_t p = g_load(n_parent(c, x));
        if (p == x) return x;
        const uint32_t gp = g_load(n_parent(c, p));
        if (gp == p) return p;
        __hip_atomic_store(n_parent(c, x), gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = gp;
    }
}
// The root whose leader (first pixel in raster order) is larger goes under the other: the root of a finished component is
// the node that holds the component's leader, which is what src/cca.h:36-57 ends up with.  A node's leader never changes.
static __device__ __forceinline__ void g_union(const CcaDev& c, uint32_t a, uint32_t b) {
    for (;;) {
        a = g_find(c, a);
        b = g_find(c, b);
        if (a == b) return;
        if (n_leader(c, a) < n_leader(c, b)) { const uint32_t t = a; a = b; b = t; }
        if (atomicCAS(n_parent(c, a), a, b) == a) return;
    }
}

// (parent, leader) of a node with ONE 8-byte load: a node's leader never changes, its parent only through the 32-bit CAS below
static __device__ __forceinline__ uint2 g_load2(const CcaDev& c, uint32_t n) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<unsigned long long*>(c.nrec + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
// x -> its root (path halving as in g_find), returns the root's leader
static __device__ __forceinline__ uint32_t g_find_leader(const CcaDev& c, uint32_t& x, uint2 r) {      // r = g_load2(x), already fetched
    for (;;) {
        if (r.x == x) return r.y;
        const uint2 g = g_load2(c, r.x);
        if (g.x == r.x) { x = r.x; return g.y; }
        __hip_atomic_store(n_parent(c, x), g.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = g.x;
        r = g_load2(c, x);
    }
}
// g_union with the two first look-ups in flight together and the leaders arriving with the parents: a contact between two nodes
// that are still roots (the rule: every node is, when this pass starts) costs one round trip for both records and one for the CAS
static __device__ __forceinline__ void g_union2(const CcaDev& c, uint32_t a, uint32_t b) {
    uint2 ra = g_load2(c, a), rb = g_load2(c, b);
    for (;;) {
        const uint32_t la = g_find_leader(c, a, ra), lb = g_find_leader(c, b, rb);
        if (a == b) return;
        if (la < lb) { const uint32_t t = a; a = b; b = t; }
        if (atomicCAS(n_parent(c, a), a, b) == a) return;
        ra = g_load2(c, a); rb = g_load2(c, b);
    }
}

// The seams between tiles.  A thread takes kSeamPx consecutive pixels of a seam (along the seam) and has everything a contact needs
// in flight at once: the labels on both sides, the left neighbours' (a contact is reported by the first pixel of a run of
// contacts only), and -- speculatively -- the component numbers on both sides.  The pass is a chain of dependent round trips
// per wavefront; it was labels -> numbers -> parents -> leaders -> CAS (57.7 us per 8 x 3840x2160, 10.8 per 8 x 1280x720, 7.6 for
// one frame) and is labels + numbers -> records -> CAS: 46.7 / 9.4 / 6.3 us.  More pixels per thread (fewer, longer wavefronts)
// were measured in the same trip (profiles/r04_assign_experiments.txt): 2: 46.0 / 11.9 / 9.7, 4: 48.4 / 16.3 / 14.2, 8: 53.5 / 23.3 / 21.9 --
// a thread's contacts are served one after the other.  One pixel per thread.
#ifndef FSLIC_SEAM_PX
#define FSLIC_SEAM_PX 1
#endif
constexpr int kSeamPx = FSLIC_SEAM_PX;
__global__ __launch_bounds__(256) void k_cca_border(CcaDev c, int nh, int nv) {
    c.select(blockIdx.y);
    constexpr int T = kCcaTile;
    constexpr int TH = kCcaTileH;
    static_assert(TH % kSeamPx == 0 && T % kSeamPx == 0, "a thread's pixels lie in one tile");
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = c.W, H = c.H;
    if (i < nh) {            // kSeamPx pixels on the first row of a tile row, merged with the pixels above
        const int cw = (W + kSeamPx - 1) / kSeamPx;
        const int sr = i / cw, x0 = (i - sr * cw) * kSeamPx;
        const int y = (sr + 1) * TH;
        const size_t p = (size_t)y * W + x0;
        uint32_t cur[kSeamPx + 1], up[kSeamPx + 1], nc[kSeamPx], nu[kSeamPx];      // [0]: the left neighbours
        const bool has_left = (x0 % T) != 0;
        cur[0] = has_left ? (uint32_t)c.in[p - 1] : 0x10000u;
        up[0] = has_left ? (uint32_t)c.in[p - W - 1] : 0x10000u;
#pragma unroll
        for (int q = 0; q < kSeamPx; ++q) {
            const bool ok = x0 + q < W;
            cur[q + 1] = ok ? (uint32_t)c.in[p + q] : 0x20000u;
            up[q + 1] = ok ? (uint32_t)c.in[p - W + q] : 0x30000u;
            nc[q] = ok ? (uint32_t)c.lid[p + q] : 0u;
            nu[q] = ok ? (uint32_t)c.lid[p - W + q] : 0u;
        }
        const uint32_t tile = (uint32_t)((sr + 1) * c.tiles_x + x0 / T) * (uint32_t)kCcaTilePx;
#pragma unroll
        for (int q = 0; q < kSeamPx; ++q) {
            const uint32_t lbl = cur[q + 1];
            if (up[q + 1] == lbl && (cur[q] != lbl || up[q] != lbl))
                g_union2(c, tile + nc[q], tile - (uint32_t)c.tiles_x * (uint32_t)kCcaTilePx + nu[q]);
        }
    } else if (i < nh + nv) { // kSeamPx pixels on the first column of a tile column, merged with the pixels to the left
        const int j = i - nh;
        const int ch = (H + kSeamPx - 1) / kSeamPx;
        const int bx = j / ch + 1, y0 = (j - (bx - 1) * ch) * kSeamPx;
        const int r0 = y0 % TH;
        const uint32_t tile_r = (uint32_t)((y0 / TH) * c.tiles_x + bx), tile_l = tile_r - 1u;
        const uint32_t* er = c.vedge + ((size_t)tile_r * TH + r0) * 2;           // first column of the right tile, row by row (stride 2 words)
        const uint32_t* el = c.vedge + ((size_t)tile_l * TH + r0) * 2 + 1;       // last column of the left tile
        uint32_t a[kSeamPx + 1], b[kSeamPx + 1];                                // [0]: the pixels above
        a[0] = r0 ? er[-2] : 0x10000u;
        b[0] = r0 ? el[-2] : 0x20000u;
#pragma unroll
        for (int q = 0; q < kSeamPx; ++q) {
            const bool ok = y0 + q < H;
            a[q + 1] = ok ? er[2 * q] : 0x10000u;
            b[q + 1] = ok ? el[2 * q] : 0x20000u;
        }
#pragma unroll
        for (int q = 0; q < kSeamPx; ++q) {
            const uint32_t lbl = a[q + 1] & 0xFFFFu;
            const bool contact = y0 + q < H && ((a[q + 1] ^ b[q + 1]) & 0xFFFFu) == 0u;
            const bool first = q + r0 == 0 || (a[q] & 0xFFFFu) != lbl || (b[q] & 0xFFFFu) != lbl;
            if (contact && first)
                g_union2(c, tile_r * (uint32_t)kCcaTilePx + (a[q + 1] >> 16), tile_l * (uint32_t)kCcaTilePx + (b[q + 1] >> 16));
        }
    }
}

// Pass over the frame's nodes (counters[0] of them, dense): flat grid, one node per thread.
//   * node -> global root (read-only walk), parent flattened, its tile-local area added to the global root's
//   * global roots get their final label reset
//   * candidates (components with area >= threshold, src/cca.cpp:213-217) are found HERE, without a second pass over
//     complete areas: a component's area only grows, so it crosses the threshold exactly once -- either it starts at or
//     above it (the root's own tile-local area: its owner appends it) or exactly one addition takes it across (old <
//     threshold <= old + a: that thread appends it).  Leaders only; the select kernel reads the final areas.
__global__ __launch_bounds__(256) void k_cca_compress(CcaDev c, int nframes) {
    // frame = fastest-varying part of the block index, as in k_cca_local: the candidate appends of blocks that run together
    // go to different frames' counters
    const uint32_t bx = blockIdx.x / (unsigned)nframes, nbx = gridDim.x / (unsigned)nframes;
    c.select((int)(blockIdx.x % (unsigned)nframes));
    const uint32_t r_first = c.roots[bx * blockDim.x + threadIdx.x];      // requested together with the count (capacity: see k_cca_chain)
    const uint32_t n = c.counters[0];
    const int lane = LANE();
    const uint32_t gstride = nbx * blockDim.x;
    const int64_t thr = (int64_t)c.min_threshold;
    // a BLOCK's trips are uniform: the candidate append below is one returning atomic per block and trip, not per wavefront.  Every
    // append of a frame lands on the same counter, and same-address returning global atomics are served one after the other at about
    // 8.5 ns each: with one per wavefront (about 800 per 1280x720 frame) the pass took 11.4 us for one frame and 15.3 for eight
    // (eight counters in parallel); per block 6.6 and 9.0 (profiles/r04_assign_experiments.txt)
    __shared__ uint32_t s_app[2];          // [0] candidates of the block in this trip, [1] their base in the list
    for (uint32_t b0 = bx * blockDim.x; b0 < n; b0 += gstride) {
        const uint32_t i = b0 + threadIdx.x;
        if (threadIdx.x == 0) s_app[0] = 0u;
        __syncthreads();
        const bool live = i < n;
        const uint32_t r = (b0 == bx * blockDim.x && live) ? r_first : c.roots[min(i, n - 1)];
        const uint32_t a = n_local(c, r);
        // READ-ONLY walk: a path-halving store of another thread (parent[r] = some ancestor) could land after this
        // thread's parent[r] = g and leave r one level short of flat, and chain / relabel rely on parent[node]
        // being the global root (an intermittent wrong label at 4K, about one run in six, was exactly that)
        uint32_t g = r;
        for (;;) {
            const uint32_t p = g_load(n_parent(c, g));
            if (p == g) break;
            g = p;
        }
        bool cand = false;
        if (live) {
            if (g == r) {
                c.nfinal[r] = 0xFFFFu;
                cand = (int64_t)a >= thr;
            } else {
                __hip_atomic_store(n_parent(c, r), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t old = atomicAdd(&c.narea[g], a);
                cand = (int64_t)old < thr && (int64_t)old + (int64_t)a >= thr;
            }
        }
        const int32_t leader = cand ? (int32_t)n_leader(c, g) : 0;
        const unsigned long long m = __ballot(cand);
        uint32_t wbase = 0;
        if (m && lane == 0) wbase = atomicAdd(&s_app[0], (uint32_t)__popcll(m));      // this wavefront's place among the block's candidates
        wbase = rfl(wbase);
        __syncthreads();
        if (threadIdx.x == 0 && s_app[0]) s_app[1] = atomicAdd(&c.counters[1], s_app[0]);
        __syncthreads();
        if (cand) {     // the leader and, beside it, its node (the component's root): k_cca_select then reaches the area in two dependent loads, not three
            const uint32_t at = s_app[1] + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            c.cand_leader[at] = leader;
            c.cand_area[at] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Device top-K + ranking (src/cca.cpp:205-237) in ONE block, for the unambiguous case:
//   M <= K            : every candidate is kept
//   M >  K            : the K largest areas are kept; the K-th largest area a* is found by an 8-bit radix
//                       select.  If only part of the components with area == a* fit, which ones
//                       std::partial_sort keeps depends on libstdc++'s heap order -> status = 1 and the host
//                       redoes the step with std::partial_sort itself.
// Kept leaders are ranked (raster buckets, LDS) and nfinal[node of the leader] = rank, i.e. labels 0,1,2.. in raster order
// of each component's first pixel (src/cca.cpp:229-237).
// status word (counters[3]): 0 = done here; 1 = area tie at the cut: the candidates have been written to
// dense_leader / dense_area SORTED BY LEADER (counters[1] = M), which is the order the reference feeds to
// std::partial_sort, and the host only has to run that one call; 2 = too many candidates for this block, the
// host gathers and sorts them itself.
// ---------------------------------------------------------------------------------------------
constexpr int kSelCap = 10240;     // candidates the device path holds in LDS (3 x 40 KB)
constexpr int kSelSortCap = 8192;  // candidates the tie path sorts in LDS (power of two <= kSelCap)
constexpr int kSelBuckets = 2048;  // ranking buckets (consecutive raster-index ranges)
constexpr int kSelBins = 4096;     // histogram bins of the selection

// block-wide exclusive scan of one value per thread (1024 threads): wave scans by shuffles, 16 wave totals in LDS
static __device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    __syncthreads();                 // s_w may still be read from a previous call
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t t = s_w[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

// K-th largest of s_area[0..M) restricted to values v with (v >> (sh + 12)) == prefix (all values when first):
// histogram of (v >> sh) & 4095 (areas spread over many bins: few same-address LDS atomics), descending scan.
// Returns through s_v[1] the bin holding the kk-th largest, through s_v[2] the rank left inside that bin.
static __device__ __forceinline__ void select_pass(const uint32_t* s_area, uint32_t M, int sh, bool first, uint32_t prefix,
                                                   uint32_t kk, uint32_t* s_hist, uint32_t* s_w, uint32_t* s_v) {
    const int tid = threadIdx.x;
    for (int i = tid; i < kSelBins; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint32_t a = s_area[i];
        if (first || (a >> (sh + 12)) == prefix) atomicAdd(&s_hist[(a >> sh) & (kSelBins - 1)], 1u);
    }
    __syncthreads();
    // thread t owns bins 4095-4t .. 4092-4t (descending order)
    uint32_t h[4], tot = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = s_hist[kSelBins - 1 - 4 * tid - q]; tot += h[q]; }
    uint32_t total;
    uint32_t run = block_excl_scan_1024(tot, s_w, total);    // count in bins above this thread's
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (run < kk && run + h[q] >= kk) {
            s_v[1] = (uint32_t)(kSelBins - 1 - 4 * tid - q);
            s_v[2] = kk - run;
        }
        run += h[q];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// The tie at the top-K cut, on the device.  libstdc++'s std::partial_sort = __heap_select + __sort_heap; only the first
// decides WHICH K components survive (the survivors are re-sorted by leader anyway, src/cca.cpp:229), and it is a
// deterministic sequence of heap operations on the candidates in ascending-leader order with comp(l, r) = area[l] > area[r]
// (src/cca.cpp:180-186): bits/stl_heap.h __make_heap / __adjust_heap / __push_heap, bits/stl_algo.h __heap_select (GCC 11;
// the tests pin the same restatement against the live library).
//   1. rank every candidate by leader (bucket ranking), permute (leader, area) into that order
//   2. __make_heap over the first K: level by level from the deepest parents up.  __adjust_heap(p) touches only p's
//      subtree and the sequential loop visits parents in descending index order, i.e. deeper levels first; parents of one
//      level have disjoint subtrees, so a level can run in parallel with identical results (the deep, populous levels one
//      thread per parent; levels 0..4, whose walks are long, one wavefront per parent: tie_wave_sift)
//   3. the __heap_select loop over candidates K..M-1 is sequential: one wavefront scans 64 candidates at a time for
//      area > area[top] (the top's area never decreases, so a candidate rejected once stays rejected) and replays
//      __pop_heap for the survivors in index order (tie_wave_sift from the root: five levels per LDS round trip)
//   4. the heap's K entries are the kept set: label = rank among the kept in leader order (src/cca.cpp:229-237)
// A heap entry is one 64-bit word (area << 32 | candidate index): one LDS access moves it.
//
// Round 5: only the SMALL entries are replayed.  With a* the K-th largest area (known from the selection above), every
// candidate with area > a* ("big") is kept whatever the heap does, and the heap is a min-heap on the area, so the small entries
// (area <= a*: the ties at the cut and whatever lies below it, M - K + a handful) form a connected region around the root.  Where
// a sift's walk leaves that region (both children big) nothing it does further down can change the position of a small entry ever
// again: a child of a big entry is big (heap property), and a walk prefers a small child to a big one whichever big one it is.  So
// every sift -- of __make_heap and of __pop_heap alike -- stops at the first preferred child whose area exceeds min(value, a*): the
// positions and contents of the small entries stay exactly libstdc++'s, the big entries' arrangement does not (and is not read).
// The kept set is then "big, or small and still in the heap", the latter as one bit per candidate (set for the first K, cleared
// when an entry leaves through the root, set when a candidate enters).  The CPU suite pins this rule against the restated library
// calls on 3000 inputs built to tie (tests/, test_sparse_sift_keeps_exactly_what_libstdcxx_heap_select_keeps).  What it buys: a replacement walks the depth of the small region (five or six levels
// for the ~50 small entries of a 1280x720 K=1600 frame) instead of the heap's eleven to thirteen, and during the sequential
// __heap_select loop the top six levels of the heap live in the REGISTERS of the one wavefront that replays it (lane j = heap node
// j - 1): a replacement that ends inside them -- nearly all -- is three comparisons, a handful of mask tests per lane and one lane
// permutation for the entries that move up, no round trip through the heap in LDS at all (round 4: three round trips of ~0.3 us
// each, 21 of the 46 us this kernel took for a 1280x720 frame with a tie).
// ---------------------------------------------------------------------------------------------
typedef unsigned long long heap_t;      // area << 32 | candidate index: one LDS access moves an entry, comp looks at the high half
constexpr int kTieWaveLevels = 4;       // __make_heap: parents of depth <= this are sifted by a wavefront each (tie_wave_sift)
static __device__ __forceinline__ heap_t hp_ld(heap_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
static __device__ __forceinline__ void hp_st(heap_t* p, heap_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
static __device__ __forceinline__ uint32_t hp_area(heap_t e) { return (uint32_t)(e >> 32); }

// Net effect of bits/stl_heap.h __adjust_heap(first, hole, len, value, comp) (which ends in __push_heap), comp(l, r) =
// area[l] > area[r], on a range whose part below `hole` is a valid heap.  libstdc++ walks the hole down to a leaf along the
// preferred children (the child with the smaller area, the right one on a tie; the walk does not look at `value`), then
// pushes `value` back up along the same path while the parent's area is larger.  Areas never decrease along that path, so
// everything below the first entry whose area exceeds value's returns to where it was: the result is the entries e_1 .. e_h
// of the path moved up one place and `value` stored at position h, h = the last path index whose area is <= value's.  That
// is a top-down sift that stops at the first larger child -- identical results, a handful of levels instead of the whole
// depth twice (pinned against the library's own heap calls in the tests, like the rest of the tie path).
// `astar` (the K-th largest area): the sparse rule of tie_heap_select -- a child larger than astar ends the walk like a child larger
// than the value does.
static __device__ __forceinline__ void tie_adjust_heap(heap_t* h, int hole, int len, heap_t value, bool writer, uint32_t astar) {
    const uint32_t vlim = min(hp_area(value), astar);
    int cur = hole;
    for (;;) {
        int child;
        heap_t em;
        if (cur < (len - 1) / 2) {
            child = 2 * (cur + 1);
            const heap_t e1 = hp_ld(h + child), e0 = hp_ld(h + child - 1);     // both children in one LDS round trip
            em = e1;
            if (hp_area(e1) > hp_area(e0)) { child--; em = e0; }              // comp(first[child], first[child - 1])
        } else if ((len & 1) == 0 && cur == (len - 2) / 2) {                   // the one node with a single (left) child
            child = 2 * cur + 1;
            em = hp_ld(h + child);
        } else {
            break;
        }
        if (hp_area(em) > vlim) break;                                        // __push_heap would bring it back here (or: the small region ends)
        if (writer) hp_st(h + cur, em);
        cur = child;
    }
    if (writer) hp_st(h + cur, value);
}

// The same sift executed by ONE wavefront (all 64 lanes active, control flow uniform): __pop_heap(first, middle, i) of the
// __heap_select loop (hole = the root, value = the candidate) and the upper levels of __make_heap (hole = a parent, value = its
// own entry).  The walk's choices do not depend on the value, only on the heap, and a dependent LDS read per level is what a
// sequential walk costs (most of the 11 levels at K = 1600: the candidates past the K-th leader are ordinary, large components
// that sink to the leaves).  Here the wavefront fetches a whole sub-tree at once -- lane j (1..63) reads the node with heap
// number j relative to the current node: five levels below it in ONE LDS round trip.  Which child a node prefers is then one
// comparison per lane against its sibling's area (a quad swap: siblings 2j, 2j+1 share a quad) and one ballot; where the walk
// stops (leaf, or a child larger than the value) another; the five steps are bit tests of those two masks on the scalar unit,
// and every lane that lies on the path stores its entry at its parent's place in one LDS instruction.  Nodes past the heap's
// end read as area 0xffffffff, which reproduces the walk's rules for the node with one child and for leaves.
// (The first form of this routine resolved the five levels with three v_readlane per level and let lanes 0..4 store: a
// replacement at the root took 0.93 us; profiles/r03_lab_select_stamps.txt has both.)
// Returns the area of the entry that ends up at `hole`.
static __device__ __forceinline__ uint32_t tie_wave_sift(heap_t* h, int len, int hole, heap_t value, int lane, uint32_t astar) {
    constexpr uint32_t kInf = 0xFFFFFFFFu;
    const uint32_t va = hp_area(value);
    const uint32_t vlim = min(va, astar);                      // (sparse rule, see tie_heap_select: astar < kInf)
    const int t = 31 - __clz(lane | 1);                       // depth of relative node `lane` (lane 0: unused)
    const int off = lane - (1 << t);
    int cur = hole;                                            // absolute index of the node the value may still end up at
    uint32_t at_hole = va;
    for (;;) {
        const long long idx = (((long long)(cur + 1)) << t) + off - 1;
        const bool ok = lane >= 1 && idx < (long long)len;
        const heap_t e = ok ? hp_ld(h + idx) : ((heap_t)kInf << 32);
        const uint32_t a = (uint32_t)(e >> 32);
        const uint32_t sib = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
        // even lanes (left children): comp(first[right], first[left]) = area[right] > area[left] -> the left one is taken
        const unsigned long long left_pref = __ballot(sib > a);
        const unsigned long long stop_at = __ballot(a > vlim);                // leaf (kInf), __push_heap would bring the value back up to here, or the small region ends
        // five steps down the preferred children, straight-line on the scalar unit (a taken branch costs more than a step's
        // arithmetic); the steps before the first stop are the moves that really happen
        int c[5];
        uint32_t stops = 0;
        {
            int jj = 1;
#pragma unroll
            for (int s5 = 0; s5 < 5; ++s5) {
                const int cl = 2 * jj;
                c[s5] = cl + (int)(((left_pref >> cl) & 1ull) ^ 1ull);
                stops |= (uint32_t)((stop_at >> c[s5]) & 1ull) << s5;
                jj = c[s5];
            }
        }
        const int nmove = __builtin_ctz(stops | 32u);
        const bool stopped = nmove < 5;
        unsigned long long path = 0;                           // relative nodes whose entries move up one place
        int j = 1;
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            path |= s5 < nmove ? 1ull << c[s5] : 0ull;
            j = s5 < nmove ? c[s5] : j;
        }
        const long long pidx = (((long long)(cur + 1)) << max(t - 1, 0)) + (off >> 1) - 1;      // the parent of relative node `lane` (t >= 1 where used)
        if ((path >> lane) & 1ull) hp_st(h + pidx, e);
        if (cur == hole && (path & 0xCull)) at_hole = (uint32_t)__builtin_amdgcn_readlane((int)a, (path & 4ull) ? 2 : 3);   // the child that moved into the hole
        const int tj = 31 - __clz(j);
        cur = (int)((((long long)(cur + 1)) << tj) + (j - (1 << tj)) - 1);                 // absolute index of relative node j
        if (stopped || 2 * cur + 1 >= len) break;                                           // (a leaf: no round trip to find that out)
    }
    if (lane == 0) hp_st(h + cur, value);
    return at_hole;
}

#if defined(FSLIC_LAB) && FSLIC_LAB == 1      // lab build only (make LAB=1): 100 MHz time stamps of the select kernel in the unused status words 8..15
#define SEL_STAMP(i_) do { if (threadIdx.x == 0) c.counters[4 + (i_)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SEL_STAMP(i_) do { } while (0)
#endif
static __device__ __forceinline__ void tie_heap_select(const CcaDev& c, uint32_t M, uint32_t K, uint32_t astar, uint32_t* s_area, uint32_t* s_lead, uint32_t* s_key,
                                                       uint32_t* s_bcnt, uint32_t* s_bstart, uint32_t* s_bfill, uint32_t* s_w) {
    const int tid = threadIdx.x, lane = tid & 63;
    SEL_STAMP(2);
    constexpr int PER = kSelSortCap / 1024;                   // candidates per thread
    // 1. leader order
    int bsh = 0;
    while ((((uint32_t)c.N - 1u) >> bsh) >= (uint32_t)kSelBuckets) bsh++;
    for (int i = tid; i < 3 * kSelBuckets; i += 1024) s_bcnt[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024) atomicAdd(&s_bcnt[s_lead[i] >> bsh], 1u);
    __syncthreads();
    {
        const uint32_t a = s_bcnt[2 * tid], b = s_bcnt[2 * tid + 1];
        uint32_t tot;
        const uint32_t ex = block_excl_scan_1024(a + b, s_w, tot);
        s_bstart[2 * tid] = ex;
        s_bstart[2 * tid + 1] = ex + a;
    }
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint32_t b = s_lead[i] >> bsh;
        s_key[s_bstart[b] + atomicAdd(&s_bfill[b], 1u)] = s_lead[i];
    }
    __syncthreads();
    uint32_t my_lead[PER], my_area[PER], my_rank[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const uint32_t i = (uint32_t)tid + (uint32_t)q * 1024u;
        my_lead[q] = 0; my_area[q] = 0; my_rank[q] = 0xFFFFFFFFu;
        if (i < M) {
            const uint32_t me = s_lead[i];
            const uint32_t b = me >> bsh;
            const uint32_t lo = s_bstart[b], hi = lo + s_bcnt[b];
            uint32_t r = lo;
            for (uint32_t j = lo; j < hi; ++j) r += (uint32_t)(s_key[j] < me);
            my_lead[q] = me; my_area[q] = s_area[i]; my_rank[q] = r;
        }
    }
    __syncthreads();
    // leaders in order -> global (read back in step 4), areas in order -> s_area; the heap (K 64-bit entries) takes over
    // s_lead and s_key, which are adjacent and dead from here on (2 * cap words >= 2 K)
    heap_t* const heap = reinterpret_cast<heap_t*>(s_lead);
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (my_rank[q] != 0xFFFFFFFFu) {
            const uint32_t r = my_rank[q];
            c.cand_leader[r] = (int32_t)my_lead[q];
            s_area[r] = my_area[q];
            if (r < K) heap[r] = ((heap_t)my_area[q] << 32) | r;
        }
    // "in the heap" bits, one per candidate (the ranking buckets are dead): the first K in leader order to begin with
    uint32_t* const s_in = s_bcnt;
    static_assert(kSelSortCap / 32 <= 3 * kSelBuckets, "one bit per candidate in the bucket arrays");
    for (uint32_t w = (uint32_t)tid; w < (uint32_t)(kSelSortCap / 32); w += 1024u) {
        const uint32_t lo = w * 32u;
        s_in[w] = K >= lo + 32u ? 0xFFFFFFFFu : (K <= lo ? 0u : (1u << (K - lo)) - 1u);
    }
    __syncthreads();
    SEL_STAMP(3);
    // 2. __make_heap(first, first + K)
    const int len = (int)K;
    if (len >= 2) {
        const int last_parent = (len - 2) / 2;
        int d = 0;
        while (((2 << d) - 1) <= last_parent) d++;           // depth of the deepest parent
        // (the deep levels subtree by subtree inside one wavefront each -- no block barrier between two levels -- were measured in
        // round 5: 9.9 instead of 6.3 us for K = 1600; the barrier is not what a level costs, the dependent LDS round trips of the
        // walks are, and a wavefront that takes two subtrees runs them one after the other)
        for (; d >= 0; --d) {
            const int lo = (1 << d) - 1, hi = min((2 << d) - 2, last_parent);
            if (d > kTieWaveLevels) {
                // deep parents: many of them, short walks -- one thread each
                for (int p = lo + tid; p <= hi; p += 1024)
                    tie_adjust_heap(heap, p, len, hp_ld(heap + p), true, astar);
            } else {
                // the upper levels: at most one parent per wavefront, walks of several levels -- five levels per LDS round trip
                for (int p = lo + (int)rfl((uint32_t)(tid >> 6)); p <= hi; p += 16)
                    tie_wave_sift(heap, len, p, hp_ld(heap + p), lane, astar);
            }
            __syncthreads();
        }
    }
    SEL_STAMP(4);
    // 3. for (i = K; i < M; ++i) if (comp(*i, *first)) __pop_heap(first, middle, i)
    if (tid < 64) {
        constexpr uint32_t kInf = 0xFFFFFFFFu;
        // the top six levels of the heap in this wavefront's registers: lane j (1..63) = heap node j - 1 (absent nodes: area kInf)
        uint32_t ea = kInf, ei = 0u;
        if (lane >= 1 && lane - 1 < len) { const heap_t e = hp_ld(heap + (lane - 1)); ea = hp_area(e); ei = (uint32_t)e; }
        uint32_t top_area = (uint32_t)__builtin_amdgcn_readlane((int)ea, 1);
        for (uint32_t base = K; base < M; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const uint32_t a = i < M ? s_area[i] : 0u;
            unsigned long long m = __ballot(i < M && a > top_area);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t aj = (uint32_t)__builtin_amdgcn_readlane((int)a, j);
                if (aj <= top_area) continue;
                const uint32_t vi = base + (uint32_t)j;
                // the entry at the root leaves the heap, the candidate enters (the bit of a big candidate is never read)
                const uint32_t ti = (uint32_t)__builtin_amdgcn_readlane((int)ei, 1);
                if (lane == 0) { atomicAnd(&s_in[ti >> 5], ~(1u << (ti & 31u))); atomicOr(&s_in[vi >> 5], 1u << (vi & 31u)); }
                const uint32_t vlim = min(aj, astar);
                // The walk, for all five levels at once and without a scalar step per level: lane j (>= 2) is the child its parent
                // prefers when its sibling's area is larger (left children; the sibling shares the lane's quad) or not smaller (right
                // children: libstdc++ takes the right one on a tie), and its entry moves up if the walk gets there and the entry is
                // not beyond the value (or the small region).  The entries that move are the lanes for which that holds at
                // themselves AND at every ancestor: five bit tests of one wavefront mask per lane.
                const uint32_t sib = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ea, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
                const unsigned long long pref = (ballot(sib > ea) & 0x5555555555555554ull) | (ballot(sib >= ea) & 0xAAAAAAAAAAAAAAA8ull);
                const unsigned long long okm = (pref & ~ballot(ea > vlim)) | 3ull;      // (bits 0, 1: the ancestors "above the root" hold nobody up)
                uint32_t on = 1u;
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) on &= (uint32_t)(okm >> (lane >> s5));
                const unsigned long long moved = ballot((on & 1u) != 0u) & ~3ull;        // a chain from level 1 down: at most one lane per level
                const int dst = moved ? 63 - __builtin_clzll(moved) : 1;                // where the value ends up (unless the walk goes on below the sixth level)
                // every entry on the chain to its parent's lane, in one pull: the parent of a moved lane fetches it
                const uint32_t kids = lane < 32 ? (uint32_t)(moved >> (2 * lane)) & 3u : 0u;
                const int from = (kids ? 2 * lane + (int)(kids >> 1) : lane) << 2;
                ea = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)ea);
                ei = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)ei);
                const int cur = dst - 1;                                       // heap index of the place the value may still end up at
                if (dst < 32 || 2 * cur + 1 >= len) {
                    if (lane == dst) { ea = aj; ei = vi; }
                } else {
                    // the small region reaches below the sixth level here (rare): the rest of the walk on the heap in LDS, whose
                    // entries from the seventh level down are current (nothing above has ever been written back, nothing above is
                    // read: the sift reads the descendants of `cur` only), then this node's new entry back into its register
                    (void)tie_wave_sift(heap, len, cur, ((heap_t)aj << 32) | (heap_t)vi, lane, astar);
                    const heap_t e = hp_ld(heap + cur);
                    if (lane == dst) { ea = hp_area(e); ei = (uint32_t)e; }
                }
                top_area = (uint32_t)__builtin_amdgcn_readlane((int)ea, 1);
            }
        }
    }
    __syncthreads();
    SEL_STAMP(5);
    // 4. kept set = the big candidates and the small ones still in the heap; label = rank among the kept in leader order
    // exclusive scan of the flags: thread t owns candidates [PER * t, PER * t + PER)
    uint32_t fl[PER], cnt = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const uint32_t i = (uint32_t)tid * PER + (uint32_t)q;
        fl[q] = i < M ? (uint32_t)(s_area[i] > astar) | ((s_in[i >> 5] >> (i & 31u)) & 1u) : 0u;
        cnt += fl[q];
    }
    uint32_t total;
    uint32_t run = block_excl_scan_1024(cnt, s_w, total);
    {   // (the kept candidates' leaders, then their pixels' component numbers, each stage's loads in flight together)
        uint32_t ld[PER], nd[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) ld[q] = (uint32_t)c.cand_leader[fl[q] ? (uint32_t)tid * PER + (uint32_t)q : 0u];
#pragma unroll
        for (int q = 0; q < PER; ++q) nd[q] = node_of(c, ld[q]);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (fl[q]) c.nfinal[nd[q]] = (uint16_t)run;
            run += fl[q];
        }
    }
    SEL_STAMP(6);
}

__global__ __launch_bounds__(1024) void k_cca_select(CcaDev c, int cap) {
    c.select(blockIdx.x);
    SEL_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];     // the tie path overlays 64-bit heap entries
    uint32_t* const s_area = s_dyn;
    uint32_t* const s_lead = s_dyn + cap;
    uint32_t* const s_key = s_dyn + 2 * cap;
    uint32_t* const s_bcnt = s_dyn + 3 * cap;           // ranking buckets; the selection's histogram lives here first
    uint32_t* const s_bstart = s_bcnt + kSelBuckets;
    uint32_t* const s_bfill = s_bstart + kSelBuckets;
    uint32_t* const s_hist = s_bcnt;                    // kSelBins <= 3 * kSelBuckets words
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_v[8];     // [0] max area, [1] bin, [2] kk, [4] ties
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int K = c.K;
    // 1. the frame's candidates -> LDS.  The first trip's (leader, node) pairs are requested together with the count: the lists hold
    // one entry per pixel of whole 64 x 32 tiles, i.e. at least max(N, 2048), and entries past the count are not used.
    const uint32_t cand_last = (uint32_t)max(c.N, 2048) - 1u;
    uint32_t l_first[4], nd_first[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { l_first[u] = (uint32_t)c.cand_leader[min((uint32_t)threadIdx.x + 1024u * u, cand_last)]; nd_first[u] = c.cand_area[min((uint32_t)threadIdx.x + 1024u * u, cand_last)]; }
    const uint32_t M = c.counters[1];
    if (tid < 8) s_v[tid] = 0;
    if (tid == 0) c.counters[2] = M;
    if (M > (uint32_t)cap) {
        if (tid == 0) c.counters[3] = 2u;
        return;
    }
    // (leader, node) -> area of that node: two dependent global loads per candidate (three until k_cca_compress wrote the node beside the
    // leader).  Four candidates per thread and trip with the loads of each stage in flight together (one candidate after the other the
    // chain was paid per candidate: 18 of the 25 us this step took for the 6 060 candidates of a 3840x2160 K=6000 frame)
    uint32_t amax = 0;
    for (uint32_t i0 = tid; i0 < M; i0 += 4096) {
        uint32_t l[4], nd[4], a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {      // (the node: written by k_cca_compress)
            if (i0 == (uint32_t)tid) { l[u] = l_first[u]; nd[u] = i0 + 1024u * u < M ? nd_first[u] : 0u; }      // (an entry past the count may hold anything: node 0 is read instead, and not used)
            else { l[u] = (uint32_t)c.cand_leader[min(i0 + 1024u * u, M - 1u)]; nd[u] = c.cand_area[min(i0 + 1024u * u, M - 1u)]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = c.narea[nd[u]];      // (a leader's node is its component's root)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 1024u * u;
            if (i < M) { s_area[i] = a[u]; s_lead[i] = l[u]; amax = max(amax, a[u]); }
        }
    }
    __syncthreads();
    if (M > (uint32_t)K) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = max(amax, (uint32_t)__shfl_xor(amax, off));
        if (lane == 0 && amax) atomicMax(&s_v[0], amax);
        __syncthreads();
    }
    // 2. K-th largest area: 12 bits per pass, starting just above the top bit of the largest area
    uint32_t astar = 0;           // keep iff area >= astar
    if (M > (uint32_t)K) {
        const uint32_t mx = s_v[0];
        int sh = 0;
        while ((mx >> sh) >= (uint32_t)kSelBins) sh += 12;
        uint32_t prefix = 0, kk = (uint32_t)K;
        bool first = true;
        for (; sh >= 0; sh -= 12) {
            select_pass(s_area, M, sh, first, prefix, kk, s_hist, s_w, s_v);
            prefix = (prefix << 12) | s_v[1];
            kk = s_v[2];
            first = false;
            __syncthreads();
        }
        astar = prefix;
        // kk of the components with area == a* are needed; unless that is all of them the choice is libstdc++'s
        uint32_t eq = 0;
        for (uint32_t i = tid; i < M; i += 1024) eq += (uint32_t)(s_area[i] == astar);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) eq += (uint32_t)__shfl_xor(eq, off);
        if (lane == 0 && eq) atomicAdd(&s_v[4], eq);
        __syncthreads();
        SEL_STAMP(1);
        if (s_v[4] != kk) {
            // Area tie at the cut: which of the tied components std::partial_sort (src/cca.cpp:225-228) keeps depends on
            // libstdc++'s heap mechanics, so its __heap_select is replayed here, operation by operation, on the candidates
            // in ascending-leader order (the order the reference feeds it).  See tie_heap_select below.
            if (M > (uint32_t)kSelSortCap) {
                if (tid == 0) c.counters[3] = 2u;
                return;
            }
            tie_heap_select(c, M, (uint32_t)K, astar, s_area, s_lead, s_key, s_bcnt, s_bstart, s_bfill, s_w);
            if (tid == 0) c.counters[3] = 0u;
            return;
        }
    }
    // 3. rank of every kept leader among the kept leaders (labels 0,1,2.. in raster order of the first pixel,
    //    src/cca.cpp:229-237): bucket by raster-index range, exclusive scan of the bucket sizes, rank inside the bucket
    int bsh = 0;
    while ((((uint32_t)c.N - 1u) >> bsh) >= (uint32_t)kSelBuckets) bsh++;
    for (int i = tid; i < 3 * kSelBuckets; i += 1024) s_bcnt[i] = 0;     // bcnt, bstart, bfill (the histogram is dead)
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024)
        if (s_area[i] >= astar) atomicAdd(&s_bcnt[s_lead[i] >> bsh], 1u);
    __syncthreads();
    {
        const uint32_t a = s_bcnt[2 * tid], b = s_bcnt[2 * tid + 1];
        uint32_t n_keep;
        const uint32_t ex = block_excl_scan_1024(a + b, s_w, n_keep);
        s_bstart[2 * tid] = ex;
        s_bstart[2 * tid + 1] = ex + a;
    }
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024)
        if (s_area[i] >= astar) {
            const uint32_t b = s_lead[i] >> bsh;
            s_key[s_bstart[b] + atomicAdd(&s_bfill[b], 1u)] = s_lead[i];
        }
    __syncthreads();
    for (uint32_t i0 = tid; i0 < M; i0 += 4096) {      // (four candidates per trip: the look-ups of their pixels' component numbers in flight together)
        uint32_t me[4], r[4], nd[4];
        bool kept[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nd[u] = c.cand_area[min(i0 + 1024u * u, M - 1u)];      // the candidates' nodes (k_cca_compress), requested before the ranks are counted
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 1024u * u;
            kept[u] = i < M && s_area[i] >= astar;
            me[u] = kept[u] ? s_lead[i] : 0u;
            r[u] = 0u;
            if (kept[u]) {
                const uint32_t b = me[u] >> bsh;
                const uint32_t lo = s_bstart[b], hi = lo + s_bcnt[b];
                r[u] = lo;
                for (uint32_t j = lo; j < hi; ++j) r[u] += (uint32_t)(s_key[j] < me[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (kept[u]) c.nfinal[nd[u]] = (uint16_t)r[u];
    }
    if (tid == 0) c.counters[3] = 0u;
}

// Grid of a pass over the roots: one root per thread for up to N / 12 roots (a structured frame has fewer: 57 k of 922 k
// pixels at 1280x720), more trips for noise; at least one block.
static int root_pass_blocks(const CcaDev& c) {
    const int b = (c.N / 12 + 255) / 256;
    return b < 1 ? 1 : b;
}

void launch_cca_phase1(const CcaDev& c, int nframes, hipStream_t st) {
    constexpr int T = kCcaTile;
    const int tx = (c.W + T - 1) / T, ty = (c.H + kCcaTileH - 1) / kCcaTileH;
    launch(k_cca_local, dim3(nframes, tx, ty), dim3(256), 0, st, c);
    const int nh = (ty - 1) * ((c.W + kSeamPx - 1) / kSeamPx), nv = (tx - 1) * ((c.H + kSeamPx - 1) / kSeamPx);      // threads: kSeamPx seam pixels each
    if (nh + nv > 0)
        launch(k_cca_border, dim3((nh + nv + 255) / 256, nframes), dim3(256), 0, st, c, nh, nv);
    launch(k_cca_compress, dim3(root_pass_blocks(c) * nframes), dim3(256), 0, st, c, nframes);
}

void launch_cca_select(const CcaDev& c, int nframes, hipStream_t st) {
    // LDS capacity: no frame has more candidates than N / max(threshold, 1)
    long long need = (long long)c.N / (c.min_threshold > 1 ? c.min_threshold : 1) + 1;
    int cap = 1024;
    while (cap < need && cap < kSelCap) cap <<= 1;
    if (cap > kSelCap) cap = kSelCap;
    const size_t lds = (3 * (size_t)cap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t);
    static_assert(kSelBins <= 3 * kSelBuckets, "the selection histogram reuses the ranking buckets");
    // (per device: a process may drive several GPUs.  A device that refuses the attribute gets a capacity that fits 64 KB; frames with more
    // candidates than that take the host's top-K step, as frames beyond kSelCap do)
    static std::atomic<int> attr_state[64];
    if (!max_dynamic_lds_on_this_device(k_cca_select, (int)((3 * (size_t)kSelCap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t)), attr_state)) {
        while (cap > 1024 && (3 * (size_t)cap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t) > 64 * 1024) cap >>= 1;
        const size_t lds_small = (3 * (size_t)cap + 3 * (size_t)kSelBuckets) * sizeof(uint32_t);
        launch(k_cca_select, dim3(nframes), dim3(1024), lds_small, st, c, cap);
        return;
    }
    launch(k_cca_select, dim3(nframes), dim3(1024), lds, st, c, cap);
}

// Host path only: forget what the device select / chain wrote, then set the host's choice.
__global__ __launch_bounds__(256) void k_cca_reset_final(CcaDev c) {
    const uint32_t n = c.counters[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t r = c.roots[i];
        if (*n_parent(c, r) == r) c.nfinal[r] = 0xFFFFu;
    }
}
__global__ void k_cca_set_kept(CcaDev c, const int32_t* __restrict__ leaders, const uint16_t* __restrict__ labels, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c.nfinal[node_of(c, (uint32_t)leaders[i])] = labels[i];
}
// Host top-K path: areas of the candidates, dense next to their leaders (the device path reads them in place).
__global__ void k_cca_gather_area(CcaDev c) {
    const uint32_t M = c.counters[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) c.cand_area[i] = c.narea[node_of(c, (uint32_t)c.cand_leader[i])];
}
void launch_cca_gather_area(const CcaDev& c, hipStream_t st) {
    launch(k_cca_gather_area, dim3(64), dim3(256), 0, st, c);
}

void launch_cca_set_kept(const CcaDev& c, const int32_t* d_leaders, const uint16_t* d_labels, int n, hipStream_t st) {
    launch(k_cca_reset_final, dim3(root_pass_blocks(c)), dim3(256), 0, st, c);
    if (n <= 0) return;
    launch(k_cca_set_kept, dim3((n + 255) / 256), dim3(256), 0, st, c, d_leaders, d_labels, n);
}

// Every component that was not kept takes the final label of the component left of (or, in column 0,
// above) its leader (src/cca.cpp:240-254).  Leaders strictly decrease along the chase, so it terminates.
// Final label of a component that was not kept (src/cca.cpp:240-254): that of the component left of (in column 0: above) its
// leader, followed along the chain until a component with a final label or pixel 0 is reached.  Leaders strictly decrease
// along the chain, so it terminates.  (Evaluating this per pixel inside the relabel pass instead of once per root was
// measured: 59 us instead of 18 + 20 for eight 1280x720 frames -- a wavefront waits for its longest chain.)
static __device__ __forceinline__ uint32_t chase_final(const CcaDev& c, uint32_t g) {
    const uint32_t W = (uint32_t)c.W;
    for (;;) {
        const uint32_t cur = n_leader(c, g);
        if (cur == 0) return 0u;                        // src/cca.cpp:238
        uint32_t a = n_adopt(c, g);
        if (a == kNoAdopt) {                            // the neighbour lies in another tile (or another wavefront's rows)
            const uint32_t t = (cur % W > 0) ? cur - 1 : cur - W;   // src/cca.cpp:243-248
            a = node_of(c, t);
        }
        g = *n_parent(c, a);
        const uint32_t f = __hip_atomic_load(c.nfinal + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f != 0xFFFFu) return f;
    }
}

__global__ __launch_bounds__(256) void k_cca_chain(CcaDev c) {
    c.select(blockIdx.y);
    // the first root is requested together with the count (the list's capacity covers the grid: root_pass_blocks): one dependent
    // round trip less at the head of a kernel that is a chain of them
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r0 = c.roots[i0];
    const uint32_t n = c.counters[0];
    const uint32_t gstride = gridDim.x * blockDim.x;
    for (uint32_t i = i0; i < n; i += gstride) {
        const uint32_t r = i == i0 ? r0 : c.roots[i];
        const uint32_t g = *n_parent(c, r);
        // the component's final label: set already when it was kept, or when its own thread got here first; otherwise
        // chased from its leader (whoever computes it finds the same value)
        uint32_t f = __hip_atomic_load(c.nfinal + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f == 0xFFFFu) f = chase_final(c, g);
        // ... stored at EVERY node (an entry of a node that is not a global root is read by nothing else: the chase only
        // ever looks at global roots), so that the relabel pass needs one table look-up per pixel
        __hip_atomic_store(c.nfinal + r, (uint16_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The last kernel of a group also carries the group's results to the host: blocks past the relabel grid copy the
// cluster state and the frame's status words into the engine's pinned blocks (ExportDev; K == 0: nothing to export).
// largest value of the wavefront's 64 lanes (DPP row steps, then the four rows through v_readlane)
static __device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define FSLIC_DPP_MAX(ctrl) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false))
    FSLIC_DPP_MAX(0xB1); FSLIC_DPP_MAX(0x4E); FSLIC_DPP_MAX(0x141); FSLIC_DPP_MAX(0x140);
#undef FSLIC_DPP_MAX
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

__global__ __launch_bounds__(256) void k_cca_relabel(CcaDev c, ExportDev ex, int relabel_blocks) {
    const int z = blockIdx.y;
    if ((int)blockIdx.x >= relabel_blocks) {
        const size_t d = (size_t)z * ex.frame_bytes;
        const uint32_t* yx = ex.yx_cur; const uint32_t* rest = ex.lab_n_moved; const uint32_t* misc = ex.misc0;
        FrameDev::adv(yx, d); FrameDev::adv(rest, d); FrameDev::adv(misc, d);
        uint32_t* out = ex.h_cl + (size_t)z * ex.h_stride;
        const int K = ex.K;
        for (int i = ((int)blockIdx.x - relabel_blocks) * blockDim.x + threadIdx.x; i < 4 * K; i += ((int)gridDim.x - relabel_blocks) * blockDim.x)
            out[i] = i < K ? yx[i] : rest[i - K];              // [0,K) positions, [K,4K) colour, member count, moved flag
        if ((int)blockIdx.x == relabel_blocks && threadIdx.x < 16) ex.h_misc[16 * (size_t)z + threadIdx.x] = misc[threadIdx.x];
        return;
    }
    c.select(z);
    // One block = one 64x32 tile, thread = 4 consecutive pixels of one row, two rows per thread: 8-byte loads of the component numbers, a look-up in the tile's (contiguous, at most
    // 4 KB) table of final labels, 8-byte stores.  Frames whose width is not a multiple of 4 take the scalar form.
    constexpr int T = kCcaTile, TH = kCcaTileH;
    const int tiles_x = c.tiles_x, tiles_y = (c.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y;
    const bool vec = (c.W & 3) == 0;
    // The tile's table of final labels goes into LDS first -- as much of it as the tile's pixels refer to (the largest component number
    // among them, found by a block-wide maximum: a few hundred entries), with coalesced 16-byte loads.  Read straight from memory every
    // pixel's look-up was a 64-address gather through the vector-memory path (eight per thread), which bounded the pass (65 us per
    // 8 x 3840x2160 = 4 TB/s of its 4 B per pixel).
    __shared__ __attribute__((aligned(16))) uint16_t s_fin[kCcaTilePx];
    __shared__ uint32_t s_max;
    for (int tile = blockIdx.x; tile < ntiles; tile += relabel_blocks) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const uint16_t* __restrict__ gfin = c.nfinal + (size_t)tile * kCcaTilePx;
        const int x0 = tx * T + 4 * (int)(threadIdx.x & 15);
        const int r0 = (int)(threadIdx.x >> 4);                 // rows r0 and r0 + 16 of the tile
        if (tile != (int)blockIdx.x) __syncthreads();           // (a block that takes several tiles: the previous tile's look-ups are done)
        if (threadIdx.x == 0) s_max = 0u;
        __syncthreads();
        if (vec) {
            uint2 v[2];
            bool ok[2];
            uint32_t mx = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = ty * TH + r0 + 16 * h;
                ok[h] = y < c.H && x0 < c.W;
                v[h] = make_uint2(0u, 0u);
                if (ok[h]) v[h] = *reinterpret_cast<const uint2*>(c.lid + (size_t)y * c.W + x0);
                mx = max(max(mx, max(v[h].x & 0xFFFFu, v[h].x >> 16)), max(v[h].y & 0xFFFFu, v[h].y >> 16));
            }
            mx = wave_max_u32(mx);
            if ((threadIdx.x & 63) == 0) atomicMax(&s_max, mx);
            __syncthreads();
            const uint32_t cnt = min(s_max + 1u, (uint32_t)kCcaTilePx);
            for (uint32_t i = threadIdx.x; i * 8u < cnt; i += 256u) reinterpret_cast<uint4*>(s_fin)[i] = reinterpret_cast<const uint4*>(gfin)[i];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!ok[h]) continue;
                const int y = ty * TH + r0 + 16 * h;
                uint2 o;
                o.x = (uint32_t)s_fin[v[h].x & 0xFFFFu] | ((uint32_t)s_fin[v[h].x >> 16] << 16);
                o.y = (uint32_t)s_fin[v[h].y & 0xFFFFu] | ((uint32_t)s_fin[v[h].y >> 16] << 16);
                st_stream(reinterpret_cast<uint2*>(c.out + (size_t)y * c.W + x0), o);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int y = ty * TH + r0 + 16 * h;
                if (y >= c.H) continue;
                for (int q = 0; q < 4; ++q) {
                    const int x = x0 + q;
                    if (x < c.W) st_stream(c.out + (size_t)y * c.W + x, gfin[c.lid[(size_t)y * c.W + x]]);
                }
            }
        }
    }
}

void launch_cca_phase2(const CcaDev& c, int nframes, hipStream_t st, const ExportDev* ex_) {
    launch(k_cca_chain, dim3(root_pass_blocks(c), nframes), dim3(256), 0, st, c);
    int blocks = c.tiles_x * ((c.H + kCcaTileH - 1) / kCcaTileH);      // one tile per block (tile-stride loop beyond the cap)
    const int cap = (16384 + nframes - 1) / nframes;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    ExportDev ex{};
    if (ex_) ex = *ex_;
    const int eb = ex.K > 0 ? (4 * ex.K + 1023) / 1024 : 0;          // a few blocks: 4K words per frame
    launch(k_cca_relabel, dim3(blocks + eb, nframes), dim3(256), 0, st, c, ex, blocks);
}

}  // namespace fslic
