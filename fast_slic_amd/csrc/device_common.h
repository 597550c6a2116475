// device_common.h -- helpers shared by the gfx950 kernel translation units (lab.hip, clusters.hip, assign.hip, cca.hip).
// Internal to the library.
#pragma once
#include "kernels.h"
#include "launch.h"
#include <cstdlib>

namespace fslic {

#define LANE() (threadIdx.x & 63)

// Kernels index the double-buffered members with the constants 0 (this pass) and 1 (the other parity): a
// run-time index into the by-value FrameDev argument would push the whole struct into scratch memory once
// select() has modified it.  The launchers rotate the arrays accordingly.
static inline FrameDev rotated(const FrameDev& f, int buf, int sbuf, int obuf = -1) {
    FrameDev g = f;
    g.cell_cnt[0] = f.cell_cnt[buf]; g.cell_cnt[1] = f.cell_cnt[buf ^ 1];
    g.cell_items[0] = f.cell_items[buf]; g.cell_items[1] = f.cell_items[buf ^ 1];
    if (obuf < 0) {      // two-buffer rotation (separate cluster pass): the spill lists follow the bins
        g.ovf_cnt[0] = f.ovf_cnt[buf]; g.ovf_cnt[1] = f.ovf_cnt[buf ^ 1];
        g.ovf_items[0] = f.ovf_items[buf]; g.ovf_items[1] = f.ovf_items[buf ^ 1];
    } else {             // three-buffer rotation (fused cluster pass): [0] read, [1] appended to, [2] cleared
        for (int q = 0; q < 3; q++) { g.ovf_cnt[q] = f.ovf_cnt[(obuf + q) % 3]; g.ovf_items[q] = f.ovf_items[(obuf + q) % 3]; }
    }
    g.sums[0] = f.sums[sbuf]; g.sums[1] = f.sums[sbuf ^ 1];
    return g;
}

// Orders this wavefront's LDS traffic for lane-to-lane exchange through LDS: without it the compiler may move one
// lane's read above another lane's write (it only has to respect a single thread's own accesses).  No instruction
// is emitted beyond the waitcnt the hardware needs; LDS itself serves a wavefront's requests in order.
static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release of GLOBAL memory too, which on
// this target is s_waitcnt vmcnt(0): every pixel load (prologue) or label store (epilogue) still in flight would have to land first.
// Only where nothing but LDS contents passes between the wavefronts of the block at that point.
static __device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Streaming stores for planes that are written once and read by a LATER kernel (LAB plane, label maps, parents,
// feature planes): written through at agent scope (`sc1`) instead of staying dirty in the XCD's L2.  A kernel's
// end-of-kernel release otherwise writes all dirty lines back in one burst that nothing overlaps -- measured
// (scripts/microbench/kernel_boundary.hip): a kernel storing 16 MB takes 8.9 us with plain stores and 3.5 us with
// sc1 stores (64 MB: 24.9 -> ~10 us).
#if defined(FSLIC_LAB) && FSLIC_LAB == 3      // lab build 3: plain (write-back) stores instead, to measure what the write-through costs / saves
template <class T> static __device__ __forceinline__ void st_stream(T* p, T v) { *p = v; }
#else
static __device__ __forceinline__ void st_stream(uint16_t* p, uint16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void st_stream(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void st_stream(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void st_stream(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
#endif

// Sum over the 64 lanes of a wavefront without the LDS crossbar: __shfl_xor is ds_bpermute, which occupies the CU's LDS
// pipe (measured: 56 % LDS-busy in a kernel that reduces 17 values per wavefront with it).  Four DPP steps leave every
// lane of a 16-lane row with its row's sum (fixed order: deterministic for floats), four readlanes combine the rows.
template <class T> static __device__ __forceinline__ T wave_reduce_add(T v) {
    static_assert(sizeof(T) == 4, "32-bit values");
#define FSLIC_DPP_ADD(ctrl) v += __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
    FSLIC_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
    FSLIC_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
    FSLIC_DPP_ADD(0x141);    // row_half_mirror
    FSLIC_DPP_ADD(0x140);    // row_mirror
#undef FSLIC_DPP_ADD
    const int b = __builtin_bit_cast(int, v);
    const T r0 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(b, 16));
    const T r2 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Value of the lane to the left (lane 0 keeps its own), as a DPP wavefront shift rather than ds_bpermute
static __device__ __forceinline__ uint32_t lane_left(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// A lane predicate that every lane computes from ONE comparison is a 64-bit mask in scalar registers (the v_cmp writes it); mask
// logic is scalar-ALU work, and a mask becomes a lane predicate again at no cost (inverse ballot: the mask is the condition
// register).  A ballot of a COMPOUND bool goes through the vector ALU instead (the booleans are and-ed as masks, materialised by
// v_cndmask and compared again), and (mask >> lane) & 1 is a 64-bit vector shift: write ballot(a) & ballot(b) and lanes(mask).
static __device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
static __device__ __forceinline__ bool lanes(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

static __device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// floor(x / n) for x < 2^31, n >= 1, through the reciprocal r = floor((2^32 - 1) / n): the estimate is at most one short
// (the five round_int means of a cluster, src/fast-slic-common.h:62-65, cost one division instead of five)
static __device__ __forceinline__ uint32_t div_by(uint32_t x, uint32_t n, uint32_t r) {
    uint32_t q = __umulhi(x, r);
    uint32_t rest = x - q * n;
    if (rest >= n) { q++; rest -= n; }
    if (rest >= n) q++;
    return q;
}

// Which assign blocks (64 px x BH visited rows) can hold pixels of a cluster centred at (cy, cx): the blocks its
// (2S+1)^2 window overlaps, in block columns [bx0, bx1] and block rows [by0, by1] (any == false: no visited row).
// The block kernel and the cluster pass evaluate this identically, which is what pairs writer and reader of
// FrameDev::cpart.
// n / d for 0 <= n < 65536 (image rows) and the subsampling strides in use: 1, 2, 3 without a division -- n / 3 as a 24-bit multiply
// ((n * 43691) >> 17 is exact below 131072 and the product stays below 2^32); with a constant d the compiler would take
// v_mul_hi_u32, which issues at a quarter of the rate, with a run-time d the whole division sequence.
static __device__ __forceinline__ int div_rows(int n, int d) {
    return d == 1 ? n : d == 2 ? (n >> 1) : d == 3 ? (int)(__umul24((uint32_t)n, 43691u) >> 17) : n / d;
}
struct SlotGeom { int bx0, bx1, by0, by1; bool any; };
static __device__ __forceinline__ SlotGeom slot_geom(int cy, int cx, int S, int H, int W, int rem, int stride, int BH, int Hv) {
    SlotGeom g;
    g.bx0 = max(cx - S, 0) >> 6;
    g.bx1 = min(cx + S, W - 1) >> 6;
    const int t = max(cy - S, 0) - rem;
    const int v0 = t <= 0 ? 0 : div_rows(t + stride - 1, stride);
    const int u = min(cy + S, H - 1) - rem;
    const int v1 = u < 0 ? -1 : min(div_rows(u, stride), Hv - 1);
    g.any = v0 <= v1;
    const int bsh = 31 - __clz(BH);          // BH = 4 wavefronts x {4, 8, 16, 32} rows: a power of two (no division in the cluster pass)
    g.by0 = v0 >> bsh;
    g.by1 = g.any ? v1 >> bsh : -1;
    return g;
}

// FrameDev::cpart addressing (see kernels.h, FrameDev::nbx).  NY - 1 for a pass of the given geometry: NY = the block rows a
// (2S+1)-row window can overlap (its visited rows span at most 2S / stride indices, i.e. at most (2S / stride) / BH + 2 blocks of BH
// visited rows), rounded up to a power of two.
static __device__ __forceinline__ uint32_t cpart_rows_mask(int S, int stride, int BH) {
    const int n = (div_rows(2 * S, stride) >> (31 - __clz(BH))) + 2;
    return (1u << (32 - __clz(n - 1))) - 1u;
}
static __device__ __forceinline__ uint32_t cpart_slot(uint32_t by, uint32_t bx, uint32_t ymask, uint32_t nbx) {
    return __umul24(by & ymask, nbx) + (bx & (nbx - 1u));
}

// integer centroid sums of one pixel through global atomics (generic kernels, stale-label pixels): count, y, x, L, a, b
static __device__ __forceinline__ void global_accumulate(uint32_t* sums, uint32_t k, uint32_t y, uint32_t x, uint32_t pix) {
    uint32_t* s = sums + 6 * (size_t)k;
    atomicAdd(s + 0, 1u);
    atomicAdd(s + 1, y);
    atomicAdd(s + 2, x);
    atomicAdd(s + 3, pix & 255u);
    atomicAdd(s + 4, (pix >> 8) & 255u);
    atomicAdd(s + 5, (pix >> 16) & 255u);
}

// preemptive mode (src/context.cpp:304-343): unless every cluster is active, only pixels of active (2S x 2S) cells take part in update()
static __device__ __forceinline__ bool in_update(const FrameDev& f, int y, int x) {
    return !f.cell_act || f.pre_flags[0] || f.cell_act[(y / (2 * f.S)) * f.pre_cw + x / (2 * f.S)];
}

// pixel window of a tile widened by S: a cluster whose centre lies inside can own a pixel of the tile
struct TileWindow { int wy_lo, wy_hi, wx_lo, wx_hi; };
static __device__ __forceinline__ bool in_window(const ClusterRec& r, const TileWindow& w) {
    const int cy = (int)(r.yx >> 16), cx = (int)(r.yx & 0xFFFF);
    return cy >= w.wy_lo && cy <= w.wy_hi && cx >= w.wx_lo && cx <= w.wx_hi;
}

// The candidate list of a block of pixels, built by ONE wavefront: every live record of the coarse cells the window can
// see plus the spill list, ballot-compacted into dst (3 x 64 words: yx, lab, tag) in arbitrary order.  Returns the
// number of live records inside the window (more than 64: the list overflowed, only the first 64 were stored).
static __device__ __forceinline__ int gather_candidates(const FrameDev& f, const TileWindow& win, uint32_t* dst, int lane) {
    const uint32_t novf = *f.ovf_cnt[0];
    const uint32_t gen = f.generation();
    const int sh = f.cs_shift;
    const ClusterRec* __restrict__ citems = f.cell_items[0];
    const ClusterRec* __restrict__ oitems = f.ovf_items[0];
    const unsigned long long lowmask = (1ull << lane) - 1ull;
    const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, f.W - 1) >> sh;
    const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
    const int ncxr = cxb - cxa + 1;
    int cnt = 0;
    auto visit = [&](const ClusterRec& rec, bool keep) {
        const unsigned long long m = __ballot(keep);
        const int pos = cnt + __popcll(m & lowmask);
        if (keep && pos < 64) { dst[pos] = rec.yx; dst[64 + pos] = rec.lab; dst[128 + pos] = rec.tag; }
        cnt += __popcll(m);
    };
    for (int cx0 = 0; cx0 < ncxr; cx0 += 8) {
        const int cxi = cx0 + (lane >> 3);
        const bool colok = cxi < ncxr;
        const ClusterRec* base = citems + ((size_t)cya * f.ncx + cxa + min(cxi, ncxr - 1)) * kCellCap + (lane & 7);
        for (int cy0 = cya; cy0 <= cyb; cy0 += 6) {
            ClusterRec rr[6];
            bool okr[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {      // unconditional loads (rows past the range re-read the last one)
                okr[u] = colok && (cy0 + u) <= cyb;
                rr[u] = base[(size_t)min(u, cyb - cy0) * f.ncx * kCellCap];
            }
            base += 6 * (size_t)f.ncx * kCellCap;
#pragma unroll
            for (int u = 0; u < 6; ++u) visit(rr[u], okr[u] && rr[u].pad == gen && in_window(rr[u], win));
        }
    }
    for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
        const uint32_t i = i0 + lane;
        ClusterRec rec;
        rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
        if (i < novf) rec = oitems[i];
        visit(rec, i < novf && in_window(rec, win));
    }
    return cnt;
}

}  // namespace fslic
