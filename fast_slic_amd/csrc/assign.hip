// assign.hip -- assign / fused update kernels (src/context.cpp:199-354, src/arch/x64/avx2.h)
// Hand-written gfx950 (CDNA4, wave64) kernels of the SLIC hot path; see kernels.h for the launch interface and
// DESIGN.md for the design.  All arithmetic on this path is integer, so results are bit-identical to the reference.
#include "device_common.h"
#include <type_traits>

namespace fslic {

// Nothing here is a translation of the reference's AVX2/NEON code.  The reference *scatters*: each cluster
// min-updates a (2S+1)^2 window, clusters visited in (phase, k) order (src/context.cpp:199-298,
// src/arch/x64/avx2.h:89-187).  These kernels *gather*: every pixel takes the lexicographic minimum of
// (distance, phase, k) over the clusters whose window covers it, which is the same function of the inputs and needs
// no min_dists plane in memory.
//   k_assign_blk      the default and the roofline kernel (block-level candidate list, packed 16-bit keys)
//   k_assign          32-bit keys, one wavefront per tile: serves what the packed kernel does not take (distances
//                     >= 1024, spatial table too large for LDS, subsample_stride > 3)
//   k_assign_generic  one thread per pixel straight from the coarse bins: fallback (S < 8, candidate-list
//                     overflow) and independent cross-check in the tests
// =============================================================================================
// 32-bit assign kernel.  One WAVEFRONT = one tile of 64 columns x R visited rows; the four
// wavefronts of a block are independent (no barrier, no shared list), so they drift apart in time and the
// loads of one overlap the argmin of another.
// Lane = column, so |x - cx| and the x window test are per-lane-per-candidate, while |y - cy| and the y window
// test are wave-uniform per row and live in SGPRs.  Per pixel-candidate the VALU does
//     a   = A + B                      (spatial term, 16 fractional bits, window folded in as kHuge)
//     w   = v_sad_hi_u8(pix, clab, a)  ((|dL|+|da|+|db|) << 16) + a
//     key = v_and_or(w, 0xffff0000, idx)   idx = rank of the candidate in (phase, k) order
//     best = min(best, key)
// so `best` ends up as the lexicographic minimum of (distance, phase, k).  Pixels that no window covers are
// not written: the previous label persists (src/context.cpp:138-145 clears the assignment plane once per
// call, not per iteration).
// Candidate list per wavefront: every slot of every coarse cell the tile's pixels can see (a slot is live
// when its generation stamp equals this pass's: one round trip fetches count and payload) plus the spill
// list; ballot compaction into 64 LDS slots; exact prune of clusters with identical (y,x,L,a,b) to an
// earlier-ordered one (same distance everywhere, loses every tie); rank by (phase,k) with readlane
// broadcasts.  More than 64 live candidates (a pile-up of seeds on one pixel): the records are streamed
// again and compared unsorted with a two-register (distance, phase<<16|k) key; no capacity limit.
// Algorithmic traffic: 4 B read + 2 B written per visited pixel; min-distances never leave VGPRs.
// With FUSE the centroid sums of update() (src/context.cpp:301-354) for the same rows are accumulated in a
// wave-private LDS region (privatised copies) and flushed with one global atomic per (candidate, quantity).
// =============================================================================================


template <int R, bool FUSE, bool LUT>
__global__ __launch_bounds__(256) void k_assign(FrameDev f, int rem, int stride, int Hv, int buf, int sbuf) {
    extern __shared__ uint32_t s_lut[];
    __shared__ uint32_t s_rawbuf[kWavesPerBlock][3 * 64];
    __shared__ uint4 s_listbuf[kWavesPerBlock][64];
    __shared__ uint32_t s_accbuf[FUSE ? kWavesPerBlock * kAccSlots * 6 * kAccCopies : 1];

    f.select(blockIdx.z);
    const uint32_t gen = f.generation();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    uint32_t* const s_raw = s_rawbuf[wave];
    uint4* const s_list = s_listbuf[wave];
    uint32_t* const s_acc = s_accbuf + (FUSE ? wave * (kAccSlots * 6 * kAccCopies) : 0);
    const int W = f.W, S = f.S;
    const int x0 = blockIdx.x * kTileW;
    const int x = x0 + lane;
    const int vrow_w0 = (blockIdx.y * kWavesPerBlock + wave) * R;
    const bool okx = x < W;
    if (LUT) {
        for (int i = tid; i < f.lut_words; i += 256) s_lut[i] = f.lut[i];
        __syncthreads();                     // the only barrier of the kernel
    }
    if (vrow_w0 >= Hv) return;               // wave-uniform
    const int yw_lo = rem + vrow_w0 * stride;
    const int nrows = min(R, Hv - vrow_w0);
    const uint32_t novf = *f.ovf_cnt[0];   // issued early, consumed after the cell gather

    // 1. pixel loads first; they do not depend on the candidate list
    uint32_t pix[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = okx && r < nrows;
        pix[r] = ok ? f.lab[(size_t)(yw_lo + r * stride) * W + x] : 0u;
    }
    if (FUSE) {
#pragma unroll
        for (int i = 0; i < kAccSlots * 6 * kAccCopies / 64; ++i) s_acc[i * 64 + lane] = 0u;
    }

    // 2. gather
    TileWindow win;
    win.wy_lo = yw_lo - S; win.wy_hi = yw_lo + (nrows - 1) * stride + S;
    win.wx_lo = x0 - S;    win.wx_hi = min(x0 + kTileW - 1, W - 1) + S;
    const int sh = f.cs_shift;
    const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, W - 1) >> sh;
    const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
    const int ncxr = cxb - cxa + 1, ncyr = cyb - cya + 1;
    const int total = ncxr * ncyr * kCellCap;
    // exact quotient for the small operands here -- for ncxr >= 2: 2^32 / 1 does not fit (the conversion saturates, + 1 wraps to 0 and every
    // cell landed in cell row 0).  One cell column is what a tile at the image's left edge sees when 63 + S is less than the cell side,
    // S = 129 .. 192 or 257 .. 448, on the inputs this kernel serves (spatial term >= 259): candidates of the lower cell rows were
    // never fetched there until round 6 (tests/fuzz_parity.py, FUZZ_SHAPES=big, seed 5 case 11: 2090x1412 K=131 compactness 80).
    const uint32_t inv = ncxr > 1 ? (uint32_t)(4294967296.0f / (float)ncxr) + 1u : 0u;
    const ClusterRec* __restrict__ citems = f.cell_items[0];
    const ClusterRec* __restrict__ oitems = f.ovf_items[0];
    const unsigned long long lowmask = (1ull << lane) - 1ull;
    auto cell_slot = [&](int e) -> size_t {
        const uint32_t ce = (uint32_t)e >> 3;
        uint32_t cyi = ncxr > 1 ? __umulhi(ce, inv) : ce;
        if (cyi * (uint32_t)ncxr > ce) cyi--;                // guard the float reciprocal
        const uint32_t cxi = ce - cyi * (uint32_t)ncxr;
        return ((size_t)(cya + (int)cyi) * f.ncx + cxa + (int)cxi) * kCellCap + (e & 7);
    };
    int nraw = 0;
    for (int c0 = 0; c0 < total; c0 += 256) {
        ClusterRec rec[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = c0 + u * 64 + lane;
            live[u] = e < total;
            if (live[u]) rec[u] = citems[cell_slot(e)];
            else { rec[u].yx = 0; rec[u].lab = 0; rec[u].tag = 0; rec[u].pad = 0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool keep = live[u] && rec[u].pad == gen && in_window(rec[u], win);
            const unsigned long long m = __ballot(keep);
            const int pos = nraw + __popcll(m & lowmask);
            if (keep && pos < 64) { s_raw[pos] = rec[u].yx; s_raw[64 + pos] = rec[u].lab; s_raw[128 + pos] = rec[u].tag; }
            nraw += __popcll(m);
        }
    }
    for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
        const uint32_t i = i0 + lane;
        ClusterRec rec;
        rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
        if (i < novf) rec = oitems[i];
        const bool keep = i < novf && in_window(rec, win);
        const unsigned long long m = __ballot(keep);
        const int pos = nraw + __popcll(m & lowmask);
        if (keep && pos < 64) { s_raw[pos] = rec.yx; s_raw[64 + pos] = rec.lab; s_raw[128 + pos] = rec.tag; }
        nraw += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();   // staging writes above, cross-lane reads below: keep program order

    const uint32_t M = f.M, Coff = f.C;
    uint32_t hi16;   // 0xffff0000 in a VGPR so that (w & hi16) | c is one v_and_or_b32 with c in an SGPR
    asm volatile("v_mov_b32 %0, 0xffff0000" : "=v"(hi16));
    // spatial + colour term of one candidate for row r, window folded in: < kHuge iff the pixel is covered
    auto spatial_A = [&](int cx) -> uint32_t {
        const int dx = x - cx;
        const uint32_t ux = (uint32_t)(dx < 0 ? -dx : dx);
        if (LUT) return ux <= (uint32_t)S ? ux : (uint32_t)f.lut_oow;
        return ux <= (uint32_t)S ? __umul24(ux, M) + Coff : kHuge;
    };
    auto dist_w = [&](int r, int cy, uint32_t clab, uint32_t A) -> uint32_t {
        const int dy = yw_lo + r * stride - cy;
        const uint32_t uy = (uint32_t)(dy < 0 ? -dy : dy);                   // SALU
        if (LUT) {
            const uint32_t B = (uy <= (uint32_t)S ? uy : (uint32_t)f.lut_oow) * (uint32_t)f.lut_row;
            return __builtin_amdgcn_sad_hi_u8(pix[r], clab, s_lut[A + B]);
        }
        const uint32_t B = uy <= (uint32_t)S ? uy * M : kHuge;
        return __builtin_amdgcn_sad_hi_u8(pix[r], clab, A + B);
    };
    const int yw_hi = yw_lo + (R - 1) * stride;

    if (nraw > 64) {
        // ---- unsorted streaming path (no capacity limit): lexicographic (distance, phase<<16|k) in two registers
        uint32_t bw[R], bt[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { bw[r] = 0xFFFFFFFFu; bt[r] = 0xFFFFFFFFu; }
        auto stream = [&](const ClusterRec& rec, bool keep) {
            unsigned long long m = __ballot(keep);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t cyx = __builtin_amdgcn_readlane(rec.yx, src);
                const uint32_t clab = __builtin_amdgcn_readlane(rec.lab, src);
                const uint32_t tag = __builtin_amdgcn_readlane(rec.tag, src);
                const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
                if (cy + S < yw_lo || cy - S > yw_hi) continue;
                const uint32_t A = spatial_A(cx);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t w = dist_w(r, cy, clab, A) & hi16;
                    const bool better = (w < bw[r]) || (w == bw[r] && tag < bt[r]);
                    bw[r] = better ? w : bw[r];
                    bt[r] = better ? tag : bt[r];
                }
            }
        };
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int e = c0 + lane;
            ClusterRec rec;
            rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
            if (e < total) rec = citems[cell_slot(e)];
            stream(rec, e < total && rec.pad == gen && in_window(rec, win));
        }
        for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
            const uint32_t i = i0 + lane;
            ClusterRec rec;
            rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
            if (i < novf) rec = oitems[i];
            stream(rec, i < novf && in_window(rec, win));
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!(okx && r < nrows)) continue;
            const int y = yw_lo + r * stride;
            const size_t p = (size_t)y * W + x;
            uint32_t lbl;
            if (bw[r] < kHuge) { lbl = bt[r] & 0xFFFFu; f.labels[p] = (uint16_t)lbl; }
            else if (f.first_visit(y)) { lbl = 0xFFFFu; f.labels[p] = (uint16_t)0xFFFFu; }      // (lazy reset of the plane, FrameDev::fv_mod)
            else lbl = f.labels[p];
            if (FUSE && lbl != 0xFFFFu) global_accumulate(f.sums[0], lbl, (uint32_t)y, (uint32_t)x, pix[r]);
        }
        return;
    }

    // 3. exact prune + rank (one record per lane, pair tests by readlane broadcast)
    int n;
    {
        const bool act = lane < nraw;
        const uint32_t yx_j = act ? s_raw[lane] : 0u;
        const uint32_t lab_j = act ? s_raw[64 + lane] : 0u;
        const uint32_t tag_j = act ? s_raw[128 + lane] : 0xFFFFFFFFu;
        bool dup = false;
        for (int i = 0; i < nraw; ++i) {
            const uint32_t ty = __builtin_amdgcn_readlane(yx_j, i);
            const uint32_t tl = __builtin_amdgcn_readlane(lab_j, i);
            const uint32_t tt = __builtin_amdgcn_readlane(tag_j, i);
            dup = dup || (act && (tt < tag_j) && (ty == yx_j) && (tl == lab_j));
        }
        const unsigned long long dupmask = __ballot(dup);
        int rank = 0;
        for (int i = 0; i < nraw; ++i) {
            const uint32_t tt = __builtin_amdgcn_readlane(tag_j, i);
            rank += (int)((((dupmask >> i) & 1ull) == 0ull) && tt < tag_j);
        }
        if (act && !dup) s_list[rank] = make_uint4(yx_j, lab_j, tag_j & 0xFFFFu, 0u);
        n = nraw - __popcll(dupmask);
    }
    __builtin_amdgcn_wave_barrier();

    // 4. the argmin over the sorted list
    uint32_t best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) best[r] = 0xFFFFFFFFu;
    uint2 nxt = *reinterpret_cast<const uint2*>(&s_list[0]);
#pragma unroll 1
    for (int c = 0; c < n; ++c) {
        const uint2 rec = nxt;
        nxt = *reinterpret_cast<const uint2*>(&s_list[min(c + 1, 63)]);   // prefetch: hides the LDS latency
        const uint32_t cyx = rfl(rec.x);
        const uint32_t clab = rfl(rec.y);
        const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
        const uint32_t A = spatial_A(cx);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t w = dist_w(r, cy, clab, A);
            best[r] = min(best[r], LUT ? (w | (uint32_t)c) : ((w & hi16) | (uint32_t)c));
        }
    }

    // 5. write labels (+ fused centroid sums)
    uint32_t slot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = yw_lo + r * stride;
        const bool ok = okx && r < nrows;
        const bool found = best[r] < kHuge;
        const uint32_t idx = found ? (best[r] & 0xFFFFu) : 0u;
        const uint32_t k = s_list[idx].z;
        slot[r] = 0xFFFFFFFFu;
        if (ok) {
            const size_t p = (size_t)y * W + x;
            if (found) {
                f.labels[p] = (uint16_t)k;
                if (FUSE) {
                    if (idx < (uint32_t)kAccSlots) slot[r] = idx;
                    else global_accumulate(f.sums[0], k, (uint32_t)y, (uint32_t)x, pix[r]);
                }
            } else if (f.first_visit(y)) {
                f.labels[p] = (uint16_t)0xFFFFu;      // nobody has looked at this row in this call yet: unlabelled (lazy reset, FrameDev::fv_mod)
            } else if (FUSE) {
                const uint32_t old = f.labels[p];     // stale label still takes part in update()
                if (old != 0xFFFFu) global_accumulate(f.sums[0], old, (uint32_t)y, (uint32_t)x, pix[r]);
            }
        }
    }
    if (FUSE) {
        bool uni = slot[0] != 0xFFFFFFFFu;
#pragma unroll
        for (int r = 1; r < R; ++r) uni = uni && (slot[r] == slot[0]);
        const uint32_t copy = (uint32_t)lane & (kAccCopies - 1);
        if (uni) {   // whole column in one superpixel: one set of LDS atomics for R pixels
            uint32_t sL = 0, sa = 0, sb = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) { sL += pix[r] & 255u; sa += (pix[r] >> 8) & 255u; sb += (pix[r] >> 16) & 255u; }
            uint32_t* a = s_acc + (slot[0] * 6) * kAccCopies + copy;
            atomicAdd(a + 0 * kAccCopies, (uint32_t)R);
            atomicAdd(a + 1 * kAccCopies, (uint32_t)(R * yw_lo + stride * (R * (R - 1) / 2)));
            atomicAdd(a + 2 * kAccCopies, (uint32_t)(R * x));
            atomicAdd(a + 3 * kAccCopies, sL);
            atomicAdd(a + 4 * kAccCopies, sa);
            atomicAdd(a + 5 * kAccCopies, sb);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (slot[r] != 0xFFFFFFFFu) {
                    uint32_t* a = s_acc + (slot[r] * 6) * kAccCopies + copy;
                    atomicAdd(a + 0 * kAccCopies, 1u);
                    atomicAdd(a + 1 * kAccCopies, (uint32_t)(yw_lo + r * stride));
                    atomicAdd(a + 2 * kAccCopies, (uint32_t)x);
                    atomicAdd(a + 3 * kAccCopies, pix[r] & 255u);
                    atomicAdd(a + 4 * kAccCopies, (pix[r] >> 8) & 255u);
                    atomicAdd(a + 5 * kAccCopies, (pix[r] >> 16) & 255u);
                }
            }
        }
        // flush (wave-private region: the LDS executes one wavefront's operations in order, no barrier needed)
        __builtin_amdgcn_wave_barrier();
        const int nacc = min(n, kAccSlots) * 6;
        for (int i = lane; i < nacc; i += 64) {
            uint32_t v = 0;
#pragma unroll
            for (int c = 0; c < kAccCopies; ++c) v += s_acc[i * kAccCopies + c];
            if (v) {
                const int sl = i / 6;
                atomicAdd(f.sums[0] + 6 * (size_t)s_list[sl].z + (i - sl * 6), v);
            }
        }
    }
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// =============================================================================================
// Block-level packed assign kernel (the default, and the roofline kernel).
// One block = 64 columns x (4 wavefronts x R) visited rows.  The key of a pixel-candidate is 16 bits,
// (distance << 6) | idx, idx = rank of the candidate in (phase, k) order within the block's list (<= 64 entries),
// distance < 1024; two rows share one VALU instruction:
//     s   = v_sad_u8   (pix[r],   clab, spatial pair)   the table's (row r, row r+1) word is the accumulator: + colour of row r
//     s   = v_sad_hi_u8(pix[r+1], clab, s)              + colour of row r+1 in the high half: two distances packed
//     key = v_pk_mad_u16(s, 64, idx|idx<<16) clamp      outside the window the table holds kTabOut: saturates to 0xffff, never wins
//     best= v_pk_min_u16(best, key)
// i.e. 2 VALU per pixel-candidate + one aligned LDS read per pair (or per 8 rows).  The spatial pair comes from an LDS copy of the
// reference's patch (src/context.cpp:22-40) laid out so that the rows of a wavefront are immediate offsets from one
// per-lane address (two layouts, see VT below).  The prologue is paid once per BLOCK (assign_blk2_body: the candidate list is
// gathered and ranked by all four wavefronts, the table arrives by LDS-DMA); then every wavefront holds the list in registers
// (lane j = candidate j), ballots the candidates whose window reaches its own rows and walks that bit mask: cluster fields arrive
// by v_readlane, no LDS latency per candidate.
// Fused update (src/context.cpp:301-354): block-level LDS accumulators (privatised copies, LDS atomics), then ONE
// plain store per (candidate, quantity) into FrameDev::cpart at the entry reserved for (cluster, this block).
// No global atomic on the common path: device-scope atomics cost 21 of the 39 us of the per-wavefront-flush
// version of this pass (8 x 720p frames).  Pixels that no window covers keep their label and reach the sums through
// global atomics (rare).
// Algorithmic traffic: 4 B read + 2 B written per visited pixel.
// =============================================================================================
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
// (Experiment builds -- make VAR=<name> DEFS=-DFSLIC_EXP=<bits>, kernels.h -- carry A/B forms of this kernel while they are being
// measured, scripts/gpu_trip.sh; what round 4 measured is in profiles/r04_assign_experiments.txt.)
constexpr int kRawBuffer = 0x00020000;     // 4th dword of a raw (stride 0) buffer descriptor on gfx9-family targets: DATA_FORMAT = 32
constexpr int kAuxSc1 = 16;                // cache policy bit of the buffer builtins: sc1 (write through at agent scope)

// key pair = min(dist * 64 + idx, 0xffff) for two rows at once: v_pk_mad_u16 with the clamp bit saturates the full-width
// result, so a distance carrying the out-of-window marker (kTabOut, >= 1024) or an idx operand of 0xffff gives 0xffff
static __device__ __forceinline__ us2 key_pair(uint32_t dist01, uint32_t idx01) {
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 64, %2 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "v"(dist01), "s"(idx01));
    return __builtin_bit_cast(us2, r);
}


// s_bitreplicate_b64_b32: every bit of the source twice, in order (bit i -> bits 2i, 2i+1)
static __device__ __forceinline__ unsigned long long sbitrep(uint32_t v) {
    unsigned long long r;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(r) : "s"(v));
    return r;
}
// One 32-bit word per pair of rows: 0xffff in the half of every row whose bit is set in `inv` (R bits), each bit
// replicated 16 times by four rounds of s_bitreplicate (scalar unit only).
// best = min(best, key) IN PLACE: with the plain builtin the compiler keeps the loop-carried minima and the pass's
// results in two register sets and copies one into the other (8-16 v_mov per candidate, a fifth of the pass)
static __device__ __forceinline__ void min_in_place(us2& best, us2 key) {
    uint32_t b = __builtin_bit_cast(uint32_t, best);
    asm("v_pk_min_u16 %0, %0, %1" : "+v"(b) : "v"(__builtin_bit_cast(uint32_t, key)));
    best = __builtin_bit_cast(us2, b);
}

template <int R>
static __device__ __forceinline__ void row_masks(uint32_t inv, uint32_t (&m)[R / 2]) {
    static_assert(R == 8 || R == 16 || R == 32, "row masks are built for 8, 16 or 32 rows");
    if constexpr (R == 32) {
        uint32_t lo[8], hi[8];
        row_masks<16>(inv & 0xFFFFu, lo);
        row_masks<16>(inv >> 16, hi);
#pragma unroll
        for (int q = 0; q < 8; ++q) { m[q] = lo[q]; m[8 + q] = hi[q]; }
        return;
    }
    const unsigned long long r1 = sbitrep(inv);                 // x2
    const unsigned long long r2 = sbitrep((uint32_t)r1);        // x4  (R <= 16: 32 significant bits of r1)
    if (R == 8) {
        const unsigned long long r3 = sbitrep((uint32_t)r2);    // x8: 64 bits
        const unsigned long long a = sbitrep((uint32_t)r3), b = sbitrep((uint32_t)(r3 >> 32));
        m[0] = (uint32_t)a; m[1] = (uint32_t)(a >> 32); m[2] = (uint32_t)b; m[3] = (uint32_t)(b >> 32);
    } else {
        const unsigned long long r3a = sbitrep((uint32_t)r2), r3b = sbitrep((uint32_t)(r2 >> 32));   // x8: 128 bits
        const unsigned long long q0 = sbitrep((uint32_t)r3a), q1 = sbitrep((uint32_t)(r3a >> 32));
        const unsigned long long q2 = sbitrep((uint32_t)r3b), q3 = sbitrep((uint32_t)(r3b >> 32));
        m[0] = (uint32_t)q0; m[1] = (uint32_t)(q0 >> 32); m[2] = (uint32_t)q1; m[3] = (uint32_t)(q1 >> 32);
        m[R / 2 - 4] = (uint32_t)q2; m[R / 2 - 3] = (uint32_t)(q2 >> 32); m[R / 2 - 2] = (uint32_t)q3; m[R / 2 - 1] = (uint32_t)(q3 >> 32);
    }
}

// VT (Manhattan patch only): the spatial term is a function of d = |dx| + |dy| alone, so the LDS table derives from the
// one-dimensional, V-shaped  V[t] = patch(|t - Cc|)  (kTabOut beyond 2S) instead of the (S+2) x (2S+2pad+1)
// column-major patch copy (26 KB at S = 72, re-read by every block).  Row r of a candidate reads V[Cc + dy_r + ux]
// when dy_r >= 0 and V[Cc + dy_r - ux] when dy_r < 0: either way the index advances with the row, so the rows are
// still fixed offsets from ONE per-lane base as long as they lie on one side of the centre; a candidate whose centre
// row lies inside the wavefront's rows is walked twice (rows above, rows below).  Lanes outside the x window use
// ux = 2S+1, which lands beyond 2S for every row; rows outside the y window are switched off by a per-pair scalar
// mask OR-ed into the idx operand on the scalar unit (0xffff + anything saturates: no extra VALU).
#if defined(FSLIC_LAB) && FSLIC_LAB == 4      // lab build 4: phase time stamps (100 MHz ticks since the block's start), 16 words per (pass, block)
#define ASG_STAMP(i_, cond_) do { if ((cond_) && lane == 0) { asm volatile("" ::: "memory"); asg_w[(i_)] = (uint32_t)__builtin_amdgcn_s_memrealtime() - asg_t0; asm volatile("" ::: "memory"); } } while (0)
#define ASG_PUT(i_, cond_, v_) do { if ((cond_) && lane == 0) asg_w[(i_)] = (uint32_t)(v_); } while (0)
#else
#define ASG_STAMP(i_, cond_) do { } while (0)
#define ASG_PUT(i_, cond_, v_) do { } while (0)
#endif
// Analysis listing only (scripts/isa_budget.py compiles this file with -DFSLIC_MARKS -S): a comment line in the assembly at every
// phase boundary of the block kernel, so that the static instruction count of each phase can be read off the ISA.  The product build
// emits nothing here.
#ifdef FSLIC_MARKS
#define ASG_MARK(name_) asm volatile("; @@MARK " name_ ::: "memory")
#else
#define ASG_MARK(name_) do { } while (0)
#endif
// =============================================================================================
// The block assign kernel's body (second form, round 3; the first form -- one wavefront builds and sorts the list while three
// stage the table through registers -- is in the history and in profiles/r02_*).  The phase stamps of lab build 4
// (profiles/r03_lab_assign_phases.txt) showed the prologue to be a third of a block's life as one wavefront's serial work:
//   * the candidate list is gathered by ALL four wavefronts (one 16-byte bin slot per thread and trip, ballot compaction into
//     per-wavefront parts of an LDS array: no atomics, no serial walk over six cell rows by one wavefront)
//   * the (phase, k) ranks are counted in four partial sums, one per wavefront, and meet in LDS (second barrier); the list stays
//     in arrival order, the rank of a candidate rides in the top byte of its colour word
//   * the spatial table is brought in by LDS-DMA (global_load_lds_dwordx4, 1 KB per wavefront instruction): no staging
//     registers and no ds_write pass
// =============================================================================================
// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to LDS at `lds_byte_addr` + lane * 16 (M0 carries the
// LDS base and is restored).  Issued through inline assembly on purpose: with the builtin the compiler waits for EVERY
// outstanding load (vmcnt(0)) at the next use of an ordinary load's result; like this its counters simply do not include the
// piece (loads return in order, so its own waits stay sufficient) and the one wait the piece needs is stated before the barrier.
static __device__ __forceinline__ void lds_dma_16(const void* gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// ---- fused cluster pass (FBIN): what the LAST assign block to deliver partial sums of a cluster does for it ----------------------
// src/context.cpp:356-373 (round_int means of the pass's sums, num_members written even when 0, position kept then) and the binning
// for the next pass (the role of k_bin_clusters<1>, clusters.hip).  One lane per cluster; lanes of a wavefront finalise different
// clusters side by side.  Every read of another block's data is an agent-scope load (sc1): the writers stored write-through
// and drained before they counted themselves in (MI355X guide: sc1 loads may stand in for the acquire when the producer stored sc1).
static __device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ unsigned long long ld_agent64(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void bin_insert_next(const FrameDev& f, uint32_t k, uint32_t yx, uint32_t lab) {
    const int y = (int)(yx >> 16), x = (int)(yx & 0xFFFFu);
    const uint32_t phase = 2u * (div_by((uint32_t)y, (uint32_t)f.T, f.T_rcp) & 1u) + (div_by((uint32_t)x, (uint32_t)f.T, f.T_rcp) & 1u);      // visit order of the reference scatter: phase, then k
    ClusterRec rec;
    rec.yx = yx; rec.lab = lab; rec.tag = (phase << 16) | k; rec.pad = f.generation() + 1u;       // the NEXT pass's stamp
    f.cl_tag[k] = rec.tag;
    const int cell = (y >> f.cs_shift) * f.ncx + (x >> f.cs_shift);
    const uint32_t slot = atomicAdd(&f.cell_cnt[1][cell], 1u);
    if (slot < (uint32_t)kCellCap) f.cell_items[1][(size_t)cell * kCellCap + slot] = rec;
    else f.ovf_items[1][atomicAdd(f.ovf_cnt[1], 1u)] = rec;
}
static __device__ __forceinline__ void fused_finalize(const FrameDev& f, uint32_t k, uint32_t yx, uint32_t lab, uint32_t nx, uint32_t ny, uint32_t b0, uint32_t ymask) {
    // the pass's sums of cluster k: what went through global atomics plus one entry per assign block its window overlaps
    uint32_t* const s0 = f.sums[0] + 6 * (size_t)k;
    uint32_t s[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) s[q] = ld_agent(s0 + q);
    // The entries four at a time with all their loads in flight (entry by entry the finaliser was a chain of nx * ny dependent round
    // trips at the end of a block's life: 2.8 us of the 9 us a one-frame launch takes); e / nx through a 16-bit reciprocal (guarded).
    const char* const cp = reinterpret_cast<const char*>(f.cpart + (size_t)k * f.cpart_ns * 6);
    const uint32_t ne = nx * ny;
    const uint32_t inv_nx = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)nx)) + 1u;
    constexpr int kBatch = 4;
    for (uint32_t e0 = 0; e0 < ne; e0 += kBatch) {
        unsigned long long v[kBatch][3];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const uint32_t e = e0 + (uint32_t)u;
            uint32_t ey = __umul24(e, inv_nx) >> 16;
            if (__umul24(ey, nx) > e) ey--;
            const uint32_t ex = e - __umul24(ey, nx);
            v[u][0] = v[u][1] = v[u][2] = 0ull;
            if (e < ne) {
                const char* p = cp + __umul24(cpart_slot((b0 >> 16) + ey, (b0 & 0xFFFFu) + ex, ymask, (uint32_t)f.nbx), 24u);      // 24-byte entries, 8-byte aligned
                v[u][0] = ld_agent64(p); v[u][1] = ld_agent64(p + 8); v[u][2] = ld_agent64(p + 16);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            s[0] += (uint32_t)v[u][0]; s[1] += (uint32_t)(v[u][0] >> 32); s[2] += (uint32_t)v[u][1]; s[3] += (uint32_t)(v[u][1] >> 32);
            s[4] += (uint32_t)v[u][2]; s[5] += (uint32_t)(v[u][2] >> 32);
        }
    }
    const uint32_t n = s[0];
    f.cl_n[k] = n;
    if (n != 0) {
        const uint32_t h = n >> 1, r = 0xFFFFFFFFu / n;
        int y = (int)div_by(s[1] + h, n, r), x = (int)div_by(s[2] + h, n, r);
        y = min(max(y, 0), f.H - 1);            // (a centroid of in-image pixels is in the image: no-op on valid sums, see k_bin_clusters)
        x = min(max(x, 0), f.W - 1);
        const uint32_t L = div_by(s[3] + h, n, r), a = div_by(s[4] + h, n, r), b = div_by(s[5] + h, n, r);
        yx = ((uint32_t)y << 16) | (uint32_t)x;
        lab = (L & 255u) | ((a & 255u) << 8) | ((b & 255u) << 16);
        f.cl_lab[k] = lab;
        f.cl_moved[k] = 1;
        f.cl_yx[k] = yx;                        // positions live in ONE buffer here (only this lane ever writes entry k during the pass)
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) s0[q] = 0u;     // this accumulator set serves the pass after next
    f.cl_arrive[k] = 0u;
    bin_insert_next(f, k, yx, lab);
    // A cluster kept out of the bins because it equals its predecessor (cl_shadow, see k_bin_clusters) has no block working for
    // it; its predecessor's finaliser looks after it.  While the predecessor receives no member it stays identical and the
    // follower stays out (exact: it can never win a pixel).  Once the predecessor has members the follower goes back into the
    // bins (src/context.cpp:362-364: no members, position unchanged) -- the rest of a run of equals keeps hiding behind IT.
    if (n != 0 && k + 1u < (uint32_t)f.K && f.cl_shadow[k + 1u]) {
        f.cl_shadow[k + 1u] = 0u;
        bin_insert_next(f, k + 1u, f.cl_yx[k + 1u], f.cl_lab[k + 1u]);
    }
}

// Blocks -> tiles: no remapping inside a frame.  The hardware hands consecutive workgroups (x fastest) to the
// eight XCDs in turn, each with its own L2, so the eight neighbours that share a frame's bins and table rows sit on eight L2s
// (1.32 x the algorithmic bytes at 1280x720).  An XCD-chunked order (an XCD works through a contiguous eighth of the tiles) was
// built and measured in round 3 (profiles/r03_xcd_mapping.txt): traffic of the fused 8 x 1280x720 launch 1.15 x, its duration
// unchanged (11.8 vs 11.5 us), the HBM-bound 3840x2160 launches 10 - 18 % SLOWER (tiles that run together then lie in eight
// frames instead of side by side in memory).  Removed.
struct BlkMap { uint32_t gx, gy; };
static inline BlkMap blk_map(dim3 g) { BlkMap m; m.gx = g.x; m.gy = g.y; return m; }
// The launch grid is (tile columns, FRAMES, tile rows): workgroups are handed out x fastest, so tile row r of every frame of the group
// runs before row r + 1 of any, and the partial last rows (a frame's visited rows are rarely a multiple of the block height: one
// wavefront of four at 3840x2160) come last, where the launch drains.  Measured against (columns, rows, frames) in round 4
// (profiles/r04_assign_experiments.txt): fused 8 x 4K 45.1 -> 44.3 us, 8 x 1280x720 11.9 -> 11.6, full pass 8 x 1280x720 15.0 -> 13.5 us;
// pipelined rates unchanged.
static inline dim3 blk_grid(dim3 g) { return dim3(g.x, g.z, g.y); }

// PRE: preemptive mode (src/preemptive.h).  The bins hold the ACTIVE clusters only (k_preempt_update), so the gather is the reference's
// scatter over the active clusters (src/context.cpp:218); unless every cluster is active the fused sums take only the pixels of active
// (2S x 2S) cells (src/context.cpp:304-343: the row-by-row epilogue with the cell test), and the block's sums leave through global
// atomics into FrameDev::sums, where k_preempt_update reads them (no partial-sum entries, no slot geometry).
template <int R, bool FUSE, int STRIDE, bool VT, bool FBIN = false, bool PRE = false>
static __device__ __forceinline__ void assign_blk2_body(FrameDev& f, int rem, int Hv, const BlkMap& bm) {
    static_assert(!FBIN || FUSE, "the cluster pass rides on the fused centroid sums");
    static_assert(!PRE || (FUSE && !FBIN), "preemptive mode: the fused pass with its own flush");
    const uint32_t bxi = blockIdx.x, byi = blockIdx.z, bzi = blockIdx.y;      // (tile column, tile row, frame): see blk_grid
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];   // spatial table (u16), whole KB (LDS-DMA lands 1 KB per wavefront instruction); 16-byte aligned: the row-vector reads are ds_read_b128
    __shared__ __attribute__((aligned(16))) uint4 s_raw[kWavesPerBlock][64];       // every wavefront's kept records {yx, lab, tag, -}
    __shared__ __attribute__((aligned(16))) uint32_t s_part[64][4];                // partial ranks: [candidate][wavefront]
    __shared__ __attribute__((aligned(16))) uint32_t s_wcnt[4];                    // kept records per wavefront
    __shared__ __attribute__((aligned(16))) uint32_t s_entry[FUSE ? 64 : 4];       // cpart entry by rank | kNoSlot
    // cluster number by a pixel's rank (the low six bits of its key).  Every wavefront writes the same values (no barrier between
    // these stores and the epilogue's reads: a wavefront's own stores precede its reads, and what the others store is identical).
    // (Until the end of round 4 the table had 256 entries indexed by the key's whole low byte -- rank | two distance bits, four copies
    // 128 bytes apart -- to save the mask: but 128 bytes is the period of the LDS banks, lanes of equal rank and different distance bits
    // met on one bank at different addresses, and the sixteen look-ups of a wavefront cost 120 LDS cycles, 90 of them bank conflicts,
    // and 8 % of the fused launch's duration: scripts/gpu_knock.sh, profiles/r04_assign_experiments.txt.  The 64 entries are one
    // pass over the banks: equal ranks read one address, different ranks different banks.)
    __shared__ __attribute__((aligned(16))) uint16_t s_klab[64];
    // centroid accumulators: per (candidate, copy) two 64-bit words of three 20-bit-spaced fields each,
    //   A = sum L | sum a << 20 | sum b << 40        B = sum (x - x0) | sum (y - y0) << 20 | count << 40
    __shared__ __attribute__((aligned(16))) unsigned long long s_acc[FUSE ? 64 * kBlkCopies * 2 : 2];
    static_assert(R % 2 == 0, "rows are processed in pairs");
    constexpr int BH = kWavesPerBlock * R;
    static_assert(!FUSE || (BH * 64 * 255 < (1 << 20) && (BH - 1) * STRIDE * BH * 64 < (1 << 20)), "accumulator fields overflow");

    ASG_MARK("entry");
    f.select((int)bzi);
    const uint32_t gen = f.generation();
    const int tid = threadIdx.x;
    if (FBIN) {

        // housekeeping the separate cluster pass used to do: the counters of the bins this pass READS (nobody looks at them
        // while the records are stamped) are cleared for the pass after next; likewise the spill counter nobody uses in this pass
        const uint32_t gi = (byi * bm.gx + bxi) * 256u + (uint32_t)tid;
        if (gi < (uint32_t)f.ncell) f.cell_cnt[0][gi] = 0u;
        if (gi == 0u) *f.ovf_cnt[2] = 0u;
    }
    const int lane = tid & 63;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    const int W = f.W, S = f.S;
    const int x0 = (int)bxi * kTileW;
    const int x = x0 + lane;
    const int vrow_b0 = (int)byi * BH;
    const int vrow_w0 = vrow_b0 + wave * R;
    const bool okx = x < W;
    const int nrows = (int)rfl((uint32_t)max(0, min(R, Hv - vrow_w0)));      // 0: this wavefront has no rows (bottom edge); (stated uniform: the clamp is a v_med3, and what follows from it stays in the scalar unit)
    const int yw_lo = rem + vrow_w0 * STRIDE;
#if defined(FSLIC_LAB) && FSLIC_LAB == 4
    const uint32_t asg_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    volatile uint32_t* const asg_w = f.lab_stamps + ((size_t)(f.gen_off - 1u) * 2048u + (size_t)(byi * bm.gx + bxi)) * 16u;
    ASG_PUT(0, wave == 0, asg_t0);
#endif

    // 1. pixel loads: nothing depends on them until the argmin, but they are issued AFTER the loads the prologue waits
    // for (bins, table).  Every wavefront of the chip starts at the same time; with the pixel loads first the few bytes on the
    // critical path queued behind the whole frame's pixel traffic (prologue 4 us of a wavefront life of 8 us at 4K).
    // Out-of-image lanes and rows re-read the last valid column / row instead of being masked off: no exec juggling
    // around the loads (their results are never stored and never reach the centroid sums).
    // CONTRACT with the table wait below (s_waitcnt vmcnt(kPixelLoads)): load_pixels() issues EXACTLY kPixelLoads vector-memory loads,
    // all of them unconditional, and nothing else may be issued to vector memory between the LDS-DMA pieces of the table and that
    // wait -- the wait counts on "at most kPixelLoads operations outstanding" meaning "the pieces, which are older, have landed"
    // (loads return in order).  A load added, removed or made conditional here changes the count the wait must state.
    constexpr int kPixelLoads = R;
    uint32_t pix[R];
    auto load_pixels = [&]() {
        const uint32_t xc = (uint32_t)min(x, W - 1);
        const int nr1 = (int)rfl((uint32_t)(max(nrows, 1) - 1));      // a wavefront without rows (bottom edge) re-reads the image's last visited row
        const int ybase = (int)rfl((uint32_t)(nrows > 0 ? yw_lo : rem + (Hv - 1) * STRIDE));   // (both wave-uniform: stated, the scalar offsets below must be SGPRs)
        // buffer loads: the row goes into the instruction's scalar offset, the column into ONE shared vector offset, so a
        // row costs no vector instruction beyond the load itself (a 64-bit address per row was 6 scalar + 1 vector)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(f.lab), 0, (int)((uint32_t)f.N * 4u), kRawBuffer);
        const uint32_t pitch = (uint32_t)(STRIDE * W) * 4u, soff0 = (uint32_t)ybase * (uint32_t)W * 4u;
        const int vo = (int)(xc * 4u);
        static_assert(kPixelLoads == R, "one buffer load per visited row: see the contract above");
#pragma unroll
        for (int r = 0; r < R; ++r) pix[r] = __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (int)(soff0 + (uint32_t)min(r, nr1) * pitch), 0);
    };

    // 2. The block's candidates, by all four wavefronts: every thread fetches ONE 16-byte bin slot per trip (all the slots the
    // block's pixels can see: 240 at 1280x720 K=1600, i.e. one trip), keeps the live records inside the window and appends them
    // to its wavefront's part of s_raw (ballot compaction, no atomics).  Meanwhile the spatial table arrives by LDS-DMA and the
    // pixel loads are in flight.
    ASG_MARK("gather");
    const int sh = f.cs_shift;
    TileWindow win;
    {
        const int vlast = min(vrow_b0 + BH - 1, Hv - 1);
        win.wy_lo = rem + vrow_b0 * STRIDE - S; win.wy_hi = rem + vlast * STRIDE + S;
        win.wx_lo = x0 - S;                     win.wx_hi = min(x0 + kTileW - 1, W - 1) + S;
    }
    const int cxa = max(win.wx_lo, 0) >> sh, cxb = min(win.wx_hi, W - 1) >> sh;
    const int cya = max(win.wy_lo, 0) >> sh, cyb = min(win.wy_hi, f.H - 1) >> sh;
    const uint32_t ncxr = (uint32_t)(cxb - cxa + 1);
    const uint32_t nslots = ncxr * (uint32_t)(cyb - cya + 1) * (uint32_t)kCellCap;
    const uint32_t inv = (uint32_t)(65536.0f * __builtin_amdgcn_rcpf((float)ncxr)) + 1u;      // cell / ncxr == (cell * inv) >> 16 for the few hundred cells here (guarded below)
    const char* const cbase = reinterpret_cast<const char*>(f.cell_items[0]);
    auto slot_ptr = [&](uint32_t e) -> const ClusterRec* {      // slot e of the block's cell range, cells row-major (slots past the end re-read slot 0..7 of the first cell)
        const uint32_t cell = e < nslots ? e >> 3 : 0u;
        // (24-bit multiplies: every factor is a cell count or a reciprocal below 2^17, and a 32-bit v_mul_lo_u32 issues at a quarter of the rate)
        uint32_t cyi = __umul24(cell, inv) >> 16;
        const uint32_t back = __umul24(cyi, ncxr);
        if (back > cell) cyi--;
        const uint32_t cxi = cell - __umul24(cyi, ncxr);
        const uint32_t off = (__umul24((uint32_t)cya + cyi, (uint32_t)f.ncx) + (uint32_t)cxa + cxi) * (uint32_t)(kCellCap * sizeof(ClusterRec)) + (e & 7u) * (uint32_t)sizeof(ClusterRec);
        return reinterpret_cast<const ClusterRec*>(cbase + off);
    };
    const ClusterRec rec0 = *slot_ptr((uint32_t)tid);
    const uint32_t novf = *f.ovf_cnt[0];
    ClusterRec rec1;
    rec1.yx = 0; rec1.lab = 0; rec1.tag = 0; rec1.pad = 0;
    if (nslots > 256u) rec1 = *slot_ptr(256u + (uint32_t)tid);                // (uniform)
    {   // spatial table -> LDS, 1 KB per wavefront instruction (no staging registers, no ds_write pass); the last piece re-reads the
        // table's last 16 bytes for the lanes past its end (they land in the padding of the KB)
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t*)s_dyn;
        const int nq = f.tab_words >> 2, nkb = (nq + 63) >> 6;
        for (int i = wave; i < nkb; i += kWavesPerBlock)
            lds_dma_16(reinterpret_cast<const uint4*>(f.tab) + min(i * 64 + lane, nq - 1), lds0 + (uint32_t)i * 1024u);
    }
    if (FUSE) {
#pragma unroll
        for (int i = 0; i < kBlkCopies / 4; ++i) reinterpret_cast<uint4*>(s_acc)[tid + 256 * i] = make_uint4(0u, 0u, 0u, 0u);    // 64 * kBlkCopies * 2 * 8 bytes = 256 threads x 16 x (copies / 4)
    }
    static_assert(kBlkCopies % 4 == 0 && 64 * kBlkCopies * 2 * 8 == 256 * 16 * (kBlkCopies / 4), "16-byte stores of all threads clear the accumulators");
    load_pixels();
    {
        int cnt = 0;
        uint4* const myraw = s_raw[wave];
        // (the tests are single comparisons whose results are wavefront masks, combined by the scalar unit: device_common.h, ballot)
        auto visit = [&](const ClusterRec& rec, unsigned long long among) {
            const int cy = (int)(rec.yx >> 16), cx = (int)(rec.yx & 0xFFFFu);
            const unsigned long long m = among & ballot(cy >= win.wy_lo) & ballot(cy <= win.wy_hi) & ballot(cx >= win.wx_lo) & ballot(cx <= win.wx_hi);
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)cnt));
            if (lanes(m & ballot(pos < 64))) myraw[pos] = make_uint4(rec.yx, rec.lab, rec.tag, 0u);
            cnt += __popcll(m);
        };
        auto first_lanes = [](int n) -> unsigned long long { return n >= 64 ? ~0ull : n <= 0 ? 0ull : (1ull << n) - 1ull; };
        const int w64 = wave * 64;
        visit(rec0, first_lanes((int)nslots - w64) & ballot(rec0.pad == gen));
        if (nslots > 256u) visit(rec1, first_lanes((int)nslots - 256 - w64) & ballot(rec1.pad == gen));
        for (uint32_t e0 = 512u; e0 < nslots; e0 += 256u) {                        // (blocks that see more than 64 cells: S < 16)
            const ClusterRec rec = *slot_ptr(e0 + (uint32_t)tid);
            visit(rec, first_lanes((int)nslots - (int)e0 - w64) & ballot(rec.pad == gen));
        }
        if (wave == kWavesPerBlock - 1) {                                          // the spill list (cells with more than kCellCap centres)
            const ClusterRec* __restrict__ oitems = f.ovf_items[0];
            for (uint32_t i0 = 0; i0 < novf; i0 += 64) {
                const uint32_t i = i0 + (uint32_t)lane;
                ClusterRec rec;
                rec.yx = 0; rec.lab = 0; rec.tag = 0; rec.pad = 0;
                if (i < novf) rec = oitems[i];
                visit(rec, first_lanes((int)novf - (int)i0));
            }
        }
        if (lane == 0) s_wcnt[wave] = (uint32_t)cnt;
    }
    ASG_STAMP(1, wave == 0);
    ASG_MARK("wait_table");
    // The table has landed once at most R loads are outstanding (the compiler's counters do not include the LDS-DMA pieces): the R pixel
    // loads were issued after them and loads return in order; whatever the gather issued later has been waited for by the compiler at
    // its use.  The pixels themselves stay in flight across both barriers of the prologue (LDS-only barriers, see lds_barrier).
#if defined(FSLIC_LAB) && FSLIC_LAB == 5      // lab build 5: the conservative wait (everything landed); results must not differ from the product's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPixelLoads) : "memory");
#endif
    lds_barrier();
    ASG_STAMP(2, wave == 0);
    ASG_MARK("list_rank");
    // lane j = candidate j of the block (raw order: wavefront 0's records, then wavefront 1's, ...)
    int bn;
    uint4 mine;
    {
        const uint4 wc = *reinterpret_cast<const uint4*>(s_wcnt);
        const int c0 = (int)rfl(wc.x), c1 = (int)rfl(wc.y), c2 = (int)rfl(wc.z), c3 = (int)rfl(wc.w);
        bn = c0 + c1 + c2 + c3;
        if (bn > 64) {                        // more live candidates than idx bits: the host redoes the frame generically
            if (tid == 0) atomicOr(f.err_flag, 1u);
            // The plane is reset lazily (FrameDev::fv_mod): rows this launch is the first to look at hold whatever the arena's previous
            // geometry left there.  The frame is redone, but a later pass of THIS run whose list fits again would take that for the
            // pixels' old labels (sums of a cluster number beyond K), and the connectivity pass runs before the host sees the flag.
            if constexpr (!FBIN) {
                if (okx) {
                    uint16_t* rowp = f.labels + (size_t)yw_lo * W + (uint32_t)x;
                    for (int r = 0; r < nrows; ++r, rowp += (size_t)STRIDE * W)
                        if (FUSE ? f.fv_mod == 1 : f.first_visit(yw_lo + r * STRIDE)) st_stream(rowp, (uint16_t)0xFFFFu);
                }
            }
            return;
        }
        const int p1 = c0, p2 = c0 + c1, p3 = c0 + c1 + c2;
        const int idx = lane + (lane >= p1 ? 64 - c0 : 0) + (lane >= p2 ? 64 - c1 : 0) + (lane >= p3 ? 64 - c2 : 0);     // [wavefront][position] flattened
        mine = (&s_raw[0][0])[lane < bn ? idx : 0];
        if (lane >= bn) mine.z = 0xFFFFFFFFu;
    }
    // rank by (phase, k) = visit order of the reference's scatter (src/context.cpp:214-242): wavefront w compares every
    // candidate with candidates w, w+4, ..; the four partial counts meet in LDS
    {
        const uint32_t tag_j = mine.z;
        uint32_t part = 0;
        for (int i = wave; i < bn; i += kWavesPerBlock) part += (uint32_t)(__builtin_amdgcn_readlane(tag_j, i) < tag_j);
        s_part[lane][wave] = part;
    }
    lds_barrier();
    uint32_t rank_j;
    {
        const uint4 pr = *reinterpret_cast<const uint4*>(s_part[lane]);
        rank_j = pr.x + pr.y + pr.z + pr.w;
    }
    if (lane < bn) s_klab[(int)rank_j] = (uint16_t)(mine.z & 0xFFFFu);
    uint32_t fin_nx = 0, fin_ny = 0, fin_b0 = 0; // FBIN, wavefront 0, lane j: blocks that deliver sums of candidate j (0: not this block), the first of them (row << 16 | column)
    if (PRE && wave == 0) {
        if (lane < bn) s_entry[rank_j] = mine.z & 0xFFFFu;
    } else if (FUSE && wave == 0) {
        // The entry of (cluster, this block): by the block's own coordinates modulo the entry grid (device_common.h, cpart_slot) -- no
        // geometry of the cluster's window needed.  Every candidate of the list has visited rows in this block: the list's window is
        // the block's first .. last VISITED row +- S (and its columns +- S), and a cluster window of 2S + 1 >= stride rows that meets
        // that extent contains one of its visited rows, so the block is always one of the blocks the cluster pass expects an entry
        // from (until round 5 this was re-derived per candidate from slot_geom by wavefront 0, ~50 vector instructions on the block's
        // critical path).  The fused cluster pass (FBIN) still needs the window's block range: how many blocks deliver.
        const uint32_t e1 = __umul24(mine.z & 0xFFFFu, (uint32_t)f.cpart_ns) + cpart_slot(byi, bxi, cpart_rows_mask(S, STRIDE, BH), (uint32_t)f.nbx);
        const uint32_t entry = (e1 << 2) + (e1 << 1);
        if constexpr (FBIN) {
            const SlotGeom g = slot_geom((int)(mine.x >> 16), (int)(mine.x & 0xFFFFu), S, f.H, W, rem, STRIDE, BH, Hv);
            if (lane < bn) { fin_nx = (uint32_t)(g.bx1 - g.bx0 + 1); fin_ny = (uint32_t)(g.by1 - g.by0 + 1); fin_b0 = ((uint32_t)g.by0 << 16) | (uint32_t)g.bx0; }
        }
        if (lane < bn) s_entry[rank_j] = entry;
    }
    ASG_STAMP(3, wave == 0);
    ASG_MARK("masks");

    // 3. the block list in registers: lane j = candidate j; candidates whose window reaches this wavefront's rows
    // (the rank travels in the top byte of the colour word: one v_readlane fetches both)
    const uint32_t yx_j = mine.x, lab_j = mine.y | (rank_j << 24);
    const int yw_hi = yw_lo + (nrows - 1) * STRIDE;
    unsigned long long todo;
    {
        const int cyj = (int)(yx_j >> 16);
        todo = nrows > 0 ? (bn >= 64 ? ~0ull : (1ull << bn) - 1ull) & ballot(cyj >= yw_lo - S) & ballot(cyj <= yw_hi + S) : 0ull;
    }

    // 4. the argmin, two rows per instruction
    us2 best[R / 2];
#pragma unroll
    for (int q = 0; q < R / 2; ++q) best[q] = (us2){0xFFFF, 0xFFFF};
    const uint32_t nrpad = (uint32_t)f.tab_nrpad;
    const int dyoff = f.tab_dyoff;
    if constexpr (VT) {
        const int Cc = f.tab_dyoff;                 // centre index of the V table
        const uint32_t xout = (uint32_t)f.tab_xout;
        // The row-range divisions: numerators made positive by BIAS rows, quotients through a 24-bit multiply (full rate; the 32-bit
        // v_mul_hi_u32 the compiler takes for n / 3 issues at a quarter of it).  |dy0| <= S + BH * STRIDE for a candidate on the block's
        // list and S <= 419 in this table mode (tables.cpp: tlen * 16 <= kTabMaxBytes for the 16-row table), so 0 < numerator < 3 * 1024 + 1300, far inside
        // the range where (n * 21846) >> 16 == n / 3 (n < 32768).
        constexpr int BIAS = 1024;
        static_assert(STRIDE >= 1 && STRIDE <= 3, "row-range divisions");
        auto div_stride = [](int n) -> int { return STRIDE == 1 ? n : STRIDE == 2 ? (n >> 1) : (int)(__umul24((uint32_t)n, 21846u) >> 16); };
        constexpr uint32_t full = R >= 32 ? 0xFFFFFFFFu : (1u << (R & 31)) - 1u;
        auto low_bits = [](int n) -> uint32_t { return n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u; };      // n in [0, 32]
        // Per-candidate row ranges, computed ONCE in the vector unit with lane j = candidate j (the scalar unit is
        // shared by the whole CU and issues about as many instructions per clock as the four SIMDs together).
        //   rows r with dy_r = dy0 + r*STRIDE: above the centre r < rs, inside the window ra <= r <= rb
        uint32_t va_j, vb_j;                        // rows above / rows at or below the centre that are in the window
        {
            const int dy0 = yw_lo - (int)(yx_j >> 16);
            const int rs = min(max(div_stride(-dy0 + STRIDE * BIAS + STRIDE - 1) - BIAS, 0), R);
            const int ra = min(max(div_stride(-S - dy0 + STRIDE * BIAS + STRIDE - 1) - BIAS, 0), R);
            const int rb = min(max(div_stride(S - dy0 + STRIDE * BIAS) - BIAS, -1), R - 1);
            const uint32_t upto_rs = low_bits(rs);
            const uint32_t in_win = rb < 0 ? 0u : (low_bits(rb + 1) & ~low_bits(ra) & full);
            va_j = in_win & upto_rs;
            vb_j = in_win & ~upto_rs;
        }
        // row-vector table: entry t = (V[t], V[t+STRIDE], .. V[t+7*STRIDE]) as one 16-byte word, so the spatial terms of
        // eight rows are ONE 16-byte-ALIGNED ds_read_b128 (a 2-byte-aligned b128 read costs 64 clk of the CU's LDS pipe
        // instead of 8, scripts/microbench/lds_unaligned.hip)
        const uint4* const s_rv = reinterpret_cast<const uint4*>(s_dyn);
        auto pass = [&](uint32_t clab, uint32_t base, uint32_t idx01, const uint32_t* mq) {
            uint32_t spw[R / 2];
            {
                const uint4 e0 = s_rv[base];
                spw[0] = e0.x; spw[1] = e0.y; spw[2] = e0.z; spw[3] = e0.w;
#pragma unroll
                for (int h = 1; h < R / 8; ++h) {
                    const uint4 e1 = s_rv[base + 8 * h * STRIDE];
                    spw[4 * h] = e1.x; spw[4 * h + 1] = e1.y; spw[4 * h + 2] = e1.z; spw[4 * h + 3] = e1.w;
                }
            }
            // sweeps over the row pairs instead of one chain per pair: a packed op that consumes the result of the
            // packed op issued just before it costs a wait state (s_nop) on this target
            uint32_t t[R / 2];
#pragma unroll
            for (int q = 0; q < R / 2; ++q) {
                const uint32_t s0 = __builtin_amdgcn_sad_u8(pix[2 * q], clab, spw[q]);          // spatial pair + colour distance of row 2q
                t[q] = __builtin_amdgcn_sad_hi_u8(pix[2 * q + 1], clab, s0);                     // + colour distance of row 2q+1 in the high half
            }
            us2 key[R / 2];
#pragma unroll
            for (int q = 0; q < R / 2; ++q) key[q] = key_pair(t[q], mq ? (mq[q] | idx01) : idx01);
#pragma unroll
            for (int q = 0; q < R / 2; ++q) min_in_place(best[q], key[q]);
        };
        // Four loops, each with ONE straight-line form of the pass (the order of the candidates is irrelevant: the key
        // carries idx).  With the forms as branches of one loop body the compiler linearised them into guarded blocks,
        // kept the running minima in two register sets and copied between them: 8-16 v_mov per candidate.
        //   whole : every row inside the window and on one side of the centre (above: index k0 - |dx|, below: k0 + |dx|): no masks
        //   above : some rows above the centre, masked          below : some rows at/below the centre, masked
        // (a candidate whose centre row lies inside this wavefront's rows is in both masked loops)
        unsigned long long todo_wa = todo & ballot(va_j == full);               // whole, above the centre
        unsigned long long todo_wb = todo & ballot(vb_j == full);               // whole, at/below
        unsigned long long todo_above = todo & ballot(va_j != 0u) & ~todo_wa;
        unsigned long long todo_below = todo & ballot(vb_j != 0u) & ~todo_wb;
        // CLAMP: lanes outside the x window get |dx| = xout (every row then reads the marker); !CLAMP: the caller runs the pass
        // under `if (uxc <= S)` instead, i.e. with those lanes switched off (one vector instruction less per candidate)
        auto fetch = [&](int c, uint32_t& clab, uint32_t& uxc, uint32_t& k0, uint32_t& idx01, auto clamp_tag) {
            constexpr bool CLAMP = decltype(clamp_tag)::value;
            const uint32_t cyx = __builtin_amdgcn_readlane(yx_j, c);
            const uint32_t clabx = __builtin_amdgcn_readlane(lab_j, c);
            clab = clabx & 0x00FFFFFFu;
            const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
            const uint32_t ux = __builtin_amdgcn_sad_u16((uint32_t)x, (uint32_t)cx, 0u);      // |x - cx|
            uxc = CLAMP ? (ux <= (uint32_t)S ? ux : xout) : ux;
            k0 = (uint32_t)(Cc + yw_lo - cy);                                                  // wave-uniform
            idx01 = (clabx >> 24) * 0x10001u;                                                  // rank | rank << 16
        };
#pragma unroll 1
        while (todo_wa) {
            const int c = __builtin_ctzll(todo_wa);
            todo_wa &= todo_wa - 1ull;
            uint32_t clab, uxc, k0, idx01;
            fetch(c, clab, uxc, k0, idx01, std::false_type{});
            if (uxc <= (uint32_t)S) pass(clab, k0 - uxc, idx01, nullptr);
        }
#pragma unroll 1
        while (todo_wb) {
            const int c = __builtin_ctzll(todo_wb);
            todo_wb &= todo_wb - 1ull;
            uint32_t clab, uxc, k0, idx01;
            fetch(c, clab, uxc, k0, idx01, std::false_type{});
            if (uxc <= (uint32_t)S) pass(clab, k0 + uxc, idx01, nullptr);
        }
        if constexpr (R == 16 && FUSE) {
            // 16 rows per wavefront (the fused passes at 3840x2160): the masked candidates by HALVES of eight rows, i.e. by the 16-byte
            // table entries.  The rows a masked pass has to look at are a prefix or a suffix of the wavefront's rows -- above the centre:
            // rows [.., rs), at / below it: [rs, ..], cut by the window -- so they lie in one half more often than not, and a candidate
            // whose centre row is inside the wavefront's rows (walked twice) needs three halves on average instead of four: a tenth of
            // the kernel's vector work at S = 72.
            auto pass_half = [&](auto half_tag, uint32_t clab, uint32_t base, uint32_t idx01, const uint32_t* mq4) {
                constexpr int HF = decltype(half_tag)::value;
                const uint4 e0 = s_rv[base + 8 * HF * STRIDE];
                const uint32_t spw[4] = {e0.x, e0.y, e0.z, e0.w};
                uint32_t t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t s0 = __builtin_amdgcn_sad_u8(pix[2 * (4 * HF + q)], clab, spw[q]);
                    t[q] = __builtin_amdgcn_sad_hi_u8(pix[2 * (4 * HF + q) + 1], clab, s0);
                }
                us2 key[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) key[q] = key_pair(t[q], mq4[q] | idx01);
#pragma unroll
                for (int q = 0; q < 4; ++q) min_in_place(best[4 * HF + q], key[q]);
            };
            // (sign: -1 the rows above the centre, index k0 - |dx|; +1 the rows at / below it, k0 + |dx|)
            auto masked = [&](unsigned long long m, const uint32_t rows_j, auto sign_tag) {
                constexpr int SG = decltype(sign_tag)::value;
                unsigned long long m_lo = m & ballot((rows_j & 0xFF00u) == 0u);         // every row in the upper half of the wavefront's rows
                unsigned long long m_hi = m & ballot((rows_j & 0x00FFu) == 0u) & ~m_lo;
                unsigned long long m_both = m & ~m_lo & ~m_hi;
#pragma unroll 1
                while (m_lo) {
                    const int c = __builtin_ctzll(m_lo);
                    m_lo &= m_lo - 1ull;
                    uint32_t clab, uxc, k0, idx01, mq[4];
                    fetch(c, clab, uxc, k0, idx01, std::true_type{});
                    row_masks<8>(~__builtin_amdgcn_readlane(rows_j, c) & 0xFFu, mq);
                    pass_half(std::integral_constant<int, 0>{}, clab, SG < 0 ? k0 - uxc : k0 + uxc, idx01, mq);
                }
#pragma unroll 1
                while (m_hi) {
                    const int c = __builtin_ctzll(m_hi);
                    m_hi &= m_hi - 1ull;
                    uint32_t clab, uxc, k0, idx01, mq[4];
                    fetch(c, clab, uxc, k0, idx01, std::true_type{});
                    row_masks<8>(~(__builtin_amdgcn_readlane(rows_j, c) >> 8) & 0xFFu, mq);
                    pass_half(std::integral_constant<int, 1>{}, clab, SG < 0 ? k0 - uxc : k0 + uxc, idx01, mq);
                }
#pragma unroll 1
                while (m_both) {
                    const int c = __builtin_ctzll(m_both);
                    m_both &= m_both - 1ull;
                    uint32_t clab, uxc, k0, idx01, mq[R / 2];
                    fetch(c, clab, uxc, k0, idx01, std::true_type{});
                    row_masks<R>(~__builtin_amdgcn_readlane(rows_j, c) & full, mq);
                    pass(clab, SG < 0 ? k0 - uxc : k0 + uxc, idx01, mq);
                }
            };
            masked(todo_above, va_j, std::integral_constant<int, -1>{});
            masked(todo_below, vb_j, std::integral_constant<int, 1>{});
        } else {
#pragma unroll 1
            while (todo_above) {
                const int c = __builtin_ctzll(todo_above);
                todo_above &= todo_above - 1ull;
                uint32_t clab, uxc, k0, idx01, mq[R / 2];
                fetch(c, clab, uxc, k0, idx01, std::true_type{});
                row_masks<R>(~__builtin_amdgcn_readlane(va_j, c) & full, mq);
                pass(clab, k0 - uxc, idx01, mq);
            }
#pragma unroll 1
            while (todo_below) {
                const int c = __builtin_ctzll(todo_below);
                todo_below &= todo_below - 1ull;
                uint32_t clab, uxc, k0, idx01, mq[R / 2];
                fetch(c, clab, uxc, k0, idx01, std::true_type{});
                row_masks<R>(~__builtin_amdgcn_readlane(vb_j, c) & full, mq);
                pass(clab, k0 + uxc, idx01, mq);
            }
        }
    } else {
        // 2-D table.  The wavefront's rows in two halves (R/2 visited rows each): a candidate whose window reaches only one half is
        // walked over that half alone -- at 1280x720 K=1600 (S = 24, eight rows spanning 22) a third of the candidates, i.e. a
        // sixth of the distance / key / min instructions.  Three loops with straight-line bodies (see the row-vector loops above).
        constexpr int HQ = R / 4;                        // row pairs per half
        static_assert(R % 4 == 0, "two halves of whole row pairs");
        unsigned long long todo_b, todo_u, todo_l;
        {
            const int cyj = (int)(yx_j >> 16);
            const int yh = yw_lo + (R / 2) * STRIDE;     // first row of the lower half
            // (cyj + S >= yw_lo holds for every candidate on the list: `todo`)
            const unsigned long long up = todo & ballot(cyj <= yw_lo + (R / 2 - 1) * STRIDE + S);
            const unsigned long long lo = todo & ballot(cyj >= yh - S) & ballot(cyj <= yh + (R / 2 - 1) * STRIDE + S);
            todo_b = up & lo;
            todo_u = up & ~lo;
            todo_l = lo & ~up;
        }
        // byte pitch of a table column in a VGPR: |dx| * pitch + row offset is then ONE v_mad_u32_u24 (two scalar operands would
        // not fit one VALU instruction) and the result is the LDS byte address itself (no shift)
        uint32_t pitch_v;
        asm volatile("v_mov_b32 %0, %1" : "=v"(pitch_v) : "s"(nrpad * 4u));
        const char* const tab_b = reinterpret_cast<const char*>(s_dyn);
        auto walk = [&](unsigned long long m, auto q0_tag, auto nq_tag) {
            constexpr int Q0 = decltype(q0_tag)::value, NQ = decltype(nq_tag)::value;
#pragma unroll 1
            while (m) {
                const int c = __builtin_ctzll(m);
                m &= m - 1ull;
                const uint32_t cyx = __builtin_amdgcn_readlane(yx_j, c);
                const uint32_t clabx = __builtin_amdgcn_readlane(lab_j, c);
                const uint32_t clab = clabx & 0x00FFFFFFu;
                const int cy = (int)(cyx >> 16), cx = (int)(cyx & 0xFFFFu);
                const uint32_t ux = __builtin_amdgcn_sad_u16((uint32_t)x, (uint32_t)cx, 0u);      // |x - cx|
                const uint32_t uxc = min(ux, (uint32_t)(S + 1));
                const uint32_t a0 = __umul24(uxc, pitch_v) + (uint32_t)(yw_lo - cy + dyoff) * 4u;  // byte offset of the pair (row 0, row 1)
                const uint32_t idx01 = (clabx >> 24) * 0x10001u;                                   // rank | rank << 16
                uint32_t t[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint32_t sp = *reinterpret_cast<const uint32_t*>(tab_b + a0 + (uint32_t)((2 * (Q0 + q)) * STRIDE * 4));   // one aligned ds_read_b32 per row pair
                    const uint32_t s0 = __builtin_amdgcn_sad_u8(pix[2 * (Q0 + q)], clab, sp);
                    t[q] = __builtin_amdgcn_sad_hi_u8(pix[2 * (Q0 + q) + 1], clab, s0);
                }
                us2 key[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) key[q] = key_pair(t[q], idx01);
#pragma unroll
                for (int q = 0; q < NQ; ++q) min_in_place(best[Q0 + q], key[q]);
            }
        };
        if constexpr (R == 16 && FUSE) {
            // 16 rows per wavefront (the fused passes of launches that fill the chip: 46 image rows at stride 3 against a window of
            // 2S + 1 = 49): QUARTERS of four visited rows.  A candidate's window is an interval of rows, so it reaches a contiguous range
            // of quarters -- ten ranges, ten loops with straight-line bodies, the masks of the ranges from four pairs of comparisons.
            // At S = 24 a candidate of the list reaches 2.5 of the four quarters on average, 3.0 when it can only be switched by halves:
            // a sixth of the distance / key / min instructions and of the table reads of these passes.
            (void)todo_b; (void)todo_u; (void)todo_l;
            unsigned long long cov[4];
            {
                const int cyj = (int)(yx_j >> 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ylo = yw_lo + q * (R / 4) * STRIDE, yhi = ylo + (R / 4 - 1) * STRIDE;
                    cov[q] = todo & ballot(cyj >= ylo - S) & ballot(cyj <= yhi + S);
                }
            }
            constexpr int P = R / 8;                         // row pairs per quarter
            auto quarters = [&](unsigned long long m, auto q0, auto nq) {
                walk(m, std::integral_constant<int, decltype(q0)::value * P>{}, std::integral_constant<int, decltype(nq)::value * P>{});
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
            quarters(cov[0] & cov[3], I0{}, I4{});
            quarters(cov[0] & cov[2] & ~cov[3], I0{}, I3{});
            quarters(cov[1] & cov[3] & ~cov[0], I1{}, I3{});
            quarters(cov[0] & cov[1] & ~cov[2], I0{}, I2{});
            quarters(cov[1] & cov[2] & ~cov[0] & ~cov[3], I1{}, I2{});
            quarters(cov[2] & cov[3] & ~cov[1], I2{}, I2{});
            quarters(cov[0] & ~cov[1], I0{}, I1{});
            quarters(cov[1] & ~cov[0] & ~cov[2], I1{}, I1{});
            quarters(cov[2] & ~cov[1] & ~cov[3], I2{}, I1{});
            quarters(cov[3] & ~cov[2], I3{}, I1{});
        } else {
            walk(todo_b, std::integral_constant<int, 0>{}, std::integral_constant<int, 2 * HQ>{});
            walk(todo_u, std::integral_constant<int, 0>{}, std::integral_constant<int, HQ>{});
            walk(todo_l, std::integral_constant<int, HQ>{}, std::integral_constant<int, HQ>{});
        }
    }

    ASG_MARK("after_walk");
    ASG_STAMP(4, wave == 0);
    ASG_PUT(9, wave == 0, __popcll(todo));
    ASG_PUT(10, wave == 0, bn);
    // vmcnt(0), stated where the compiler sees it: the pixels are in registers from here on.  Without it the compiler (the loops above may
    // run zero times) keeps "pixel load pending" marks on pix[] and places s_waitcnt vmcnt(R-1) .. vmcnt(0) in front of the sums of row
    // 0 .. R-1 -- which, with R label stores issued in between, made every row's sums wait for a store's acknowledgement.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // 5. write labels (+ fused centroid sums)
    bool epilogue_done = false;
    {
        us2 worst = best[0];
#pragma unroll
        for (int q = 1; q < R / 2; ++q) worst = __builtin_elementwise_max(worst, best[q]);
        const bool all_found = (ballot(worst.x == 0xFFFF) | ballot(worst.y == 0xFFFF)) == 0ull;
        bool fast_ok = all_found && nrows == R;
        if constexpr (PRE) fast_ok = fast_ok && rfl(f.pre_flags[0]) != 0u;      // (inactive cells: the row-by-row form below applies the cell test)
        if (fast_ok) {          // interior wavefront with every pixel assigned: the rule, by far
            ASG_MARK("labels");
            epilogue_done = true;
            if (okx) {
                uint32_t kk[R];
                const char* const ktab = reinterpret_cast<const char*>(s_klab);
#pragma unroll
                for (int q = 0; q < R / 2; ++q) {
                    const uint32_t b = __builtin_bit_cast(uint32_t, best[q]) & 0x003F003Fu;       // (the fused sums below need the ranks alone as well)
                    kk[2 * q] = *reinterpret_cast<const uint16_t*>(ktab + ((b & 0xFFu) << 1));
                    kk[2 * q + 1] = *reinterpret_cast<const uint16_t*>(ktab + (((b >> 16) & 0xFFu) << 1));
                }
                const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(f.labels, 0, (int)((uint32_t)f.N * 2u), kRawBuffer);
                const uint32_t so0 = (uint32_t)yw_lo * (uint32_t)W * 2u, pitch2 = (uint32_t)(STRIDE * W) * 2u;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    __builtin_amdgcn_raw_buffer_store_b16((uint16_t)kk[r], ws, (int)((uint32_t)x * 2u), (int)(so0 + (uint32_t)r * pitch2), kAuxSc1);
                ASG_STAMP(5, wave == 0);
                ASG_MARK("sums");
                if (FUSE) {
                    // Sums of update() (src/context.cpp:301-354), third form.  A lane keeps the PREFIX sums of its column (three v_dot4 per
                    // row, nothing else: count, sum x and sum y of a prefix are constants of the row number).  Where the label changes
                    // between rows r and r+1 the run that ends there is (prefix now) - (prefix at the lane's previous change): two packed
                    // 64-bit differences (three 20-bit-spaced fields each, see s_acc; field by field non-negative, so no borrow crosses a
                    // field), added to that run's accumulator; the column's last run at the end.  TWO 64-bit LDS atomics and ~14 vector
                    // instructions per row gap in which some lane changes its label; the second form (+ prefix to the upper run, - prefix
                    // to the lower run, no per-lane state) issued four atomics for the same vector work.
                    // What this epilogue costs, measured in round 5 (profiles/r05_experiments.txt): without the work at the label changes
                    // the launch of 16 x 1280x720 takes 12.3 instead of 16.2 us (knock-out build 1) -- but halving the atomics (this form)
                    // gave 16.66 -> 16.54 us, so it is the ~14 vector instructions a wavefront executes per gap for a handful of lanes,
                    // 15 gaps per wavefront of 16 rows, not the LDS pipe (a 64-bit LDS atomic occupies it for 6.5 clk whatever the number
                    // of lanes, profiles/r05_microbench_valu_lds_overlap.txt).  Lanes HOLDING their finished run until a common flush
                    // point every five rows (8 - 12 atomics per wavefront instead of 32): 18.3 us -- columns that run along a jagged
                    // boundary change label at nearly every gap and force the flush there and then, on top of the capture.
                    const uint32_t copy16 = ((uint32_t)lane & (kBlkCopies - 1)) * 16u;
                    const uint32_t dyw = (uint32_t)(wave * R * STRIDE);
                    char* const accb = reinterpret_cast<char*>(s_acc);
                    uint32_t sL = 0, sa = 0, sb = 0;
                    uint32_t wq[R / 2];
#pragma unroll
                    for (int q = 0; q < R / 2; ++q) wq[q] = __builtin_bit_cast(uint32_t, best[q]) & 0x003F003Fu;
                    auto words = [&](int cnt, unsigned long long& A, unsigned long long& B) {      // the column's first cnt rows
                        const uint32_t a_lo = sL | (sa << 20), a_hi = (sa >> 12) | (sb << 8);
                        const uint32_t ys = (uint32_t)cnt * dyw + (uint32_t)(STRIDE * (cnt * (cnt - 1) / 2));     // sum (y - y0) < 4096 (see the static_assert on BH)
                        const uint32_t b_lo = __umul24((uint32_t)lane, (uint32_t)cnt) + (ys << 20), b_hi = (uint32_t)cnt << 8;
                        A = (unsigned long long)a_lo | ((unsigned long long)a_hi << 32);
                        B = (unsigned long long)b_lo | ((unsigned long long)b_hi << 32);
                    };
                    unsigned long long prevA = 0ull, prevB = 0ull;      // the packed prefixes at the lane's last label change
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const uint32_t p = pix[r];
                        sL = __builtin_amdgcn_udot4(p, 0x00000001u, sL, false);
                        sa = __builtin_amdgcn_udot4(p, 0x00000100u, sa, false);
                        sb = __builtin_amdgcn_udot4(p, 0x00010000u, sb, false);
                        if (r == R - 1) break;
                        const uint32_t cur = (r & 1) ? (wq[r / 2] >> 16) : (wq[r / 2] & 0xFFFFu);
                        const uint32_t nxt = (r & 1) ? (wq[r / 2 + 1] & 0xFFFFu) : (wq[r / 2] >> 16);
#if FSLIC_EXP & 1      // (knock-out build 1: no accumulator updates at the label changes -- sums wrong on purpose)
                        if (false) {
#else
                        if (cur != nxt) {
#endif
                            unsigned long long A, B;
                            words(r + 1, A, B);
                            unsigned long long* const run = reinterpret_cast<unsigned long long*>(accb + (cur * (uint32_t)(kBlkCopies * 16) + copy16));
                            atomicAdd(run, A - prevA);
                            atomicAdd(run + 1, B - prevB);
                            prevA = A; prevB = B;
                        }
                    }
                    {
                        unsigned long long A, B;
                        words(R, A, B);
                        unsigned long long* const last = reinterpret_cast<unsigned long long*>(accb + ((wq[R / 2 - 1] >> 16) * (uint32_t)(kBlkCopies * 16) + copy16));
                        atomicAdd(last, A - prevA);
                        atomicAdd(last + 1, B - prevB);
                    }
                }
            }
        }
    }
    ASG_MARK("slow_epilogue");
    if (!epilogue_done) {
        // The exception: bottom rows missing, or a pixel no window reached (it keeps its label and still takes part in the update,
        // src/context.cpp:138-145): row by row with the full rules, sums per run of equal slot.
        uint32_t slot[R];
        uint16_t* rowp = f.labels + (size_t)yw_lo * W;
        const uint32_t xo = (uint32_t)x;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = okx && r < nrows;
            const uint32_t key = (r & 1) ? (uint32_t)best[r / 2].y : (uint32_t)best[r / 2].x;
            const bool found = key != 0xFFFFu;
            slot[r] = kNoSlot;
            if (ok && found) st_stream(rowp + xo, s_klab[key & 63u]);
            // lazy reset of the plane (FrameDev::fv_mod): the first pass to look at a row stores the 0xFFFF of pixels no window covers
            // (a fused pass is a subsampled pass: ALL of its rows are first visits or none is, one scalar test; only the full pass
            // has to look at the row's residue)
            // (the launches that carry the cluster pass -- FBIN -- belong to groups whose plane the LAB kernel still fills, group.cpp: compiled out there)
            const bool unlabelled = !FBIN && ok && !found && (FUSE ? f.fv_mod == 1 : f.first_visit(yw_lo + r * STRIDE));
            if (unlabelled) st_stream(rowp + xo, (uint16_t)0xFFFFu);
            bool summed = FUSE && ok && !unlabelled;
            if constexpr (PRE) summed = summed && in_update(f, yw_lo + r * STRIDE, x);
            if (summed) {
                if (found) slot[r] = key & 63u;
                else {
                    const uint32_t old = rowp[xo];     // stale label still takes part in update()
                    if (old != 0xFFFFu) {
                        global_accumulate(f.sums[0], old, (uint32_t)(yw_lo + r * STRIDE), xo, pix[r]);
                        // the owner of that label may lie outside this block's candidates: nothing orders these atomics before its
                        // finaliser.  Rare (no window covers the pixel); the host redoes the frame with the separate cluster pass.
                        if (FBIN) atomicOr(f.err_flag, 2u);
                    }
                }
            }
            rowp += (size_t)STRIDE * W;
        }
        ASG_STAMP(5, wave == 0);
        if (FUSE) {
            const uint32_t copy = (uint32_t)lane & (kBlkCopies - 1);
            const uint32_t dyw = (uint32_t)(wave * R * STRIDE);            // first row of this wavefront relative to the block's
            uint32_t cur = slot[0], a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0;
            auto flush = [&]() {
                if (cur != kNoSlot) {
                    unsigned long long* a = s_acc + (cur * kBlkCopies + copy) * 2;
                    atomicAdd(a, (unsigned long long)a_lo | ((unsigned long long)a_hi << 32));
                    atomicAdd(a + 1, (unsigned long long)b_lo | ((unsigned long long)b_hi << 32));
                }
            };
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r > 0 && slot[r] != cur) {
                    flush();
                    cur = slot[r]; a_lo = 0; a_hi = 0; b_lo = 0; b_hi = 0;
                }
                const uint32_t p = pix[r];
                a_lo += (p & 255u) | ((p & 0xFF00u) << 12);                // sum L | sum a << 20
                a_hi += (p >> 8) & 0xFF00u;                               // sum b << 40
                b_lo += (uint32_t)lane | ((dyw + (uint32_t)(r * STRIDE)) << 20);   // sum (x - x0) | sum (y - y0) << 20
                b_hi += 256u;                                             // count << 40
            }
            flush();
        }
    }
    ASG_MARK("block_flush");
    if (FUSE) {
        ASG_STAMP(6, wave == 0);
        lds_barrier();
        ASG_STAMP(7, wave == 0);
        // The block's sums leave through wavefront 0, lane j = candidate j: the 24-byte entry reserved for (cluster, this block) as
        // three 8-byte stores, zeros included.  (One thread per (candidate, quantity) spread the stores over three wavefronts, but each
        // of them paid an index division, the whole sum and a six-way select for one value: ~100 issue slots in three of the four
        // wavefronts, a tenth of the kernel's vector work; three threads per candidate with one 8-byte store each had been measured
        // by duration alone: no difference, profiles/r04_assign_experiments.txt.)
        if (wave == 0 && lane < bn) {
            const uint32_t entry = s_entry[lane];
            if (entry != kNoSlot) {
                unsigned long long A = 0, B = 0;
#pragma unroll
                for (int cpy = 0; cpy < kBlkCopies; ++cpy) { A += s_acc[(lane * kBlkCopies + cpy) * 2]; B += s_acc[(lane * kBlkCopies + cpy) * 2 + 1]; }
                const uint32_t yb0 = (uint32_t)(rem + vrow_b0 * STRIDE);
                const uint32_t cnt = (uint32_t)(B >> 40);                 // (at most 64 x BH pixels: the products below are 24-bit multiplies)
                const uint2 q01 = make_uint2(cnt, ((uint32_t)(B >> 20) & 0xFFFFFu) + __umul24(yb0, cnt));
                const uint2 q23 = make_uint2(((uint32_t)B & 0xFFFFFu) + __umul24((uint32_t)x0, cnt), (uint32_t)A & 0xFFFFFu);
                const uint2 q45 = make_uint2((uint32_t)(A >> 20) & 0xFFFFFu, (uint32_t)(A >> 40));
                if constexpr (PRE) {                   // (`entry` is the cluster number here)
                    uint32_t* const sm = f.sums[0] + 6 * (size_t)entry;
                    if (cnt) { atomicAdd(sm, q01.x); atomicAdd(sm + 1, q01.y); atomicAdd(sm + 2, q23.x); atomicAdd(sm + 3, q23.y); atomicAdd(sm + 4, q45.x); atomicAdd(sm + 5, q45.y); }
                } else {
                    uint2* const dst = reinterpret_cast<uint2*>(f.cpart + (size_t)entry);      // (entries are 24 bytes apart in an 8-byte aligned array)
                    if (FBIN) { st_stream(dst, q01); st_stream(dst + 1, q23); st_stream(dst + 2, q45); }      // written through: another block's lane reads it
                    else { dst[0] = q01; dst[1] = q23; dst[2] = q45; }
                }
            }
        }
        if (FBIN) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront drains (sums, stale-pixel atomics) ..
            __syncthreads();                                       // .. before the block counts itself in
            if (wave == 0 && fin_nx != 0u) {
                const uint32_t k = mine.z & 0xFFFFu;
                const uint32_t before = __hip_atomic_fetch_add(f.cl_arrive + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1u == fin_nx * fin_ny) fused_finalize(f, k, mine.x, mine.y, fin_nx, fin_ny, fin_b0, cpart_rows_mask(S, STRIDE, BH));
            }
        }
    }
    ASG_STAMP(8, wave == 0);
}

// The kernels proper.  Register budgets are part of the design (amdgpu_waves_per_eu): the bodies need 24 - 48 VGPRs with 8 rows per
// wavefront (8 wavefronts per SIMD), 38 - 40 for the 16-row full pass (8) and 74 for the 16-row fused pass (6); left to itself the
// compiler schedules them into more (at 3840x2160 that cost 17 % with the first form: 51 vs 60 us per fused launch of eight frames).
template <int R, bool FUSE, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(7, 8))) void k_assign_blk2(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, FUSE, STRIDE, VT>(f, rem, Hv, bm);
}
template <int R, bool FUSE, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void k_assign_blk2_w8(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, FUSE, STRIDE, VT>(f, rem, Hv, bm);
}
template <int R, bool FUSE, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(6, 8))) void k_assign_blk2_w6(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, FUSE, STRIDE, VT>(f, rem, Hv, bm);
}

template <int R, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(7, 8))) void k_assign_bin(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, true, STRIDE, VT, true>(f, rem, Hv, bm);
}
template <int R, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(6, 8))) void k_assign_bin_w6(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, true, STRIDE, VT, true>(f, rem, Hv, bm);
}

template <int R, int STRIDE, bool VT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(7, 8))) void k_assign_pre(FrameDev f, int rem, int Hv, BlkMap bm) {
    assign_blk2_body<R, true, STRIDE, VT, false, true>(f, rem, Hv, bm);
}

template <int R, bool FUSE, int STRIDE>
static void launch_assign_blk_t(const FrameDev& f_, int nframes, int rem, int Hv, hipStream_t st) {
    FrameDev f = f_;
    if (FUSE) { f.tab = f.tabs; f.tab_words = f.tabs_words; f.tab_dyoff = f.tabs_dyoff; f.tab_nrpad = f.tabs_nrpad; }   // table of the subsampled stride
    if (FUSE && R == 16 && !f.tab_vmode) { f.tab = f.tabs16; f.tab_words = f.tabs16_words; f.tab_dyoff = f.tabs16_dyoff; f.tab_nrpad = f.tabs16_nrpad; }
    // row-vector mode, full pass with at most 16 rows per wavefront: the shorter stride-1 table (48 entries = 768 bytes of LDS per block less)
    if (!FUSE && R <= 16 && f.tab_vmode && f.tabv16_words > 0) { f.tab = f.tabv16; f.tab_words = f.tabv16_words; f.tab_dyoff = f.tabv16_dyoff; }
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * R - 1) / (kWavesPerBlock * R), nframes);
    const size_t lds2 = ((size_t)f.tab_words * sizeof(uint32_t) + 1023) / 1024 * 1024;       // whole KB: what the LDS-DMA pieces fill
    if constexpr (R == 32) {
        static_assert(!FUSE, "32 rows per wavefront: full pass only");
        launch((k_assign_blk2_w6<R, FUSE, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    } else if constexpr (R == 16 && !FUSE) {
        if (f.tab_vmode) launch((k_assign_blk2_w8<R, FUSE, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
        else launch((k_assign_blk2_w8<R, FUSE, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    } else if constexpr (R == 16 && FUSE) {
        if (f.tab_vmode) launch((k_assign_blk2_w6<R, FUSE, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
        else launch((k_assign_blk2_w6<R, FUSE, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    } else {
        if (f.tab_vmode) launch((k_assign_blk2<R, FUSE, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
        else launch((k_assign_blk2<R, FUSE, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    }
}

template <int R, int STRIDE>
static void launch_assign_bin_t(const FrameDev& f_, int nframes, int rem, int Hv, hipStream_t st) {
    FrameDev f = f_;
    f.tab = f.tabs; f.tab_words = f.tabs_words; f.tab_dyoff = f.tabs_dyoff; f.tab_nrpad = f.tabs_nrpad;     // table of the subsampled stride
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * R - 1) / (kWavesPerBlock * R), nframes);
    const size_t lds2 = ((size_t)f.tab_words * sizeof(uint32_t) + 1023) / 1024 * 1024;
    if constexpr (R == 16) {
        launch((k_assign_bin_w6<R, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    } else if constexpr (R == 4) {
        launch((k_assign_bin<R, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));      // (2-D table only)
    } else {
        if (f.tab_vmode) launch((k_assign_bin<R, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
        else launch((k_assign_bin<R, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    }
}

template <int STRIDE>
static void launch_assign_pre_t(const FrameDev& f_, int nframes, int rem, int Hv, hipStream_t st) {
    FrameDev f = f_;
    f.tab = f.tabs; f.tab_words = f.tabs_words; f.tab_dyoff = f.tabs_dyoff; f.tab_nrpad = f.tabs_nrpad;     // table of the subsampled stride
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * 8 - 1) / (kWavesPerBlock * 8), nframes);
    const size_t lds2 = ((size_t)f.tab_words * sizeof(uint32_t) + 1023) / 1024 * 1024;
    if (f.tab_vmode) launch((k_assign_pre<8, STRIDE, true>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
    else launch((k_assign_pre<8, STRIDE, false>), blk_grid(grid), dim3(256), lds2, st, f, rem, Hv, blk_map(grid));
}
// preemptive mode: the fused pass of the block kernel on the bins of the active clusters (see assign_blk2_body, PRE); false: the
// geometry does not take the block kernel (the caller uses the generic kernel)
bool launch_assign_pre(const FrameDev& f_, int nframes, int rem, int stride, int buf, int sbuf, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (!(f.tab_words > 0 && stride >= 1 && stride <= 3 && (unsigned long long)f.N * 4ull < 0x7FFFFFFFull)) return false;
    if (Hv <= 0) return true;
    if (stride == 1) launch_assign_pre_t<1>(f, nframes, rem, Hv, st);
    else if (stride == 2) launch_assign_pre_t<2>(f, nframes, rem, Hv, st);
    else launch_assign_pre_t<3>(f, nframes, rem, Hv, st);
    return true;
}

static bool blk_kernel_applies(const FrameDev& f, int stride) {
    return f.tab_words > 0 && stride >= 1 && stride <= 3 && (unsigned long long)f.N * 4ull < 0x7FFFFFFFull;
}

bool assign_fuses_cluster_pass(const FrameDev& f, int stride) { return blk_kernel_applies(f, stride); }

void launch_assign_fused_bin(const FrameDev& f_, int nframes, int rem, int stride, int it, hipStream_t st) {
    const FrameDev f = rotated(f_, it & 1, it & 1, it % 3);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const int tiles_x = (f.W + kTileW - 1) / kTileW;
    const int blocks8 = nframes * tiles_x * ((Hv + kWavesPerBlock * 8 - 1) / (kWavesPerBlock * 8));
    const bool r16 = f.tab_vmode && blocks8 > 3072;      // as launch_assign
    // Launches that leave most of the chip idle even with 8-row wavefronts (one or two 1280x720 frames: 160 blocks each on 256
    // CUs): 4 rows per wavefront.  A wavefront alone on its SIMD issues a dependent instruction every 8 - 10 clocks, so a
    // block's life there is its instruction chain; half the rows halve the argmin, label and sum phases of that chain.
    const bool r4 = !f.tab_vmode && stride == 3 && blocks8 <= 384;      // (measured: one 1280x720 frame per group 179.5 -> 175.7 us, profiles/r04_assign_experiments.txt)
    if (r4) {
        launch_assign_bin_t<4, 3>(f, nframes, rem, Hv, st);
    } else if (r16) {
        if (stride == 1) launch_assign_bin_t<16, 1>(f, nframes, rem, Hv, st);
        else if (stride == 2) launch_assign_bin_t<16, 2>(f, nframes, rem, Hv, st);
        else launch_assign_bin_t<16, 3>(f, nframes, rem, Hv, st);
    } else {
        if (stride == 1) launch_assign_bin_t<8, 1>(f, nframes, rem, Hv, st);
        else if (stride == 2) launch_assign_bin_t<8, 2>(f, nframes, rem, Hv, st);
        else launch_assign_bin_t<8, 3>(f, nframes, rem, Hv, st);
    }
}

static int assign_rows_per_wave(const FrameDev& f, int nframes, int Hv, bool fuse_update) {
    const int tiles_x = (f.W + kTileW - 1) / kTileW;
    const int blocks8 = nframes * tiles_x * ((Hv + kWavesPerBlock * 8 - 1) / (kWavesPerBlock * 8));
    // 32 rows (row-vector table only): launches that keep the chip busy for three rounds of 16-row blocks and more -- the block
    // prologue and the per-candidate fetch are then paid once per 8192 pixels instead of 4096
    if (!fuse_update && f.tab_vmode && f.tab_rows32 && blocks8 > 4 * 6144) return 32;
    return (!fuse_update && blocks8 > 3072) ? 16 : 8;
}

template <int R>
static void launch_assign_r(const FrameDev& f, int nframes, int rem, int stride, int Hv, int buf, int sbuf, bool fuse_update, hipStream_t st) {
    dim3 grid((f.W + kTileW - 1) / kTileW, (Hv + kWavesPerBlock * R - 1) / (kWavesPerBlock * R), nframes);
    const size_t lds = (size_t)f.lut_words * sizeof(uint32_t);
    if (f.lut_words > 0) {
        if (fuse_update) launch((k_assign<R, true, true>), grid, dim3(256), lds, st, f, rem, stride, Hv, buf, sbuf);
        else launch((k_assign<R, false, true>), grid, dim3(256), lds, st, f, rem, stride, Hv, buf, sbuf);
    } else {
        if (fuse_update) launch((k_assign<R, true, false>), grid, dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
        else launch((k_assign<R, false, false>), grid, dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
    }
}

PassGeom launch_assign(const FrameDev& f_, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st, int obuf) {
    const FrameDev f = rotated(f_, buf, sbuf, obuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    PassGeom pg;
    pg.rem = rem; pg.stride = stride; pg.Hv = Hv; pg.BH = kWavesPerBlock * 8; pg.use_slots = 0;
    if (Hv <= 0) return pg;
    if (f.tab_words > 0 && stride >= 1 && stride <= 3 && (unsigned long long)f.N * 4ull < 0x7FFFFFFFull) {       // block-level packed kernel (its buffer loads address a plane with 31-bit byte offsets; larger planes take the 32-bit kernel)
        if (!fuse_update && stride == 1) {
            const int rpw = assign_rows_per_wave(f, nframes, Hv, false);
            if (rpw == 32) launch_assign_blk_t<32, false, 1>(f, nframes, rem, Hv, st);
            else if (rpw == 16) launch_assign_blk_t<16, false, 1>(f, nframes, rem, Hv, st);
            else launch_assign_blk_t<8, false, 1>(f, nframes, rem, Hv, st);
            return pg;
        }
        if (fuse_update) {
            pg.use_slots = 1;
            // Launches of many blocks (3840x2160 frames in groups) with the row-vector table: 16 rows per wavefront, i.e. 64
            // visited rows per block -- the block prologue (candidate list, table) is paid once per 4096 pixels instead of
            // 2048.  Small launches keep 8 rows (more, shorter-lived blocks fill the chip better); so does the 2-D table,
            // whose LDS footprint grows with the rows a wavefront spans.
            const int tiles_x = (f.W + kTileW - 1) / kTileW;
            const int fblocks8 = nframes * tiles_x * ((Hv + kWavesPerBlock * 8 - 1) / (kWavesPerBlock * 8));
            // (2-D table: 16 rows from 2048 eight-row blocks on -- the launch then is one round of blocks instead of one and a bit)
            const bool r16 = (f.tab_vmode && fblocks8 > 3072) || (!f.tab_vmode && f.tabs16_words > 0 && fblocks8 > 2048);
            if (r16) {
                pg.BH = kWavesPerBlock * 16;
                if (stride == 1) launch_assign_blk_t<16, true, 1>(f, nframes, rem, Hv, st);
                else if (stride == 2) launch_assign_blk_t<16, true, 2>(f, nframes, rem, Hv, st);
                else launch_assign_blk_t<16, true, 3>(f, nframes, rem, Hv, st);
                return pg;
            }
            if (stride == 1) launch_assign_blk_t<8, true, 1>(f, nframes, rem, Hv, st);
            else if (stride == 2) launch_assign_blk_t<8, true, 2>(f, nframes, rem, Hv, st);
            else launch_assign_blk_t<8, true, 3>(f, nframes, rem, Hv, st);
            return pg;
        }
    }
    if (assign_rows_per_wave(f, nframes, Hv, fuse_update) >= 16) launch_assign_r<16>(f, nframes, rem, stride, Hv, buf, sbuf, fuse_update, st);
    else launch_assign_r<8>(f, nframes, rem, stride, Hv, buf, sbuf, fuse_update, st);
    return pg;
}

// ---------------------------------------------------------------------------------------------
// Generic gather: one thread per visited pixel, candidates read straight from the coarse cells,
// 64-bit key (u16 distance << 32 | phase << 16 | k), spatial term from the (2S+1)^2 patch itself and
// the distance wrapped to 16 bits exactly like src/context.cpp:285-286.  It is the fallback for inputs
// the tiled kernel does not take (candidate-list overflow, S < 8, non-Manhattan patch, no exact
// fixed-point multiplier) and an independent cross-check of the tiled kernel in the tests.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ void generic_test(const ClusterRec& rec, int y, int x, int S, int P, uint32_t pix,
                                                    const uint16_t* __restrict__ patch, unsigned long long& best) {
    const int dy = y - (int)(rec.yx >> 16), dx = x - (int)(rec.yx & 0xFFFFu);
    if (dy < -S || dy > S || dx < -S || dx > S) return;
    const uint32_t sp = patch[(dy + S) * P + (dx + S)];
    const uint32_t d = (sp + __builtin_amdgcn_sad_u8(pix, rec.lab, 0u)) & 0xFFFFu;
    const unsigned long long key = ((unsigned long long)d << 32) | rec.tag;
    best = key < best ? key : best;
}

template <bool FUSE>
__global__ __launch_bounds__(256) void k_assign_generic(FrameDev f, int rem, int stride, int Hv, int buf, int sbuf) {
    f.select(blockIdx.y);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Hv * f.W) return;
    const int v = (int)(i / f.W), x = (int)(i - (size_t)v * f.W);
    const int y = rem + v * stride;
    const size_t p = (size_t)y * f.W + x;
    const uint32_t pix = f.lab[p];
    const int S = f.S, P = 2 * S + 1, sh = f.cs_shift;
    unsigned long long best = ~0ull;
    const int cxa = max(x - S, 0) >> sh, cxb = min(x + S, f.W - 1) >> sh;
    const int cya = max(y - S, 0) >> sh, cyb = min(y + S, f.H - 1) >> sh;
    for (int cyi = cya; cyi <= cyb; ++cyi)
        for (int cxi = cxa; cxi <= cxb; ++cxi) {
            const int cell = cyi * f.ncx + cxi;
            const uint32_t cnt = min(f.cell_cnt[0][cell], (uint32_t)kCellCap);
            for (uint32_t s = 0; s < cnt; ++s)
                generic_test(f.cell_items[0][(size_t)cell * kCellCap + s], y, x, S, P, pix, f.patch2d, best);
        }
    const uint32_t novf = *f.ovf_cnt[0];
    for (uint32_t s = 0; s < novf; ++s) generic_test(f.ovf_items[0][s], y, x, S, P, pix, f.patch2d, best);
    uint32_t lbl;
    if (best != ~0ull) { lbl = (uint32_t)(best & 0xFFFFull); f.labels[p] = (uint16_t)lbl; }
    else lbl = f.labels[p];
    // preemptive mode: only pixels of active (2S x 2S) cells take part in the update unless every cluster is active
    // (src/context.cpp:304-343)
    if (FUSE && lbl != 0xFFFFu && (!f.cell_act || f.pre_flags[0] || f.cell_act[(y / (2 * S)) * f.pre_cw + x / (2 * S)]))
        global_accumulate(f.sums[0], lbl, (uint32_t)y, (uint32_t)x, pix);
}

void launch_assign_generic(const FrameDev& f_, int nframes, int rem, int stride, int buf, int sbuf, bool fuse_update, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int Hv = (f.H - rem + stride - 1) / stride;
    if (Hv <= 0) return;
    const size_t n = (size_t)Hv * f.W;
    const int blocks = (int)((n + 255) / 256);
    if (fuse_update) launch(k_assign_generic<true>, dim3(blocks, nframes), dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
    else launch(k_assign_generic<false>, dim3(blocks, nframes), dim3(256), 0, st, f, rem, stride, Hv, buf, sbuf);
}

}  // namespace fslic
