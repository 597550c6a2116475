// graph.hip -- superpixel-graph utilities on a finished label map (src/fast-slic.cpp), gfx950.
//   k_mask_sums        : per-cluster sums of a mask plane            (fast_slic_get_mask_density, src/fast-slic.cpp:141-150)
//   k_density_to_mask  : per-pixel broadcast of a per-cluster value  (fast_slic_cluster_density_to_mask, :156-168)
//   k_adjacent_pairs   : the distinct adjacent label pairs of the map with the raster position of their FIRST
//                        occurrence in the reference's scan (fast_slic_get_connectivity, :16-78); the host orders them
//                        and applies the sequential 12-neighbour cap (engine.cpp)
// All integer work; HBM-bound (2-3 B per pixel), no reshaping.
#include "device_common.h"
#include <algorithm>

namespace fslic {

// ---- mask density -----------------------------------------------------------------------------
// One wavefront per tile of 64 columns x 16 rows.  A superpixel is taller than that, so most pixels of a column carry the
// label of the column's first row: those are summed in a register (no atomic at all) and the 64 registers are merged per
// distinct label at the end (DPP reduction, ONE global atomic per (tile, label)).  Pixels whose label differs from their
// column's first row take the same merge at once, row by row.  Kt = (uint16_t)K as in the reference's comparison
// (src/fast-slic.cpp:147).  A flat one-atomic-per-(64 pixels, label) version took 141 us at 3840x2160.
constexpr int kMaskRows = 16;
static __device__ __forceinline__ void merge_by_label(uint32_t lab, uint32_t val, bool take, uint32_t* __restrict__ sums, int lane) {
    unsigned long long todo = __ballot(take);
    while (todo) {                                                  // wave-uniform: every lane takes part in the reduction
        const int l = __builtin_ctzll(todo);
        const uint32_t cur = __builtin_amdgcn_readlane(lab, l);
        const bool mine = take && lab == cur;
        const unsigned long long same = __ballot(mine);
        const uint32_t s = wave_reduce_add<uint32_t>(mine ? val : 0u);
        if (lane == l && s != 0u) atomicAdd(&sums[cur], s);
        todo &= ~same;
    }
}
__global__ __launch_bounds__(256) void k_mask_sums(const uint16_t* __restrict__ labels, const uint8_t* __restrict__ mask,
                                                   uint32_t* __restrict__ sums, int H, int W, uint32_t Kt) {
    const int lane = LANE();
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const unsigned long long ntx = (unsigned long long)((W + 63) / 64), nty = (unsigned long long)((H + kMaskRows - 1) / kMaskRows);
    const unsigned long long ntiles = ntx * nty, nwaves = (unsigned long long)gridDim.x * 4ull;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 4ull + wave; t < ntiles; t += nwaves) {
        const int ty = (int)(t / ntx), tx = (int)(t - (unsigned long long)ty * ntx);
        const int x = tx * 64 + lane, y0 = ty * kMaskRows;
        const bool okx = x < W;
        const int nrows = min(kMaskRows, H - y0);
        const size_t p0 = (size_t)y0 * W + (size_t)min(x, W - 1);
        const uint32_t lab0 = labels[p0];
        const bool live0 = okx && lab0 < Kt;
        uint32_t labr[kMaskRows], mr[kMaskRows];           // every row's loads in flight at once (rows past the image re-read the last one)
#pragma unroll
        for (int r = 0; r < kMaskRows; ++r) {
            const size_t p = p0 + (size_t)min(r, nrows - 1) * W;
            labr[r] = labels[p];
            mr[r] = mask[p];
        }
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < kMaskRows; ++r) {
            const bool live = okx && r < nrows && labr[r] < Kt;
            const bool same0 = live && labr[r] == lab0;
            acc += same0 ? mr[r] : 0u;
            merge_by_label(labr[r], mr[r], live && !same0, sums, lane);
        }
        merge_by_label(lab0, acc, live0, sums, lane);
    }
}

__global__ __launch_bounds__(256) void k_density_to_mask(const uint16_t* __restrict__ labels, const uint8_t* __restrict__ dens,
                                                         uint8_t* __restrict__ out, unsigned long long N, uint32_t Kt) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long p = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; p < N; p += stride) {
        const uint32_t lab = labels[p];
        out[p] = lab < Kt ? dens[lab] : (uint8_t)0;
    }
}

// ---- adjacency --------------------------------------------------------------------------------
// The reference scans the pixels (i < H-1, j < W-1) in raster order and looks right, down and down-right
// (src/fast-slic.cpp:69-71).  Event key of an occurrence = pixel index * 4 + direction: ascending key == scan order.
// Open-addressing table keyed by the unordered pair (lo << 16 | hi); the value is the minimum event key.
// Plain pre-reads only ever see values that are not newer than the truth (keys only decrease, a filled slot never
// changes), so skipping on them is safe; the atomics decide.
constexpr uint32_t kPairEmpty = 0xFFFFFFFFu;
constexpr uint32_t kMaxProbe = 255u;          // (linear probing at a load of at most 1/2: runs of a few dozen slots are already rare)
static __device__ __forceinline__ uint32_t pair_hash(uint32_t v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}
__global__ __launch_bounds__(256) void k_adjacent_pairs(const uint16_t* __restrict__ labels, int H, int W, uint32_t K,
                                                        uint32_t* __restrict__ tab_pair, unsigned long long* __restrict__ tab_key,
                                                        uint32_t cap_mask, uint32_t* __restrict__ counters) {
    const unsigned long long npx = (unsigned long long)(H - 1) * (unsigned long long)(W - 1);
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long q = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; q < npx; q += stride) {
        // The table is declared full -- more than half of it taken, or a probe sequence longer than kMaxProbe -- as soon as one thread
        // finds it so, and every thread leaves at its next pixel: the host doubles the table and starts over.  (Until round 6 a thread
        // probed the WHOLE table before it gave up, and nobody else stopped: a noise map of 2072x2348 with K = 20 000 -- 14 million distinct
        // pairs against a first table of a million entries -- did not finish in 13 minutes; tests/fuzz_parity.py, graph mode, big shapes.)
        if (__hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        const unsigned long long i = q / (unsigned long long)(W - 1), j = q - i * (unsigned long long)(W - 1);
        const unsigned long long p = i * (unsigned long long)W + j;
        const uint32_t s = labels[p];
        if (s >= K) continue;
        const uint32_t t3[3] = {labels[p + 1], labels[p + (unsigned long long)W], labels[p + (unsigned long long)W + 1]};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t t = t3[d];
            if (t >= K || t == s) continue;
            if (d == 1 && t == t3[0]) continue;                     // same pair as a smaller key of this very pixel
            if (d == 2 && (t == t3[0] || t == t3[1])) continue;
            const uint32_t pair = (min(s, t) << 16) | max(s, t);
            const unsigned long long key = p * 4ull + (unsigned long long)d;
            uint32_t h = pair_hash(pair) & cap_mask;
            bool done = false;
            for (uint32_t probe = 0; probe <= min(cap_mask, kMaxProbe); ++probe) {
                uint32_t cur = tab_pair[h];
                if (cur == kPairEmpty) {
                    cur = atomicCAS(&tab_pair[h], kPairEmpty, pair);
                    if (cur == kPairEmpty) {                                                // counters[0]: distinct pairs stored
                        if (atomicAdd(&counters[0], 1u) >= (cap_mask >> 1)) atomicExch(&counters[1], 1u);      // (the host wants it at most half full)
                        cur = pair;
                    }
                }
                if (cur == pair) {
                    if (tab_key[h] > key) atomicMin(&tab_key[h], key);
                    done = true;
                    break;
                }
                h = (h + 1u) & cap_mask;
            }
            if (!done) atomicExch(&counters[1], 1u);                // counters[1]: table full
        }
    }
}

// ---- launches ---------------------------------------------------------------------------------
static inline int grid_for(unsigned long long items, unsigned long long per_block) {
    const unsigned long long want = (items + per_block - 1) / per_block;
    return (int)std::max<unsigned long long>(1ull, std::min<unsigned long long>(want, 256ull * 32ull));   // grid-stride beyond 32 blocks per CU
}
void launch_mask_sums(const uint16_t* labels, const uint8_t* mask, uint32_t* sums, int H, int W, int K, hipStream_t st) {
    const unsigned long long tiles = (unsigned long long)((W + 63) / 64) * (unsigned long long)((H + kMaskRows - 1) / kMaskRows);
    launch(k_mask_sums, dim3(grid_for(tiles, 4)), dim3(256), 0, st, labels, mask, sums, H, W, (uint32_t)(uint16_t)K);
}
void launch_density_to_mask(const uint16_t* labels, const uint8_t* dens, uint8_t* out, size_t N, int K, hipStream_t st) {
    launch(k_density_to_mask, dim3(grid_for(N, 256)), dim3(256), 0, st, labels, dens, out, (unsigned long long)N, (uint32_t)(uint16_t)K);
}
void launch_adjacent_pairs(const uint16_t* labels, int H, int W, int K, uint32_t* tab_pair, unsigned long long* tab_key,
                           uint32_t cap, uint32_t* counters, hipStream_t st) {
    if (H < 2 || W < 2) return;
    const unsigned long long npx = (unsigned long long)(H - 1) * (unsigned long long)(W - 1);
    launch(k_adjacent_pairs, dim3(grid_for(npx, 256)), dim3(256), 0, st, labels, H, W, (uint32_t)K, tab_pair, tab_key, cap - 1u, counters);
}

// ---- measured HBM copy rate (bench.py's "measured peak": fslic_hip_copy_bandwidth) -------------------------------------------------
// The plain streaming copy the MI355X guide quotes its achievable HBM rate for (6.29 TB/s), in the form that reaches it here
// (scripts/microbench/copy_rate.hip, profiles/r05_microbench_copy_rate.txt): every block copies ONE contiguous 16 KB chunk -- four
// 16-byte loads per lane in flight, then four stores, all marked non-temporal -- and the grid covers the buffer (65 536 blocks per
// GiB).  Grid-stride loops over fewer, longer-lived blocks reach 4.4 - 5.7 TB/s, hipMemcpyAsync 5.3; this form 6.3.
typedef uint32_t copy_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy16(const copy_u4* __restrict__ src, copy_u4* __restrict__ dst, size_t n16) {
    const size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x;
    copy_u4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i0 + u * 256 < n16) v[u] = __builtin_nontemporal_load(src + i0 + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i0 + u * 256 < n16) __builtin_nontemporal_store(v[u], dst + i0 + u * 256);
}
void launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t st) {
    const size_t n16 = bytes / 16;
    launch(k_copy16, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, st, reinterpret_cast<const copy_u4*>(src), reinterpret_cast<copy_u4*>(dst), n16);
}

}  // namespace fslic
