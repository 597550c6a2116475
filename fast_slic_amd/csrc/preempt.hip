// preempt.hip -- preemptive mode (src/preemptive.h, PreemptiveGrid): clusters that have stopped moving drop out of the
// assignment, tiles without a moving cluster nearby drop out of the update.
// Hand-written gfx950 kernel; see kernels.h for the launch interface and DESIGN.md for the design.
//
// One block per frame does what BaseContext::update (src/context.cpp:356-387) and PreemptiveGrid::set_new_clusters
// (src/preemptive.h:114-178) do after every assignment pass, in the reference's order:
//   A  integer centroid update of the clusters whose is_updatable counter is non-zero (the others keep position,
//      colour AND num_members, src/context.cpp:360)
//   B  |old.x - new.x| + |old.y - new.y| (integer abs of the truncated differences, as the reference build computes it)
//      < l1_thres: is_updatable--, else is_updatable = cooldown (2)
//   C  every cluster within 2S (Chebyshev) of a cluster that is still updatable becomes active, and so does the
//      (2S x 2S) cell it lies in
//   D  all_active = every cluster active
//   E  the active clusters are binned for the next assignment pass
// K^2 / 1024 neighbour tests per thread: O(10 us) for K = 1600.  The mode is an approximation meant to save CPU time;
// the Slic variant takes the block assign kernel on the bins of the active clusters (assign.hip, PRE; the generic kernel where the
// geometry is outside it), the
// float-distance variants (src/context.h:100-125: BaseContext<float>::iterate is the same template, src/context.cpp:152-181) their
// own assign kernels on the bins of the active clusters, with the member sums restricted to the active cells.
#include "device_common.h"
#include <atomic>

namespace fslic {

// LISTS: the activity marking (C) through per-cell lists of the updatable clusters in LDS -- the structure the reference itself uses
// (cluster_grid, src/preemptive.h:118-123, :141-163): a cluster can only be activated from the 3 x 3 cells of side 2S around its own,
// so it tests the handful of updatable clusters listed there instead of all K (K^2 / 1024 tests per thread were 470 us per call at
// K = 1600: 93 % of a preemptive group once the assignment ran on the block kernel).  Positions (4 B), lists (2 B) and cell offsets in
// dynamic LDS: K <= 8192 and at most 4096 cells; larger problems keep the all-pairs form.
template <bool LISTS>
__global__ __launch_bounds__(1024) void k_preempt_update(FrameDev f, int rebin_all, float l1_thres) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int ncell2 = f.pre_cw * f.pre_ch;
    uint32_t* const s_pos = s_dyn;                                     // K: new positions
    uint32_t* const s_start = s_dyn + (LISTS ? f.K : 0);               // ncell2 + 1: first list entry of every cell
    uint32_t* const s_fill = s_start + (LISTS ? ncell2 + 1 : 0);       // ncell2
    uint16_t* const s_list = reinterpret_cast<uint16_t*>(s_fill + (LISTS ? ncell2 : 0));      // K: updatable clusters, cell by cell
    f.select(blockIdx.x);
    const int tid = threadIdx.x, K = f.K, S2 = 2 * f.S;
    for (int gi = tid; gi < f.ncell; gi += 1024) f.cell_cnt[1][gi] = 0;
    if (tid == 0) *f.ovf_cnt[1] = 0;
    const uint32_t* __restrict__ pos = rebin_all ? f.cl_yx : f.cl_yx_out;
    if (!rebin_all) {
        if (LISTS) {
            for (int c = tid; c <= ncell2; c += 1024) s_start[c] = 0u;
            for (int c = tid; c < ncell2; c += 1024) s_fill[c] = 0u;
            __syncthreads();
        }
        for (int k = tid; k < K; k += 1024) {
            const uint32_t yx_old = f.cl_yx[k];
            const uint32_t* s = f.sums[0] + 6 * (size_t)k;
            uint32_t* z0 = f.sums[1] + 6 * (size_t)k;
            uint32_t yx = yx_old;
            uint32_t upd = f.cl_upd[k];
            if (upd) {
                const int32_t n = (int32_t)s[0];
                f.cl_n[k] = (uint32_t)n;
                // |old.x - new.x| + |old.y - new.y| as the reference build evaluates it (src/preemptive.h:131-133): the FLOAT
                // differences truncated to int (cvttss2si), integer abs -- the same thing as below for the quantised centroids,
                // not the same for the float centroids of the 'noq' variant
                int l1 = 0;
                if (n != 0) {
                    const int32_t h = n / 2;
                    int y = min(max(((int32_t)s[1] + h) / n, 0), f.H - 1);
                    int x = min(max(((int32_t)s[2] + h) / n, 0), f.W - 1);
                    const int32_t L = ((int32_t)s[3] + h) / n, a = ((int32_t)s[4] + h) / n, b = ((int32_t)s[5] + h) / n;
                    if (f.cl_f) {
                        // centroid_quantization_enabled() == false (src/context.cpp:374-380): (float)sum / n per component; bins,
                        // windows and the activity grid use the integer parts
                        float* c = f.cl_f + 8 * (size_t)k;
                        const float fn = (float)n, oy = c[0], ox = c[1];
                        const float fy = (float)(int32_t)s[1] / fn, fx = (float)(int32_t)s[2] / fn;
                        c[0] = fy; c[1] = fx;
                        c[2] = (float)(int32_t)s[3] / fn; c[3] = (float)(int32_t)s[4] / fn; c[4] = (float)(int32_t)s[5] / fn;
                        y = min(max((int)fy, 0), f.H - 1);
                        x = min(max((int)fx, 0), f.W - 1);
                        l1 = abs((int)(ox - fx)) + abs((int)(oy - fy));
                    } else {
                        l1 = abs((int)(yx_old & 0xFFFFu) - x) + abs((int)(yx_old >> 16) - y);
                    }
                    yx = ((uint32_t)y << 16) | (uint32_t)x;
                    f.cl_lab[k] = (uint32_t)(L & 255) | ((uint32_t)(a & 255) << 8) | ((uint32_t)(b & 255) << 16);
                    f.cl_moved[k] = 1;
                }
                upd = ((float)l1 < l1_thres) ? upd - 1u : 2u;
                f.cl_upd[k] = upd;
            }
            f.cl_yx_out[k] = yx;
            f.cl_act[k] = 0u;
            if (LISTS) {
                s_pos[k] = yx;
                if (upd) atomicAdd(&s_start[1 + ((int)(yx >> 16) / S2) * f.pre_cw + (int)(yx & 0xFFFFu) / S2], 1u);
            }
            z0[0] = 0; z0[1] = 0; z0[2] = 0; z0[3] = 0; z0[4] = 0; z0[5] = 0;
        }
        for (int c = tid; c < f.pre_cw * f.pre_ch; c += 1024) f.cell_act[c] = 0u;
        __syncthreads();
        if (LISTS) {
            // inclusive scan of the per-cell counts by one wavefront (a lane sums its chunk of cells, the 64 chunk totals are scanned
            // across the lanes, the lane writes its chunk's running totals), then the lists
            if (tid < 64) {
                const int chunk = (ncell2 + 63) / 64;
                const int lo = 1 + tid * chunk, hi = min(lo + chunk, ncell2 + 1);
                uint32_t sum = 0;
                for (int c = lo; c < hi; ++c) sum += s_start[c];
                uint32_t inc = sum;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t u = __shfl_up(inc, off);
                    if (tid >= off) inc += u;
                }
                uint32_t run = inc - sum;
                for (int c = lo; c < hi; ++c) { run += s_start[c]; s_start[c] = run; }
            }
            __syncthreads();
            for (int k = tid; k < K; k += 1024) {
                if (!f.cl_upd[k]) continue;
                const uint32_t yx = s_pos[k];
                const int cell = ((int)(yx >> 16) / S2) * f.pre_cw + (int)(yx & 0xFFFFu) / S2;
                s_list[s_start[cell] + atomicAdd(&s_fill[cell], 1u)] = (uint16_t)k;
            }
            __syncthreads();
            for (int j = tid; j < K; j += 1024) {
                const uint32_t yxj = s_pos[j];
                const int yj = (int)(yxj >> 16), xj = (int)(yxj & 0xFFFFu);
                const int cy = yj / S2, cx = xj / S2;
                bool act = false;
                for (int ny = max(cy - 1, 0); ny <= min(cy + 1, f.pre_ch - 1) && !act; ++ny)
                    for (int nx = max(cx - 1, 0); nx <= min(cx + 1, f.pre_cw - 1) && !act; ++nx) {
                        const int c = ny * f.pre_cw + nx;
                        for (uint32_t q = s_start[c]; q < s_start[c + 1] && !act; ++q) {
                            const uint32_t yxk = s_pos[s_list[q]];
                            act = abs(yj - (int)(yxk >> 16)) <= S2 && abs(xj - (int)(yxk & 0xFFFFu)) <= S2;
                        }
                    }
                if (act) { f.cl_act[j] = 1u; f.cell_act[cy * f.pre_cw + cx] = 1u; }
            }
        } else {
        for (int k = tid; k < K; k += 1024) {
            if (!f.cl_upd[k]) continue;
            const uint32_t yxk = f.cl_yx_out[k];
            const int yk = (int)(yxk >> 16), xk = (int)(yxk & 0xFFFFu);
            for (int j = 0; j < K; ++j) {
                const uint32_t yxj = f.cl_yx_out[j];
                const int yj = (int)(yxj >> 16), xj = (int)(yxj & 0xFFFFu);
                if (abs(yj - yk) <= S2 && abs(xj - xk) <= S2) {
                    f.cl_act[j] = 1u;
                    f.cell_act[(yj / S2) * f.pre_cw + xj / S2] = 1u;
                }
            }
        }
        }
        __syncthreads();
        __shared__ int s_cnt;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int mine = 0;
        for (int k = tid; k < K; k += 1024) mine += (int)f.cl_act[k];
        if (mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        if (tid == 0) f.pre_flags[0] = s_cnt == K ? 1u : 0u;
    }
    const uint32_t gen = f.generation();
    for (int k = tid; k < K; k += 1024) {
        const uint32_t yx = pos[k];
        const int y = (int)(yx >> 16), x = (int)(yx & 0xFFFFu);
        const uint32_t phase = (uint32_t)(2 * ((y / f.T) & 1) + ((x / f.T) & 1));
        ClusterRec rec;
        rec.yx = yx; rec.lab = f.cl_lab[k]; rec.tag = (phase << 16) | (uint32_t)k; rec.pad = gen;
        const bool out = !rebin_all && !f.cl_act[k];       // src/context.cpp:218
        // (the kernels that fall back to a walk over ALL clusters when a block sees more than 64 -- realdist.hip, lsc.hip -- select by
        // the tag's phase: an inactive cluster carries a phase no walk asks for)
        f.cl_tag[k] = out ? (0xFFFF0000u | (uint32_t)k) : rec.tag;
        f.cl_shadow[k] = 0u;
        if (out) continue;
        const int cell = (y >> f.cs_shift) * f.ncx + (x >> f.cs_shift);
        const uint32_t slot = atomicAdd(&f.cell_cnt[0][cell], 1u);
        if (slot < (uint32_t)kCellCap) f.cell_items[0][(size_t)cell * kCellCap + slot] = rec;
        else f.ovf_items[0][atomicAdd(f.ovf_cnt[0], 1u)] = rec;
    }
}

void launch_preempt_update(const FrameDev& f_, int nframes, int buf, int sbuf, bool rebin_all, float l1_thres, hipStream_t st) {
    const FrameDev f = rotated(f_, buf, sbuf);
    const int ncell2 = f.pre_cw * f.pre_ch;
    if (f.K <= 8192 && ncell2 <= 4096) {
        const size_t lds = ((size_t)f.K + 2 * (size_t)ncell2 + 1) * 4 + (size_t)f.K * 2 + 16;
        // the attribute belongs to the device that is current when it is set: once per DEVICE (launch.h), and a device on which it could
        // not be set takes the all-pairs kernel instead of failing the launch
        constexpr int kMaxLds = (8192 + 2 * 4096 + 1) * 4 + 8192 * 2 + 16;
        static std::atomic<int> attr_state[64];
        if (!max_dynamic_lds_on_this_device(k_preempt_update<true>, kMaxLds, attr_state) && lds > 64 * 1024) {
            launch(k_preempt_update<false>, dim3(nframes), dim3(1024), 0, st, f, rebin_all ? 1 : 0, l1_thres);
            return;
        }
        launch(k_preempt_update<true>, dim3(nframes), dim3(1024), lds, st, f, rebin_all ? 1 : 0, l1_thres);
    } else {
        launch(k_preempt_update<false>, dim3(nframes), dim3(1024), 0, st, f, rebin_all ? 1 : 0, l1_thres);
    }
}

}  // namespace fslic
