// lab.hip -- RGB -> CIELAB prepass (src/cielab.h)
// Hand-written gfx950 (CDNA4, wave64) kernels of the SLIC hot path; see kernels.h for the launch interface and
// DESIGN.md for the design.  All arithmetic on this path is integer, so results are bit-identical to the reference.
#include <cstdio>
#include <cstdlib>
#include "device_common.h"

namespace fslic {

// =============================================================================================
// RGB -> CIELAB prepass.  Restates FastCIELabCvt::convert (src/cielab.h:309-325) with both lookup
// tables staged in LDS (16.5 KB); 4 pixels per thread: 3 dword loads in, one 16-byte store out.
// Algorithmic traffic: 3 B read + 4 B written per pixel, + 2 B written per pixel where the kernel also fills the label plane (the
// first kernel of every group): 9 B per pixel.  8 x 3840x2160 take 99 - 102 us = 5.9 TB/s, the rate of a plain streaming copy on this
// part (6.2 TB/s, k_copy16): HBM-bound there.  16 x 1280x720: 29 - 31 us = 4.3 TB/s (512 blocks, ramp and tail).  Measured without
// effect on either (round 5, profiles/r05_experiments.txt): 12 % fewer vector instructions (the dot2 rows below), two quads in
// flight per thread.
// =============================================================================================
struct LabConst { uint32_t c01[3], c2[3]; };      // per matrix row: coefficient 0 | coefficient 1 << 16, coefficient 2 (all below 2^16: launch_rgb_to_lab)
typedef unsigned short lab_us2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t lab_px(uint32_t R, uint32_t G, uint32_t B,
                                                  const uint32_t* sg, const uint16_t* sl, const LabConst& k) {
    // A matrix row is c0 * r + c1 * g + c2 * b with the gamma values below 2^14, the coefficients below 2^16 and the sum below 2^30, so
    // the results are the reference's ints whichever way the products are formed: one v_dot2_u32_u16 on (r | g << 16) with the 24-bit
    // product of the blue term as its addend -- 7 instructions per pixel for the three rows instead of 12 (three 24-bit products and a
    // three-operand add per row; 32-bit v_mul_lo_u32, a quarter of the rate, before that: nine of them were more than half of this
    // kernel's vector time).
    // gamma table: entry e of the lane's own copy, (e * 32 + lane % 32) dwords into the array -- copy c lies entirely in LDS bank c, and
    // the two lanes that share a copy (i, i + 32) are served in different halves of the instruction: no bank conflicts, where the 128
    // dwords of a single copy met ~3.5 lanes per bank for arbitrary pixel values (63 % of this kernel's LDS time were conflicts,
    // profiles/r06_pipeline_stalls.txt).  `sg` is the lane's base (copy offset applied), one v_lshl_add per look-up.
    const uint32_t sr = sg[R << 5], sgv = sg[G << 5], sb = sg[B << 5];
    const lab_us2 rg = __builtin_bit_cast(lab_us2, sr | (sgv << 16));
    auto row = [&](int i) -> int {
        return (int)(__builtin_amdgcn_udot2(rg, __builtin_bit_cast(lab_us2, k.c01[i]), __umul24(k.c2[i], sb), false) >> 16);
    };
    const int xr = row(0), yr = row(1), zr = row(2);
    const int fx = sl[xr], fy = sl[yr], fz = sl[zr];
    const int ciel = 116 * fy - (16 << 13);
    const int ciea = 500 * (fx - fy) + (128 << 13);
    const int cieb = 200 * (fy - fz) + (128 << 13);
    int l = (int)((unsigned)ciel >> 12);
    int a = (int)((unsigned)ciea >> 12) - 128;
    int b = (int)((unsigned)cieb >> 12) - 128;
    l = min(max(l, 0), 255);
    a = min(max(a, 0), 255);
    b = min(max(b, 0), 255);
    return (uint32_t)l | ((uint32_t)a << 8) | ((uint32_t)b << 16);
}

template <bool CONVERT>
__global__ __launch_bounds__(1024) void k_rgb_to_lab(FrameDev f, const uint16_t* __restrict__ g_gamma,
                                                     const uint16_t* __restrict__ g_labtbl, LabConst kc, int init_labels, uint32_t gen_step,
                                                     char* zero_a, unsigned zero_a_quads, uint32_t* zero_b, unsigned zero_b_words,
                                                     const uint32_t* __restrict__ stage_yx, size_t stage_stride, int stage_K, int conv_blocks) {
    __shared__ uint32_t s_gamma32[CONVERT ? 256 * 32 : 1];      // 32 copies of the gamma table, one per LDS bank (see lab_px): 32 KB
    __shared__ uint16_t s_lab[8194];
    // first kernel of a group: new generation of bin stamps (nothing in this kernel reads them; the stream orders the rest)
    if (gen_step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *f.gen_base += gen_step;
    f.select(blockIdx.y);
    if ((int)blockIdx.x >= conv_blocks) {
        // blocks behind the conversion's: the group's cluster centres from where the host staged them (pinned memory) into device memory
        // -- the read over the bus (~2 us) runs beside the conversion instead of at the head of the first cluster pass
        const uint32_t* src = stage_yx + (size_t)blockIdx.y * stage_stride;
        for (int i = ((int)blockIdx.x - conv_blocks) * (int)blockDim.x + (int)threadIdx.x; i < stage_K; i += ((int)gridDim.x - conv_blocks) * (int)blockDim.x)
            f.cl_yx[i] = src[i];
        return;
    }
    const uint8_t* __restrict__ rgb = f.rgb;
    uint32_t* __restrict__ lab = f.lab;
    uint16_t* __restrict__ labels_init = init_labels ? f.labels : nullptr;
    const int N = f.N;
    const int aligned4 = ((reinterpret_cast<uintptr_t>(rgb) & 3) == 0) ? 1 : 0;
    const int nquad = aligned4 ? (N >> 2) : 0;
    const int gstride = conv_blocks * blockDim.x;
    // Software-pipelined over the thread's quads: the next quad's loads are issued BEFORE this quad's stores, so that
    // waiting for them does not also wait for the acknowledgement of the written-through stores (one vector-memory
    // counter covers both; a load issued after a store cannot be waited for without the store).
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    if (q < nquad) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb) + 3 * (size_t)q;
        w0 = src[0]; w1 = src[1]; w2 = src[2];
    }
    // The tables are requested with the first quad's loads in flight (a one-frame launch is 225 blocks of one quad per thread: the
    // two round trips were one after the other).  1024 threads per block (the launcher).
    uint32_t tg0 = 0, tg1 = 0, tg2 = 0, tg3 = 0, tl0 = 0, tl1 = 0, tl2 = 0, tl3 = 0, tl4 = 0;
    if (CONVERT) {
        const uint32_t* gt = reinterpret_cast<const uint32_t*>(g_labtbl);   // 8194 entries (last one padding) = 4097 words
        // thread t fills copy t % 32 of the gamma entries 8 * (t / 32) .. + 7 (four dwords of the source)
        const uint32_t* gg = reinterpret_cast<const uint32_t*>(g_gamma) + 4 * (threadIdx.x >> 5);
        tg0 = gg[0]; tg1 = gg[1]; tg2 = gg[2]; tg3 = gg[3];
        tl0 = gt[threadIdx.x]; tl1 = gt[1024 + threadIdx.x]; tl2 = gt[2048 + threadIdx.x]; tl3 = gt[3072 + threadIdx.x];
        if (threadIdx.x == 0) tl4 = gt[4096];
    }
    const uint32_t* const s_gamma = s_gamma32 + (CONVERT ? (threadIdx.x & 31) : 0);      // the lane's copy
    if (CONVERT) {
        uint32_t* sl = reinterpret_cast<uint32_t*>(s_lab);
        uint32_t* const gd = s_gamma32 + ((threadIdx.x >> 5) * 8) * 32 + (threadIdx.x & 31);
        gd[0 * 32] = tg0 & 0xFFFFu; gd[1 * 32] = tg0 >> 16; gd[2 * 32] = tg1 & 0xFFFFu; gd[3 * 32] = tg1 >> 16;
        gd[4 * 32] = tg2 & 0xFFFFu; gd[5 * 32] = tg2 >> 16; gd[6 * 32] = tg3 & 0xFFFFu; gd[7 * 32] = tg3 >> 16;
        sl[threadIdx.x] = tl0; sl[1024 + threadIdx.x] = tl1; sl[2048 + threadIdx.x] = tl2; sl[3072 + threadIdx.x] = tl3;
        if (threadIdx.x == 0) sl[4096] = tl4;
        __syncthreads();
    }
    while (q < nquad) {
        const int qn = q + gstride;
        uint32_t n0 = 0, n1 = 0, n2 = 0;
        if (qn < nquad) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(rgb) + 3 * (size_t)qn;
            n0 = src[0]; n1 = src[1]; n2 = src[2];
        }
        uint32_t r0 = w0 & 255, g0 = (w0 >> 8) & 255, b0 = (w0 >> 16) & 255;
        uint32_t r1 = w0 >> 24, g1 = w1 & 255, b1 = (w1 >> 8) & 255;
        uint32_t r2 = (w1 >> 16) & 255, g2 = w1 >> 24, b2 = w2 & 255;
        uint32_t r3 = (w2 >> 8) & 255, g3 = (w2 >> 16) & 255, b3 = w2 >> 24;
        uint4 o;
        if (CONVERT) {
            o.x = lab_px(r0, g0, b0, s_gamma, s_lab, kc);
            o.y = lab_px(r1, g1, b1, s_gamma, s_lab, kc);
            o.z = lab_px(r2, g2, b2, s_gamma, s_lab, kc);
            o.w = lab_px(r3, g3, b3, s_gamma, s_lab, kc);
        } else {   // raw copy, src/context.cpp:117-127
            o.x = r0 | (g0 << 8) | (b0 << 16);
            o.y = r1 | (g1 << 8) | (b1 << 16);
            o.z = r2 | (g2 << 8) | (b2 << 16);
            o.w = r3 | (g3 << 8) | (b3 << 16);
        }
        st_stream(reinterpret_cast<uint4*>(lab) + q, o);
        // assignment plane <- 0xFFFF once per call (src/context.cpp:138-145)
        if (labels_init) st_stream(reinterpret_cast<uint2*>(labels_init) + q, make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));
        w0 = n0; w1 = n1; w2 = n2;
        q = qn;
    }
    // tail (and the whole image when the source pointer is not 4-byte aligned)
    for (int p = (nquad << 2) + blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gstride) {
        const uint32_t r = rgb[3 * (size_t)p], g = rgb[3 * (size_t)p + 1], b = rgb[3 * (size_t)p + 2];
        st_stream(lab + p, CONVERT ? lab_px(r, g, b, s_gamma, s_lab, kc) : (r | (g << 8) | (b << 16)));
        if (labels_init) st_stream(labels_init + p, (uint16_t)0xFFFFu);
    }
    // Last, the zeroing of the group's accumulators, bin counters and flags of this frame (nothing reads them before the next
    // kernel): the one vector-memory counter is in order, so a wait for a load also waits for every store issued before it --
    // at the top of the kernel these stores sat in front of the first quad's loads.
    {
        const size_t fo = (size_t)blockIdx.y * f.frame_bytes;
        const unsigned t0 = blockIdx.x * blockDim.x + threadIdx.x, ts = (unsigned)conv_blocks * blockDim.x;
        if (zero_a) { uint4* za = reinterpret_cast<uint4*>(zero_a + fo); for (unsigned i = t0; i < zero_a_quads; i += ts) za[i] = make_uint4(0u, 0u, 0u, 0u); }
        if (zero_b) { uint32_t* zb = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(zero_b) + fo); for (unsigned i = t0; i < zero_b_words; i += ts) zb[i] = 0u; }
    }
}

void launch_rgb_to_lab(const FrameDev& f, int nframes, int convert, const LabTables& t, bool init_labels, hipStream_t st, uint32_t gen_step,
                       void* zero_a, size_t zero_a_bytes, uint32_t* zero_b, size_t zero_b_words, const uint32_t* stage_yx, size_t stage_stride) {
    LabConst kc;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            if (t.cb[3 * i + j] < 0 || t.cb[3 * i + j] > 0xFFFF) { fprintf(stderr, "fslic: RGB->XYZ coefficient outside 16 bits\n"); abort(); }      // (src/cielab.h:300-301: all below 1.0 * 65536)
        kc.c01[i] = (uint32_t)t.cb[3 * i] | ((uint32_t)t.cb[3 * i + 1] << 16);
        kc.c2[i] = (uint32_t)t.cb[3 * i + 2];
    }
    int blocks = (f.N / 4 + 1023) / 1024;
    const int cap = (512 + nframes - 1) / nframes;   // about two resident blocks per CU over the whole group, grid-stride
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const int stage_blocks = stage_yx ? (f.K + 1023) / 1024 : 0;      // the centres' prefetch (see the kernel)
    if (convert)
        launch(k_rgb_to_lab<true>, dim3(blocks + stage_blocks, nframes), dim3(1024), 0, st, f, t.gamma, t.labtbl, kc, init_labels ? 1 : 0, gen_step, (char*)zero_a, (unsigned)(zero_a_bytes / 16), zero_b, (unsigned)zero_b_words,
               stage_yx, stage_stride, f.K, blocks);
    else
        launch(k_rgb_to_lab<false>, dim3(blocks + stage_blocks, nframes), dim3(1024), 0, st, f, t.gamma, t.labtbl, kc, init_labels ? 1 : 0, gen_step, (char*)zero_a, (unsigned)(zero_a_bytes / 16), zero_b, (unsigned)zero_b_words,
               stage_yx, stage_stride, f.K, blocks);
}

}  // namespace fslic
