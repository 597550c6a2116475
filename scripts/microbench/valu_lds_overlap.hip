// Microbenchmark: do a CU's vector ALUs and its LDS pipe work at the same time (gfx950)?  Every kernel of the hot path takes about
// (vector-ALU time alone) + (LDS time alone), not the larger of the two (profiles/r05_pmc_group_instruction_mix_*.txt): is that the
// hardware or the kernels' phases?  Per iteration NV independent v_sad_u8 and NL independent conflict-free LDS operations (ds_read_b32,
// or 64-bit ds_add without return), the LDS operations issued first and waited for last, so even one wavefront could overlap them;
// 8 wavefronts per SIMD.  Times by HIP events over a chip-filling grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define ITER 2048

template <int NV, int NL, int ATOM>
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, uint32_t seed) {
    __shared__ unsigned long long s_tab[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) s_tab[i] = i * seed;
    __syncthreads();
    uint32_t v[16], r[8];
    for (int i = 0; i < 16; ++i) v[i] = seed * (threadIdx.x + 1) + i * 77;
    for (int i = 0; i < 8; ++i) r[i] = 0;
    const uint32_t s1 = seed | 0x01010101u;
    const uint32_t addr = (threadIdx.x & 63) * (ATOM ? 8 : 4) + (threadIdx.x >> 6) * 1024;      // a wavefront's lanes: consecutive words
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (ATOM) asm volatile("ds_add_u64 %0, %1 offset:%2" :: "v"(addr), "v"(*reinterpret_cast<unsigned long long*>(&v[2 * (i & 7)])), "n"((i & 1) * 512) : "memory");
            else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i & 7]) : "v"(addr), "n"((i & 3) * 256));
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i & 15]) : "s"(s1), "v"(v[(i + 1) & 15]));
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint32_t acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= v[i];
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    if (acc == 0x12345678u) out[threadIdx.x] = acc + (uint32_t)s_tab[threadIdx.x];
}

template <int NV, int NL, int ATOM>
static int run(const char* what, uint32_t* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((k_mix<NV, NL, ATOM>), dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CHK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<NV, NL, ATOM>), dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // per SIMD: 8 wavefronts x ITER iterations
    const double cyc_per_iter_simd = best * 1e-3 * 2.4e9 / (8.0 * ITER);
    printf("%-44s %8.3f ms   %7.1f clk per iteration and wavefront (8 wavefronts per SIMD, at 2.4 GHz)\n", what, best, cyc_per_iter_simd);
    return 0;
}

int main() {
    uint32_t* d_out; CHK(hipMalloc(&d_out, 4096));
    run<16, 0, 0>("16 v_sad_u8", d_out);
    run<0, 4, 0>("4 ds_read_b32", d_out);
    run<16, 4, 0>("16 v_sad_u8 + 4 ds_read_b32", d_out);
    run<0, 8, 0>("8 ds_read_b32", d_out);
    run<16, 8, 0>("16 v_sad_u8 + 8 ds_read_b32", d_out);
    run<0, 16, 0>("16 ds_read_b32", d_out);
    run<16, 16, 0>("16 v_sad_u8 + 16 ds_read_b32", d_out);
    run<0, 2, 1>("2 ds_add_u64", d_out);
    run<16, 2, 1>("16 v_sad_u8 + 2 ds_add_u64", d_out);
    run<0, 4, 1>("4 ds_add_u64", d_out);
    run<16, 4, 1>("16 v_sad_u8 + 4 ds_add_u64", d_out);
    run<32, 4, 1>("32 v_sad_u8 + 4 ds_add_u64", d_out);
    return 0;
}
