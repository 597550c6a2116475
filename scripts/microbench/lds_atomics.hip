// Microbenchmark: what an LDS instruction costs the CU's LDS pipe as a function of its ACTIVE LANES -- non-returning 64-bit and
// 32-bit atomic adds (the fused assign kernel's run-boundary flushes), 32-bit gathers (the tile connectivity kernel's finds),
// 16-byte writes of a few lanes (its staged node records).  All 32 wavefronts of a CU issue the same instruction back to back; the time
// per instruction and CU is the pipe's cost of one such instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// OP: 7 ds_write_b32, 8 ds_write_b64, 9 ds_write2st64_b32, 10 ds_read_b32 at a pseudo-random word of the wavefront's 2 KB; 0 ds_add_u64, 1 ds_add_u32, 2 ds_read_b32 (dependent address chain broken: fixed address), 3 ds_write_b128, 4 ds_read_u16, 5 ds_read_b128, 6 ds_min_u32
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int nlanes, int stride_bytes, int same) {
    __shared__ __attribute__((aligned(16))) uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = (uint32_t)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t addr = (uint32_t)((wave * 2048 + (same ? (lane / same) : lane) * stride_bytes) & 0x7FF0);
    if (OP == 10) addr = (uint32_t)(wave * 2048 + ((lane * 2654435761u >> 7) & 0x7FC) % 2048);
    if (OP == 9) addr &= 0x1FFF;
    uint32_t acc = 0;
    if (lane < nlanes) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        unsigned long long v = 1; uint32_t w = 1; u4 q = {1, 2, 3, 4};
#pragma unroll 1
        for (int it = 0; it < ITER; it += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (OP == 0) asm volatile("ds_add_u64 %0, %1" :: "v"(addr), "v"(v) : "memory");
                else if (OP == 1) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(w) : "memory");
                else if (OP == 2) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc ^= r; }
                else if (OP == 3) asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(q) : "memory");
                else if (OP == 4) { uint32_t r; asm volatile("ds_read_u16 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc ^= r; }
                else if (OP == 5) { u4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc ^= r.x; }
                else if (OP == 6) asm volatile("ds_min_u32 %0, %1" :: "v"(addr), "v"(w) : "memory");
                else if (OP == 7) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(w) : "memory");
                else if (OP == 8) asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
                else if (OP == 9) asm volatile("ds_write2st64_b32 %0, %1, %1 offset1:32" :: "v"(addr), "v"(w) : "memory");
                else if (OP == 10) { uint32_t r; asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr) : "memory"); acc ^= r; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + s[threadIdx.x];
}

template <int OP>
static int run(const char* name, int nlanes, int stride, int same, uint32_t* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, nlanes, stride, same);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d_out, nlanes, stride, same);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = (double)ITER * 32;      // instructions per CU (8 blocks x 4 wavefronts, one round)
    printf("%-34s lanes %2d stride %3d B same-address groups of %2d: %7.2f ns = %6.1f clk @2.4 GHz per instruction and CU\n", name, nlanes, stride, same, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    return 0;
}

int main() {
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    const int ns[] = {1, 2, 4, 8, 16, 32, 64};
    for (int n : ns) run<0>("ds_add_u64 (no return)", n, 16, 0, d_out);
    for (int n : ns) run<0>("ds_add_u64 (no return)", n, 8, 0, d_out);
    run<0>("ds_add_u64 (no return)", 64, 16, 2, d_out);
    run<0>("ds_add_u64 (no return)", 64, 16, 8, d_out);
    run<0>("ds_add_u64 (no return)", 8, 16, 2, d_out);
    for (int n : ns) run<1>("ds_add_u32 (no return)", n, 4, 0, d_out);
    run<1>("ds_add_u32 (no return)", 64, 4, 4, d_out);
    for (int n : ns) run<6>("ds_min_u32 (no return)", n, 4, 0, d_out);
    for (int n : ns) run<2>("ds_read_b32", n, 4, 0, d_out);
    run<2>("ds_read_b32", 64, 4, 8, d_out);
    run<2>("ds_read_b32", 64, 132, 0, d_out);
    for (int n : ns) run<4>("ds_read_u16", n, 2, 0, d_out);
    for (int n : ns) run<3>("ds_write_b128", n, 16, 0, d_out);
    for (int n : ns) run<5>("ds_read_b128", n, 16, 0, d_out);
    for (int n : {1, 8, 64}) run<7>("ds_write_b32", n, 4, 0, d_out);
    for (int n : {1, 8, 64}) run<8>("ds_write_b64", n, 8, 0, d_out);
    for (int n : {1, 8, 64}) run<9>("ds_write2st64_b32", n, 4, 0, d_out);
    for (int n : {8, 64}) run<10>("ds_read_b32 (scattered words)", n, 0, 0, d_out);
    run<2>("ds_read_b32", 64, 8, 0, d_out);
    run<2>("ds_read_b32", 64, 16, 0, d_out);
    run<2>("ds_read_b32", 64, 36, 0, d_out);
    return 0;
}
