// Microbenchmark: cost of ds_read_b128 / b64 / u16 with 2-byte-aligned per-lane addresses (the assign kernel's
// spatial-table reads) against the 16-byte-aligned case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int lane_stride_bytes, int base_off) {
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s_tab[i] = (uint16_t)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // lane -> table entry: 0 one entry per lane, 1 V shape |lane - 20| (the assign kernel's |x - cx|), 2 one entry for all
    // lanes, 3 min(lane, 30) (lanes outside the window share an entry), 4 lane pairs share an entry
    const int pattern = base_off >> 16;
    base_off &= 0xffff;
    const int li = pattern == 1 ? abs(lane - 20) : pattern == 2 ? 0 : pattern == 3 ? min(lane, 30) : pattern == 4 ? lane >> 1 : lane;
    uint32_t addr = (uint32_t)(base_off + li * lane_stride_bytes);      // byte address inside s_tab
    uint32_t acc = 0;
    const char* basep = reinterpret_cast<const char*>(s_tab);
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
        const uint32_t a = (addr + (it & 7) * 2 * (MODE >= 10 ? 0 : 1)) & 0x3fff;
        if (MODE % 10 == 0) {          // 2 x b128 (16 rows)
            uint4 v0, v1;
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(v0), "=v"(v1) : "v"(a + (uint32_t)(size_t)0) : "memory");
            acc += v0.x ^ v0.w ^ v1.y ^ v1.z;
        } else if (MODE % 10 == 1) {   // 4 x b64
            uint2 v0, v1, v2, v3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
            acc += v0.x ^ v1.y ^ v2.x ^ v3.y;
        } else if (MODE % 10 == 2) {   // 8 x b32
            uint32_t v[8];
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                         "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a) : "memory");
            acc += v[0] ^ v[3] ^ v[5] ^ v[7];
        }
    }
    (void)basep;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static int run(const char* name, int stride, int off, uint32_t* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, stride, off);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, stride, off);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    // 32 bytes per lane per iteration
    const double groups = (double)ITER * grid * 4;      // wave-level 32-byte-per-lane reads
    printf("%-44s %8.1f us   %.2f ns per 32B/lane wave-read per CU (=%.1f clk @2.1GHz)\n", name, ms * 1e3,
           ms * 1e6 / (groups / 256.0), ms * 1e6 / (groups / 256.0) * 2.1);
    return 0;
}

int main() {
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    run<10>("2 x b128, 16B aligned (lane*32)", 32, 0, d_out);
    run<10>("2 x b128, 16B aligned (lane*16: RV table)", 16, 0, d_out);
    run<10>("2 x b128, 16B aligned (lane*176: 2-D, pitch 44 banks)", 176, 0, d_out);
    run<10>("2 x b128, 4B aligned  (lane*176 + 4)", 176, 4, d_out);
    run<10>("2 x b128, 8B aligned  (lane*176 + 8)", 176, 8, d_out);
    run<10>("2 x b128, 2B aligned  (lane*176 + 2)", 176, 2, d_out);
    run<11>("4 x b64,  8B aligned  (lane*176 + 8)", 176, 8, d_out);
    run<11>("4 x b64,  4B aligned  (lane*176 + 4)", 176, 4, d_out);
    run<12>("8 x b32,  4B aligned  (lane*176 + 4)", 176, 4, d_out);
    run<12>("8 x b32,  2B aligned  (lane*176 + 2)", 176, 2, d_out);
    run<10>("2 x b128, 16B entries, V shape |lane-20|", 16, 1 << 16, d_out);
    run<10>("2 x b128, 16B entries, all lanes one entry", 16, 2 << 16, d_out);
    run<10>("2 x b128, 16B entries, min(lane,30)", 16, 3 << 16, d_out);
    run<10>("2 x b128, 16B entries, lane pairs share", 16, 4 << 16, d_out);
    run<12>("8 x b32,  V shape, 16B entries", 16, 1 << 16, d_out);
    run<12>("8 x b32,  all lanes one entry", 16, 2 << 16, d_out);
    run<0>("2 x b128, lane*2 bytes (+0..14 phase)", 2, 0, d_out);
    run<0>("2 x b128, lane*2 bytes, base+2", 2, 2, d_out);
    run<0>("2 x b128, lane*(178*2) bytes (2-D table)", 356, 0, d_out);
    run<11>("4 x b64, 8B aligned (lane*32)", 32, 0, d_out);
    run<1>("4 x b64, lane*2 bytes", 2, 0, d_out);
    run<12>("8 x b32, 4B aligned (lane*32)", 32, 0, d_out);
    run<2>("8 x b32, lane*2 bytes", 2, 0, d_out);
    return 0;
}
