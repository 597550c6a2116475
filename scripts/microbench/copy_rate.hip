// Microbenchmark: what a plain streaming copy reaches on this GPU, by kernel form (the "measured HBM rate" bench.py prints comes
// from the best of these, csrc/graph.hip k_copy16).  1 GiB source and destination (far beyond the 256 MB infinity cache).
//   hipcc --offload-arch=gfx950 -O3 -o copy_rate.out copy_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// MODE 0: grid-stride, U loads in flight per lane, plain   1: nontemporal loads and stores   2: sc1 stores   3: nt loads, plain stores
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_stride(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (MODE == 1 || MODE == 3) ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 1) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + i + u * stride), "v"(v[u]) : "memory");
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
// every block owns one contiguous chunk of U x 4 KB (no grid-stride loop: one trip per block)
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_chunk(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
    const size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * 256 < n16) v[u] = MODE == 1 ? __builtin_nontemporal_load(src + i0 + u * 256) : src[i0 + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * 256 < n16) { if (MODE == 1) __builtin_nontemporal_store(v[u], dst + i0 + u * 256); else dst[i0 + u * 256] = v[u]; }
}

template <class F> static double best_of(F launch, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
    double best = 0;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms > 0) { const double g = 2.0 * bytes / (ms * 1e-3) / 1e9; if (g > best) best = g; }
    }
    return best;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    u4 *a, *b;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("streaming copy of 1 GiB, GB/s = (bytes read + bytes written) / time, best of 5\n");
    printf("hipMemcpyAsync device to device                 %8.1f\n", best_of([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes, e0, e1));
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        printf("grid-stride, %6d blocks: 1 in flight %8.1f", grid, best_of([&] { k_stride<1, 0><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
        printf("   4 in flight %8.1f", best_of([&] { k_stride<4, 0><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
        printf("   8 in flight %8.1f", best_of([&] { k_stride<8, 0><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
        printf("   4, nontemporal %8.1f", best_of([&] { k_stride<4, 1><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
        printf("   4, sc1 stores %8.1f", best_of([&] { k_stride<4, 2><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
        printf("   4, nt loads %8.1f\n", best_of([&] { k_stride<4, 3><<<grid, 256>>>(a, b, n16); }, bytes, e0, e1));
    }
    {
        const int g1 = (int)((n16 + 255) / 256), g4 = (int)((n16 + 1023) / 1024), g8 = (int)((n16 + 2047) / 2048);
        printf("one chunk per block: 4 KB %8.1f   16 KB %8.1f   32 KB %8.1f   16 KB nontemporal %8.1f\n",
               best_of([&] { k_chunk<1, 0><<<g1, 256>>>(a, b, n16); }, bytes, e0, e1), best_of([&] { k_chunk<4, 0><<<g4, 256>>>(a, b, n16); }, bytes, e0, e1),
               best_of([&] { k_chunk<8, 0><<<g8, 256>>>(a, b, n16); }, bytes, e0, e1), best_of([&] { k_chunk<4, 1><<<g4, 256>>>(a, b, n16); }, bytes, e0, e1));
    }
    // read-only and write-only rates (what a kernel with another read : write mix can reach)
    return 0;
}
