// Microbenchmark: what does a kernel boundary cost on one in-order stream, and do boundaries of several streams overlap?
//   hipcc --offload-arch=gfx950 -O3 -o kernel_boundary.out kernel_boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_hold(int hold, uint32_t* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < hold) __builtin_amdgcn_s_sleep(4);
    if (hold < 0) sink[0] = 1;
}
// every thread writes `per` 16-byte words, coalesced
__global__ __launch_bounds__(256) void k_write(uint4* dst, int per, uint32_t v) {
    uint4* p = dst + (size_t)blockIdx.x * 256 * per + threadIdx.x;
    for (int i = 0; i < per; ++i) p[(size_t)i * 256] = make_uint4(v, v + 1, v + 2, v + 3);
}
// the same with store cache policies: 1 = nontemporal builtin, 2 = sc1, 3 = sc0 sc1, 4 = nt sc0 sc1
template <int MODE>
__global__ __launch_bounds__(256) void k_write_pol(uint4* dst, int per, uint32_t v) {
    uint4* p = dst + (size_t)blockIdx.x * 256 * per + threadIdx.x;
    for (int i = 0; i < per; ++i) {
        const uint4 q = make_uint4(v, v + 1, v + 2, v + 3);
        uint4* a = p + (size_t)i * 256;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 qq = {q.x, q.y, q.z, q.w};
        if (MODE == 1) __builtin_nontemporal_store(qq, reinterpret_cast<u4*>(a));
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(a), "v"(qq) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(a), "v"(qq) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(a), "v"(qq) : "memory");
    }
}
__global__ __launch_bounds__(256) void k_read(const uint4* src, int per, uint32_t* sink) {
    const uint4* p = src + (size_t)blockIdx.x * 256 * per + threadIdx.x;
    uint32_t a = 0;
    for (int i = 0; i < per; ++i) { const uint4 q = p[(size_t)i * 256]; a += q.x ^ q.w; }
    if (a == 0x12345u) sink[0] = a;
}

int main() {
    uint32_t* d_sink; CHK(hipMalloc((void**)&d_sink, 64));
    uint4* d_buf; CHK(hipMalloc((void**)&d_buf, (size_t)256 << 20));
    hipStream_t st[8];
    for (auto& s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int N = 400;
    auto timeit = [&](const char* name, int nstreams, auto&& launch) -> int {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) launch(st[i % nstreams], i);
            CHK(hipDeviceSynchronize());
            const float us = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
            best = us < best ? us : best;
        }
        printf("%-64s streams=%d: %7.2f us per kernel (wall, %d launches)\n", name, nstreams, best / N, N);
        return 0;
    };
    for (int ns : {1, 4}) {
        timeit("empty kernel, 64 blocks", ns, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_hold, dim3(64), dim3(256), 0, s, 0, d_sink); });
        timeit("empty kernel, 2040 blocks", ns, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_hold, dim3(2040), dim3(256), 0, s, 0, d_sink); });
        timeit("hold 10 us, 64 blocks", ns, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_hold, dim3(64), dim3(256), 0, s, 24000, d_sink); });
        timeit("hold 10 us, 1280 blocks (5 per CU)", ns, [&](hipStream_t s, int) { hipLaunchKernelGGL(k_hold, dim3(1280), dim3(256), 0, s, 24000, d_sink); });
        timeit("write 16 MB (own region per stream)", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("write 16 MB, nontemporal builtin", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write_pol<1>, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("write 16 MB, sc1", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write_pol<2>, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("write 16 MB, sc0 sc1", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write_pol<3>, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("write 16 MB, sc0 sc1 nt", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write_pol<4>, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("write 64 MB (4096 blocks)", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (4 << 20), 4, (uint32_t)i); });
        timeit("write 64 MB nontemporal (4096 blocks)", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write_pol<1>, dim3(4096), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (4 << 20), 4, (uint32_t)i); });
        timeit("write 2 MB (128 blocks)", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_write, dim3(128), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
        timeit("read 16 MB (own region per stream)", ns, [&](hipStream_t s, int i) { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, d_sink); });
        timeit("write 16 MB then read it (alternating kernels)", ns, [&](hipStream_t s, int i) {
            if ((i / ns) & 1) hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, d_sink);
            else hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, d_buf + (size_t)(i % ns) * (1 << 20), 4, (uint32_t)i); });
    }
    return 0;
}
