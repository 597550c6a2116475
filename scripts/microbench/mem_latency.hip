// Microbenchmark: dependent-load latency seen by one wavefront (pointer chase) for working sets that live in L2, in
// the Infinity Cache and in HBM; the s_memtime tick; and the time from wavefront start to the first load's data when
// a whole grid starts at once (the assign kernels' prologue).   hipcc --offload-arch=gfx950 -O3 -o mem_latency.out mem_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_chase(const uint32_t* __restrict__ next, int steps, unsigned long long* out) {
    uint32_t p = threadIdx.x == 0 ? 0u : 0u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < steps; ++i) p = __builtin_nontemporal_load(next + p * 32u);      // one 128-byte line per node
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = p; }
}

// every wavefront: stamp, one load of its own line from `buf`, stamp
__global__ __launch_bounds__(256) void k_first(const uint32_t* __restrict__ buf, size_t stride_words, unsigned long long* out, uint32_t* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t v = buf[wv * stride_words + (threadIdx.x & 63)];
    if (v == 0xdeadbeefu) sink[0] = v;
    __builtin_amdgcn_s_waitcnt(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[wv] = t1 - t0;
}

int main() {
    unsigned long long* d_out; uint32_t* d_sink;
    CHK(hipMalloc((void**)&d_out, 8 * 65536)); CHK(hipMalloc((void**)&d_sink, 64));
    std::mt19937 rng(1);
    for (size_t bytes : {(size_t)64 << 10, (size_t)2 << 20, (size_t)32 << 20, (size_t)192 << 20, (size_t)2048 << 20}) {
        const size_t nodes = bytes / 128;
        std::vector<uint32_t> perm(nodes); std::iota(perm.begin(), perm.end(), 0u);
        std::shuffle(perm.begin() + 1, perm.end(), rng);
        std::vector<uint32_t> h(nodes * 32, 0u);
        for (size_t i = 0; i < nodes; ++i) h[(size_t)perm[i] * 32] = perm[(i + 1) % nodes];
        uint32_t* d; CHK(hipMalloc((void**)&d, bytes)); CHK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
        const int steps = 4096;
        unsigned long long r[3];
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, steps, d_out);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(r, d_out, 24, hipMemcpyDeviceToHost));
        printf("chase %8.2f MB: %7.1f s_memtime ticks / load, %7.1f ns / load (wall_clock64 @100 MHz) => s_memtime tick = %.3f ns\n",
               bytes / 1048576.0, (double)r[0] / steps, (double)r[1] * 10.0 / steps, (double)r[1] * 10.0 / (double)r[0]);
        CHK(hipFree(d));
    }
    // first-load latency at grid start: L2-hot 8 KB region vs a distinct line per wavefront (HBM / Infinity Cache)
    const size_t big = (size_t)512 << 20;
    uint32_t* d; CHK(hipMalloc((void**)&d, big)); CHK(hipMemset(d, 0, big));
    for (int blocks : {256, 2048, 8192}) {
        for (int mode = 0; mode < 2; ++mode) {
            const size_t stride = mode == 0 ? 0 : 4096;       // words: same line for all / own 16 KB-spaced line
            std::vector<unsigned long long> h((size_t)blocks * 4);
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(k_first, dim3(blocks), dim3(256), 0, 0, d, stride, d_out, d_sink);
                CHK(hipDeviceSynchronize());
            }
            CHK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            printf("first load, %5d blocks, %s: median %6llu ticks, p90 %6llu, max %6llu\n", blocks, mode == 0 ? "one shared line   " : "own line per wave ",
                   h[h.size() / 2], h[h.size() * 9 / 10], h.back());
        }
    }
    return 0;
}
