// Microbenchmark: cost of a barrier among the blocks that work on ONE frame inside a persistent kernel
// (grid = blocks_per_frame x frames; every frame has its own counter, frames never wait for each other).
//   hipcc --offload-arch=gfx950 -O3 -o frame_barrier.out frame_barrier.hip
// Variants: 0 = release fence + agent-scope atomic add + spin on an agent-scope load + acquire fence
//           1 = the same without the fences (cost of the fences alone, not a usable barrier)
//           2 = two-level: 8 sub-counters per frame (block index mod 8 = XCD), the last arrival of a sub-counter
//               bumps the frame counter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void k_loop(uint32_t* bar, uint32_t* payload, int iters, int hold, unsigned long long* out) {
    const int z = blockIdx.y, nb = gridDim.x;
    uint32_t* ctr = bar + z * 64;           // 256 B apart
    uint32_t* sub = bar + 4096 + z * 64 * 8;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (hold) { const unsigned long long c0 = __builtin_readcyclecounter(); while ((long long)(__builtin_readcyclecounter() - c0) < hold) __builtin_amdgcn_s_sleep(2); }
        // some payload written by every block and read by the next one after the barrier (keeps the fences honest)
        if (threadIdx.x < 16) __hip_atomic_store(payload + ((size_t)z * nb + blockIdx.x) * 16 + threadIdx.x, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t target = (uint32_t)(it + 1) * (uint32_t)nb;
            if (VARIANT == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            } else if (VARIANT == 1) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                const int g = blockIdx.x & 7;
                const int ng = (nb - g + 7) / 8;            // blocks of this sub-group
                const uint32_t old = __hip_atomic_fetch_add(sub + g * 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (uint32_t)(it + 1) * (uint32_t)ng - 1u) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t t2 = (uint32_t)(it + 1) * (uint32_t)(nb < 8 ? nb : 8);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t2) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        // read the neighbour's payload: must be the value of this iteration
        if (threadIdx.x == 0) {
            const uint32_t v = __hip_atomic_load(payload + ((size_t)z * nb + (blockIdx.x + 1) % nb) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (uint32_t)it && v != (uint32_t)it + 1u) atomicAdd(bar + 8000, 1u);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[z] = wall_clock64() - t0;
}

template <int VARIANT>
static int run(int nb, int frames, int iters, int hold, uint32_t* d_bar, uint32_t* d_pay, unsigned long long* d_out) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHK(hipMemset(d_bar, 0, 4 * 16384));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_loop<VARIANT>, dim3(nb, frames), dim3(256), 0, 0, d_bar, d_pay, iters, hold, d_out);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    uint32_t bad = 0; CHK(hipMemcpy(&bad, d_bar + 8000, 4, hipMemcpyDeviceToHost));
    printf("variant %d: %4d blocks/frame x %2d frames, hold %5d cyc, %d barriers: kernel %8.1f us => %6.2f us per iteration (stale reads %u)\n",
           VARIANT, nb, frames, hold, iters, ms * 1e3, ms * 1e3 / iters, bad);
    return 0;
}

int main() {
    uint32_t *d_bar, *d_pay; unsigned long long* d_out;
    CHK(hipMalloc((void**)&d_bar, 4 * 16384)); CHK(hipMalloc((void**)&d_pay, 4 * 16 * 2048 * 16)); CHK(hipMalloc((void**)&d_out, 8 * 64));
    const int iters = 200;
    for (int frames : {1, 8, 16}) {
        run<0>(160, frames, iters, 0, d_bar, d_pay, d_out);
        run<1>(160, frames, iters, 0, d_bar, d_pay, d_out);
        run<2>(160, frames, iters, 0, d_bar, d_pay, d_out);
    }
    run<0>(40, 8, iters, 0, d_bar, d_pay, d_out);
    run<0>(80, 8, iters, 0, d_bar, d_pay, d_out);
    run<0>(256, 4, iters, 0, d_bar, d_pay, d_out);
    run<0>(160, 8, iters, 12000, d_bar, d_pay, d_out);   // 5 us of "work" per iteration
    run<2>(160, 8, iters, 12000, d_bar, d_pay, d_out);
    return 0;
}
