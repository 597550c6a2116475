// Microbenchmark: how fast does the hardware start the wavefronts of a grid that fits the chip in one round?
// Every wavefront spins for `hold` shader cycles (s_memtime), so no slot is reused: event-timed kernel duration minus
// the hold time is launch + dispatch ramp + completion.  (s_memtime is per XCD, not chip-wide: start stamps of
// different XCDs cannot be compared, so only durations are reported.)
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_rate.out dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_spin(unsigned long long* out, int hold, int lds_words) {
    extern __shared__ uint32_t s_dyn[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lds_words > 0 && threadIdx.x == 0) s_dyn[lds_words - 1] = 1;
    while ((long long)(__builtin_readcyclecounter() - t0) < hold) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)] = t0;
}

template <int BLOCK>
static int run(int blocks, int hold, int lds_bytes, unsigned long long* d_out) {
    const size_t nw = (size_t)blocks * (BLOCK / 64);
    std::vector<unsigned long long> h(nw);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<BLOCK>, dim3(blocks), dim3(BLOCK), lds_bytes, 0, d_out, hold, lds_bytes / 4);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHK(hipMemcpy(h.data(), d_out, nw * 8, hipMemcpyDeviceToHost));
    const double ghz = 2.39;
    printf("block %4d x %5d blocks (%5zu waves), lds %5d B, hold %5d cyc = %5.2f us: kernel %7.2f us (events) => launch + ramp + completion %5.2f us\n",
           BLOCK, blocks, nw, lds_bytes, hold, hold / ghz / 1e3, ms * 1e3, ms * 1e3 - hold / ghz / 1e3);
    return 0;
}

int main() {
    unsigned long long* d_out; CHK(hipMalloc((void**)&d_out, 8 * 131072));
    const int hold = 60000;   // 25 us: longer than any ramp
    run<256>(460, hold, 0, d_out);
    run<256>(2040, hold, 0, d_out);
    run<256>(2040, hold, 10240, d_out);
    run<256>(1024, hold, 10240, d_out);
    run<512>(1020, hold, 20480, d_out);
    run<1024>(510, hold, 40960, d_out);
    run<64>(8160, hold, 0, d_out);
    run<256>(4080, hold, 0, d_out);     // two rounds
    run<256>(2040, 2000, 0, d_out);
    return 0;
}
