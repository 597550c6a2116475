// Microbenchmark: host cost of a 40-kernel in-order sequence launched kernel by kernel vs replayed as one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o graph_launch.out graph_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_hold(int hold, uint32_t* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < hold) __builtin_amdgcn_s_sleep(4);
    if (hold < 0) sink[0] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint32_t* d_sink; CHK(hipMalloc((void**)&d_sink, 64));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NK = 40, REP = 200;
    for (int hold : {0, 12000}) {       // empty kernels / 5 us kernels
        auto seq = [&]() { for (int i = 0; i < NK; ++i) hipLaunchKernelGGL(k_hold, dim3(1280), dim3(256), 0, st, hold, d_sink); };
        seq(); CHK(hipStreamSynchronize(st));
        double t0 = now_us(), host = 0;
        for (int r = 0; r < REP; ++r) { const double a = now_us(); seq(); host += now_us() - a; CHK(hipStreamSynchronize(st)); }
        const double direct = (now_us() - t0) / REP;
        printf("hold %5d cyc: direct   %8.1f us per %d-kernel sequence (host enqueue %.1f us)\n", hold, direct, NK, host / REP);
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        seq();
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
        t0 = now_us(); host = 0;
        for (int r = 0; r < REP; ++r) { const double a = now_us(); CHK(hipGraphLaunch(ge, st)); host += now_us() - a; CHK(hipStreamSynchronize(st)); }
        printf("hold %5d cyc: hipGraph %8.1f us per %d-kernel sequence (host enqueue %.1f us)\n", hold, (now_us() - t0) / REP, NK, host / REP);
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    }
    return 0;
}
