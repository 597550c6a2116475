// Microbenchmark: aggregate rate of DEPENDENT small kernels when several streams replay 40-kernel graphs at once
// (one host thread per stream, as the engine's slots do).  If the chip retires kernels at a bounded rate whatever
// the number of streams, the number of kernels per group bounds the pipeline's throughput.
//   hipcc --offload-arch=gfx950 -O3 -pthread -o kernel_rate.out kernel_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <thread>
#include <vector>
#include <chrono>
#include <atomic>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_work(uint32_t* buf, int hold, int words_per_block) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    uint32_t* p = buf + (size_t)blockIdx.x * words_per_block;
    for (int i = threadIdx.x; i < words_per_block; i += 256) p[i] = p[i] + 1u;     // read-modify-write of the block's own region
    while ((long long)(__builtin_readcyclecounter() - t0) < hold) __builtin_amdgcn_s_sleep(4);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int NK = 40, REPS = 60;
    struct Cfg { int blocks, hold, words; const char* what; };
    const Cfg cfgs[] = {{64, 0, 0, "empty, 64 blocks"}, {1280, 0, 0, "empty, 1280 blocks"}, {1280, 12000, 0, "hold 5 us, 1280 blocks"},
                        {1280, 0, 2048, "rmw 8 KB per block (10 MB), 1280 blocks"}, {1280, 12000, 2048, "hold 5 us + rmw 10 MB, 1280 blocks"}};
    for (const Cfg& c : cfgs) {
        for (int ns : {1, 2, 4, 6, 8}) {
            std::vector<std::thread> th;
            std::atomic<int> ready{0};
            std::atomic<bool> go{false};
            std::vector<double> t_end(ns);
            double t_start = 0;
            for (int s = 0; s < ns; ++s) {
                th.emplace_back([&, s] {
                    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                    uint32_t* buf; CHK(hipMalloc((void**)&buf, (size_t)1280 * 2048 * 4 + 256)); CHK(hipMemset(buf, 0, (size_t)1280 * 2048 * 4));
                    hipGraph_t g; hipGraphExec_t ge;
                    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(k_work, dim3(c.blocks), dim3(256), 0, st, buf, c.hold, c.words);
                    CHK(hipStreamEndCapture(st, &g));
                    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    for (int r = 0; r < 3; ++r) { CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st)); }
                    ready++;
                    while (!go.load()) std::this_thread::yield();
                    for (int r = 0; r < REPS; ++r) { CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st)); }
                    t_end[s] = now_us();
                    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(buf); hipStreamDestroy(st);
                });
            }
            while (ready.load() < ns) std::this_thread::yield();
            t_start = now_us();
            go = true;
            for (auto& t : th) t.join();
            double te = 0; for (double v : t_end) te = v > te ? v : te;
            const double total_k = (double)ns * REPS * NK;
            printf("%-42s streams=%d: %7.2f us per graph per stream, aggregate %6.2f us per kernel (%5.2f kernels/us)\n",
                   c.what, ns, (te - t_start) / REPS, (te - t_start) / total_k, total_k / (te - t_start));
        }
    }
    return 0;
}
