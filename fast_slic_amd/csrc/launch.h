// launch.h -- one launch call for the stream and for the graph of a group
// A slot replays its group (about 30 kernels at 10 iterations) from a hipGraph.  The graph is BUILT node by node from the very
// launch calls that enqueue a group directly: while a GraphRecorder is current on the calling thread, fslic::launch() adds a
// kernel node (chained behind the previous one: program order, like the stream) instead of launching.  No stream capture:
// for as long as any stream of the process captures, hipDeviceSynchronize and other calls of OTHER threads (a framework
// synchronising, allocating) fail with "operation not permitted when stream is capturing", whatever the capture mode.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <tuple>
#include <type_traits>
#include <utility>

namespace fslic {

struct GraphRecorder {
    hipGraph_t graph = nullptr;
    hipGraphNode_t last = nullptr;     // the chain's tail
    hipError_t err = hipSuccess;       // first failure; the group is then enqueued directly and the graph dropped
    int nodes = 0;
    void chain(hipGraphNode_t n) { last = n; nodes++; }
};
// The recorder of the calling thread (nullptr: launches go to the stream).  Defined in group.cpp.
GraphRecorder*& current_recorder();

template <typename Tuple, size_t... I>
static inline void add_kernel_node(GraphRecorder& r, void* func, dim3 grid, dim3 block, unsigned shmem, Tuple& a, std::index_sequence<I...>) {
    void* ptrs[sizeof...(I) + 1] = {(void*)&std::get<I>(a)..., nullptr};     // the values are copied into the node here
    hipKernelNodeParams p{};
    p.func = func; p.gridDim = grid; p.blockDim = block; p.sharedMemBytes = shmem; p.kernelParams = ptrs; p.extra = nullptr;
    hipGraphNode_t node = nullptr;
    const hipError_t e = hipGraphAddKernelNode(&node, r.graph, r.last ? &r.last : nullptr, r.last ? 1 : 0, &p);
    if (e != hipSuccess) { if (r.err == hipSuccess) r.err = e; return; }
    r.chain(node);
}

template <typename... KArgs, typename... Args>
static inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count of the kernel");
    GraphRecorder* r = current_recorder();
    if (!r) {
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, static_cast<KArgs>(args)...);
        return;
    }
    if (r->err != hipSuccess) return;
    std::tuple<std::remove_cv_t<std::remove_reference_t<KArgs>>...> a(static_cast<KArgs>(args)...);
    add_kernel_node(*r, reinterpret_cast<void*>(kernel), grid, block, shmem, a, std::index_sequence_for<KArgs...>{});
}

// Row-wise clear (hipMemset2DAsync on the stream, a memset node in the graph).
static inline hipError_t clear_rows(void* dst, size_t pitch, size_t width_bytes, size_t rows, hipStream_t st) {
    GraphRecorder* r = current_recorder();
    if (!r) return hipMemset2DAsync(dst, pitch, 0, width_bytes, rows, st);
    if (r->err != hipSuccess) return hipSuccess;
    hipMemsetParams m{};
    m.dst = dst; m.pitch = pitch; m.value = 0; m.elementSize = 1; m.width = width_bytes; m.height = rows;
    hipGraphNode_t node = nullptr;
    const hipError_t e = hipGraphAddMemsetNode(&node, r->graph, r->last ? &r->last : nullptr, r->last ? 1 : 0, &m);
    if (e != hipSuccess) { r->err = e; return hipSuccess; }
    r->chain(node);
    return hipSuccess;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the DEVICE that is current when it is set, and a process may hold an engine per
// GPU (the patched reference binding: arch "hip/gfx950:N"): set once per device and call site, result kept (`state`: 64 zero-initialised
// words of the call site; 1 set, -1 refused).  false: the kernel must not be launched with more than 64 KB of dynamic LDS on this device.
template <class F> static inline bool max_dynamic_lds_on_this_device(F* kernel, int bytes, std::atomic<int>* state) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    int s = state[dev].load(std::memory_order_acquire);
    if (s == 0) {
        s = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : -1;
        state[dev].store(s, std::memory_order_release);
    }
    return s > 0;
}

// A stream operation the recorder has no node for: direct launches only for such a group.
static inline bool recording_unsupported() {
    GraphRecorder* r = current_recorder();
    if (!r) return false;
    if (r->err == hipSuccess) r->err = hipErrorNotSupported;
    return true;
}

}  // namespace fslic
