// copypool.cpp -- the engine's helper threads for host-side copies between the caller's pageable buffers and pinned staging.
// Part of the host engine, see engine_internal.h.  (The reference has no such step: its iterate() works on the caller's memory,
// cfast_slic.pyx:150-197; a GPU engine behind the same surface has to move 3N + 2N bytes per call, and one thread's memcpy
// is slower than the device-side work.)
#include "engine_internal.h"

#include <cstring>

namespace fslic {

struct CopyPool {
    struct Job {
        char* dst; const char* src; size_t bytes, chunk;
        std::atomic<size_t> next{0};      // next chunk to take
        std::atomic<size_t> done{0};      // chunks finished
        size_t nchunks = 0;
        int holders = 0;                  // helpers currently working on this job (guarded by mu): the job outlives them
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job*> jobs;                // jobs with chunks left to hand out
    std::vector<std::thread> helpers;
    bool quit = false;

    static void work_on(Job* j) {         // take chunks of j until none are left
        for (;;) {
            const size_t c = j->next.fetch_add(1);
            if (c >= j->nchunks) break;
            const size_t off = c * j->chunk, n = std::min(j->chunk, j->bytes - off);
            std::memcpy(j->dst + off, j->src + off, n);
            j->done.fetch_add(1);
        }
    }
    void helper() {
        for (;;) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return quit || !jobs.empty(); });
                if (quit) return;
                j = jobs.front();
                if (j->next.load() >= j->nchunks) { jobs.pop_front(); continue; }      // all chunks handed out: nothing left to take
                j->holders++;
            }
            work_on(j);
            {
                std::lock_guard<std::mutex> lk(mu);
                j->holders--;
            }
            cv.notify_all();
        }
    }
};

CopyPool* copy_pool_create(int helpers) {
    CopyPool* p = new CopyPool();
    for (int i = 0; i < helpers; i++) p->helpers.emplace_back([p] { p->helper(); });
    return p;
}

void copy_pool_destroy(CopyPool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->quit = true;
    }
    p->cv.notify_all();
    for (auto& t : p->helpers) t.join();
    delete p;
}

void copy_pool_run(CopyPool* p, void* dst, const void* src, size_t bytes) {
    constexpr size_t kChunk = 256 * 1024;
    if (!p || p->helpers.empty() || bytes <= 2 * kChunk) { std::memcpy(dst, src, bytes); return; }
    CopyPool::Job job;
    job.dst = (char*)dst; job.src = (const char*)src; job.bytes = bytes; job.chunk = kChunk;
    job.nchunks = (bytes + kChunk - 1) / kChunk;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->jobs.push_back(&job);
    }
    p->cv.notify_all();
    CopyPool::work_on(&job);                        // the caller copies too
    std::unique_lock<std::mutex> lk(p->mu);
    // every chunk handed out; wait for the helpers still copying theirs, and make sure no helper still holds the job's address
    p->cv.wait(lk, [&] { return job.done.load() == job.nchunks && job.holders == 0; });
    for (auto it = p->jobs.begin(); it != p->jobs.end(); ++it)
        if (*it == &job) { p->jobs.erase(it); break; }
}

}  // namespace fslic
