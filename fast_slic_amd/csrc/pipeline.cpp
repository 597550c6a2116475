// pipeline.cpp -- who may use a slot, and the asynchronous forms of a group.  Part of the host engine, see
// engine_internal.h.
//
// The reference runs iterate() `with nogil` on a per-call Context (cfast_slic.pyx:188-193; thread count and timer are
// thread_local, src/parallel.cpp:14, src/timer.cpp:45), so concurrent calls on different models are legal.  Here every
// call needs one of the engine's slots (stream + arena): synchronous entry points take a free slot for their duration
// and wait while all are taken, so threads sharing one engine run concurrently up to n_slots and are serialised beyond.
// Asynchronous groups (fslic_hip_submit_group / fslic_hip_pipeline_submit) run on the slot's own host thread.
#include "engine_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <sys/prctl.h>

namespace fslic {

namespace {
thread_local std::string g_err;
}
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const std::string& last_error() { return g_err; }
void set_last_error(const std::string& msg) { g_err = msg; }

namespace {
Knobs read_knobs() {
    Knobs k;
    const char* g = getenv("FSLIC_GROUP");
    k.group_size = g ? std::min(std::max(atoi(g), 1), (int)kMaxGroup) : 8;
    const char* gr = getenv("FSLIC_GRAPH");
    k.use_graphs = !(gr && atoi(gr) == 0);
    k.poison = getenv("FSLIC_POISON") != nullptr;
    k.host_timing = getenv("FSLIC_HOST_TIMING") != nullptr;
    const char* fb = getenv("FSLIC_FUSEBIN");
    k.fuse_bin = fb ? std::min(std::max(atoi(fb), 0), 2) : 1;
    return k;
}
const Knobs g_knobs = read_knobs();      // once, when the library is loaded
}  // namespace
const Knobs& knobs() { return g_knobs; }

// ---- slot ownership ------------------------------------------------------------------------------------------------
namespace {
// caller holds e->mu: the slot's pending group came from the submit queue, i.e. its own thread collects it
bool self_collecting(const Slot& s) { return s.pending && s.async && s.async->from_queue; }
}  // namespace

namespace {
// wakes the slot threads when a caller that had been counted in sync_waiters leaves (declared BEFORE the lock: runs after it is released)
struct WakeWorkers {
    fslic_engine* e; bool armed = false;
    explicit WakeWorkers(fslic_engine* e_) : e(e_) {}
    ~WakeWorkers() { if (armed) e->cv_work.notify_all(); }
};
}  // namespace

int acquire_slot(fslic_engine* e, int want, int& slot) {
    WakeWorkers wake(e);
    std::unique_lock<std::mutex> lk(e->mu);
    const int ns = (int)e->slots.size();
    if (want >= ns) return fail(FSLIC_E_INVALID, "slot out of range");
    // A caller that has to wait is counted (sync_waiters, or Slot::wanted when it asked for one slot): while the count is non-zero the
    // slot threads concerned leave the submit queue alone, so the slot that frees next goes to the waiting caller instead of straight
    // back to the queue (under a continuous pipeline_submit load a synchronous call could otherwise wait indefinitely).
    auto got = [&](int i) { e->slots[i].busy = true; slot = i; return FSLIC_OK; };
    for (;;) {
        // A slot serving the submit / drain queue collects its group itself: it is busy for a while, not lost.  Only a
        // group submitted by slot number (collected by the caller's fslic_hip_wait_group) is a reason to refuse.
        if (want >= 0) {
            Slot& s = e->slots[want];
            if (s.pending && !self_collecting(s)) return fail(FSLIC_E_INVALID, "the slot still owns an unfinished group (call fslic_hip_wait_group first)");
            if (!s.pending && !s.busy) return got(want);
        } else {
            bool any_usable = false;
            for (int i = 0; i < ns; i++) {
                Slot& s = e->slots[i];
                if (s.pending) { any_usable = any_usable || self_collecting(s); continue; }
                any_usable = true;
                if (!s.busy) return got(i);
            }
            if (!any_usable) return fail(FSLIC_E_INVALID, "every slot owns an unfinished asynchronous group");
        }
        // counted where it matters: a caller that wants ONE slot holds back that slot's thread only (the others keep serving the
        // submit queue); a caller that takes any slot holds back whichever thread becomes idle next
        int& waiting = want >= 0 ? e->slots[want].wanted : e->sync_waiters;
        waiting++;
        e->cv.wait(lk);
        waiting--;
        wake.armed = true;
    }
}

int acquire_all_slots(fslic_engine* e) {
    WakeWorkers wake(e);
    std::unique_lock<std::mutex> lk(e->mu);
    for (;;) {
        bool free_all = true;
        for (const Slot& s : e->slots) {
            if (s.pending && !self_collecting(s)) return fail(FSLIC_E_INVALID, "a slot still owns an unfinished group");
            free_all = free_all && !s.busy && !s.pending;
        }
        if (free_all) break;
        e->sync_waiters++;
        e->cv.wait(lk);
        e->sync_waiters--;
        wake.armed = true;
    }
    for (Slot& s : e->slots) s.busy = true;
    return FSLIC_OK;
}

void release_slot(fslic_engine* e, int slot) {
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->slots[slot].busy = false;
    }
    e->cv.notify_all(); e->cv_work.notify_all();      // callers waiting for a slot; the slot's own thread may serve the queue again
}

void release_all_slots(fslic_engine* e) {
    {
        std::lock_guard<std::mutex> lk(e->mu);
        for (Slot& s : e->slots) s.busy = false;
    }
    e->cv.notify_all(); e->cv_work.notify_all();
}

// ---- the slot's host thread: one group at a time, begin + finish ---------------------------------------------------
namespace {

// caller holds e->mu: fold a completed pipeline group into the engine's totals and free its slot
void pipeline_collect(fslic_engine* e, Slot& s) {
    Slot::Async& a = *s.async;
    if (a.rc != FSLIC_OK && e->pipe_rc == FSLIC_OK) { e->pipe_rc = a.rc; e->pipe_err = a.err; }
    e->pipe_device_ms += s.total_ms;
    e->pipe_groups += 1;
    e->pipe_frames += s.nframes;
    e->pipe_host_topk += s.n_host_topk;
    s.pending = false;
}

bool same_work(const fslic_engine::PipeJob& j, const Slot::Async& a) {
    return j.H == a.H && j.W == a.W && j.K == a.K && memcmp(&j.p, &a.p, sizeof(fslic_params)) == 0;
}

void slot_worker(fslic_engine* e, Slot* s) {
    Slot::Async& a = *s->async;
    (void)hipSetDevice(e->device);
    (void)prctl(PR_SET_TIMERSLACK, 1000UL);      // this thread's 10 us naps (group.cpp, nap_wait) are not rounded up to the default 50 us slack
    for (;;) {
        bool took = false;
        {
            std::unique_lock<std::mutex> lk(e->mu);
            e->cv_work.wait(lk, [&] { return a.has_job || a.quit || (!e->pipe_q.empty() && !e->pipe_gathering && !s->busy && !s->pending && e->sync_waiters == 0 && s->wanted == 0); });
            if (a.quit) return;
            if (!a.has_job) {
                // the submit queue's head, plus -- with batching on -- the submissions behind it that ask for the same work,
                // as long as the group stays within the frame limit
                const fslic_engine::PipeJob& j0 = e->pipe_q.front();
                a.p = j0.p; a.H = j0.H; a.W = j0.W; a.K = j0.K; a.n = 0; a.jobs = 0;
                const int limit = std::min(e->pipe_batch_frames, (int)kMaxGroup);
                const int unit = j0.n;
                auto gather = [&](int max_jobs) {
                    while (!e->pipe_q.empty() && a.jobs < max_jobs) {
                        const fslic_engine::PipeJob& j = e->pipe_q.front();
                        if (a.jobs > 0 && (!same_work(j, a) || a.n + j.n > limit)) break;
                        for (int i = 0; i < j.n; i++) { a.d_rgb[a.n + i] = j.d_rgb[i]; a.clusters[a.n + i] = j.clusters[i]; a.d_out[a.n + i] = j.d_out[i]; }
                        a.n += j.n; a.jobs++;
                        e->pipe_q.pop_front();
                    }
                };
                // (Guided self-scheduling -- a slot takes ceil(waiting / slots) submissions, so that the last submissions of a burst spread
                // over the slots -- was measured in round 5 and not kept: the driver's 20 steps 45.6 against 46.7 GP/s, mean of four runs each,
                // profiles/r05_burst_schedule.txt: 8-frame groups cost more per frame than the balance gains, and which slots share a
                // hardware queue -- the runtime's choice: two of four queues carry two streams each with six slots -- decides the tail.)
                gather(kMaxGroup);
                // the slot is taken and the group counted BEFORE the wait below: a drain must not take the queue for served
                // meanwhile, nor a synchronous call this slot for free
                s->pending = true;
                a.done = false; a.from_queue = true;
                e->pipe_inflight++;
                // Room for another submission of this size and nothing waiting: a caller in the middle of a burst delivers the
                // next one within microseconds.  This thread waits for it briefly (the other idle threads leave the queue alone
                // meanwhile), so that the first groups of a burst are as full as the later ones.
                // (only while the caller IS in the middle of a burst -- its last submission is a few microseconds old: the slot that takes
                // the last submission of a burst would otherwise wait 100 us for a companion that never comes, 3 % of the driver's region)
                if (a.n + unit <= limit && e->pipe_q.empty() && !e->pipe_gathering && now_us() - e->pipe_last_submit_us < 50.0) {
                    e->pipe_gathering = true;
                    e->cv_work.wait_for(lk, std::chrono::microseconds(100), [&] { return !e->pipe_q.empty() || a.quit; });
                    e->pipe_gathering = false;
                    gather(kMaxGroup);
                }
                s->launch_timing = e->launch_timing;
                a.has_job = true;
                took = true;
            }
        }
        if (took) { e->cv.notify_all(); e->cv_work.notify_all(); }      // room in the queue (callers); the gathering wait is over (the other slot threads)
        int rc = group_begin(e, *s, &a.p, a.H, a.W, a.K, a.n, a.d_rgb, a.clusters, a.d_out);
        s->nap_wait = true;
        if (rc == FSLIC_OK) rc = group_finish(e, *s);
        s->nap_wait = false;
        if (rc != FSLIC_OK) {        // nothing may still run against the caller's buffers once the group is reported done
            const std::string msg = last_error();
            (void)hipStreamSynchronize(s->st);
            (void)hipGetLastError();
            set_last_error(msg);
        }
        {
            std::lock_guard<std::mutex> lk(e->mu);
            a.rc = rc;
            a.err = rc ? last_error() : std::string();
            a.has_job = false;
            a.done = true;
            if (a.from_queue) {                  // nobody waits for this slot by number: collected here
                pipeline_collect(e, *s);
                a.from_queue = false;
                e->pipe_inflight--;
            }
        }
        e->cv.notify_all();
    }
}

// caller holds e->mu
void ensure_worker(fslic_engine* e, Slot& s) {
    if (s.async) return;
    s.async.reset(new Slot::Async());
    s.async->worker = std::thread(slot_worker, e, &s);
}

// caller holds e->mu; the slot is neither busy nor pending
void hand_over(fslic_engine* e, Slot& s, const fslic_params* p, int H, int W, int K, int n,
               const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels) {
    ensure_worker(e, s);
    Slot::Async& a = *s.async;
    a.p = *p; a.H = H; a.W = W; a.K = K; a.n = n; a.from_queue = false; a.jobs = 1;
    s.launch_timing = e->launch_timing;      // sampled on the caller's thread: the worker may start later
    for (int i = 0; i < n; i++) { a.d_rgb[i] = d_rgb[i]; a.clusters[i] = clusters[i]; a.d_out[i] = d_labels[i]; }
    a.done = false;
    a.has_job = true;
    s.pending = true;
}

int check_group_args(fslic_engine* e, const fslic_params* p, int H, int W, int K, int n_frames,
                     const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (n_frames < 1 || n_frames > (int)kMaxGroup || !d_rgb || !clusters || !d_labels) return fail(FSLIC_E_INVALID, "bad group arguments");
    int S = 0;
    const int rc = validate(p, H, W, K, S);          // argument errors surface on the caller's thread
    if (rc) return rc;
    for (int i = 0; i < n_frames; i++)
        if (!clusters[i] || !d_rgb[i] || !d_labels[i]) return fail(FSLIC_E_INVALID, "NULL pointer argument");
    return FSLIC_OK;
}

}  // namespace

void stop_slot_thread(fslic_engine* e, Slot& s) {
    if (!s.async) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        s.async->quit = true;
    }
    e->cv_work.notify_all();
    if (s.async->worker.joinable()) s.async->worker.join();
    s.async.reset();
}

}  // namespace fslic

using namespace fslic;

extern "C" {

int fslic_hip_submit_group(fslic_engine* e, int slot, const fslic_params* p, int H, int W, int K, int n_frames,
                           const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels) {
    int rc = check_group_args(e, p, H, W, K, n_frames, d_rgb, clusters, d_labels);
    if (rc) return rc;
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    {
        std::unique_lock<std::mutex> lk(e->mu);
        Slot& s = e->slots[slot];
        // checked again after every wake-up: while this thread waited for a synchronous call to leave the slot, the slot's
        // own thread may have taken a submission from the queue, or another thread's submit_group may have got in first
        for (;;) {
            if (s.pending && !self_collecting(s)) return fail(FSLIC_E_INVALID, "the slot still owns an unfinished group (call fslic_hip_wait_group first)");
            if (!s.busy && !s.pending) break;
            e->cv.wait(lk);
        }
        hand_over(e, s, p, H, W, K, n_frames, d_rgb, clusters, d_labels);
    }
    e->cv_work.notify_all();          // the slot's thread
    return FSLIC_OK;
}

int fslic_hip_wait_group(fslic_engine* e, int slot) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (slot < 0 || slot >= (int)e->slots.size()) return fail(FSLIC_E_INVALID, "slot out of range");
    Slot& s = e->slots[slot];
    int rc;
    {
        std::unique_lock<std::mutex> lk(e->mu);
        if (!s.pending) return FSLIC_OK;
        Slot::Async& a = *s.async;
        e->cv.wait(lk, [&] { return a.done; });
        rc = a.rc;
        if (rc) set_last_error(a.err);
        s.pending = false;
        if (rc == FSLIC_OK) set_thread_timing_report(make_timing_report(s));
    }
    e->cv.notify_all(); e->cv_work.notify_all();      // the slot is free again: callers, and its own thread (the queue)
    return rc;
}

int fslic_hip_group_done(fslic_engine* e, int slot) {
    if (!e || slot < 0 || slot >= (int)e->slots.size()) return -1;
    std::lock_guard<std::mutex> lk(e->mu);
    Slot& s = e->slots[slot];
    if (!s.pending) return 1;
    return s.async->done ? 1 : 0;
}

// The submit / drain pipeline: submissions go into the engine's queue and the slot threads serve it; submit blocks only
// while the queue is full (two submissions per slot).
int fslic_hip_pipeline_submit(fslic_engine* e, const fslic_params* p, int H, int W, int K, int n_frames,
                              const uint8_t* const* d_rgb, fslic_cluster* const* clusters, uint16_t* const* d_labels) {
    int rc = check_group_args(e, p, H, W, K, n_frames, d_rgb, clusters, d_labels);
    if (rc) return rc;
    {
        std::unique_lock<std::mutex> lk(e->mu);
        for (Slot& s : e->slots) ensure_worker(e, s);
        const size_t cap = 2 * e->slots.size();
        e->cv.wait(lk, [&] { return e->pipe_rc != FSLIC_OK || e->pipe_q.size() < cap; });
        if (e->pipe_rc != FSLIC_OK) { set_last_error(e->pipe_err); return e->pipe_rc; }   // reported once more by drain
        e->pipe_q.emplace_back();
        fslic_engine::PipeJob& j = e->pipe_q.back();
        e->pipe_last_submit_us = now_us();
        j.p = *p; j.H = H; j.W = W; j.K = K; j.n = n_frames;
        for (int i = 0; i < n_frames; i++) { j.d_rgb[i] = d_rgb[i]; j.clusters[i] = clusters[i]; j.d_out[i] = d_labels[i]; }
    }
    e->cv_work.notify_all();          // the slot threads (and the one waiting for a companion)
    return FSLIC_OK;
}

int fslic_hip_pipeline_batching(fslic_engine* e, int max_frames_per_group) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    if (max_frames_per_group < 0 || max_frames_per_group > (int)kMaxGroup) return fail(FSLIC_E_INVALID, "frames per group out of range");
    std::lock_guard<std::mutex> lk(e->mu);
    e->pipe_batch_frames = max_frames_per_group;
    e->reserve_frames.store(max_frames_per_group);
    return FSLIC_OK;
}

int fslic_hip_pipeline_drain(fslic_engine* e, double* device_ms, long long* groups, long long* frames, long long* host_topk_frames) {
    if (!e) return fail(FSLIC_E_INVALID, "engine is NULL");
    int rc;
    {
        std::unique_lock<std::mutex> lk(e->mu);
        // everything submitted has been taken by a slot thread and collected (a slot that a caller drives by number, or a
        // synchronous call, may keep a queued submission waiting: the queue is served as soon as a slot is free)
        e->cv.wait(lk, [&] { return e->pipe_q.empty() && e->pipe_inflight == 0; });
        rc = e->pipe_rc;
        if (rc) set_last_error(e->pipe_err);
        if (device_ms) *device_ms = e->pipe_device_ms;
        if (groups) *groups = e->pipe_groups;
        if (frames) *frames = e->pipe_frames;
        if (host_topk_frames) *host_topk_frames = e->pipe_host_topk;
        e->pipe_rc = FSLIC_OK; e->pipe_err.clear();
        e->pipe_device_ms = 0; e->pipe_groups = 0; e->pipe_frames = 0; e->pipe_host_topk = 0;
    }
    e->cv.notify_all();
    return rc;
}

}  // extern "C"
