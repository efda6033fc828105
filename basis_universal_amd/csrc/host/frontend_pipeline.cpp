// frontend_pipeline.cpp -- bu_frontend_pipeline_* (include/basisu_hip_frontend.h): N ETC1S frontends in flight on one GPU, driven by ONE host thread.
//
// The reference reaches "several images in flight" only through basis_parallel_compress (encoder/basisu_comp.cpp:5466-5559): one host thread per image, each
// blocked in its accelerator calls. Half of a frontend step is a chain of small dependent launches (the codebook builders' rounds) that leaves most of the chip
// idle, the other half is chip-filling per-block kernels: two or more images interleave well on the device, but a host thread per image spends its time waiting.
// Here every image in flight is a TASK with a stack of its own (ucontext) on the pipeline's one driver thread. The frontend code is unchanged -- it still reads
// as a sequence of blocking calls -- but its context carries a wait hook (bu_hip_set_wait_hook): wherever a call would block on the device it yields to the
// driver, which resumes the next task. One thread launches the kernels of all lanes; nobody blocks in the runtime.
#include <ucontext.h>
#include <sys/mman.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/basisu_hip_frontend.h"

extern thread_local std::string bu_last_exception_text;   // frontend_capi.cpp

struct bu_frontend_pipeline;

namespace {

using clock_type = std::chrono::steady_clock;
inline double now_s() { return std::chrono::duration<double>(clock_type::now().time_since_epoch()).count(); }

struct job {
    uint64_t ticket = 0;
    bu_frontend_job d{};
    bu_frontend* fe = nullptr;
    bu_hip_context* ctx = nullptr;
    bool ok = false, done = false, handed_out = false;
    std::string error;
    double t_submit = 0, t_start = 0, t_done = 0;
};

struct driver_state;
struct lane {
    bu_frontend_pipeline* owner = nullptr;
    driver_state* drv = nullptr;
    ucontext_t uc;
    void* stack = nullptr;
    size_t stack_bytes = 0;
    job* j = nullptr;
    bool finished = false;
    uint64_t yields = 0;
};

const size_t kStackBytes = (size_t)8 << 20;   // the frontend keeps its arrays on the heap; this is address space, touched pages only

// One driver thread and the lanes it serves (a pipeline has one unless asked for more: bu_frontend_pipeline_create_n).
struct driver_state {
    bu_frontend_pipeline* p = nullptr;
    ucontext_t sched;
    std::thread th;
    std::thread::id tid;            // set by the thread itself before it runs its first task
    uint32_t first_lane = 0, n_lanes = 0, active = 0;
    // counters of the thread itself (nobody else touches them) ...
    uint64_t n_switches = 0, n_idle_sleeps = 0;
    double busy_s = 0, idle_s = 0;
    // ... and the copy bu_frontend_pipeline_stats reads, refreshed under the pipeline's mutex whenever one of this thread's jobs finishes
    struct snapshot { uint64_t switches = 0, naps = 0, yields = 0; double busy_s = 0, idle_s = 0, cpu_s = 0; } pub;
};

} // namespace

struct bu_frontend_pipeline {
    int device = 0;
    std::vector<lane> lanes;
    std::deque<driver_state> drivers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<job*> pending;
    std::map<uint64_t, job*> jobs;        // everything submitted and not yet released
    std::map<bu_frontend*, job*> by_frontend;
    uint64_t next_ticket = 1;
    bool stop = false;
    double idle_spin_s = 50e-6, idle_sleep_s = 10e-6;
    std::string error;
    // counters (bu_frontend_pipeline_stats)
    uint64_t n_done = 0;
};

namespace {

void lane_yield(void* user) {
    lane* l = static_cast<lane*>(user);
    // A task's stack can only be left from the thread that runs it. Should a helper thread of the frontend ever wait on the context (none does today: the host loops
    // that fan out over BU_HOST_THREADS touch no device call), it simply gives up its time slice and looks again -- an ordinary polite wait.
    if (std::this_thread::get_id() != l->drv->tid) { std::this_thread::yield(); return; }
    l->yields++;
    swapcontext(&l->uc, &l->drv->sched);
}

// Self-test task (bu_frontend_pipeline_selftest, no GPU): keeps a pattern on ITS stack across `n_blocks` yields, throws and catches an exception on the way when
// asked to, and reports whether the stack came back intact every time -- what the scheduler must guarantee to the real tasks.
void run_debug_job(lane& l) {
    job* j = l.j;
    volatile uint32_t pattern[256];
    const uint32_t seed = (uint32_t)j->ticket * 2654435761u;
    for (uint32_t i = 0; i < 256; i++) pattern[i] = seed + i;
    bool ok = true;
    for (uint32_t y = 0; y < j->d.n_blocks; y++) {
        lane_yield(&l);
        for (uint32_t i = 0; i < 256; i++) ok = ok && pattern[i] == seed + i + y;
        for (uint32_t i = 0; i < 256; i++) pattern[i] = pattern[i] + 1;
        if (j->d.max_threads && y == j->d.n_blocks / 2) {
            try { throw std::runtime_error("inside a task"); } catch (const std::runtime_error&) { ok = ok && pattern[0] == seed + y + 1; }
        }
    }
    if (j->d.compression_level == 99) throw std::runtime_error("a task that fails");
    j->ok = ok;
    if (!ok) j->error = "stack pattern damaged";
}

void run_job(lane& l) {
    bu_frontend_pipeline* p = l.owner;
    job* j = l.j;
    j->t_start = now_s();
    if (j->d.flags & 0x80000000u) { run_debug_job(l); j->t_done = now_s(); return; }
    j->ctx = bu_hip_create_context_on(p->device);   // a parked context comes back warm (its pools, arenas, pinned rings, streams)
    if (!j->ctx) { const char* e = bu_hip_last_error(nullptr); j->error = e ? e : "bu_hip_create_context_on failed"; return; }
    bu_hip_set_wait_hook(j->ctx, lane_yield, &l);
    // However this task ends -- normally, or through an exception that lane_entry turns into a failed job -- the hook comes off the context: the lane goes on to another
    // job, and whoever waits for the ticket (or frees the failed job) uses the context from a thread of its own with ordinary blocking calls.
    struct unhook { bu_hip_context* c; ~unhook() { bu_hip_set_wait_hook(c, nullptr, nullptr); } } unhook_guard{j->ctx};
    j->fe = bu_frontend_create();
    bool ok = j->fe != nullptr;
    if (ok && (j->d.flags & BU_FRONTEND_JOB_VIDEO)) ok = bu_frontend_set_video(j->fe, 1) != 0;
    if (ok && j->d.max_threads) ok = bu_frontend_set_max_threads(j->fe, j->d.max_threads) != 0;
    if (ok) ok = bu_frontend_init(j->fe, j->ctx, j->d.h_blocks, j->d.d_blocks, j->d.n_blocks, j->d.max_endpoint_clusters, j->d.max_selector_clusters,
                                  j->d.compression_level, j->d.perceptual) != 0;
    if (ok) ok = bu_frontend_compress(j->fe) != 0;
    if (!ok) {
        j->error = j->fe ? bu_frontend_error(j->fe) : "bu_frontend_create failed";
        if (j->error.empty()) j->error = bu_last_exception_text;
    }
    j->ok = ok;   // (from here on the frontend and its context belong to whoever waits for the ticket)
    j->t_done = now_s();
}

void lane_entry(unsigned hi, unsigned lo) {
    lane* l = reinterpret_cast<lane*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    try { run_job(*l); } catch (const std::exception& e) { l->j->error = e.what(); l->j->ok = false; } catch (...) { l->j->error = "unknown exception"; l->j->ok = false; }
    l->finished = true;
    // returning resumes uc_link = the driver's scheduler context
}

void start_lane(lane& l, job* j) {
    l.j = j; l.finished = false;
    getcontext(&l.uc);
    l.uc.uc_stack.ss_sp = l.stack;
    l.uc.uc_stack.ss_size = l.stack_bytes;
    l.uc.uc_link = &l.drv->sched;
    const uintptr_t a = reinterpret_cast<uintptr_t>(&l);
    makecontext(&l.uc, reinterpret_cast<void (*)()>(lane_entry), 2, (unsigned)(a >> 32), (unsigned)(a & 0xFFFFFFFFu));
}

void drive(driver_state* d) {
    bu_frontend_pipeline* p = d->p;
    d->tid = std::this_thread::get_id();
    double idle_since = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> g(p->m);
            for (;;) {
                for (uint32_t k = 0; k < d->n_lanes; k++) {
                    lane& l = p->lanes[d->first_lane + k];
                    if (!l.j && !p->pending.empty()) { job* j = p->pending.front(); p->pending.pop_front(); start_lane(l, j); d->active++; }
                }
                if (d->active) break;
                if (p->stop && p->pending.empty()) return;
                p->cv_work.wait(g);
                idle_since = 0;
            }
        }
        // one round: every lane with a task gets the thread until its next wait
        const double t0 = now_s();
        bool finished_any = false;
        uint32_t ran = 0;
        for (uint32_t k = 0; k < d->n_lanes; k++) {
            lane& l = p->lanes[d->first_lane + k];
            if (!l.j) continue;
            ran++;
            d->n_switches++;
            swapcontext(&d->sched, &l.uc);
            if (l.finished) {
                job* j = l.j;
                l.j = nullptr;
                {
                    timespec ts;
                    const double cpu = clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec : 0.0;
                    uint64_t yields = 0;
                    for (uint32_t q = 0; q < d->n_lanes; q++) yields += p->lanes[d->first_lane + q].yields;
                    std::lock_guard<std::mutex> g(p->m);
                    j->done = true; d->active--; p->n_done++;
                    d->pub = driver_state::snapshot{d->n_switches, d->n_idle_sleeps, yields, d->busy_s, d->idle_s, cpu};
                }
                p->cv_done.notify_all();
                finished_any = true;
            }
        }
        const double t1 = now_s();
        // A round in which every task only looked at its stream and yielded again did no work: after idle_spin_s of such rounds the thread naps between looks
        // (the device is busy with launches already queued; the naps cost their length in latency at most once per wait).
        const bool worked = finished_any || (t1 - t0) > 4e-6 * ran;
        if (worked) { idle_since = 0; d->busy_s += t1 - t0; }
        else {
            d->idle_s += t1 - t0;
            if (idle_since == 0) idle_since = t0;
            else if (p->idle_sleep_s > 0 && t1 - idle_since > p->idle_spin_s) {
                d->n_idle_sleeps++;
                std::this_thread::sleep_for(std::chrono::duration<double>(p->idle_sleep_s));
                d->idle_s += now_s() - t1;
            }
        }
    }
}

void free_job(job* j) {
    if (j->fe) bu_frontend_destroy(j->fe);      // first: its device buffers go back to the context's pool
    if (j->ctx) bu_hip_destroy_context(j->ctx); // parked for the next job
    delete j;
}

} // namespace

extern "C" {

#define BU_PIPE_CATCH(fail_value) catch (const std::exception& e_) { bu_last_exception_text = e_.what(); return fail_value; } catch (...) { bu_last_exception_text = "unknown exception"; return fail_value; }

void bu_frontend_pipeline_destroy(bu_frontend_pipeline* p);

static bu_frontend_pipeline* pipeline_new(int device, uint32_t lanes, bool need_device, uint32_t n_drivers = 1) {
    if (lanes < 1 || lanes > 16) { bu_last_exception_text = "bu_frontend_pipeline_create: 1..16 lanes"; return nullptr; }
    if (n_drivers < 1 || n_drivers > lanes) { bu_last_exception_text = "bu_frontend_pipeline_create: 1..lanes driver threads"; return nullptr; }
    if (need_device && !bu_hip_is_available() && !bu_hip_init(0)) { bu_last_exception_text = std::string("bu_frontend_pipeline_create: ") + bu_hip_last_error(nullptr); return nullptr; }
    bu_frontend_pipeline* p = new bu_frontend_pipeline();
    p->device = device;
    if (const char* e = std::getenv("BU_PIPELINE_SPIN_US")) p->idle_spin_s = std::max(0.0, std::atof(e)) * 1e-6;
    if (const char* e = std::getenv("BU_PIPELINE_SLEEP_US")) p->idle_sleep_s = std::max(0.0, std::atof(e)) * 1e-6;
    p->lanes.resize(lanes);
    for (auto& l : p->lanes) {
        l.owner = p;
        l.stack_bytes = kStackBytes;
        void* s = mmap(nullptr, l.stack_bytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
        if (s == MAP_FAILED) {
            for (auto& k : p->lanes) if (k.stack) munmap(static_cast<char*>(k.stack) - 4096, k.stack_bytes + 4096);
            delete p;
            bu_last_exception_text = "bu_frontend_pipeline_create: mmap of a task stack failed";
            return nullptr;
        }
        mprotect(s, 4096, PROT_NONE);   // guard page below the stack
        l.stack = static_cast<char*>(s) + 4096;
    }
    for (uint32_t k = 0, at = 0; k < n_drivers; k++) {   // the lanes dealt out evenly
        p->drivers.emplace_back();
        driver_state& d = p->drivers.back();
        d.p = p; d.first_lane = at; d.n_lanes = lanes / n_drivers + (k < lanes % n_drivers ? 1u : 0u);
        for (uint32_t i = 0; i < d.n_lanes; i++) p->lanes[at + i].drv = &d;
        at += d.n_lanes;
    }
    try {
        for (auto& d : p->drivers) d.th = std::thread(drive, &d);
    } catch (...) {   // a thread could not be started: stop the ones that were, give everything back
        bu_frontend_pipeline_destroy(p);
        throw;
    }
    return p;
}

bu_frontend_pipeline* bu_frontend_pipeline_create(int device, uint32_t lanes) try { return pipeline_new(device, lanes, true); } BU_PIPE_CATCH(nullptr)
bu_frontend_pipeline* bu_frontend_pipeline_create_n(int device, uint32_t lanes, uint32_t driver_threads) try { return pipeline_new(device, lanes, true, driver_threads); } BU_PIPE_CATCH(nullptr)

// Test hook (no GPU): `tasks` self-test tasks of `yields` yields each through a `lanes`-lane pipeline; every third one throws and catches inside, `failing` of them end in
// an exception that must surface as a failed job. 1 = every task saw its own stack intact at every resume, failures were reported as failures, nothing else was.
int bu_frontend_pipeline_selftest(uint32_t lanes, uint32_t tasks, uint32_t yields, uint32_t failing) try {
    bu_frontend_pipeline* p = pipeline_new(0, lanes, false, lanes >= 4 ? 2u : 1u);   // two driver threads from four lanes up: both forms are exercised
    if (!p) return 0;
    p->idle_sleep_s = 0;
    std::vector<uint64_t> t;
    bool ok = true;
    for (uint32_t i = 0; i < tasks; i++) {
        job* j = new job();
        j->d.flags = 0x80000000u; j->d.n_blocks = yields + i % 7; j->d.max_threads = i % 3 == 0; j->d.compression_level = i < failing ? 99 : 0;
        {
            std::lock_guard<std::mutex> g(p->m);
            j->ticket = p->next_ticket++; p->jobs[j->ticket] = j; p->pending.push_back(j);
        }
        p->cv_work.notify_all();
        t.push_back(j->ticket);
    }
    for (uint32_t i = 0; i < tasks; i++) {
        std::unique_lock<std::mutex> g(p->m);
        job* j = p->jobs[t[i]];
        p->cv_done.wait(g, [&] { return j->done; });
        ok = ok && (j->ok == (i >= failing)) && (i >= failing || j->error == "a task that fails");
    }
    uint64_t yields_seen = 0;
    for (auto& l : p->lanes) yields_seen += l.yields;
    uint64_t want = 0;
    for (uint32_t i = 0; i < tasks; i++) want += yields + i % 7;
    ok = ok && yields_seen == want && p->n_done == tasks;
    bu_frontend_pipeline_destroy(p);
    return ok ? 1 : 0;
} BU_PIPE_CATCH(0)

uint64_t bu_frontend_pipeline_submit(bu_frontend_pipeline* p, const bu_frontend_job* d, uint32_t struct_bytes) try {
    if (!p || !d || struct_bytes < offsetof(bu_frontend_job, max_threads)) return 0;
    job* j = new job();
    std::memcpy(&j->d, d, std::min<size_t>(struct_bytes, sizeof(j->d)));   // fields a caller's older header does not have stay 0
    if (!j->d.n_blocks || (!j->d.h_blocks == !j->d.d_blocks)) { delete j; bu_last_exception_text = "bu_frontend_pipeline_submit: exactly one of h_blocks / d_blocks, n_blocks > 0"; return 0; }
    if (j->d.flags & ~(uint32_t)BU_FRONTEND_JOB_VIDEO) {   // unknown bits are refused, not ignored (the top one is the library's own self-test task: only bu_frontend_pipeline_selftest sets it)
        delete j; bu_last_exception_text = "bu_frontend_pipeline_submit: unknown bits in bu_frontend_job::flags"; return 0;
    }
    j->t_submit = now_s();
    {
        std::lock_guard<std::mutex> g(p->m);
        if (p->stop) { delete j; return 0; }
        j->ticket = p->next_ticket++;
        p->jobs[j->ticket] = j;
        p->pending.push_back(j);
    }
    p->cv_work.notify_all();
    return j->ticket;
} BU_PIPE_CATCH(0)

bu_frontend* bu_frontend_pipeline_wait(bu_frontend_pipeline* p, uint64_t ticket) try {
    if (!p) return nullptr;
    std::unique_lock<std::mutex> g(p->m);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) { p->error = "bu_frontend_pipeline_wait: unknown ticket"; return nullptr; }
    job* j = it->second;
    if (j->handed_out) { p->error = "bu_frontend_pipeline_wait: this ticket has a waiter or has been collected already"; return nullptr; }
    j->handed_out = true;   // the ticket is this caller's from here on (a second wait on it is refused instead of sharing -- or outliving -- the job)
    p->cv_done.wait(g, [&] { return j->done; });
    if (!j->ok) {
        p->error = "job " + std::to_string(ticket) + ": " + j->error;
        p->jobs.erase(it);
        g.unlock();
        free_job(j);
        return nullptr;
    }
    p->by_frontend[j->fe] = j;
    return j->fe;
} BU_PIPE_CATCH(nullptr)

int bu_frontend_pipeline_poll(bu_frontend_pipeline* p, uint64_t ticket) {
    if (!p) return -1;
    std::lock_guard<std::mutex> g(p->m);
    auto it = p->jobs.find(ticket);
    return it == p->jobs.end() ? -1 : (it->second->done ? 1 : 0);
}

bu_hip_context* bu_frontend_pipeline_context(bu_frontend_pipeline* p, bu_frontend* fe) {
    if (!p || !fe) return nullptr;
    std::lock_guard<std::mutex> g(p->m);
    auto it = p->by_frontend.find(fe);
    return it == p->by_frontend.end() ? nullptr : it->second->ctx;
}

int bu_frontend_pipeline_release(bu_frontend_pipeline* p, bu_frontend* fe) try {
    if (!p || !fe) return 0;
    job* j = nullptr;
    {
        std::lock_guard<std::mutex> g(p->m);
        auto it = p->by_frontend.find(fe);
        if (it == p->by_frontend.end()) { p->error = "bu_frontend_pipeline_release: not a frontend of this pipeline (released twice?)"; return 0; }
        j = it->second;
        p->by_frontend.erase(it);
        p->jobs.erase(j->ticket);
    }
    free_job(j);
    return 1;
} BU_PIPE_CATCH(0)

void bu_frontend_pipeline_destroy(bu_frontend_pipeline* p) try {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(p->m);
        p->stop = true;    // the driver finishes what is queued and in flight, then leaves
    }
    p->cv_work.notify_all();
    for (auto& d : p->drivers) if (d.th.joinable()) d.th.join();
    for (auto& kv : p->jobs) free_job(kv.second);   // results nobody collected or released
    for (auto& l : p->lanes) if (l.stack) munmap(static_cast<char*>(l.stack) - 4096, l.stack_bytes + 4096);
    delete p;
} catch (...) {}

const char* bu_frontend_pipeline_error(const bu_frontend_pipeline* cp) {   // a copy per calling thread: other threads' failures do not move it under the caller
    if (!cp) return "null pipeline";
    static thread_local std::string mine;
    bu_frontend_pipeline* p = const_cast<bu_frontend_pipeline*>(cp);
    std::lock_guard<std::mutex> g(p->m);
    mine = p->error;
    return mine.c_str();
}

uint32_t bu_frontend_pipeline_stats(bu_frontend_pipeline* p, double* out, uint32_t cap) {
    if (!p) return 0;
    std::lock_guard<std::mutex> g(p->m);
    uint64_t yields = 0, switches = 0, naps = 0; double busy = 0, idle = 0, cpu = 0;
    for (auto& d : p->drivers) { switches += d.pub.switches; naps += d.pub.naps; yields += d.pub.yields; busy += d.pub.busy_s; idle += d.pub.idle_s; cpu += d.pub.cpu_s; }
    const double v[7] = {(double)p->n_done, (double)switches, (double)yields, (double)naps, busy, idle, cpu};
    for (uint32_t i = 0; i < 7 && i < cap; i++) out[i] = v[i];
    return 7;
}

} // extern "C"
