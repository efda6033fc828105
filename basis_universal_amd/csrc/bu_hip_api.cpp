// bu_hip_api.cpp -- context management and the C ABI of libbasisu_hip.so (include/basisu_hip.h).
//
// Replaces encoder/basisu_opencl.cpp of the reference: same entry points, same ownership and error conventions
// (opencl.cpp:730-1213), but one HIP stream per context, persistent scratch arenas instead of per-call cl buffers,
// and a device-resident layer (section 2 of the header) underneath the blocking host-pointer layer (section 1).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <thread>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/basisu_hip.h"
#include "etc1s_kernels.h"
#include "tsvq_kernels.h"
#include "tsvq_bufs.h"
#include "uastc_kernels.h"
#include "mipmap_kernels.h"
#include "unique_kernels.h"
#include "bookkeeping_kernels.h"
#include "kmeans_kernels.h"

namespace {

std::mutex g_init_mutex;
bool g_initialized = false;
int g_device_count = 0;
std::string g_global_error;

// A grow-only device buffer: the per-call temporaries of the blocking layer live here so that repeated calls
// (one per frontend stage, several per refinement iteration) do not hit hipMalloc/hipFree.
struct arena {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = std::max(bytes, (size_t)4096);
        want += want / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace

struct bu_hip_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // a second stream for work that is independent of what the main stream is doing (the one-workgroup TSVQ splits of a round next to
    // its many-workgroup ones); joined back through the two events before anything reads the results
    hipStream_t side_stream = nullptr; hipEvent_t side_fork = nullptr, side_join = nullptr;
    bool dedicated_queue = false;         // own_stream was made with a (full) CU mask: a hardware queue of its own instead of a share of the runtime's pool
    // UASTC pipeline lanes with reserved walk CUs (bu_hip_tuning::uastc_walk_cus): the lean strip walk of uastc_rdo goes to walk_stream, whose CU mask is the reserved
    // set; own_stream (and with it everything that fills the chip) is masked to the OTHER CUs, side_stream (the walk with the refit in it) to the reserved ones
    hipStream_t walk_stream = nullptr; hipEvent_t walk_join = nullptr; uint32_t walk_cus = 0;
    arena refine_lists;                   // the sorted candidate lists of refine_endpoint_clusterization (etc1s_kernels.hip, k_refine_sort_lists)
    const void* d_pixel_blocks = nullptr; // resident tiles (a1): 64 B per block
    size_t total_blocks = 0;
    arena pixel_arena;                    // owns the tiles when they were uploaded through bu_hip_set_pixel_blocks
    arena scratch[6];
    // pinned staging ring for host -> device uploads of pageable caller memory (see h2d below)
    void* stage = nullptr; size_t stage_cap = 0, stage_used = 0;
    void* bounce = nullptr; size_t bounce_cap = 0;   // pinned bounce buffer of device -> host downloads under a wait hook (bu_hip_memcpy_d2h)
    // small results (mail_fetch): a coherent page-locked buffer a one-workgroup kernel copies them into, followed by a word the host looks at; -1 = not available
    void* mail = nullptr; char* mail_dev = nullptr; int mail_state = 0; uint32_t mail_seq = 0;
    // pipelined tile upload (bu_hip_k_upload_and_encode_etc1s_blocks): a pinned ring of UP_SLOTS pieces the helper threads fill, one event per piece in flight
    void* up_ring = nullptr; size_t up_ring_cap = 0; std::vector<hipEvent_t> up_events;
    // background downloads (bu_hip_download_*): their own stream, so that a copy never sits in front of the side stream's kernels; events recycled; handles not yet waited for
    hipStream_t copy_stream = nullptr; std::vector<hipEvent_t> down_events;
    // ... carried out by ONE helper thread per context, started with the first download and parked on a condition variable between them (starting a thread per
    // download cost the calling thread 30-40 us each, on the step's critical path)
    std::thread down_thread; std::mutex down_mu; std::condition_variable down_cv, down_done_cv; std::deque<struct bu_hip_download*> down_queue; bool down_stop = false;
    std::string error;
    // bu_hip_malloc / bu_hip_free recycle blocks per context: an encoder frees and re-allocates the same dozen buffers for every
    // image, and hipMalloc/hipFree cost 0.1-1 ms each (hipFree also synchronises the device). Reuse is stream-ordered: everything
    // that touches these blocks is enqueued on the context's stream.
    void* tsvq_pinned = nullptr; size_t tsvq_pinned_cap = 0;  // recycled by bu_tsvq objects (one alive at a time per stream in practice)
    struct pooled { void* p; size_t cap; };
    std::vector<pooled> pool_free;
    std::vector<pooled> pool_live;
    size_t pool_free_bytes = 0;
    // optional per-kernel timing with HIP events on the launch stream (bu_hip_profile_*)
    int profiling = 0;   // bu_hip_profile_enable: 0 off, 1 every region, 2 the regions that are one kernel launch each
    struct prof_rec { const char* name; hipEvent_t start, stop; };
    std::vector<prof_rec> prof_pending;
    std::vector<hipEvent_t> prof_events;   // recycled (creating and destroying two events per timed region cost more host time than recording them)
    struct prof_sum { const char* name; double ms; uint32_t launches; };
    std::vector<prof_sum> prof_totals;
    bu_hip_tuning tuning{};               // bu_hip_set_tuning; starts as the process defaults (measured values, environment overrides read once)
    // cooperative waiting (bu_hip_set_wait_hook): called between looks at the stream wherever a call on this context would block its host thread
    bu_hip_wait_fn wait_hook = nullptr; void* wait_user = nullptr;
    // bu_hip_on_destroy registrations
    std::mutex closing_lock;
    std::vector<std::pair<bu_hip_destroy_fn, void*>> closing;
};

static std::atomic<int> g_live_contexts{0};   // contexts in use (not parked): decides how a waiting host thread waits

namespace {

void set_error(bu_hip_context* ctx, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (ctx) ctx->error = buf; else g_global_error = buf;
}

#define BU_TRY(ctx, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { set_error(ctx, "%s: %s", #expr, hipGetErrorString(e__)); return 0; } } while (0)

// Every wait of a call for its context's own stream. Default: block the host thread. With a wait hook (a host that runs several contexts as cooperative tasks
// on one thread: bu_frontend_pipeline_*) the stream is only ever QUERIED and the hook runs between the looks -- it switches to another task and returns when it is
// this one's turn again.
hipError_t stream_wait(bu_hip_context* ctx, hipStream_t s) {
    if (!ctx->wait_hook) return hipStreamSynchronize(s);
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        ctx->wait_hook(ctx->wait_user);
    }
}

// Device -> host copy into PAGEABLE caller memory, enqueued on the context's stream. The runtime blocks the calling thread inside such a copy until the stream has
// drained; under a wait hook the draining is waited for cooperatively first, so that what blocks is only the (microseconds of a) copy from an idle stream.
hipError_t d2h_pageable(bu_hip_context* ctx, void* h, const void* d, size_t bytes) {
    if (ctx->wait_hook) { const hipError_t e = stream_wait(ctx, ctx->stream); if (e != hipSuccess) return e; }
    return hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
}


// Waits for the word a one-thread kernel (k_tsvq_signal / the tail of k_mail_copy) stores into a coherent page-locked buffer. 1 = seen, 0 = the stream failed (error text set).
int wait_flag(bu_hip_context* ctx, int poll_mode, volatile uint32_t* round_flag, uint32_t seq) {
    // One context in the process: spin (the round trip is what the step waits for). Several (basis_parallel_compress, images in flight): the device is shared, a round
    // can take milliseconds, and a spinning waiter takes a core from another image's host backend -- after 30 us the core is offered to whoever wants it, after
    // 2 ms the thread sleeps between looks. bu_hip_tuning::tsvq_poll (BU_TSVQ_POLL=spin|yield) overrides.
    const bool polite = !ctx->wait_hook && (poll_mode == 2 || (poll_mode == 0 && g_live_contexts.load(std::memory_order_relaxed) > 1));
    const auto t_wait0 = std::chrono::steady_clock::now();
    auto last_query = t_wait0;
    for (;;) {
        if (*round_flag == seq) break;
        if (ctx->wait_hook) {   // cooperative: another task of this host thread runs while the round is on the device
            ctx->wait_hook(ctx->wait_user);
            if (*round_flag == seq) break;
        }
        const auto t_now = std::chrono::steady_clock::now();
        if (polite && t_now - t_wait0 > std::chrono::microseconds(30)) {
            if (t_now - t_wait0 > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));
            else std::this_thread::yield();
        }
        if (t_now - last_query > std::chrono::microseconds(200)) {   // every 200 us: did the stream die, or finish without the flag becoming visible?
            last_query = t_now;
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) { __atomic_thread_fence(__ATOMIC_SEQ_CST); if (*round_flag != seq) BU_TRY(ctx, stream_wait(ctx, ctx->stream)); break; }
            if (e != hipErrorNotReady) { set_error(ctx, "tsvq_split: %s", hipGetErrorString(e)); return 0; }
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 1;
}


// Small device results for the host WITHOUT a copy command: a one-workgroup kernel copies them into a coherent page-locked buffer and stores a sequence number behind them,
// the host looks at that word (wait_flag: spinning, yielding or running the wait hook) and copies them out. A hipMemcpyAsync into pageable memory + hipStreamSynchronize
// costs a blit launch by the runtime, its completion signal and the wake-up: 25-40 us between the producing kernel and the host's next launch; this is ~10.
// Up to four parts per wait (results that live in different arrays); what does not fit, or a context without the buffer, takes the copy.
constexpr size_t MAIL_BYTES = (size_t)64 << 10, MAIL_FLAG_AT = MAIL_BYTES;
struct mail_fetch {
    bu_hip_context* ctx;
    struct part { void* h; const void* d; size_t at, bytes; } parts[4];
    int n = 0; size_t used = 0; bool copied = false;
    explicit mail_fetch(bu_hip_context* c) : ctx(c) {}
    bool usable() {
        if (ctx->mail_state == 0) {
            ctx->mail_state = -1;
            static const bool on = [] { const char* e = std::getenv("BU_MAIL_FETCH"); return !e || e[0] != '0'; }();   // A/B switch
            if (!on) return false;
            void* p = nullptr;
            if (hipHostMalloc(&p, MAIL_BYTES + 256, hipHostMallocCoherent) == hipSuccess) {
                void* dp = nullptr;
                if (hipHostGetDevicePointer(&dp, p, 0) == hipSuccess) { ctx->mail = p; ctx->mail_dev = static_cast<char*>(dp); ctx->mail_state = 1; *reinterpret_cast<volatile uint32_t*>(static_cast<char*>(p) + MAIL_FLAG_AT) = 0; }
                else { (void)hipGetLastError(); (void)hipHostFree(p); }
            } else (void)hipGetLastError();
        }
        return ctx->mail_state == 1;
    }
    hipError_t add(void* h, const void* d, size_t bytes) {
        if (!bytes) return hipSuccess;
        const size_t need = (bytes + 15) & ~(size_t)15;
        if (n == 4 || used + need > MAIL_BYTES || !usable()) { copied = true; return d2h_pageable(ctx, h, d, bytes); }
        parts[n++] = part{h, d, used, bytes};
        used += need;
        return hipSuccess;
    }
    // 1 = everything added is in the caller's memory
    int wait() {
        if (n) {
            const uint32_t seq = ++ctx->mail_seq ? ctx->mail_seq : ++ctx->mail_seq;   // never 0
            volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(static_cast<char*>(ctx->mail) + MAIL_FLAG_AT);
            for (int i = 0; i < n; i++)
                BU_TRY(ctx, bu::launch_mail_copy(ctx->stream, ctx->mail_dev + parts[i].at, parts[i].d, parts[i].bytes, i + 1 == n ? reinterpret_cast<uint32_t*>(ctx->mail_dev + MAIL_FLAG_AT) : nullptr, seq));
            if (!wait_flag(ctx, (int)ctx->tuning.tsvq_poll, flag, seq)) return 0;
            for (int i = 0; i < n; i++) std::memcpy(parts[i].h, static_cast<const char*>(ctx->mail) + parts[i].at, parts[i].bytes);
        }
        if (copied || !n) BU_TRY(ctx, stream_wait(ctx, ctx->stream));
        return 1;
    }
};
// device -> host + wait, one result
int fetch(bu_hip_context* ctx, void* h, const void* d, size_t bytes) {
    mail_fetch f(ctx);
    BU_TRY(ctx, f.add(h, d, bytes));
    return f.wait();
}

struct device_guard {
    int prev = -1; bool ok = false;
    explicit device_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
    }
    ~device_guard() { /* leave the context's device current: callers (torch) re-select theirs explicitly */ }
};

// RAII bracket around one kernel launch sequence: records a start/stop event pair on the launch stream when profiling is on.
struct prof_scope {
    bu_hip_context* ctx; const char* name; hipEvent_t start = nullptr, stop = nullptr;
    prof_scope(bu_hip_context* c, const char* n) : ctx(c), name(n) {
        if (!ctx->profiling) return;
        // level 2: only the regions that are ONE kernel launch; the many-launch regions (codebook builders' rounds, de-duplication sorts, list bookkeeping) go untimed,
        // and with them the events that would sit between their kernels
        if (ctx->profiling == 2 && (std::strncmp(n, "tsvq_", 5) == 0 || std::strncmp(n, "unique_", 7) == 0 || std::strncmp(n, "map_", 4) == 0 || std::strncmp(n, "kmeans_", 7) == 0)) return;
        if (ctx->prof_events.size() < 2 && ctx->prof_pending.size() >= 64) {   // reap the oldest finished regions: their events are the next ones recorded
            size_t done = 0;
            while (done < ctx->prof_pending.size() && done < 8 && hipEventQuery(ctx->prof_pending[done].stop) == hipSuccess) {
                const bu_hip_context::prof_rec& r = ctx->prof_pending[done++];
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
                    bool found = false;
                    for (auto& t : ctx->prof_totals) if (t.name == r.name) { t.ms += ms; t.launches++; found = true; break; }
                    if (!found) ctx->prof_totals.push_back({r.name, (double)ms, 1u});
                }
                ctx->prof_events.push_back(r.start); ctx->prof_events.push_back(r.stop);
            }
            if (done) ctx->prof_pending.erase(ctx->prof_pending.begin(), ctx->prof_pending.begin() + (long)done);
            (void)hipGetLastError();   // a hipErrorNotReady of the last look must not be what the next launcher's hipGetLastError() finds
        }
        auto take = [&](hipEvent_t& e) {
            if (!ctx->prof_events.empty()) { e = ctx->prof_events.back(); ctx->prof_events.pop_back(); return true; }
            return hipEventCreate(&e) == hipSuccess;
        };
        if (!take(start)) { start = nullptr; return; }
        if (!take(stop)) { ctx->prof_events.push_back(start); start = stop = nullptr; return; }
        (void)hipEventRecord(start, ctx->stream);
    }
    ~prof_scope() {
        if (!start) return;
        (void)hipEventRecord(stop, ctx->stream);
        ctx->prof_pending.push_back({name, start, stop});
    }
};

void prof_drain(bu_hip_context* ctx) {
    for (auto& r : ctx->prof_pending) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            bool found = false;
            for (auto& t : ctx->prof_totals) if (t.name == r.name) { t.ms += ms; t.launches++; found = true; break; }
            if (!found) ctx->prof_totals.push_back({r.name, (double)ms, 1u});
        }
        ctx->prof_events.push_back(r.start); ctx->prof_events.push_back(r.stop);
    }
    ctx->prof_pending.clear();
}

// Host -> device upload of caller-owned (pageable) memory, ordered on the context's stream. Pageable memory is never handed to
// hipMemcpyAsync: on this stack (ROCm 7.2, MI355X) a kernel launched right behind such a copy was observed to read the
// destination before the data had landed (tools/tsvq_root_repeat.py). Small uploads go through a pinned staging ring (a real
// stream-ordered DMA; the ring is recycled only after a stream synchronise), large ones through a blocking hipMemcpy.
hipError_t h2d(bu_hip_context* ctx, void* d, const void* h, size_t bytes) {
    if (!bytes) return hipSuccess;
    hipError_t e;
    if (bytes > ((size_t)4 << 20)) {
        // large uploads (an image's tiles) go through the ring in 4 MiB pieces: a blocking hipMemcpy + hipDeviceSynchronize here made every image's upload wait for
        // every OTHER context's kernels (basis_parallel_compress: one context per image in flight), which serialised the images
        for (size_t at = 0; at < bytes; at += (size_t)4 << 20) {
            const size_t piece = std::min(bytes - at, (size_t)4 << 20);
            if ((e = h2d(ctx, static_cast<char*>(d) + at, static_cast<const char*>(h) + at, piece)) != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > ctx->stage_cap - ctx->stage_used) {
        if ((e = stream_wait(ctx, ctx->stream)) != hipSuccess) return e; // every copy out of the ring has completed
        ctx->stage_used = 0;
        if (need > ctx->stage_cap) {
            if (ctx->stage) { (void)hipHostFree(ctx->stage); ctx->stage = nullptr; ctx->stage_cap = 0; }
            const size_t want = std::max(need * 2, (size_t)16 << 20);
            if ((e = hipHostMalloc(&ctx->stage, want, hipHostMallocDefault)) != hipSuccess) { ctx->stage = nullptr; return e; }
            ctx->stage_cap = want;
        }
    }
    char* slot = static_cast<char*>(ctx->stage) + ctx->stage_used;
    std::memcpy(slot, h, bytes);
    ctx->stage_used += need;
    return hipMemcpyAsync(d, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
}

int quality_from_perms(uint32_t total_perms) {
    // frontend.cpp:746-752 / etc.cpp:792-800: {4,16,64,165} <-> {fast, medium, slow, uber}
    if (total_perms <= 4) return bu::BU_Q_FAST;
    if (total_perms <= 16) return bu::BU_Q_MEDIUM;
    if (total_perms <= 64) return bu::BU_Q_SLOW;
    return bu::BU_Q_UBER;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------------------- init

int bu_hip_init(int /*force_serialization*/) {
    // (No environment is touched here. Streams of a process share a few hardware queues -- GPU_MAX_HW_QUEUES, ROCm's default: 4 -- and two streams that land on one
    // queue run their kernels one after the other; a host that wants more than four lanes side by side sets that variable itself before its first HIP call: INTEGRATION.md.)
    std::lock_guard<std::mutex> lock(g_init_mutex);
    if (g_initialized) return 1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error(nullptr, "bu_hip_init: no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return 0;
    }
    g_device_count = n;
    g_initialized = true;
    return 1;
}

// Contexts are PARKED, not torn down, when they are destroyed: the reference's throughput driver (basis_parallel_compress, comp.cpp:5466-5559) creates one accelerator
// context per image and destroys it with the image, and a context's worth of device buffers costs ~20 hipMalloc calls to build and as many hipFree calls -- each one a
// DEVICE-wide synchronisation that stalls every other image's stream -- to tear down. A parked context keeps its stream, its workspaces and its block pool; the next
// bu_hip_create_context on that device gets it back, warm. At most BU_HIP_PARKED_CONTEXTS (default 16, 0 = off) are kept; bu_hip_deinit releases them.
static std::mutex g_park_lock;
static std::vector<bu_hip_context*> g_parked;
// contexts handed out and not yet given back: with more than one, a host thread that waits for its device round shares the cores with the other contexts' host work
static void context_release(bu_hip_context* ctx);   // the real teardown
static const bu_hip_tuning& default_tuning();
static bool ensure_side_stream(bu_hip_context* ctx);
static hipStream_t make_dedicated_stream(int device, uint32_t reserve, bool reserved_side);
static size_t park_limit() {
    static const size_t n = [] { const char* e = std::getenv("BU_HIP_PARKED_CONTEXTS"); const long v = e ? std::atol(e) : 16; return (size_t)(v < 0 ? 0 : (v > 64 ? 64 : v)); }();
    return n;
}

void bu_hip_deinit(void) {
    std::vector<bu_hip_context*> parked;
    { std::lock_guard<std::mutex> g(g_park_lock); parked.swap(g_parked); }
    for (bu_hip_context* c : parked) context_release(c);
    std::lock_guard<std::mutex> lock(g_init_mutex);
    g_initialized = false;
}

int bu_hip_is_available(void) { return g_initialized ? 1 : 0; }

// A parked context keeps its stream, and with it the KIND of queue the stream sits on: the lanes of a UASTC pipeline run on hardware queues of their own (dedicated_queue),
// everybody else on the runtime's pooled ones -- ETC1S frontend jobs measured 15-20 % slower with every context on its own queue. So a caller gets a parked context of
// the kind it asks for: the public create calls never a dedicated-queue one, the UASTC pipeline those first.
static bu_hip_context* create_context_kind(int device, bool want_dedicated) {
    if (!g_initialized) { set_error(nullptr, "bu_hip_create_context: bu_hip_init() has not succeeded"); return nullptr; }
    if (device < 0 || device >= g_device_count) { set_error(nullptr, "bu_hip_create_context: bad device %d", device); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { set_error(nullptr, "hipSetDevice(%d) failed", device); return nullptr; }
    {
        std::lock_guard<std::mutex> g(g_park_lock);
        for (int pass = 0; pass < (want_dedicated ? 2 : 1); pass++)   // (a lane that finds no parked context of its own kind takes a pooled one and moves it onto a queue of its own)
            for (size_t i = 0; i < g_parked.size(); i++)
                if (g_parked[i]->device == device && g_parked[i]->dedicated_queue == (want_dedicated && pass == 0)) {
                    bu_hip_context* c = g_parked[i]; g_parked.erase(g_parked.begin() + (long)i); g_live_contexts.fetch_add(1); return c;
                }
    }
    bu_hip_context* ctx = new (std::nothrow) bu_hip_context();
    if (!ctx) return nullptr;
    ctx->device = device;
    ctx->tuning = default_tuning();
    // (a pooled queue, not a dedicated one: with EVERY context on a queue of its own the frontend pipeline lost 15-20 %, measured)
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error(nullptr, "hipStreamCreate failed"); delete ctx; return nullptr; }
    ctx->stream = ctx->own_stream;
    hipError_t e = bu::upload_etc1s_tables(device);
    if (e != hipSuccess) { set_error(nullptr, "constant table upload failed: %s", hipGetErrorString(e)); (void)hipStreamDestroy(ctx->own_stream); delete ctx; return nullptr; }
    g_live_contexts.fetch_add(1);
    return ctx;
}

bu_hip_context* bu_hip_create_context_on(int device) { return create_context_kind(device, false); }

bu_hip_context* bu_hip_create_context(void) {
    int dev = 0;
    if (!g_initialized) { set_error(nullptr, "bu_hip_create_context: bu_hip_init() has not succeeded"); return nullptr; }
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return bu_hip_create_context_on(dev);
}

int bu_hip_on_destroy(bu_hip_context* ctx, bu_hip_destroy_fn fn, void* user) {
    if (!ctx || !fn) return 0;
    std::lock_guard<std::mutex> g(ctx->closing_lock);
    ctx->closing.emplace_back(fn, user);
    return 1;
}
void bu_hip_cancel_on_destroy(bu_hip_context* ctx, bu_hip_destroy_fn fn, void* user) {
    if (!ctx) return;
    std::lock_guard<std::mutex> g(ctx->closing_lock);
    for (size_t i = 0; i < ctx->closing.size(); i++)
        if (ctx->closing[i].first == fn && ctx->closing[i].second == user) { ctx->closing.erase(ctx->closing.begin() + (long)i); break; }
}

void bu_hip_destroy_context(bu_hip_context* ctx) {
    if (!ctx) return;
    g_live_contexts.fetch_sub(1);
    (void)hipSetDevice(ctx->device);
    for (;;) {  // dependents first (a callback may cancel others; each runs once, outside the lock)
        std::pair<bu_hip_destroy_fn, void*> cb;
        {
            std::lock_guard<std::mutex> g(ctx->closing_lock);
            if (ctx->closing.empty()) break;
            cb = ctx->closing.back(); ctx->closing.pop_back();
        }
        cb.first(cb.second);
    }
    bool healthy = stream_wait(ctx, ctx->stream) == hipSuccess;
    if (ctx->side_stream) healthy = hipStreamSynchronize(ctx->side_stream) == hipSuccess && healthy;
    if (ctx->own_stream != ctx->stream) healthy = hipStreamSynchronize(ctx->own_stream) == hipSuccess && healthy;
    if (!healthy) (void)hipGetLastError();
    prof_drain(ctx);
    if (park_limit() && healthy) {   // a context whose streams ended in an error is torn down, never handed to the next creator
        // back to the state bu_hip_create_context_on hands out, with the memory kept: blocks the caller leaked join the free list (the context owns all device memory it handed out)
        for (auto& b : ctx->pool_live) { ctx->pool_free.push_back(b); ctx->pool_free_bytes += b.cap; }
        ctx->pool_live.clear();
        ctx->stream = ctx->own_stream;
        ctx->d_pixel_blocks = nullptr; ctx->total_blocks = 0;
        ctx->stage_used = 0;
        ctx->error.clear();
        ctx->profiling = 0; ctx->prof_totals.clear();
        ctx->wait_hook = nullptr; ctx->wait_user = nullptr;
        ctx->tuning = default_tuning();
        std::lock_guard<std::mutex> g(g_park_lock);
        if (g_parked.size() < park_limit()) { g_parked.push_back(ctx); return; }
    }
    context_release(ctx);
}

static void context_release(bu_hip_context* ctx) {
    (void)hipSetDevice(ctx->device);
    ctx->pixel_arena.release();
    for (auto& a : ctx->scratch) a.release();
    ctx->refine_lists.release();
    if (ctx->tsvq_pinned) (void)hipHostFree(ctx->tsvq_pinned);
    for (auto& b : ctx->pool_free) (void)hipFree(b.p);
    for (auto& b : ctx->pool_live) (void)hipFree(b.p);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    if (ctx->bounce) (void)hipHostFree(ctx->bounce);
    if (ctx->mail) (void)hipHostFree(ctx->mail);
    if (ctx->up_ring) (void)hipHostFree(ctx->up_ring);
    if (ctx->down_thread.joinable()) {
        { std::lock_guard<std::mutex> lk(ctx->down_mu); ctx->down_stop = true; }
        ctx->down_cv.notify_all();
        ctx->down_thread.join();
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->down_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->up_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->walk_stream) (void)hipStreamDestroy(ctx->walk_stream);
    if (ctx->walk_join) (void)hipEventDestroy(ctx->walk_join);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int bu_hip_context_device(const bu_hip_context* ctx) { return ctx ? ctx->device : -1; }
int bu_hip_set_stream(bu_hip_context* ctx, void* s) {
    if (!ctx) return 0;
    (void)stream_wait(ctx, ctx->stream); // staged uploads still in flight belong to the old stream
    ctx->stage_used = 0;
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return 1;
}
void* bu_hip_get_stream(bu_hip_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
const char* bu_hip_last_error(const bu_hip_context* ctx) { return ctx ? ctx->error.c_str() : g_global_error.c_str(); }

// The process defaults of bu_hip_tuning: measured values, each overridable ONCE per process by the environment variable named in basisu_hip.h.
static const bu_hip_tuning& default_tuning() {
    static const bu_hip_tuning t = [] {
        bu_hip_tuning d;
        std::memset(&d, 0, sizeof(d));
        d.struct_bytes = (uint32_t)sizeof(d);
        d.tsvq_wide_min = 8192; d.tsvq_wide6_min = 8192; d.tsvq_wide_cov_min = 98304; d.tsvq_windows = 0; d.tsvq_dense_min = 257; d.tsvq_zero_copy = 1; d.tsvq_deep_levels = 0; d.uastc_walk_cus = 0; d.codebook_wide_min = 32768;
        auto num = [](const char* name, long lo, long hi, uint32_t* out) { if (const char* e = std::getenv(name)) { const long v = std::atol(e); if (v >= lo && v <= hi) *out = (uint32_t)v; } };
        num("BU_TSVQ_WIDE_MIN", 512, 1l << 30, &d.tsvq_wide_min);
        num("BU_TSVQ_WIDE6_MIN", 512, 1l << 30, &d.tsvq_wide6_min);
        num("BU_TSVQ_WIDE_COV_MIN", 0, 1l << 30, &d.tsvq_wide_cov_min);
        num("BU_TSVQ_DENSE_MIN", 0, 1l << 30, &d.tsvq_dense_min);
        num("BU_TSVQ_ZEROCOPY", 0, 1, &d.tsvq_zero_copy);
        num("BU_TSVQ_DEEP", 0, (long)bu::TSVQ_MAX_DEEP_LEVELS, &d.tsvq_deep_levels);
        num("BU_UASTC_WALK_CUS", 0, 128, &d.uastc_walk_cus);
        num("BU_CODEBOOK_WIDE_MIN", 0, 1l << 30, &d.codebook_wide_min);
        if (const char* e = std::getenv("BU_TSVQ_WIDE")) if (std::atoi(e) == 0) d.tsvq_wide_min = d.tsvq_wide6_min = 0;
        if (const char* e = std::getenv("BU_TSVQ_WIDE6")) if (std::atoi(e) == 0) d.tsvq_wide6_min = 0;
        if (const char* e = std::getenv("BU_TSVQ_WINDOWS")) d.tsvq_windows = e[0] == '0' ? 2u : 1u;
        if (std::getenv("BU_TSVQ_CHAINED")) d.tsvq_chained_only = 1;
        if (const char* e = std::getenv("BU_TSVQ_POLL")) d.tsvq_poll = e[0] == 's' ? 1u : 2u;
        d.debug = (std::getenv("BU_TSVQ_ROUNDS") ? 1u : 0u) | (std::getenv("BU_TSVQ_SERIAL") ? 2u : 0u) | (std::getenv("BU_TSVQ_STATS") ? 4u : 0u);
        return d;
    }();
    return t;
}

void bu_hip_get_tuning(const bu_hip_context* ctx, bu_hip_tuning* out, uint32_t struct_bytes) {
    if (!out || struct_bytes < 8) return;
    const bu_hip_tuning& t = ctx ? ctx->tuning : default_tuning();
    std::memcpy(out, &t, std::min<size_t>(struct_bytes, sizeof(t)));
    out->struct_bytes = (uint32_t)std::min<size_t>(struct_bytes, sizeof(t));
}

int bu_hip_set_tuning(bu_hip_context* ctx, const bu_hip_tuning* t) {
    if (!ctx) return 0;
    if (!t) { ctx->tuning = default_tuning(); return 1; }
    if (t->struct_bytes < 8 || t->struct_bytes > 4096) { set_error(ctx, "bu_hip_set_tuning: struct_bytes %u", t->struct_bytes); return 0; }
    bu_hip_tuning n = default_tuning();   // fields a caller's older header does not have keep their defaults
    std::memcpy(&n, t, std::min<size_t>(t->struct_bytes, sizeof(n)));
    n.struct_bytes = (uint32_t)sizeof(n);
    if ((n.tsvq_wide_min && n.tsvq_wide_min < 512) || (n.tsvq_wide6_min && n.tsvq_wide6_min < 512) || n.tsvq_windows > 2 || n.tsvq_poll > 2 || n.tsvq_deep_levels > bu::TSVQ_MAX_DEEP_LEVELS) {
        set_error(ctx, "bu_hip_set_tuning: value out of range (many-workgroup thresholds are 0 or >= 512, windows / poll 0..2, deep levels 0..2)");
        return 0;
    }
    ctx->tuning = n;
    return 1;
}

int bu_hip_set_wait_hook(bu_hip_context* ctx, bu_hip_wait_fn fn, void* user) {
    if (!ctx) return 0;
    ctx->wait_hook = fn; ctx->wait_user = fn ? user : nullptr;
    return 1;
}

int bu_hip_sync(bu_hip_context* ctx) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));
    return 1;
}

void* bu_hip_malloc(bu_hip_context* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    device_guard g(ctx->device);
    const size_t want = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
    // best fit among the cached blocks, but never more than twice (+1 MiB) what was asked for
    int best = -1;
    for (size_t i = 0; i < ctx->pool_free.size(); i++) {
        const size_t cap = ctx->pool_free[i].cap;
        if (cap >= want && cap <= want * 2 + ((size_t)1 << 20) && (best < 0 || cap < ctx->pool_free[(size_t)best].cap)) best = (int)i;
    }
    if (best >= 0) {
        const bu_hip_context::pooled b = ctx->pool_free[(size_t)best];
        ctx->pool_free.erase(ctx->pool_free.begin() + best);
        ctx->pool_free_bytes -= b.cap;
        ctx->pool_live.push_back(b);
        return b.p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        // out of memory: drop the cache and retry once
        (void)stream_wait(ctx, ctx->stream);
        for (auto& b : ctx->pool_free) (void)hipFree(b.p);
        ctx->pool_free.clear(); ctx->pool_free_bytes = 0;
        if (hipMalloc(&p, want) != hipSuccess) { set_error(ctx, "hipMalloc(%zu) failed", bytes); (void)hipGetLastError(); return nullptr; }
    }
    ctx->pool_live.push_back({p, want});
    return p;
}
void bu_hip_free(bu_hip_context* ctx, void* p) {
    if (!ctx || !p) return;
    device_guard g(ctx->device);
    for (size_t i = 0; i < ctx->pool_live.size(); i++)
        if (ctx->pool_live[i].p == p) {
            const bu_hip_context::pooled b = ctx->pool_live[i];
            ctx->pool_live.erase(ctx->pool_live.begin() + (long)i);
            if (ctx->pool_free_bytes + b.cap <= ((size_t)16 << 30)) { ctx->pool_free.push_back(b); ctx->pool_free_bytes += b.cap; return; }
            (void)stream_wait(ctx, ctx->stream);
            (void)hipFree(p);
            return;
        }
    // not a live block of this context: a second free of a pooled block (it is in pool_free and will be handed out again -- hipFree here would
    // turn that into a use after free) or a foreign pointer. Leave it alone and say so.
    set_error(ctx, "bu_hip_free: %p is not a live allocation of this context (double free?)", p);
}
int bu_hip_memcpy_h2d(bu_hip_context* ctx, void* d, const void* h, size_t bytes) {
    if (!ctx) return 0;
    if (!bytes) return 1;
    device_guard g(ctx->device);
    BU_TRY(ctx, h2d(ctx, d, h, bytes));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream)); // h may be pageable and released by the caller right after
    return 1;
}
// the same, stream-ordered: on return `h` has been copied out (into the context's pinned ring) and may be released; the device side is ordered with everything enqueued on
// the context's stream before and after. No host synchronisation (the ring synchronises the stream only when it wraps).
int bu_hip_memcpy_h2d_async(bu_hip_context* ctx, void* d, const void* h, size_t bytes) {
    if (!ctx) return 0;
    if (!bytes) return 1;
    device_guard g(ctx->device);
    BU_TRY(ctx, h2d(ctx, d, h, bytes));
    return 1;
}
int bu_hip_memcpy_d2h(bu_hip_context* ctx, void* h, const void* d, size_t bytes) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    if (ctx->wait_hook && bytes > 4096) {
        // cooperative host: hipMemcpyAsync into pageable memory would block this thread for the whole transfer (the runtime stages it synchronously). Through a pinned
        // bounce buffer the transfer is a real stream-ordered DMA and the thread's other tasks run while it is in flight.
        const size_t piece_max = (size_t)32 << 20;
        const size_t want = std::min(bytes, piece_max);
        if (want > ctx->bounce_cap) {
            if (ctx->bounce) { (void)hipHostFree(ctx->bounce); ctx->bounce = nullptr; ctx->bounce_cap = 0; }
            BU_TRY(ctx, hipHostMalloc(&ctx->bounce, want, hipHostMallocDefault));
            ctx->bounce_cap = want;
        }
        for (size_t at = 0; at < bytes; at += piece_max) {
            const size_t piece = std::min(bytes - at, piece_max);
            BU_TRY(ctx, hipMemcpyAsync(ctx->bounce, static_cast<const char*>(d) + at, piece, hipMemcpyDeviceToHost, ctx->stream));
            BU_TRY(ctx, stream_wait(ctx, ctx->stream));
            std::memcpy(static_cast<char*>(h) + at, ctx->bounce, piece);
        }
        return 1;
    }
    // small results travel without a copy command (mail_fetch); the rest: (under a wait hook the stream is drained cooperatively first: the runtime would block in the copy until it has)
    return fetch(ctx, h, d, bytes);
}
void* bu_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void bu_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---- background downloads
struct bu_hip_download {
    bu_hip_context* ctx; hipEvent_t ready; void* h; const void* d; size_t bytes; bool done; hipError_t result;
};
static void download_worker(bu_hip_context* c) {
    (void)hipSetDevice(c->device);
    for (;;) {
        bu_hip_download* dl = nullptr;
        {
            std::unique_lock<std::mutex> lk(c->down_mu);
            c->down_cv.wait(lk, [&] { return c->down_stop || !c->down_queue.empty(); });
            if (c->down_queue.empty()) return;   // stop, and nothing left to do
            dl = c->down_queue.front(); c->down_queue.pop_front();
        }
        // this thread is what blocks in the copy (into pageable memory the runtime holds the calling thread for the whole transfer); the copy waits for `ready` on the device
        hipError_t e = hipStreamWaitEvent(c->copy_stream, dl->ready, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(dl->h, dl->d, dl->bytes, hipMemcpyDeviceToHost, c->copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->copy_stream);
        { std::lock_guard<std::mutex> lk(c->down_mu); dl->result = e; dl->done = true; }
        c->down_done_cv.notify_all();
    }
}
bu_hip_download* bu_hip_download_begin(bu_hip_context* ctx, void* h, const void* d, size_t bytes) {
    if (!ctx || !h || !d || !bytes || ctx->wait_hook) return nullptr;
    device_guard g(ctx->device);
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); ctx->copy_stream = nullptr; return nullptr; }
    if (!ctx->down_thread.joinable()) {
        try { ctx->down_thread = std::thread(download_worker, ctx); } catch (...) { return nullptr; }
    }
    hipEvent_t ev = nullptr;
    if (!ctx->down_events.empty()) { ev = ctx->down_events.back(); ctx->down_events.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipEventRecord(ev, ctx->stream) != hipSuccess) { (void)hipGetLastError(); ctx->down_events.push_back(ev); return nullptr; }
    bu_hip_download* dl = new (std::nothrow) bu_hip_download{ctx, ev, h, d, bytes, false, hipSuccess};
    if (!dl) { ctx->down_events.push_back(ev); return nullptr; }
    { std::lock_guard<std::mutex> lk(ctx->down_mu); ctx->down_queue.push_back(dl); }
    ctx->down_cv.notify_one();
    return dl;
}
int bu_hip_download_wait(bu_hip_download* dl) {
    if (!dl) return 0;
    bu_hip_context* ctx = dl->ctx;
    { std::unique_lock<std::mutex> lk(ctx->down_mu); ctx->down_done_cv.wait(lk, [&] { return dl->done; }); }
    ctx->down_events.push_back(dl->ready);
    const hipError_t e = dl->result;
    delete dl;
    if (e != hipSuccess) { set_error(ctx, "bu_hip_download: %s", hipGetErrorString(e)); (void)hipGetLastError(); return 0; }
    return 1;
}

int bu_hip_memcpy_d2d(bu_hip_context* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return 0;
    if (!bytes) return 1;
    device_guard g(ctx->device);
    BU_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 1;
}
int bu_hip_memset(bu_hip_context* ctx, void* d, int value, size_t bytes) {
    if (!ctx) return 0;
    if (!bytes) return 1;
    device_guard g(ctx->device);
    BU_TRY(ctx, hipMemsetAsync(d, value, bytes, ctx->stream));
    return 1;
}

int bu_hip_profile_enable(bu_hip_context* ctx, int on) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_drain(ctx);
    ctx->prof_totals.clear();
    ctx->profiling = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return 1;
}

uint32_t bu_hip_profile_read(bu_hip_context* ctx, const char** names, double* total_ms, uint32_t* launches, uint32_t cap) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_drain(ctx);
    const uint32_t n = (uint32_t)std::min<size_t>(ctx->prof_totals.size(), cap);
    for (uint32_t i = 0; i < n; i++) { names[i] = ctx->prof_totals[i].name; total_ms[i] = ctx->prof_totals[i].ms; launches[i] = ctx->prof_totals[i].launches; }
    return (uint32_t)ctx->prof_totals.size();
}

// ---------------------------------------------------------------------------------------------------------------- tiles

int bu_hip_set_pixel_blocks(bu_hip_context* ctx, size_t total_blocks, const bu_pixel_block* blocks) {
    if (!ctx) return 0;
    if (total_blocks > 0xFFFFFFFFull) { set_error(ctx, "too many blocks"); return 0; }
    device_guard g(ctx->device);
    BU_TRY(ctx, ctx->pixel_arena.reserve(total_blocks * sizeof(bu_pixel_block)));
    if (total_blocks) BU_TRY(ctx, h2d(ctx, ctx->pixel_arena.p, blocks, total_blocks * sizeof(bu_pixel_block)));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream)); // the caller may free its copy right after (frontend.cpp:67-79)
    ctx->d_pixel_blocks = ctx->pixel_arena.p;
    ctx->total_blocks = total_blocks;
    return 1;
}

int bu_hip_set_pixel_blocks_device(bu_hip_context* ctx, size_t total_blocks, const void* d_blocks) {
    if (!ctx || total_blocks > 0xFFFFFFFFull) return 0;
    ctx->d_pixel_blocks = d_blocks;
    ctx->total_blocks = total_blocks;
    return 1;
}

const void* bu_hip_get_pixel_blocks_device(const bu_hip_context* ctx, size_t* total_blocks) {
    if (!ctx) return nullptr;
    if (total_blocks) *total_blocks = ctx->total_blocks;
    return ctx->d_pixel_blocks;
}

// ---------------------------------------------------------------------------------------------------------------- section 2

int bu_hip_k_encode_etc1s_blocks(bu_hip_context* ctx, const void* d_px, uint32_t n, int quality, int perceptual, void* d_out) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "encode_etc1s_blocks");
    BU_TRY(ctx, bu::launch_encode_etc1s_blocks(ctx->stream, d_px, n, quality, perceptual != 0, d_out));
    return 1;
}

// Tiles from HOST memory and their first kernel as one pipeline (SURVEY 8d figure (i): the hot path with the host-to-device transfer inside). The upload goes in pieces of
// UP_PIECE_BLOCKS tiles on the context's side stream (the copy engine), the etc1_optimizer kernel of piece i is launched on the main stream behind piece i's event: while piece
// i is encoded, piece i + 1 is on the link and -- for pageable source memory -- pieces i + 2 .. are being copied into the pinned ring by helper threads (one host thread
// copies at 12-17 GB/s, a third of the link). Page-locked source memory is handed to the copy engine as it is. On return h_px may be released (every piece has left it), the
// device side is ordered on the context's stream like any other launch. Same bytes in d_out as bu_hip_k_encode_etc1s_blocks over the uploaded tiles.
namespace {
constexpr uint32_t UP_PIECE_BLOCKS = 65536;   // 4 MiB of tiles: 0.08 ms on the link, 0.11 ms of the kernel
constexpr uint32_t UP_SLOTS = 8, UP_EVENTS = 64;
unsigned upload_threads() {
    static const unsigned t = [] {
        unsigned want = 4;
        if (const char* e = std::getenv("BU_UPLOAD_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 16) want = (unsigned)v; }
        const unsigned hw = std::thread::hardware_concurrency();
        return hw ? std::min(want, std::max(1u, hw / 2)) : 1u;
    }();
    return t;
}
}
int bu_hip_k_upload_and_encode_etc1s_blocks(bu_hip_context* ctx, void* d_px, const void* h_px, uint32_t n, int quality, int perceptual, void* d_out) {
    if (!ctx) return 0;
    if (!n) return 1;
    if (!d_px || !h_px || !d_out) { set_error(ctx, "upload_and_encode_etc1s_blocks: null argument"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n_pieces = (n + UP_PIECE_BLOCKS - 1) / UP_PIECE_BLOCKS;
    // a cooperative host (wait hook) must never block in an event wait, and one piece is no pipeline: upload, then one launch
    if (ctx->wait_hook || n_pieces < 2 || !ensure_side_stream(ctx)) {
        BU_TRY(ctx, h2d(ctx, d_px, h_px, (size_t)n * 64));
        prof_scope ps(ctx, "encode_etc1s_blocks");
        BU_TRY(ctx, bu::launch_encode_etc1s_blocks(ctx->stream, d_px, n, quality, perceptual != 0, d_out));
        return 1;
    }
    while (ctx->up_events.size() < UP_EVENTS) {
        hipEvent_t e = nullptr;
        BU_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->up_events.push_back(e);
    }
    bool pinned_src = false;
    {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, h_px) == hipSuccess) pinned_src = a.type == hipMemoryTypeHost;
        else (void)hipGetLastError();   // ordinary pageable memory is "invalid value" to the runtime: not an error of ours
    }
    const size_t piece_bytes = (size_t)UP_PIECE_BLOCKS * 64;
    if (!pinned_src && ctx->up_ring_cap < piece_bytes * UP_SLOTS) {
        if (ctx->up_ring) { (void)hipHostFree(ctx->up_ring); ctx->up_ring = nullptr; ctx->up_ring_cap = 0; }
        BU_TRY(ctx, hipHostMalloc(&ctx->up_ring, piece_bytes * UP_SLOTS, hipHostMallocDefault));
        ctx->up_ring_cap = piece_bytes * UP_SLOTS;
    }
    // the destination may be a recycled block that earlier launches on the main stream still read: the copies start behind them
    BU_TRY(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
    BU_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
    const char* src = static_cast<const char*>(h_px);
    char* dst = static_cast<char*>(d_px);
    char* out = static_cast<char*>(d_out);
    // pageable source: helper thread t copies pieces t, t + T, ... into ring slot (piece % UP_SLOTS) as soon as the piece that used the slot before has left it
    std::vector<std::atomic<int>> ready(pinned_src ? 0 : n_pieces), issued(pinned_src ? 0 : n_pieces);
    std::atomic<int> stop{0};
    std::vector<std::thread> helpers;
    if (!pinned_src) {
        for (auto& r : ready) r.store(0, std::memory_order_relaxed);
        for (auto& r : issued) r.store(0, std::memory_order_relaxed);
        const unsigned T = std::min<unsigned>(upload_threads(), n_pieces);
        try {
            for (unsigned t = 0; t < T; t++)
                helpers.emplace_back([&, t, T] {
                    (void)hipSetDevice(ctx->device);
                    for (uint32_t i = t; i < n_pieces && !stop.load(std::memory_order_acquire); i += T) {
                        if (i >= UP_SLOTS) {
                            while (!issued[i - UP_SLOTS].load(std::memory_order_acquire)) { if (stop.load(std::memory_order_acquire)) return; std::this_thread::yield(); }
                            (void)hipEventSynchronize(ctx->up_events[(i - UP_SLOTS) % UP_EVENTS]);
                        }
                        const size_t at = (size_t)i * piece_bytes, bytes = std::min(piece_bytes, (size_t)n * 64 - at);
                        std::memcpy(static_cast<char*>(ctx->up_ring) + (size_t)(i % UP_SLOTS) * piece_bytes, src + at, bytes);
                        ready[i].store(1, std::memory_order_release);
                    }
                });
        } catch (...) {
            stop.store(1); for (auto& h : helpers) h.join();
            set_error(ctx, "upload_and_encode_etc1s_blocks: could not start the helper threads");
            return 0;
        }
    }
    hipError_t err = hipSuccess;
    {
        prof_scope ps(ctx, "upload_and_encode_etc1s_blocks");
        for (uint32_t i = 0; i < n_pieces && err == hipSuccess; i++) {
            const size_t at = (size_t)i * piece_bytes, bytes = std::min(piece_bytes, (size_t)n * 64 - at);
            const uint32_t blocks = (uint32_t)(bytes / 64);
            const void* from = src + at;
            if (!pinned_src) {
                while (!ready[i].load(std::memory_order_acquire)) std::this_thread::yield();
                from = static_cast<char*>(ctx->up_ring) + (size_t)(i % UP_SLOTS) * piece_bytes;
            }
            hipEvent_t ev = ctx->up_events[i % UP_EVENTS];
            if ((err = hipMemcpyAsync(dst + at, from, bytes, hipMemcpyHostToDevice, ctx->side_stream)) != hipSuccess) break;
            if ((err = hipEventRecord(ev, ctx->side_stream)) != hipSuccess) break;
            if (!pinned_src) issued[i].store(1, std::memory_order_release);
            if ((err = hipStreamWaitEvent(ctx->stream, ev, 0)) != hipSuccess) break;
            err = bu::launch_encode_etc1s_blocks(ctx->stream, dst + at, blocks, quality, perceptual != 0, out + (size_t)i * UP_PIECE_BLOCKS * 8);
        }
    }
    if (err != hipSuccess) stop.store(1, std::memory_order_release);
    for (auto& h : helpers) h.join();
    // the source (or the ring) must have been read completely before the caller releases it (or the next call refills the ring): the last copy is the one to wait for
    const hipError_t drained = hipStreamSynchronize(ctx->side_stream);
    if (err == hipSuccess) err = drained;
    if (err != hipSuccess) { set_error(ctx, "upload_and_encode_etc1s_blocks: %s", hipGetErrorString(err)); return 0; }
    return 1;
}

int bu_hip_k_endpoint_training_vectors(bu_hip_context* ctx, const void* d_etc, uint32_t n, float* d_out6) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "endpoint_training_vectors");
    BU_TRY(ctx, bu::launch_endpoint_training_vectors(ctx->stream, d_etc, n, d_out6));
    return 1;
}

// The clusters of a codebook fit, split by size: one workgroup per cluster for the many small ones (largest first, so the big ones do not start last), the
// many-workgroup passes of etc1s_codebook_wide.inc for those of bu_hip_tuning::codebook_wide_min texels and more (a sky, a flat wall, a constant alpha plane: one
// workgroup would walk 10^5-10^7 texels 17 times while the rest of the chip waits). `order`: the call's clusters, largest first.
static int codebook_fit_split(bu_hip_context* ctx, const std::vector<uint32_t>& order, const uint32_t* h_offsets, const void* d_px, const void* d_enc, const uint32_t* d_offsets,
                              const uint32_t* d_indices, int quality, bool perceptual, bool forced, uint32_t step, uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid,
                              uint64_t* d_cur_err, const char* label) {
    const uint32_t wide_min = ctx->tuning.codebook_wide_min;
    std::vector<uint32_t> small_ones, big_cluster, big_first, big_sub;
    for (uint32_t c : order) {
        const uint32_t sub = h_offsets[c + 1] - h_offsets[c];
        if (wide_min && sub && (uint64_t)sub * 8u >= wide_min && sub < (1u << 28)) { big_cluster.push_back(c); big_first.push_back(h_offsets[c]); big_sub.push_back(sub); }
        else small_ones.push_back(c);
    }
    std::vector<unsigned char> image;
    bu::cb_wide_layout L{};
    if (!big_cluster.empty()) L = bu::codebook_wide_prepare(big_cluster.data(), big_first.data(), big_sub.data(), (uint32_t)big_cluster.size(), image);
    arena& ord = ctx->scratch[5];
    const size_t ord_bytes = (small_ones.size() * sizeof(uint32_t) + 255) & ~(size_t)255;
    BU_TRY(ctx, ord.reserve(ord_bytes + L.total));
    char* work = static_cast<char*>(ord.p) + ord_bytes;
    if (!small_ones.empty()) BU_TRY(ctx, h2d(ctx, ord.p, small_ones.data(), small_ones.size() * sizeof(uint32_t)));
    if (!image.empty()) BU_TRY(ctx, h2d(ctx, work, image.data(), image.size()));
    {
        prof_scope ps(ctx, label);
        if (!small_ones.empty()) {
            if (forced) BU_TRY(ctx, bu::launch_refit_endpoints_given_selectors(ctx->stream, d_px, d_enc, (uint32_t)small_ones.size(), static_cast<const uint32_t*>(ord.p), d_offsets, d_indices,
                                                                               quality, perceptual, d_params, d_err, d_valid, d_cur_err));
            else BU_TRY(ctx, bu::launch_generate_endpoint_codebook(ctx->stream, d_px, (uint32_t)small_ones.size(), static_cast<const uint32_t*>(ord.p), d_offsets, d_indices,
                                                                   quality, perceptual, step, d_params, d_err, d_valid));
        }
        if (L.n_big) BU_TRY(ctx, bu::launch_codebook_wide(ctx->stream, d_px, d_indices, work, L, quality, perceptual, forced, step, d_enc, d_params, d_err, d_valid, d_cur_err));
    }
    // (no wait here: h2d has copied the uploads' sources into the context's page-locked ring before it returned, and whoever wants the results waits for them)
    return 1;
}

// Test hook: the colour mean the cluster fit starts from (etc.cpp:1034-1041: a running float sum in texel order, divided by the count) of every cluster, through the
// many-workgroup path's order-free evaluation of that sum (etc1s_codebook_wide.inc, cbw_ordered_sum) whatever the clusters' sizes. h_out: 3 floats per cluster.
int bu_hip_k_cluster_colour_means(bu_hip_context* ctx, const void* d_px, uint32_t n_clusters, const uint32_t* h_offsets, const uint32_t* d_indices, float* h_out) {
    if (!ctx) return 0;
    if (!n_clusters) return 1;
    if (!d_px || !h_offsets || !d_indices || !h_out) { set_error(ctx, "cluster_colour_means: null argument"); return 0; }
    device_guard g(ctx->device);
    std::vector<uint32_t> cl(n_clusters), first(n_clusters), sub(n_clusters);
    for (uint32_t c = 0; c < n_clusters; c++) {
        cl[c] = c; first[c] = h_offsets[c]; sub[c] = h_offsets[c + 1] - h_offsets[c];
        if (!sub[c]) { set_error(ctx, "cluster_colour_means: empty cluster %u", c); return 0; }
    }
    std::vector<unsigned char> image;
    const bu::cb_wide_layout L = bu::codebook_wide_prepare(cl.data(), first.data(), sub.data(), n_clusters, image);
    arena& ws = ctx->scratch[5];
    const size_t out_at = (L.total + 255) & ~(size_t)255;
    BU_TRY(ctx, ws.reserve(out_at + (size_t)n_clusters * 12));
    BU_TRY(ctx, h2d(ctx, ws.p, image.data(), image.size()));
    float* d_out = reinterpret_cast<float*>(static_cast<char*>(ws.p) + out_at);
    BU_TRY(ctx, bu::launch_codebook_wide_means(ctx->stream, d_px, d_indices, ws.p, L, d_out));
    if (!fetch(ctx, h_out, d_out, (size_t)n_clusters * 12)) return 0;
    return 1;
}

// The clusters by descending size, ties in index order (= std::stable_sort with that comparator), as an LSD radix sort of the sizes: the device waits while this runs
// (one workgroup per cluster, the big ones must not start last), and the comparison sort of a few thousand indirect keys was 50-80 us of that wait.
static std::vector<uint32_t> size_descending_order(const uint32_t* h_offsets, uint32_t n) {
    std::vector<uint32_t> order(n), tmp(n), key(n);
    uint32_t largest = 0;
    for (uint32_t i = 0; i < n; i++) largest = std::max(largest, h_offsets[i + 1] - h_offsets[i]);
    for (uint32_t i = 0; i < n; i++) { order[i] = i; key[i] = largest - (h_offsets[i + 1] - h_offsets[i]); }   // ascending key = descending size
    for (uint32_t shift = 0; shift < 32 && (largest >> shift); shift += 11) {
        uint32_t count[2049] = {};
        for (uint32_t i = 0; i < n; i++) count[((key[order[i]] >> shift) & 2047u) + 1]++;
        for (uint32_t b = 0; b < 2048; b++) count[b + 1] += count[b];
        for (uint32_t i = 0; i < n; i++) tmp[count[(key[order[i]] >> shift) & 2047u]++] = order[i];
        order.swap(tmp);
    }
    return order;
}

int bu_hip_k_generate_endpoint_codebook_part(bu_hip_context* ctx, const void* d_px, uint32_t n_clusters, const uint32_t* h_offsets,
                                             const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual, uint32_t step,
                                             uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid, uint32_t part, uint32_t parts) {
    if (!ctx) return 0;
    if (!n_clusters) return 1;
    if (!parts || part >= parts) { set_error(ctx, "generate_endpoint_codebook: bad part %u of %u", part, parts); return 0; }
    device_guard g(ctx->device);
    // largest clusters first: one workgroup per cluster, so the big ones must not start last. With parts > 1 this call handles the
    // clusters at positions part, part + parts, ... of that order (the same order on every rank: the sort is stable and deterministic).
    const std::vector<uint32_t> order = size_descending_order(h_offsets, n_clusters);
    std::vector<uint32_t> mine;
    for (uint32_t i = part; i < n_clusters; i += parts) mine.push_back(order[i]);
    if (mine.empty()) return 1;
    return codebook_fit_split(ctx, mine, h_offsets, d_px, nullptr, d_offsets, d_indices, quality, perceptual != 0, false, step, d_params, d_err, d_valid, nullptr, "generate_endpoint_codebook");
}

int bu_hip_k_generate_endpoint_codebook(bu_hip_context* ctx, const void* d_px, uint32_t n_clusters, const uint32_t* h_offsets,
                                        const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual, uint32_t step,
                                        uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid) {
    return bu_hip_k_generate_endpoint_codebook_part(ctx, d_px, n_clusters, h_offsets, d_offsets, d_indices, quality, perceptual, step, d_params, d_err, d_valid, 0, 1);
}

int bu_hip_k_refit_endpoints_given_selectors_q(bu_hip_context* ctx, const void* d_px, const void* d_enc, uint32_t n_clusters, const uint32_t* h_offsets,
                                               const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual, uint8_t* d_params, uint64_t* d_err,
                                             uint8_t* d_valid, uint64_t* d_cur_err) {
    if (!ctx) return 0;
    if (!n_clusters) return 1;
    device_guard g(ctx->device);
    const std::vector<uint32_t> order = size_descending_order(h_offsets, n_clusters);
    return codebook_fit_split(ctx, order, h_offsets, d_px, d_enc, d_offsets, d_indices, quality == BU_ETC_QUALITY_SLOW ? BU_ETC_QUALITY_SLOW : BU_ETC_QUALITY_UBER, perceptual != 0, true, 0u,
                              d_params, d_err, d_valid, d_cur_err, "refit_endpoints_given_selectors");
}

int bu_hip_k_refit_endpoints_given_selectors(bu_hip_context* ctx, const void* d_px, const void* d_enc, uint32_t n_clusters, const uint32_t* h_offsets,
                                             const uint32_t* d_offsets, const uint32_t* d_indices, int perceptual, uint8_t* d_params, uint64_t* d_err,
                                             uint8_t* d_valid, uint64_t* d_cur_err) {
    return bu_hip_k_refit_endpoints_given_selectors_q(ctx, d_px, d_enc, n_clusters, h_offsets, d_offsets, d_indices, BU_ETC_QUALITY_UBER, perceptual, d_params, d_err, d_valid, d_cur_err);
}

int bu_hip_k_subblock_errors(bu_hip_context* ctx, const void* d_px, uint32_t n_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                             int perceptual, uint64_t* d_out) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "subblock_errors");
    BU_TRY(ctx, bu::launch_subblock_errors(ctx->stream, d_px, n_blocks, d_block_cluster, d_cluster_params, perceptual != 0, d_out));
    return 1;
}

int bu_hip_k_backend_block_errors(bu_hip_context* ctx, const void* d_px, const void* d_etc_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                                  uint32_t first_block, uint32_t num_blocks_x, uint32_t num_blocks_y, uint32_t n_clusters, int perceptual, int with_neighbours,
                                  uint32_t* d_own_err, uint32_t* d_neighbour_err) {
    if (!ctx) return 0;
    if (!d_px || !d_etc_blocks || !d_block_cluster || !d_cluster_params || !d_own_err || (with_neighbours && !d_neighbour_err)) { set_error(ctx, "backend_block_errors: null argument"); return 0; }
    device_guard g(ctx->device);
    prof_scope ps(ctx, "backend_block_errors");
    BU_TRY(ctx, bu::launch_backend_block_errors(ctx->stream, d_px, d_etc_blocks, d_block_cluster, d_cluster_params, first_block, num_blocks_x, num_blocks_y, n_clusters, perceptual != 0,
                                                with_neighbours != 0, d_own_err, d_neighbour_err));
    return 1;
}

int bu_hip_k_refine_endpoint_clusterization(bu_hip_context* ctx, const void* d_px, uint32_t n_blocks, const uint32_t* d_block_cluster,
                                            const uint8_t* d_cluster_params, uint32_t n_clusters, uint32_t n_parents, const uint32_t* d_cand_offsets,
                                            const uint32_t* d_cand_indices, const uint8_t* d_block_parent, int perceptual, uint32_t* d_out_best) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    void* work = nullptr;
    if (const size_t wb = ctx->tuning.refine_unsorted ? 0 : bu::refine_workspace_bytes(n_clusters, n_parents)) { BU_TRY(ctx, ctx->refine_lists.reserve(wb)); work = ctx->refine_lists.p; }
    prof_scope ps(ctx, "refine_endpoint_clusterization");
    BU_TRY(ctx, bu::launch_refine_endpoint_clusterization(ctx->stream, d_px, n_blocks, d_block_cluster, d_cluster_params, n_clusters, n_parents,
                                                          d_cand_offsets, d_cand_indices, d_block_parent, perceptual != 0, d_out_best, work));
    return 1;
}

int bu_hip_k_extract_blocks(bu_hip_context* ctx, const void* d_rgba, uint32_t width, uint32_t height, uint32_t pitch_bytes, void* d_out) {
    if (!ctx) return 0;
    if (!d_rgba || !d_out || !width || !height || pitch_bytes < width * 4u) { set_error(ctx, "extract_blocks: bad arguments"); return 0; }
    device_guard g(ctx->device);
    prof_scope ps(ctx, "extract_blocks");
    BU_TRY(ctx, bu::launch_extract_blocks(ctx->stream, d_rgba, width, height, pitch_bytes, d_out));
    return 1;
}

int bu_hip_k_resample_rgba8(bu_hip_context* ctx, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h,
                            const uint32_t* x_first, const uint16_t* x_pixel, const float* x_weight, const uint32_t* y_first, const uint16_t* y_pixel, const float* y_weight,
                            int x_after_y, int srgb, const float* srgb_to_linear, const uint8_t* linear_to_srgb, uint32_t num_comps) {
    if (!ctx) return 0;
    if (!d_src || !d_dst || !src_w || !src_h || !dst_w || !dst_h || !x_first || !x_pixel || !x_weight || !y_first || !y_pixel || !y_weight || !srgb_to_linear ||
        !linear_to_srgb || num_comps < 3 || num_comps > 4 || src_w > 16384 || src_h > 16384) { set_error(ctx, "resample_rgba8: bad arguments"); return 0; }
    device_guard g(ctx->device);
    // everything the kernels read besides the image, packed into one upload: lists of both axes, then the tables
    const size_t nx = x_first[dst_w], ny = y_first[dst_h];
    for (uint32_t i = 0; i < nx; i++) if (x_pixel[i] >= src_w) { set_error(ctx, "resample_rgba8: x contributor out of range"); return 0; }
    for (uint32_t i = 0; i < ny; i++) if (y_pixel[i] >= src_h) { set_error(ctx, "resample_rgba8: y contributor out of range"); return 0; }
    auto pad = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_xf = 0, o_xw = pad(o_xf + (dst_w + 1) * 4), o_xp = pad(o_xw + nx * 4), o_yf = pad(o_xp + nx * 2), o_yw = pad(o_yf + (dst_h + 1) * 4), o_yp = pad(o_yw + ny * 4),
                 o_t0 = pad(o_yp + ny * 2), o_t1 = o_t0 + 1024, total = o_t1 + 8192;
    std::vector<uint8_t> pack(total, 0);
    std::memcpy(&pack[o_xf], x_first, (dst_w + 1) * 4); std::memcpy(&pack[o_xw], x_weight, nx * 4); std::memcpy(&pack[o_xp], x_pixel, nx * 2);
    std::memcpy(&pack[o_yf], y_first, (dst_h + 1) * 4); std::memcpy(&pack[o_yw], y_weight, ny * 4); std::memcpy(&pack[o_yp], y_pixel, ny * 2);
    std::memcpy(&pack[o_t0], srgb_to_linear, 1024); std::memcpy(&pack[o_t1], linear_to_srgb, 8192);
    arena &lists = ctx->scratch[4], &tmp = ctx->scratch[5];
    const size_t tmp_px = std::max((size_t)dst_w * src_h, (size_t)src_w * dst_h);
    BU_TRY(ctx, lists.reserve(total));
    BU_TRY(ctx, tmp.reserve(tmp_px * 16));
    BU_TRY(ctx, h2d(ctx, lists.p, pack.data(), total));
    const char* b = static_cast<const char*>(lists.p);
    {
        prof_scope ps(ctx, "resample_rgba8");
        BU_TRY(ctx, bu::launch_resample_rgba8(ctx->stream, d_src, src_w, src_h, d_dst, dst_w, dst_h, (const uint32_t*)(b + o_xf), (const uint16_t*)(b + o_xp), (const float*)(b + o_xw),
                                              (const uint32_t*)(b + o_yf), (const uint16_t*)(b + o_yp), (const float*)(b + o_yw), x_after_y != 0, srgb != 0,
                                              (const float*)(b + o_t0), (const uint8_t*)(b + o_t1), num_comps, tmp.p));
    }
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));  // the packed lists are reused by the next call
    return 1;
}

int bu_hip_k_determine_selectors(bu_hip_context* ctx, const void* d_px, uint32_t n_blocks, const uint8_t* d_color5_inten,
                                 const uint32_t* d_block_cluster, int perceptual, void* d_out) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "determine_selectors");
    BU_TRY(ctx, bu::launch_determine_selectors(ctx->stream, d_px, n_blocks, d_color5_inten, d_block_cluster, perceptual != 0, d_out));
    return 1;
}

int bu_hip_k_selector_training_vectors(bu_hip_context* ctx, const void* d_enc, uint32_t n_blocks, int perceptual, float* d_out16, uint64_t* d_w) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "selector_training_vectors");
    BU_TRY(ctx, bu::launch_selector_training_vectors(ctx->stream, d_enc, n_blocks, perceptual != 0, d_out16, d_w));
    return 1;
}

int bu_hip_k_create_optimized_selector_codebook(bu_hip_context* ctx, const void* d_px, const void* d_enc, uint32_t n_clusters,
                                                const uint32_t* d_offsets, const uint32_t* d_block_indices, int perceptual, void* d_selector_blocks) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    if (!n_clusters) return 1;
    // (no look at the offsets from here: the accumulation kernel is a fixed number of waves that find the span on the device)
    arena& ws = ctx->scratch[4];
    BU_TRY(ctx, ws.reserve(bu::create_optimized_selector_codebook_workspace_bytes(n_clusters)));
    prof_scope ps(ctx, "create_optimized_selector_codebook");
    BU_TRY(ctx, bu::launch_create_optimized_selector_codebook(ctx->stream, d_px, d_enc, n_clusters, d_offsets, d_block_indices, perceptual != 0, ws.p,
                                                              d_selector_blocks));
    return 1;
}

int bu_hip_k_find_optimal_selector_clusters(bu_hip_context* ctx, const void* d_px, void* d_enc, uint32_t n_blocks, const void* d_selector_blocks,
                                            uint32_t n_selectors, uint32_t n_parents, const uint32_t* d_cand_offsets, const uint32_t* d_cand_indices,
                                            const uint8_t* d_block_parent, int perceptual, uint32_t chunk, uint32_t* d_out) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "find_optimal_selector_clusters");
    arena& tmp = ctx->scratch[4];
    // behind the per-block scratch: room for the candidates' selector words in list order (at most parents x selectors of them; left out beyond 64 MiB)
    const size_t idx_bytes = ((size_t)n_blocks * sizeof(uint32_t) + 255) & ~(size_t)255;
    size_t words = (size_t)(n_parents ? n_parents : 1u) * n_selectors;
    if (words * 4 > ((size_t)64 << 20)) words = 0;
    BU_TRY(ctx, tmp.reserve(idx_bytes + words * 4));
    BU_TRY(ctx, bu::launch_find_optimal_selector_clusters(ctx->stream, d_px, d_enc, n_blocks, d_selector_blocks, n_selectors, n_parents, d_cand_offsets,
                                                          d_cand_indices, d_block_parent, perceptual != 0, chunk, static_cast<uint32_t*>(tmp.p), d_out,
                                                          words ? reinterpret_cast<uint32_t*>(static_cast<char*>(tmp.p) + idx_bytes) : nullptr, words));
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------- cluster bookkeeping on the device

int bu_hip_k_map_blocks_from_groups(bu_hip_context* ctx, const uint32_t* d_goffs, const uint32_t* d_idx, uint32_t n, uint32_t u_total, const uint32_t* d_leaf,
                                    const uint32_t* d_first_pos, const uint32_t* d_parent_of_unique, uint32_t* d_cluster, uint32_t* d_pos, uint8_t* d_parent) {
    if (!ctx) return 0;
    if (n && (!d_goffs || !d_idx || !d_leaf || !d_cluster || (d_pos && !d_first_pos))) { set_error(ctx, "map_blocks_from_groups: null pointer"); return 0; }
    device_guard g(ctx->device);
    prof_scope ps(ctx, "map_blocks_from_groups");
    BU_TRY(ctx, bu::launch_blocks_from_groups(ctx->stream, d_goffs, d_idx, n, u_total, d_leaf, d_first_pos, d_parent_of_unique, d_cluster, d_pos, d_parent));
    return 1;
}

int bu_hip_k_map_rank_blocks(bu_hip_context* ctx, const uint32_t* d_cluster, uint32_t n, uint32_t k, uint32_t* d_sizes, uint32_t* d_offsets, uint32_t* d_sorted, uint32_t* d_pos) {
    if (!ctx) return 0;
    if (n && (!d_cluster || !d_sizes || !d_offsets || !d_sorted)) { set_error(ctx, "map_rank_blocks: null pointer"); return 0; }
    device_guard g(ctx->device);
    arena& ws = ctx->scratch[4];
    BU_TRY(ctx, ws.reserve(bu::rank_blocks_workspace_bytes(n, k)));
    prof_scope ps(ctx, "map_rank_blocks");
    BU_TRY(ctx, bu::launch_rank_blocks(ctx->stream, d_cluster, n, k, ws.p, d_sizes, d_offsets, d_sorted, d_pos));
    return 1;
}

int bu_hip_k_map_endpoint_csr(bu_hip_context* ctx, const uint32_t* d_cluster, const uint32_t* d_pos, uint32_t n, const uint32_t* d_offsets, uint32_t* d_indices) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "map_endpoint_csr");
    BU_TRY(ctx, bu::launch_endpoint_csr_fill(ctx->stream, d_cluster, d_pos, n, d_offsets, d_indices));
    return 1;
}

int bu_hip_k_map_remap(bu_hip_context* ctx, uint32_t* d_cluster, uint32_t* d_pos, uint32_t n, const uint32_t* d_new_index, const uint32_t* d_base) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    prof_scope ps(ctx, "map_remap");
    BU_TRY(ctx, bu::launch_remap_clusters(ctx->stream, d_cluster, d_pos, n, d_new_index, d_base));
    return 1;
}

int bu_hip_k_map_count_differences(bu_hip_context* ctx, const uint32_t* d_a, const uint32_t* d_b, uint32_t n, uint32_t* d_count) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    BU_TRY(ctx, bu::launch_count_differences(ctx->stream, d_a, d_b, n, d_count));
    return 1;
}

int bu_hip_k_map_membership(bu_hip_context* ctx, const uint8_t* d_parent, const uint32_t* d_cluster, uint32_t n, uint32_t parents, uint32_t clusters, uint8_t* d_flags) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    BU_TRY(ctx, bu::launch_membership(ctx->stream, d_parent, d_cluster, n, parents, clusters, d_flags));
    return 1;
}

int bu_hip_k_map_gather(bu_hip_context* ctx, const uint32_t* d_table, const uint32_t* d_index, uint32_t n, uint32_t* d_out) {
    if (!ctx) return 0;
    device_guard g(ctx->device);
    BU_TRY(ctx, bu::launch_gather_u32(ctx->stream, d_table, d_index, n, d_out));
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------- f3: k-means codebooks (fast mode)

int bu_hip_kmeans_codebook(bu_hip_context* ctx, int kind, const void* d_keys, const uint64_t* d_weights, const uint32_t* d_goffs, uint32_t n, uint32_t max_clusters,
                           uint32_t n_parents, uint32_t iterations, uint32_t* d_cluster, uint32_t* d_parent, uint32_t* out_clusters, uint32_t* out_parents) {
    if (!ctx) return 0;
    if (!d_keys || !d_cluster || !out_clusters || !n || !max_clusters || (kind == 0 && !d_weights) || (kind != 0 && !d_goffs) || (n_parents && !d_parent)) {
        set_error(ctx, "kmeans_codebook: bad arguments");
        return 0;
    }
    device_guard g(ctx->device);
    const uint32_t k = std::min(max_clusters, n);
    arena& ws = ctx->scratch[3];
    BU_TRY(ctx, ws.reserve(bu::kmeans_workspace_bytes(n, k) + (size_t)k * 8 + 256));
    const bu::kmeans_buffers b = bu::kmeans_carve(ws.p, n, k);
    uint32_t* d_tab = reinterpret_cast<uint32_t*>(static_cast<char*>(ws.p) + bu::kmeans_workspace_bytes(n, k));
    {
        prof_scope ps(ctx, kind ? "kmeans_endpoints" : "kmeans_selectors");
        BU_TRY(ctx, bu::launch_kmeans(ctx->stream, kind, d_keys, d_weights, d_goffs, n, k, iterations, b, d_cluster));
    }
    std::vector<uint64_t> sums((size_t)k * 17);
    std::vector<float> cen((size_t)k * 16);
    {
        mail_fetch f(ctx);
        BU_TRY(ctx, f.add(sums.data(), b.sums, sums.size() * 8));
        BU_TRY(ctx, f.add(cen.data(), b.cen, cen.size() * 4));
        if (!f.wait()) return 0;
    }
    // non-empty clusters, in index order
    std::vector<uint32_t> old_to_new(k, 0), live;
    for (uint32_t c = 0; c < k; c++) if (sums[(size_t)c * 17 + 16]) { old_to_new[c] = (uint32_t)live.size(); live.push_back(c); }
    const uint32_t kl = (uint32_t)live.size();
    *out_clusters = kl;
    uint32_t parents = 0;
    std::vector<uint32_t> parent_of_old(k, 0);
    if (n_parents && kl) {
        // the parent level: weighted k-means over the live clusters' final centres (the true means of their members), a few thousand points, on the host
        const int D = 16;
        std::vector<double> pts((size_t)kl * D), wts(kl);
        for (uint32_t i = 0; i < kl; i++) {
            const uint64_t w = sums[(size_t)live[i] * 17 + 16];
            wts[i] = (double)w;
            for (int d = 0; d < D; d++) pts[(size_t)i * D + d] = (double)sums[(size_t)live[i] * 17 + d] / (double)w;
        }
        const uint32_t P = std::min(n_parents, kl);
        std::vector<double> pc((size_t)P * D);
        for (uint32_t p = 0; p < P; p++) std::memcpy(&pc[(size_t)p * D], &pts[(size_t)(((uint64_t)p * 2 + 1) * kl / (2ull * P)) * D], D * sizeof(double));
        std::vector<uint32_t> owner(kl, 0);
        for (int it = 0; it < 12; it++) {
            for (uint32_t i = 0; i < kl; i++) {
                double bd = 1e300; uint32_t bp = 0;
                for (uint32_t p = 0; p < P; p++) {
                    double dd = 0;
                    for (int d = 0; d < D; d++) { const double t = pts[(size_t)i * D + d] - pc[(size_t)p * D + d]; dd += t * t; }
                    if (dd < bd) { bd = dd; bp = p; }
                }
                owner[i] = bp;
            }
            std::vector<double> acc((size_t)P * D, 0.0), aw(P, 0.0);
            for (uint32_t i = 0; i < kl; i++) { aw[owner[i]] += wts[i]; for (int d = 0; d < D; d++) acc[(size_t)owner[i] * D + d] += wts[i] * pts[(size_t)i * D + d]; }
            for (uint32_t p = 0; p < P; p++) if (aw[p] > 0) for (int d = 0; d < D; d++) pc[(size_t)p * D + d] = acc[(size_t)p * D + d] / aw[p];
        }
        std::vector<int32_t> renum(P, -1);   // parents that own something, in index order
        for (uint32_t i = 0; i < kl; i++) if (renum[owner[i]] < 0) renum[owner[i]] = 0;
        for (uint32_t p = 0; p < P; p++) if (renum[p] == 0) renum[p] = (int32_t)parents++;
        for (uint32_t i = 0; i < kl; i++) parent_of_old[live[i]] = (uint32_t)renum[owner[i]];
    }
    if (out_parents) *out_parents = parents;
    // per distinct vector: parent first (from the raw assignment), then the compacted cluster index in place
    if (n_parents) {
        BU_TRY(ctx, h2d(ctx, d_tab, parent_of_old.data(), (size_t)k * 4));
        BU_TRY(ctx, bu::launch_gather_u32(ctx->stream, d_tab, d_cluster, n, d_parent));
    }
    BU_TRY(ctx, h2d(ctx, d_tab + k, old_to_new.data(), (size_t)k * 4));
    BU_TRY(ctx, bu::launch_gather_u32(ctx->stream, d_tab + k, d_cluster, n, d_cluster));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));
    (void)cen;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------- a8: TSVQ

} // extern "C" (reopened below)

struct bu_tsvq {
    uint32_t dim = 0, n = 0;
    bool packed = false;
    void* rows = nullptr;       // float[n][dim], or uint32[n] when packed
    uint64_t* w64 = nullptr;
    uint32_t* perm[2] = {nullptr, nullptr};   // perm[0]: ONE block of TSVQ_BUFS buffers of n indices each, perm[1] = perm[0] + n (tsvq_bufs.h: the pair gives every kernel base and stride)
    uint8_t* side = nullptr;
    arena nodes, outs;
    arena deep_nodes;           // deep rounds: the node records of generations 1.. (made on the device by k_tsvq_children)
    // the knobs below are copies of the context's bu_hip_tuning at creation (basisu_hip.h): one tree never changes paths half way
    bool force_chained = false; // tsvq_chained_only: never use the exact (integer-reduced) kernel variants (tests compare both)
    // Nodes with at least wide_min members go through the many-workgroup path (tsvq_wide_kernels.hip for packed rows, tsvq_wide6_kernels.hip for 6-float rows).
    uint32_t wide_min = 0;      // 0: off
    uint32_t wide_cov_min = 0;  // batches whose largest node is smaller run the covariance pass chained (tsvq_wide_cov_min)
    int windows = 0; uint32_t dense_min = 257; int poll = 0;
    uint32_t wide_blocks_cap = 0, wide_nodes_cap = 0;
    void* xchg = nullptr; size_t xchg_cap = 0;   // staging of bu_hip_tsvq_exchange_* (multi-GPU)
    void* wide_ws = nullptr; void* wide_packed = nullptr; bu::tsvq_wide_node* wide_nodes = nullptr; bu::tsvq_wide_ctrl* wide_ctrl = nullptr; void* wide_ctrl_raw = nullptr;
    // Pinned staging for the per-round node / result records: hipMemcpyAsync on PAGEABLE host memory followed directly by a
    // kernel on the same stream was observed to let the kernel read the destination before the copy landed (MI355X, ROCm 7.2:
    // tools/tsvq_root_repeat.py, 2 of 10 runs), so nothing on this path hands pageable memory to an asynchronous copy.
    void* pinned = nullptr; size_t pinned_cap = 0;
    // Zero-copy rounds (default; tsvq_zero_copy = 0 switches back to staged copies + hipStreamSynchronize): the one-workgroup kernel reads its node records
    // from, and every split kernel writes its result records to, the page-locked buffer directly; a one-thread kernel behind them raises `round_flag`
    // (system scope) and the host spins on it. That takes two copy launches and a blocking synchronisation out of every round of the tree build.
    bool zero_copy = true;
    uint32_t round_seq = 0;
    bool dbg_rounds = false, dbg_serial = false, dbg_stats = false;   // bu_hip_tuning::debug bits
    hipError_t reserve_pinned(size_t bytes) {
        if (bytes <= pinned_cap) return hipSuccess;
        if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; pinned_cap = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        // coherent (fine-grained) host memory: the zero-copy rounds have kernels write result records and the completion word straight into this buffer while the host polls
        // it; with a non-coherent mapping the word would only become visible when the kernel retires
        hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocCoherent);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostMalloc(&pinned, want, hipHostMallocDefault); zero_copy = false; }
        if (e != hipSuccess) { pinned = nullptr; return e; }
        pinned_cap = want;
        return hipSuccess;
    }
};

static_assert(sizeof(bu_tsvq_root) == sizeof(bu::tsvq_root_out), "layout");
static_assert(sizeof(bu_tsvq_node) == sizeof(bu::tsvq_node_in), "layout");
static_assert(sizeof(bu_tsvq_split) == sizeof(bu::tsvq_split_out), "layout");


extern "C" {

void bu_hip_tsvq_destroy(bu_hip_context* ctx, bu_tsvq* q) {
    if (!ctx || !q) return;
    device_guard g(ctx->device);
    (void)stream_wait(ctx, ctx->stream);
    for (void* p : {q->rows, (void*)q->w64, (void*)q->perm[0], (void*)q->side, q->nodes.p, q->outs.p, q->deep_nodes.p, q->xchg, q->wide_ws, q->wide_packed, (void*)q->wide_nodes, q->wide_ctrl_raw}) if (p) bu_hip_free(ctx, p);
    q->nodes.p = nullptr; q->outs.p = nullptr; q->deep_nodes.p = nullptr;
    if (q->pinned) {  // hand the pinned staging buffer back to the context (keep the larger one)
        if (q->pinned_cap > ctx->tsvq_pinned_cap) { if (ctx->tsvq_pinned) (void)hipHostFree(ctx->tsvq_pinned); ctx->tsvq_pinned = q->pinned; ctx->tsvq_pinned_cap = q->pinned_cap; }
        else (void)hipHostFree(q->pinned);
    }
    delete q;
}

static bu_tsvq* tsvq_create_common(bu_hip_context* ctx, uint32_t dim, bool packed, const void* h_rows, const uint64_t* h_weights, uint32_t n, bu_tsvq_root* out_root,
                                   bool source_on_device = false, const uint64_t* d_endpoint_keys = nullptr, const uint32_t* d_endpoint_goffs = nullptr) {
    if (!ctx || !n || !out_root || (dim != 6 && dim != 16) || (packed && dim != 16)) { if (ctx) set_error(ctx, "tsvq_create: bad arguments"); return nullptr; }
    device_guard g(ctx->device);
    bu_tsvq* q = new (std::nothrow) bu_tsvq();
    if (!q) return nullptr;
    q->dim = dim; q->n = n; q->packed = packed;
    const bu_hip_tuning& tune = ctx->tuning;
    q->force_chained = tune.tsvq_chained_only != 0;
    q->zero_copy = tune.tsvq_zero_copy != 0;
    q->dbg_rounds = (tune.debug & 1) != 0; q->dbg_serial = (tune.debug & 2) != 0; q->dbg_stats = (tune.debug & 4) != 0;
    q->windows = (int)tune.tsvq_windows; q->dense_min = tune.tsvq_dense_min; q->poll = (int)tune.tsvq_poll;
    const size_t row_bytes = packed ? 4 : (size_t)dim * 4;
    auto fail = [&](const char* what) -> bu_tsvq* { set_error(ctx, "tsvq_create: %s", what); bu_hip_tsvq_destroy(ctx, q); return nullptr; };
    if (ctx->tsvq_pinned) { q->pinned = ctx->tsvq_pinned; q->pinned_cap = ctx->tsvq_pinned_cap; ctx->tsvq_pinned = nullptr; ctx->tsvq_pinned_cap = 0; }
    // all device blocks come from (and return to) the context's pool; the node / result records are sized for the largest batch
    // a codebook of cMaxSelectorClusters can ask for, so they never grow
    const size_t rec_cap = (size_t)16384 * std::max(sizeof(bu_tsvq_node), sizeof(bu_tsvq_split));
    q->rows = bu_hip_malloc(ctx, (size_t)n * row_bytes); q->w64 = (uint64_t*)bu_hip_malloc(ctx, (size_t)n * 8);
    q->perm[0] = (uint32_t*)bu_hip_malloc(ctx, (size_t)n * 4 * bu::TSVQ_BUFS); q->perm[1] = q->perm[0] ? q->perm[0] + n : nullptr;
    q->side = (uint8_t*)bu_hip_malloc(ctx, n);
    q->nodes.p = bu_hip_malloc(ctx, rec_cap); q->outs.p = bu_hip_malloc(ctx, rec_cap);
    if (!q->rows || !q->w64 || !q->perm[0] || !q->perm[1] || !q->side || !q->nodes.p || !q->outs.p) return fail("allocation");
    q->nodes.cap = q->outs.cap = rec_cap;
    if (!packed && dim == 6 && !q->force_chained) {
        // the endpoint tree's large nodes through the many-workgroup path for 6-float rows (tsvq_wide6_kernels.hip): tsvq_wide6_min, default 8,192
        // (6,144 / 10,000 / 14,000 side by side on one box: 1.68 / 1.69 / 1.69 ms for the endpoint tree's splits, 2.15 without the path); 0 = off (tests compare both)
        const uint32_t wide_min = tune.tsvq_wide6_min;
        if (wide_min && n >= wide_min && n < (1u << 22)) {
            q->wide_min = wide_min;
            q->wide_nodes_cap = n / wide_min + 1;
            q->wide_blocks_cap = (n + 255) / 256 + q->wide_nodes_cap;
            q->wide_ws = bu_hip_malloc(ctx, bu::tsvq_wide_workspace_bytes(q->wide_blocks_cap));
            q->wide_nodes = (bu::tsvq_wide_node*)bu_hip_malloc(ctx, (size_t)q->wide_nodes_cap * sizeof(bu::tsvq_wide_node));
            q->wide_ctrl_raw = bu_hip_malloc(ctx, (size_t)q->wide_nodes_cap * sizeof(bu::tsvq_wide_ctrl));
            q->wide_ctrl = static_cast<bu::tsvq_wide_ctrl*>(q->wide_ctrl_raw);
            q->wide_packed = bu_hip_malloc(ctx, (size_t)n * 32);   // the list-order copies of the per-member addends: 6 n floats + n doubles
            if (!q->wide_ws || !q->wide_nodes || !q->wide_ctrl || !q->wide_packed) return fail("allocation");
        }
    }
    if (packed && !q->force_chained) {
        const uint32_t wide_min = tune.tsvq_wide_min;   // default 8,192 (16,384 until round 3: the one-workgroup launches of the smaller nodes are the longer of the two concurrent streams)
        if (wide_min && n >= wide_min && n < (1u << 22)) {   // above 2^22 members the binade prediction loses its margin; the chained kernel takes those
            q->wide_min = wide_min;
            q->wide_cov_min = tune.tsvq_wide_cov_min;   // default 98,304 (side by side on one box with the register-composed stretches kernel: 131,072 / 98,304 / 65,536 / 49,152 -> 5.10 / 4.98 / 5.04 / 5.04 ms of many-workgroup rounds per 4096^2 step)
            q->wide_nodes_cap = n / wide_min + 1;
            q->wide_blocks_cap = (n + 255) / 256 + q->wide_nodes_cap;
            q->wide_ws = bu_hip_malloc(ctx, bu::tsvq_wide_workspace_bytes(q->wide_blocks_cap));
            q->wide_nodes = (bu::tsvq_wide_node*)bu_hip_malloc(ctx, (size_t)q->wide_nodes_cap * sizeof(bu::tsvq_wide_node));
            q->wide_ctrl_raw = bu_hip_malloc(ctx, (size_t)q->wide_nodes_cap * sizeof(bu::tsvq_wide_ctrl));
            q->wide_ctrl = static_cast<bu::tsvq_wide_ctrl*>(q->wide_ctrl_raw);
            q->wide_packed = bu_hip_malloc(ctx, (size_t)n * 8);
            if (!q->wide_ws || !q->wide_packed || !q->wide_nodes || !q->wide_ctrl) return fail("allocation");
        }
    }
    if (d_endpoint_keys) {   // the rows are made on the device from the de-duplication's keys (bu_hip_k_unique_endpoint_vectors)
        if (bu::launch_endpoint_rows(ctx->stream, d_endpoint_keys, d_endpoint_goffs, n, static_cast<float*>(q->rows), q->w64) != hipSuccess) return fail("endpoint rows");
    } else if (source_on_device) {  // stream-ordered device copies: the vectors were produced on this context's stream
        if (hipMemcpyAsync(q->rows, h_rows, (size_t)n * row_bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(q->w64, h_weights, (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return fail("device copy");
    } else if (stream_wait(ctx, ctx->stream) != hipSuccess || hipMemcpy(q->rows, h_rows, (size_t)n * row_bytes, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(q->w64, h_weights, (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return fail("upload");  // blocking copies (the sources are pageable, see bu_tsvq::pinned); the root kernel below needs both anyway
    // the page-locked buffer here: [0] the root's node record (many-workgroup variant), [256] the root record the kernels produce, [768] the completion word
    constexpr size_t ROOT_AT = 256, FLAG_AT = 768;
    static_assert(sizeof(bu::tsvq_wide_node) <= ROOT_AT && ROOT_AT + sizeof(bu_tsvq_root) <= FLAG_AT, "layout of the root's page-locked records");
    if (q->reserve_pinned(1024) != hipSuccess) return fail("pinned allocation");
    // Zero-copy (as the rounds, tsvq_split_impl): the kernels read the node record from and write the root record into the page-locked buffer and a last one-thread kernel
    // stores a word there that this thread looks at -- no copy commands, no hipStreamSynchronize between the root and the first round.
    char* d_pinned = nullptr;
    if (q->zero_copy && hipHostGetDevicePointer(reinterpret_cast<void**>(&d_pinned), q->pinned, 0) != hipSuccess) { (void)hipGetLastError(); d_pinned = nullptr; }
    bu::tsvq_root_out* d_root = d_pinned ? reinterpret_cast<bu::tsvq_root_out*>(d_pinned + ROOT_AT) : static_cast<bu::tsvq_root_out*>(q->outs.p);
    const bu_tsvq_root* h_root = reinterpret_cast<const bu_tsvq_root*>(static_cast<const char*>(q->pinned) + (d_pinned ? ROOT_AT : 0));
    volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(static_cast<char*>(q->pinned) + FLAG_AT);
    // many-workgroup variant first where it applies, then the exact (integer-reduced) one-workgroup variant; a record flagged
    // pad == 1 left the exact range -> next variant, the chained one last
    for (int attempt = q->wide_min ? -1 : 0; attempt < 2; attempt++) {
        const bool exact = packed && attempt == 0 && !q->force_chained;
        if (d_pinned) { *flag = 0; }
        if (attempt < 0) {
            bu::tsvq_wide_node wn; std::memset(&wn, 0, sizeof(wn));
            wn.count = n; wn.n_blocks = (n + 255) / 256;
            std::memcpy(q->pinned, &wn, sizeof(wn));
            if (d_pinned) {
                __atomic_thread_fence(__ATOMIC_SEQ_CST);
                if (bu::launch_tsvq_wide_prologue(ctx->stream, reinterpret_cast<const bu::tsvq_wide_node*>(d_pinned), q->wide_nodes, q->wide_ctrl, 1) != hipSuccess) return fail("root upload");
            } else if (hipMemcpyAsync(q->wide_nodes, q->pinned, sizeof(wn), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return fail("root upload");
            prof_scope ps(ctx, packed ? "tsvq_root_packed16" : "tsvq_root_float6");
            if (!packed) {
                if (bu::launch_tsvq_wide6_root(ctx->stream, static_cast<const float*>(q->rows), q->w64, n, q->perm[0], q->side, q->wide_nodes, q->wide_ctrl, q->wide_ws, wn.n_blocks,
                                               d_root, static_cast<float*>(q->wide_packed),
                                               reinterpret_cast<double*>(static_cast<char*>(q->wide_packed) + (size_t)n * 24), d_pinned != nullptr) != hipSuccess) return fail("wide root launch");
            } else
            if (bu::launch_tsvq_wide_root(ctx->stream, static_cast<const uint32_t*>(q->rows), q->w64, n, q->perm[0], q->wide_nodes, q->wide_ctrl, q->wide_ws, wn.n_blocks,
                                          d_root, q->windows, d_pinned != nullptr) != hipSuccess) return fail("wide root launch");
        } else {
            if (d_pinned) __atomic_thread_fence(__ATOMIC_SEQ_CST);
            prof_scope ps(ctx, packed ? "tsvq_root_packed16" : "tsvq_root_float6");
            if (bu::launch_tsvq_root(ctx->stream, (int)dim, packed, exact, q->rows, q->w64, n, q->perm[0], d_root) != hipSuccess) return fail("root launch");
        }
        if (d_pinned) {
            const uint32_t seq = ++q->round_seq ? q->round_seq : ++q->round_seq;   // never 0
            if (bu::launch_tsvq_signal(ctx->stream, reinterpret_cast<uint32_t*>(d_pinned + FLAG_AT), seq) != hipSuccess || !wait_flag(ctx, q->poll, flag, seq)) return fail("root wait");
        } else if (hipMemcpyAsync(q->pinned, q->outs.p, sizeof(bu_tsvq_root), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || stream_wait(ctx, ctx->stream) != hipSuccess)
            return fail("root download");
        if ((attempt >= 0 && !exact) || h_root->pad == 0) break;
    }
    std::memcpy(out_root, h_root, sizeof(bu_tsvq_root));
    return q;
}

bu_tsvq* bu_hip_tsvq_create(bu_hip_context* ctx, uint32_t dim, const float* h_rows, const uint64_t* h_weights, uint32_t n, bu_tsvq_root* out_root) {
    return tsvq_create_common(ctx, dim, false, h_rows, h_weights, n, out_root);
}

bu_tsvq* bu_hip_tsvq_create_packed16(bu_hip_context* ctx, const uint32_t* h_keys, const uint64_t* h_weights, uint32_t n, bu_tsvq_root* out_root) {
    return tsvq_create_common(ctx, 16, true, h_keys, h_weights, n, out_root);
}

bu_tsvq* bu_hip_tsvq_create_packed16_device(bu_hip_context* ctx, const uint32_t* d_keys, const uint64_t* d_weights, uint32_t n, bu_tsvq_root* out_root) {
    return tsvq_create_common(ctx, 16, true, d_keys, d_weights, n, out_root, true);
}

bu_tsvq* bu_hip_tsvq_create_endpoint_device(bu_hip_context* ctx, const uint64_t* d_unique_keys, const uint32_t* d_group_offsets, uint32_t n, bu_tsvq_root* out_root) {
    if (ctx && (!d_unique_keys || !d_group_offsets)) { set_error(ctx, "tsvq_create_endpoint_device: null pointer"); return nullptr; }
    return tsvq_create_common(ctx, 6, false, nullptr, nullptr, n, out_root, true, d_unique_keys, d_group_offsets);
}

int bu_hip_k_unique_endpoint_vectors(bu_hip_context* ctx, const void* d_etc1_blocks, uint32_t n_blocks, uint32_t* d_sorted_block_idx, uint64_t* d_unique_keys,
                                     uint32_t* d_group_offsets, uint32_t* out_unique) {
    if (!ctx) return 0;
    if (!out_unique || (n_blocks && (!d_etc1_blocks || !d_sorted_block_idx || !d_unique_keys || !d_group_offsets))) { set_error(ctx, "unique_endpoint_vectors: null pointer"); return 0; }
    *out_unique = 0;
    if (!n_blocks) return 1;
    device_guard g(ctx->device);
    arena& ws = ctx->scratch[4];
    BU_TRY(ctx, ws.reserve(bu::unique_endpoint_vectors_workspace_bytes(n_blocks)));
    uint32_t* d_n = nullptr;
    {
        prof_scope ps(ctx, "unique_endpoint_vectors");
        BU_TRY(ctx, bu::launch_unique_endpoint_vectors(ctx->stream, d_etc1_blocks, n_blocks, ws.p, d_sorted_block_idx, d_unique_keys, d_group_offsets, &d_n));
    }
    if (!fetch(ctx, out_unique, d_n, 4)) return 0;
    return 1;
}

int bu_hip_k_unique_selector_vectors(bu_hip_context* ctx, const void* d_enc_blocks, const uint64_t* d_weights, uint32_t n_blocks, uint32_t* d_sorted_block_idx,
                                     uint32_t* d_unique_keys, uint64_t* d_unique_weights, uint32_t* d_group_offsets, uint32_t* out_unique) {
    if (!ctx) return 0;
    if (!out_unique || (n_blocks && (!d_enc_blocks || !d_weights || !d_sorted_block_idx || !d_unique_keys || !d_unique_weights || !d_group_offsets))) {
        set_error(ctx, "unique_selector_vectors: null pointer");
        return 0;
    }
    *out_unique = 0;
    if (!n_blocks) return 1;
    device_guard g(ctx->device);
    arena& ws = ctx->scratch[4];
    BU_TRY(ctx, ws.reserve(bu::unique_selector_vectors_workspace_bytes(n_blocks)));
    uint32_t* d_n = nullptr;
    {
        prof_scope ps(ctx, "unique_selector_vectors");
        BU_TRY(ctx, bu::launch_unique_selector_vectors(ctx->stream, d_enc_blocks, d_weights, n_blocks, ws.p, d_sorted_block_idx, d_unique_keys, d_unique_weights,
                                                        d_group_offsets, &d_n));
    }
    if (!fetch(ctx, out_unique, d_n, 4)) return 0;
    return 1;
}

// One round of splits. levels > 0 (deep round, zero-copy rounds only): the one-workgroup nodes' children, grandchildren, ... are split in the same round trip --
// every generation's node records are made on the device from the results of the one before (k_tsvq_children), so `levels` more launches follow the batch's own
// without the host. h_deep: generation g (1..levels) of batch node i, path p (the sides taken, first step in the top bit) at h_deep[n_nodes * (2^g - 2) + i * 2^g + p];
// ok == 3 = not attempted (the parent's split failed or went through the many-workgroup passes, one member, variance below the floor in h_nodes[i].pad).
static int tsvq_split_impl(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_split* h_out, uint32_t levels, bu_tsvq_split* h_deep) {
    if (!ctx || !q) return 0;
    if (!n_nodes) return 1;
    if (levels > bu::TSVQ_MAX_DEEP_LEVELS) levels = bu::TSVQ_MAX_DEEP_LEVELS;   // tsvq_bufs.h: a write must not reach a list that may still become a leaf
    const uint32_t h_deep_levels = h_deep ? levels : 0;   // what the caller's array is laid out for (the round may attempt fewer)
    device_guard g(ctx->device);
    const bool round_stats = q->dbg_rounds;   // development aid: one line per round on stderr
    const auto round_t0 = std::chrono::steady_clock::now();
    if ((size_t)n_nodes * sizeof(bu_tsvq_node) > q->nodes.cap || (size_t)n_nodes * sizeof(bu_tsvq_split) > q->outs.cap) { set_error(ctx, "tsvq_split: batch of %u nodes exceeds the record buffers", n_nodes); return 0; }
    // Large nodes go through the many-workgroup path, the rest one workgroup each; both write one result array
    // (narrow records first, in batch order, then the wide ones).
    std::vector<uint32_t> order; order.reserve(n_nodes);
    uint32_t n_wide = 0, wide_blocks = 0, wide_max_count = 0;
    uint64_t wide_max_weight = 0;   // chain addends are value (0..3) x weight: 3 x a node's weight bounds every side chain's total
    if (q->wide_min) {
        std::vector<uint32_t> wide;
        for (uint32_t i = 0; i < n_nodes; i++) {
            const uint32_t nb = (h_nodes[i].count + 255) / 256;
            if (h_nodes[i].count >= q->wide_min && (q->packed || h_nodes[i].weight < (1ull << 52)) && wide.size() < q->wide_nodes_cap && wide_blocks + nb <= q->wide_blocks_cap) { wide.push_back(i); wide_blocks += nb; wide_max_count = std::max(wide_max_count, h_nodes[i].count); wide_max_weight = std::max<uint64_t>(wide_max_weight, h_nodes[i].weight); }
            else order.push_back(i);
        }
        n_wide = (uint32_t)wide.size();
        order.insert(order.end(), wide.begin(), wide.end());
    } else for (uint32_t i = 0; i < n_nodes; i++) order.push_back(i);
    const uint32_t n_narrow = n_nodes - n_wide;
    const size_t in_bytes = (size_t)n_narrow * sizeof(bu_tsvq_node), wide_bytes = (size_t)n_wide * sizeof(bu::tsvq_wide_node), out_bytes = (size_t)n_nodes * sizeof(bu_tsvq_split);
    const size_t wide_at = (in_bytes + 63) & ~(size_t)63;
    const bool zero_copy = q->zero_copy;
    // staged: the result records come back over the node records; zero-copy: the kernels write them while others still read their nodes, so they get their own place
    if (!zero_copy || !n_narrow || q->dbg_serial || !h_deep) levels = 0;
    while (levels && (size_t)n_narrow * ((2u << levels) - 2u) * sizeof(bu_tsvq_split) > ((size_t)64 << 20)) levels--;   // (never in practice: 64 MiB of records)
    const size_t deep_recs = (size_t)n_narrow * ((2u << levels) - 2u);   // 2 + 4 + ... + 2^levels per one-workgroup node
    const size_t out_at = zero_copy ? ((wide_at + wide_bytes + 63) & ~(size_t)63) : 0, deep_at = out_at + out_bytes,
                 flag_at = (deep_at + deep_recs * sizeof(bu_tsvq_split) + 63) & ~(size_t)63;
    BU_TRY(ctx, q->reserve_pinned(std::max(wide_at + wide_bytes, flag_at + 128)));   // the round's flag
    if (deep_recs * sizeof(bu_tsvq_node) > q->deep_nodes.cap) {
        if (q->deep_nodes.p) { BU_TRY(ctx, stream_wait(ctx, ctx->stream)); bu_hip_free(ctx, q->deep_nodes.p); q->deep_nodes.p = nullptr; q->deep_nodes.cap = 0; }
        const size_t want = deep_recs * sizeof(bu_tsvq_node) * 2;
        q->deep_nodes.p = bu_hip_malloc(ctx, want);
        if (!q->deep_nodes.p) { set_error(ctx, "tsvq_split: allocation of %zu bytes of node records failed", want); return 0; }
        q->deep_nodes.cap = want;
    }
    char* d_pinned = nullptr;   // the page-locked buffer as the device addresses it
    if (zero_copy) BU_TRY(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&d_pinned), q->pinned, 0));
    volatile uint32_t* round_flag = reinterpret_cast<volatile uint32_t*>(static_cast<char*>(q->pinned) + flag_at);
    const bu::tsvq_node_in* d_nodes_in = zero_copy ? reinterpret_cast<const bu::tsvq_node_in*>(d_pinned) : static_cast<const bu::tsvq_node_in*>(q->nodes.p);
    bu::tsvq_split_out* d_outs = zero_copy ? reinterpret_cast<bu::tsvq_split_out*>(d_pinned + out_at) : static_cast<bu::tsvq_split_out*>(q->outs.p);
    {
        bu_tsvq_node* pn = static_cast<bu_tsvq_node*>(q->pinned);
        for (uint32_t i = 0; i < n_narrow; i++) pn[i] = h_nodes[order[i]];
        bu::tsvq_wide_node* pw = reinterpret_cast<bu::tsvq_wide_node*>(static_cast<char*>(q->pinned) + wide_at);
        uint32_t first = 0;
        for (uint32_t i = 0; i < n_wide; i++) {
            const bu_tsvq_node& s = h_nodes[order[n_narrow + i]];
            bu::tsvq_wide_node w; std::memset(&w, 0, sizeof(w));
            w.buf = s.buf; w.start = s.start; w.count = s.count; w.out_index = n_narrow + i; w.first_block = first; w.n_blocks = (s.count + 255) / 256; w.weight = s.weight;
            std::memcpy(w.origin, s.origin, sizeof(w.origin));
            first += w.n_blocks;
            pw[i] = w;
        }
    }
    if (zero_copy) { *round_flag = 0; __atomic_thread_fence(__ATOMIC_SEQ_CST); }
    if (n_narrow && !zero_copy) BU_TRY(ctx, hipMemcpyAsync(q->nodes.p, q->pinned, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    // the wide nodes' records + their cleared state: one kernel that reads the page-locked records, where the device can address them (otherwise a copy here and a fill in the launcher)
    const bool wide_prologue = n_wide && zero_copy;
    if (wide_prologue) BU_TRY(ctx, bu::launch_tsvq_wide_prologue(ctx->stream, reinterpret_cast<const bu::tsvq_wide_node*>(d_pinned + wide_at), q->wide_nodes, q->wide_ctrl, n_wide));
    else if (n_wide) BU_TRY(ctx, hipMemcpyAsync(q->wide_nodes, static_cast<char*>(q->pinned) + wide_at, wide_bytes, hipMemcpyHostToDevice, ctx->stream));
    const bool exact = q->packed && !q->force_chained;
    // The two kinds of node of a round do not touch each other's data: when both are present the one-workgroup kernel runs on the side
    // stream, under the many small launches of the wide path. (Not while kernels are being timed one by one.)
    bool narrow_on_side = n_wide && n_narrow && ctx->profiling != 1 && !q->dbg_serial;   // (not while the rounds' kernels are being timed one region after the other)
    if (narrow_on_side && !ensure_side_stream(ctx)) narrow_on_side = false;
    // deep round: generation g's records from generation g - 1's results, then its splits, on the stream the batch's own one-workgroup launch went to
    auto deep_generations = [&](hipStream_t st) -> bool {
        const bu::tsvq_node_in* parents = d_nodes_in;
        const bu::tsvq_split_out* parent_outs = d_outs;
        bu::tsvq_node_in* children = static_cast<bu::tsvq_node_in*>(q->deep_nodes.p);
        bu::tsvq_split_out* child_outs = reinterpret_cast<bu::tsvq_split_out*>(d_pinned + deep_at);
        uint32_t n_parents = n_narrow;
        for (uint32_t gen = 1; gen <= levels; gen++) {
            if (bu::launch_tsvq_children(st, parents, parent_outs, n_parents, children, child_outs) != hipSuccess ||
                bu::launch_tsvq_split(st, (int)q->dim, q->packed, exact, q->rows, q->w64, q->perm[0], q->perm[1], q->side, children, 2 * n_parents, child_outs, q->dense_min) != hipSuccess) {
                set_error(ctx, "tsvq_split: deep generation %u: %s", gen, hipGetErrorString(hipGetLastError()));
                return false;
            }
            parents = children; parent_outs = child_outs;
            children += 2 * (size_t)n_parents; child_outs += 2 * (size_t)n_parents;
            n_parents *= 2;
        }
        return true;
    };
    // From the fork on, every early return must wait for the side stream first: the caller's guard destroys q (its buffers go back to the
    // pool without a synchronisation) while the one-workgroup kernel may still be running on them.
    struct side_joiner { hipStream_t s; bool armed; ~side_joiner() { if (armed) (void)hipStreamSynchronize(s); } } side_join_guard{ctx->side_stream, false};
    if (narrow_on_side) {
        side_join_guard.s = ctx->side_stream; side_join_guard.armed = true;
        BU_TRY(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
        BU_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
        BU_TRY(ctx, bu::launch_tsvq_split(ctx->side_stream, (int)q->dim, q->packed, exact, q->rows, q->w64, q->perm[0], q->perm[1], q->side, d_nodes_in, n_narrow, d_outs, q->dense_min));
        if (levels && !deep_generations(ctx->side_stream)) return 0;
        BU_TRY(ctx, hipEventRecord(ctx->side_join, ctx->side_stream));
    }
    if (n_wide) {
        prof_scope ps(ctx, q->packed ? "tsvq_split_packed16_wide" : "tsvq_split_float6_wide");
        if (!q->packed)
            BU_TRY(ctx, bu::launch_tsvq_wide6_split(ctx->stream, static_cast<const float*>(q->rows), q->w64, q->n, q->perm[0], q->perm[1], q->side, q->wide_nodes, n_wide, q->wide_ctrl, q->wide_ws,
                                                    wide_blocks, d_outs, static_cast<float*>(q->wide_packed), reinterpret_cast<double*>(static_cast<char*>(q->wide_packed) + (size_t)q->n * 24), wide_prologue));
        else
        BU_TRY(ctx, bu::launch_tsvq_wide_split(ctx->stream, static_cast<const uint32_t*>(q->rows), q->w64, q->perm[0], q->perm[1], q->side, q->wide_packed, q->wide_nodes, n_wide, q->wide_ctrl,
                                               q->wide_ws, wide_blocks, d_outs, wide_max_count < q->wide_cov_min,
                                               wide_max_weight * 3ull < (1ull << 24), q->windows, wide_prologue));
    }
    if (n_wide && q->dbg_stats && q->packed) {   // development aid: how the last pass's walks went, per wide node
        std::vector<bu::tsvq_wide_ctrl> hc(n_wide);
        if (d2h_pageable(ctx, hc.data(), q->wide_ctrl, hc.size() * sizeof(bu::tsvq_wide_ctrl)) == hipSuccess && stream_wait(ctx, ctx->stream) == hipSuccess)
            for (uint32_t i = 0; i < n_wide; i++) {
                uint32_t ms = 0, mr = 0, ts = 0, tr = 0, ex = 0;
                for (int c = 0; c < 32; c++) { ms = std::max<uint32_t>(ms, hc[i].stat_scans[c]); mr = std::max<uint32_t>(mr, hc[i].stat_raw[c]); ts += hc[i].stat_scans[c]; tr += hc[i].stat_raw[c]; ex += hc[i].exact[c]; }
                std::fprintf(stderr, "[tsvq stats] wide node %u: count %u blocks %u iter %d | last pass, 32 side chains: scans max %u avg %.1f, raw blocks max %u avg %.1f, exact chains %u\n",
                             i, h_nodes[order[n_narrow + i]].count, (h_nodes[order[n_narrow + i]].count + 255) / 256, hc[i].iter, ms, ts / 32.0, mr, tr / 32.0, ex);
                uint32_t cms = 0, cmr = 0, cts = 0, ctr = 0, diag_r = 0; int x = 0, y = 0;
                for (int c = 0; c < 136; c++) {
                    cms = std::max<uint32_t>(cms, hc[i].stat_cov_scans[c]); cmr = std::max<uint32_t>(cmr, hc[i].stat_cov_raw[c]); cts += hc[i].stat_cov_scans[c]; ctr += hc[i].stat_cov_raw[c];
                    if (x == y) diag_r += hc[i].stat_cov_raw[c];
                    if (++y == 16) { x++; y = x; }
                }
                if (cts) std::fprintf(stderr, "[tsvq stats]      covariance pass, 136 chains: scans max %u avg %.1f, raw blocks max %u avg %.1f (diagonal avg %.1f)\n", cms, cts / 136.0, cmr, ctr / 136.0, diag_r / 16.0);
            }
    }
    if (narrow_on_side) { BU_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_join, 0)); side_join_guard.armed = false; }
    else if (n_narrow) {
        prof_scope ps(ctx, q->packed ? "tsvq_split_packed16" : "tsvq_split_float6");
        BU_TRY(ctx, bu::launch_tsvq_split(ctx->stream, (int)q->dim, q->packed, exact, q->rows, q->w64, q->perm[0], q->perm[1], q->side, d_nodes_in, n_narrow, d_outs, q->dense_min));
        if (levels && !deep_generations(ctx->stream)) return 0;
    }
    if (zero_copy) {
        const uint32_t seq = ++q->round_seq ? q->round_seq : ++q->round_seq;   // never 0
        BU_TRY(ctx, bu::launch_tsvq_signal(ctx->stream, reinterpret_cast<uint32_t*>(d_pinned + flag_at), seq));
        if (!wait_flag(ctx, q->poll, round_flag, seq)) return 0;
    } else {
        BU_TRY(ctx, hipMemcpyAsync(q->pinned, q->outs.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        BU_TRY(ctx, stream_wait(ctx, ctx->stream));
    }
    if (round_stats) {
        uint32_t mx = 0; uint64_t tot = 0;
        for (uint32_t i = 0; i < n_nodes; i++) { mx = std::max(mx, h_nodes[i].count); tot += h_nodes[i].count; }
        std::fprintf(stderr, "[tsvq round] dim %u: %u nodes (%u wide), largest %u, members %llu: %.0f us\n", q->dim, n_nodes, n_wide, mx, (unsigned long long)tot,
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - round_t0).count());
    }
    {
        const bu_tsvq_split* po = reinterpret_cast<const bu_tsvq_split*>(static_cast<const char*>(q->pinned) + out_at);
        for (uint32_t i = 0; i < n_nodes; i++) h_out[order[i]] = po[i];
    }
    if (h_deep) {   // generation g of batch node i, path p: h_deep[n_nodes * (2^g - 2) + i * 2^g + p]; here generation g of one-workgroup node j sits at pd[n_narrow * (2^g - 2) + j * 2^g + p]
        const bu_tsvq_split* pd = reinterpret_cast<const bu_tsvq_split*>(static_cast<const char*>(q->pinned) + deep_at);
        for (uint32_t gen = 1; gen <= h_deep_levels; gen++) {
            bu_tsvq_split* dst = h_deep + (size_t)n_nodes * ((1u << gen) - 2u);
            const uint32_t w = 1u << gen;
            if (gen > levels) { for (size_t k = 0; k < (size_t)n_nodes * w; k++) dst[k].ok = 3; continue; }
            const bu_tsvq_split* src = pd + (size_t)n_narrow * (w - 2u);
            for (uint32_t j = 0; j < n_nodes; j++) {
                bu_tsvq_split* d = dst + (size_t)order[j] * w;
                if (j >= n_narrow) { for (uint32_t p2 = 0; p2 < w; p2++) d[p2].ok = 3; continue; }
                for (uint32_t p2 = 0; p2 < w; p2++) {
                    const bu_tsvq_split& r = src[(size_t)j * w + p2];
                    if (r.ok == 1 || r.ok == 0) d[p2] = r; else d[p2].ok = 3;   // (ok == 2, data outside the exact kernel's range: left to a later round, which has the redo path)
                }
            }
        }
    }
    if (exact || n_wide) { // nodes whose data left the exact range, or that a wide path handed back (ok == 2), go through the one-workgroup kernel: packed wide ones through its exact variant first
        for (int attempt = exact ? 0 : 1; attempt < 2; attempt++) {
            std::vector<uint32_t> redo;
            for (uint32_t i = 0; i < n_nodes; i++) if (h_out[i].ok == 2) redo.push_back(i);
            if (redo.empty()) break;
            bu_tsvq_node* pn = static_cast<bu_tsvq_node*>(q->pinned);
            for (size_t j = 0; j < redo.size(); j++) pn[j] = h_nodes[redo[j]];
            BU_TRY(ctx, hipMemcpyAsync(q->nodes.p, q->pinned, redo.size() * sizeof(bu_tsvq_node), hipMemcpyHostToDevice, ctx->stream));
            {
                prof_scope ps(ctx, q->packed ? "tsvq_split_packed16" : "tsvq_split_float6");
                BU_TRY(ctx, bu::launch_tsvq_split(ctx->stream, (int)q->dim, q->packed, exact && attempt == 0 && n_wide != 0, q->rows, q->w64, q->perm[0], q->perm[1], q->side,
                                                  static_cast<const bu::tsvq_node_in*>(q->nodes.p), (uint32_t)redo.size(), static_cast<bu::tsvq_split_out*>(q->outs.p), q->dense_min));
            }
            BU_TRY(ctx, hipMemcpyAsync(q->pinned, q->outs.p, redo.size() * sizeof(bu_tsvq_split), hipMemcpyDeviceToHost, ctx->stream));
            BU_TRY(ctx, stream_wait(ctx, ctx->stream));
            const bu_tsvq_split* po = static_cast<const bu_tsvq_split*>(q->pinned);
            for (size_t j = 0; j < redo.size(); j++) h_out[redo[j]] = po[j];
        }
    }
    return 1;
}

int bu_hip_tsvq_split(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_split* h_out) {
    return tsvq_split_impl(ctx, q, h_nodes, n_nodes, h_out, 0, nullptr);
}

int bu_hip_tsvq_split_deep(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_split* h_out, uint32_t levels, bu_tsvq_split* h_deep) {
    if (levels && !h_deep) { if (ctx) set_error(ctx, "tsvq_split_deep: no array for the deeper generations"); return 0; }
    if (levels > bu::TSVQ_MAX_DEEP_LEVELS) { if (ctx) set_error(ctx, "tsvq_split_deep: %u levels (at most BU_TSVQ_BUFFERS - 2)", levels); return 0; }
    return tsvq_split_impl(ctx, q, h_nodes, n_nodes, h_out, levels, h_deep);
}

// prepare_root (enc.h:1708-1735) of member spans: what a tree_vector_quant whose training set is that span, in list order, starts from --
// the roots of the T independent trees of generate_hierarchical_codebook_threaded_internal (enc.h:2137-2152).
int bu_hip_tsvq_roots(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_root* h_out) {
    if (!ctx || !q || (n_nodes && (!h_nodes || !h_out))) return 0;
    if (!n_nodes) return 1;
    device_guard g(ctx->device);
    if ((size_t)n_nodes * sizeof(bu_tsvq_node) > q->nodes.cap) { set_error(ctx, "tsvq_roots: %u spans exceed the record buffer", n_nodes); return 0; }
    for (uint32_t i = 0; i < n_nodes; i++)
        if (h_nodes[i].buf >= bu::TSVQ_BUFS || !h_nodes[i].count || (uint64_t)h_nodes[i].start + h_nodes[i].count > q->n) { set_error(ctx, "tsvq_roots: span outside the training set"); return 0; }
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));   // the pinned staging buffer may still feed an earlier copy
    BU_TRY(ctx, q->reserve_pinned((size_t)n_nodes * std::max(std::max(sizeof(bu_tsvq_node), sizeof(bu_tsvq_root)), sizeof(bu::tsvq_wide_node))));
    std::vector<uint32_t> todo;
    // large spans of packed rows: the many-workgroup root pass (one batch); a record flagged pad == 1 left its exact range -> the one-workgroup kernels below
    if (q->wide_min && q->packed) {
        std::vector<uint32_t> wide;
        uint32_t blocks = 0;
        for (uint32_t i = 0; i < n_nodes; i++) {
            const uint32_t nb = (h_nodes[i].count + 255) / 256;
            if (h_nodes[i].count >= q->wide_min && wide.size() < q->wide_nodes_cap && blocks + nb <= q->wide_blocks_cap) { wide.push_back(i); blocks += nb; }
            else todo.push_back(i);
        }
        if (!wide.empty()) {
            bu::tsvq_wide_node* pw = static_cast<bu::tsvq_wide_node*>(q->pinned);
            uint32_t first = 0;
            for (size_t j = 0; j < wide.size(); j++) {
                const bu_tsvq_node& sp = h_nodes[wide[j]];
                bu::tsvq_wide_node w; std::memset(&w, 0, sizeof(w));
                w.buf = sp.buf; w.start = sp.start; w.count = sp.count; w.out_index = (uint32_t)j; w.first_block = first; w.n_blocks = (sp.count + 255) / 256;
                first += w.n_blocks;
                pw[j] = w;
            }
            BU_TRY(ctx, hipMemcpyAsync(q->wide_nodes, q->pinned, wide.size() * sizeof(bu::tsvq_wide_node), hipMemcpyHostToDevice, ctx->stream));
            {
                prof_scope ps(ctx, "tsvq_root_packed16");
                BU_TRY(ctx, bu::launch_tsvq_wide_span_roots(ctx->stream, static_cast<const uint32_t*>(q->rows), q->w64, q->perm[0], q->perm[1], q->wide_packed, q->wide_nodes,
                                                            (uint32_t)wide.size(), q->wide_ctrl, q->wide_ws, blocks, static_cast<bu::tsvq_root_out*>(q->outs.p), q->windows));
            }
            BU_TRY(ctx, stream_wait(ctx, ctx->stream));   // the node records were read from the pinned buffer the results come back to
            BU_TRY(ctx, hipMemcpyAsync(q->pinned, q->outs.p, wide.size() * sizeof(bu_tsvq_root), hipMemcpyDeviceToHost, ctx->stream));
            BU_TRY(ctx, stream_wait(ctx, ctx->stream));
            const bu_tsvq_root* po = static_cast<const bu_tsvq_root*>(q->pinned);
            for (size_t j = 0; j < wide.size(); j++) {
                if (po[j].pad) todo.push_back(wide[j]); else h_out[wide[j]] = po[j];
            }
            std::sort(todo.begin(), todo.end());
        }
    } else {
        todo.resize(n_nodes);
        for (uint32_t i = 0; i < n_nodes; i++) todo[i] = i;
    }
    // the exact (integer-reduced) variant first where it applies; a record flagged pad == 1 left the exact range -> the chained one
    for (int attempt = (q->packed && !q->force_chained) ? 0 : 1; attempt < 2 && !todo.empty(); attempt++) {
        bu_tsvq_node* pn = static_cast<bu_tsvq_node*>(q->pinned);
        for (size_t j = 0; j < todo.size(); j++) pn[j] = h_nodes[todo[j]];
        BU_TRY(ctx, hipMemcpyAsync(q->nodes.p, q->pinned, todo.size() * sizeof(bu_tsvq_node), hipMemcpyHostToDevice, ctx->stream));
        {
            prof_scope ps(ctx, q->packed ? "tsvq_root_packed16" : "tsvq_root_float6");
            BU_TRY(ctx, bu::launch_tsvq_span_roots(ctx->stream, (int)q->dim, q->packed, attempt == 0, q->rows, q->w64, q->perm[0], q->perm[1],
                                                   static_cast<const bu::tsvq_node_in*>(q->nodes.p), (uint32_t)todo.size(), static_cast<bu::tsvq_root_out*>(q->outs.p)));
        }
        BU_TRY(ctx, stream_wait(ctx, ctx->stream));
        BU_TRY(ctx, hipMemcpyAsync(q->pinned, q->outs.p, todo.size() * sizeof(bu_tsvq_root), hipMemcpyDeviceToHost, ctx->stream));
        BU_TRY(ctx, stream_wait(ctx, ctx->stream));
        const bu_tsvq_root* po = static_cast<const bu_tsvq_root*>(q->pinned);
        std::vector<uint32_t> redo;
        for (size_t j = 0; j < todo.size(); j++) {
            if (attempt == 0 && po[j].pad) redo.push_back(todo[j]);
            else h_out[todo[j]] = po[j];
        }
        todo.swap(redo);
    }
    return 1;
}

static_assert(sizeof(bu_tsvq_span) == sizeof(bu::bk_span), "layout");
int bu_hip_tsvq_scatter_spans(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_span* h_spans, uint32_t n_spans, uint32_t* d_out) {
    if (!ctx || !q || (n_spans && (!h_spans || !d_out))) return 0;
    if (!n_spans) return 1;
    device_guard g(ctx->device);
    const size_t bytes = (size_t)n_spans * sizeof(bu_tsvq_span);
    if (bytes > q->nodes.cap) { set_error(ctx, "tsvq_scatter_spans: %u spans exceed the record buffer", n_spans); return 0; }
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));   // the pinned staging buffer may still feed an earlier copy
    BU_TRY(ctx, q->reserve_pinned(bytes));
    std::memcpy(q->pinned, h_spans, bytes);
    BU_TRY(ctx, hipMemcpyAsync(q->nodes.p, q->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    BU_TRY(ctx, bu::launch_scatter_spans(ctx->stream, q->perm[0], q->perm[1], static_cast<const bu::bk_span*>(q->nodes.p), n_spans, d_out));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));   // q may be destroyed (and the pinned buffer recycled) right after
    return 1;
}

int bu_hip_tsvq_finish_spans(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_span* h_spans, uint32_t n_spans, uint32_t* d_leaf_of, uint32_t* d_parent_of, const uint32_t* d_group_offsets,
                             uint32_t* d_first_pos, uint32_t* d_sizes) {
    if (!ctx || !q || (n_spans && (!h_spans || !d_leaf_of)) || (d_group_offsets && (!d_first_pos || !d_sizes))) return 0;
    if (!n_spans) return 1;
    device_guard g(ctx->device);
    const size_t bytes = (size_t)n_spans * sizeof(bu_tsvq_span);
    if (bytes > q->nodes.cap) { set_error(ctx, "tsvq_finish_spans: %u spans exceed the record buffer", n_spans); return 0; }
    BU_TRY(ctx, h2d(ctx, q->nodes.p, h_spans, bytes));   // through the context's pinned ring: the caller's array may go when this returns
    BU_TRY(ctx, bu::launch_finish_spans(ctx->stream, q->perm[0], q->perm[1], static_cast<const bu::bk_span*>(q->nodes.p), n_spans, d_leaf_of, d_parent_of, d_group_offsets, d_first_pos,
                                        d_sizes));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));   // q may be destroyed right after
    return 1;
}

// staging layout: [children of node 0 | children of node 1 | ... ] u32, padded to a u64 boundary, then n_nodes result records, then the node table + flags
static int tsvq_exchange_layout(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, uint32_t n_nodes, std::vector<bu::bk_span>& table, size_t& rec_at, size_t& tab_at, size_t& total) {
    table.resize(n_nodes);
    uint64_t run = 0;
    for (uint32_t i = 0; i < n_nodes; i++) {
        if ((uint64_t)h_nodes[i].start + h_nodes[i].count > q->n) { set_error(ctx, "tsvq_exchange: node outside the training set"); return 0; }
        table[i] = bu::bk_span{h_nodes[i].buf, h_nodes[i].start, h_nodes[i].count, (uint32_t)run};
        run += h_nodes[i].count;
    }
    if (run > (uint64_t)bu::TSVQ_BUFS * q->n) { set_error(ctx, "tsvq_exchange: overlapping nodes"); return 0; }   // (every member buffer once over: the whole-tree exchange of the one-tree-per-rank build)
    rec_at = ((size_t)run * 4 + 7) & ~(size_t)7;
    tab_at = rec_at + (size_t)n_nodes * sizeof(bu_tsvq_split);
    total = tab_at + (size_t)n_nodes * sizeof(bu::bk_span) + ((size_t)n_nodes + 7 & ~(size_t)7);
    if (total > q->xchg_cap) {
        if (q->xchg) bu_hip_free(ctx, q->xchg);
        q->xchg_cap = total + total / 4 + 4096;
        q->xchg = bu_hip_malloc(ctx, q->xchg_cap);
        if (!q->xchg) { q->xchg_cap = 0; set_error(ctx, "tsvq_exchange: allocation"); return 0; }
    }
    return 1;
}

int bu_hip_tsvq_exchange_pack(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, const uint8_t* h_mine, const bu_tsvq_split* h_records, uint32_t n_nodes,
                              void** d_staging, uint64_t* n_u64) {
    if (!ctx || !q || !h_nodes || !h_mine || !h_records || !d_staging || !n_u64 || !n_nodes) return 0;
    device_guard g(ctx->device);
    std::vector<bu::bk_span> table;
    size_t rec_at, tab_at, total;
    if (!tsvq_exchange_layout(ctx, q, h_nodes, n_nodes, table, rec_at, tab_at, total)) return 0;
    char* base = static_cast<char*>(q->xchg);
    // host part of the staging buffer: records (zero where not mine), node table, flags -- one upload
    std::vector<char> host(total - rec_at, 0);
    for (uint32_t i = 0; i < n_nodes; i++) if (h_mine[i]) std::memcpy(&host[(size_t)i * sizeof(bu_tsvq_split)], &h_records[i], sizeof(bu_tsvq_split));
    std::memcpy(&host[tab_at - rec_at], table.data(), (size_t)n_nodes * sizeof(bu::bk_span));
    std::memcpy(&host[tab_at - rec_at + (size_t)n_nodes * sizeof(bu::bk_span)], h_mine, n_nodes);
    BU_TRY(ctx, h2d(ctx, base + rec_at, host.data(), host.size()));
    if (rec_at >= 8) BU_TRY(ctx, hipMemsetAsync(base + rec_at - 8, 0, 8, ctx->stream));   // the padding word of an odd child count
    BU_TRY(ctx, bu::launch_exchange_children(ctx->stream, q->perm[0], q->perm[1], reinterpret_cast<const bu::bk_span*>(base + tab_at),
                                             reinterpret_cast<const uint8_t*>(base + tab_at + (size_t)n_nodes * sizeof(bu::bk_span)), n_nodes, reinterpret_cast<uint32_t*>(base), 0));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));
    *d_staging = q->xchg;
    *n_u64 = tab_at / 8;
    return 1;
}

int bu_hip_tsvq_exchange_unpack(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_node* h_nodes, const uint8_t* h_mine, bu_tsvq_split* h_records, uint32_t n_nodes) {
    if (!ctx || !q || !h_nodes || !h_mine || !h_records || !n_nodes || !q->xchg) return 0;
    device_guard g(ctx->device);
    std::vector<bu::bk_span> table;
    size_t rec_at, tab_at, total;
    if (!tsvq_exchange_layout(ctx, q, h_nodes, n_nodes, table, rec_at, tab_at, total)) return 0;
    char* base = static_cast<char*>(q->xchg);
    std::vector<uint8_t> theirs(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) theirs[i] = h_mine[i] ? 0 : 1;
    BU_TRY(ctx, h2d(ctx, base + tab_at + (size_t)n_nodes * sizeof(bu::bk_span), theirs.data(), n_nodes));
    BU_TRY(ctx, bu::launch_exchange_children(ctx->stream, q->perm[0], q->perm[1], reinterpret_cast<const bu::bk_span*>(base + tab_at),
                                             reinterpret_cast<const uint8_t*>(base + tab_at + (size_t)n_nodes * sizeof(bu::bk_span)), n_nodes, reinterpret_cast<uint32_t*>(base), 1));
    if (!fetch(ctx, h_records, base + rec_at, (size_t)n_nodes * sizeof(bu_tsvq_split))) return 0;
    return 1;
}

int bu_hip_tsvq_read_members(bu_hip_context* ctx, bu_tsvq* q, uint32_t buf, uint32_t start, uint32_t count, uint32_t* h_out) {
    if (!ctx || !q || buf >= bu::TSVQ_BUFS || (uint64_t)start + count > q->n) return 0;
    device_guard g(ctx->device);
    if (count) BU_TRY(ctx, d2h_pageable(ctx, h_out, q->perm[0] + (size_t)buf * q->n + start, (size_t)count * 4));
    BU_TRY(ctx, stream_wait(ctx, ctx->stream));
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------- section 1 (blocking, host pointers)

// ---------------------------------------------------------------- UASTC (rows a16-a19)

size_t bu_hip_uastc_workspace_bytes(uint32_t n_blocks, uint32_t flags) { return bu::uastc_workspace_bytes(n_blocks, flags); }

int bu_hip_k_encode_uastc_blocks(bu_hip_context* ctx, const void* d_px, uint32_t n_blocks, uint32_t flags, void* d_out) {
    if (!ctx) return 0;
    if (!d_px || !d_out) { set_error(ctx, "encode_uastc: null device pointer"); return 0; }
    device_guard g(ctx->device);
    arena& ws = ctx->scratch[5];
    BU_TRY(ctx, ws.reserve(bu::uastc_workspace_bytes(n_blocks, flags)));
    static const char* const names[4] = { "uastc_classify", "uastc_candidates", "uastc_score", "uastc_finish" };
    for (int phase = 0; phase < 4; phase++) {
        prof_scope ps(ctx, names[phase]);
        BU_TRY(ctx, bu::launch_uastc_phase(ctx->stream, phase, d_px, n_blocks, flags, ws.p, d_out));
    }
    return 1;
}

int bu_hip_encode_uastc_blocks(bu_hip_context* ctx, bu_uastc_block* out, uint32_t flags) {
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    arena& o = ctx->scratch[0];
    BU_TRY(ctx, o.reserve((size_t)n * 16));
    if (!bu_hip_k_encode_uastc_blocks(ctx, ctx->d_pixel_blocks, n, flags, o.p)) return 0;
    if (!fetch(ctx, out, o.p, (size_t)n * 16)) return 0;
    return 1;
}

void bu_hip_uastc_rdo_default_params(bu_uastc_rdo_params* p) {
    if (!p) return;
    p->m_lz_dict_size = 4096; p->m_lambda = 0.5f; p->m_max_allowed_rms_increase_ratio = 10.0f; p->m_skip_block_rms_thresh = 8.0f;
    p->m_endpoint_refinement = 1; p->m_lz_literal_cost = 100; p->m_max_smooth_block_std_dev = 18.0f; p->m_smooth_block_max_error_scale = 10.0f;
}

// A stream with a hardware queue of its own: hipExtStreamCreateWithCUMask with every CU enabled (the runtime does not pool queues that carry a CU mask). nullptr on failure.
// reserve = 0: every CU. Otherwise the device's CUs are split into a RESERVED set of about `reserve` CUs -- every (CUs / reserve)-th one, so that whatever order the
// mask's bits have over XCDs and shader engines, every one of them gives its share -- and the rest; reserved_side picks which of the two the stream may use.
static hipStream_t make_dedicated_stream(int device, uint32_t reserve = 0, bool reserved_side = false) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || prop.multiProcessorCount <= 0) { (void)hipGetLastError(); return nullptr; }
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    std::vector<uint32_t> mask((cus + 31) / 32, 0xFFFFFFFFu);
    if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
    if (reserve && reserve < cus) {
        const uint32_t stride = cus / reserve;
        for (uint32_t i = 0; i < cus; i++) {
            const bool is_reserved = stride >= 2 ? (i % stride == stride - 1) : (i < reserve);
            if (is_reserved != reserved_side) mask[i / 32] &= ~(1u << (i % 32));
        }
    }
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}

// The context's second stream (and the two events that fork it off the main stream and join it back), made on first use.
static bool ensure_side_stream(bu_hip_context* ctx) {
    if (ctx->side_stream) return true;
    if (hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess || (!ctx->side_fork && hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming) != hipSuccess) ||
        (!ctx->side_join && hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming) != hipSuccess)) {
        (void)hipGetLastError();
        if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr; }
        if (ctx->side_fork) { (void)hipEventDestroy(ctx->side_fork); ctx->side_fork = nullptr; }
        if (ctx->side_join) { (void)hipEventDestroy(ctx->side_join); ctx->side_join = nullptr; }
        return false;
    }
    return true;
}

// The strip walks of uastc_rdo behind its prepare pass: the lean build (strips without a block of a sensitive mode: four waves per SIMD) on the context's stream and,
// when endpoint refinement is on, the build with the refit in it (the flagged strips) on the side stream beside it -- forked and joined with events, nobody waits on
// the host. Without a side stream the two launches simply follow each other.
static int uastc_rdo_walks(bu_hip_context* ctx, void* d_blocks, const void* d_px, uint32_t n_blocks, const float* fp, const uint32_t* up, uint32_t flags, uint32_t total_jobs, void* ws) {
    const bool refit = up[2] != 0;
    if (ctx->walk_stream) {
        // a pipeline lane with reserved walk CUs: both builds of the walk on streams of their own whose CU masks are the reserved set, forked off and joined back to the
        // lane's stream (which may not use those CUs): the walks' waves never wait for a slot behind a chip-filling kernel, and never share a SIMD with one
        BU_TRY(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
        BU_TRY(ctx, hipStreamWaitEvent(ctx->walk_stream, ctx->side_fork, 0));
        BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->walk_stream, 1, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws));
        BU_TRY(ctx, hipEventRecord(ctx->walk_join, ctx->walk_stream));
        if (refit) {
            BU_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
            BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->side_stream, 3, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws));
            BU_TRY(ctx, hipEventRecord(ctx->side_join, ctx->side_stream));
            BU_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
        }
        BU_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->walk_join, 0));
        return 1;
    }
    const bool side = refit && ensure_side_stream(ctx);
    if (side) {
        BU_TRY(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
        BU_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
        BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->side_stream, 3, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws));
        BU_TRY(ctx, hipEventRecord(ctx->side_join, ctx->side_stream));
    }
    BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->stream, 1, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws));
    if (side) BU_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
    else if (refit) BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->stream, 3, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws));
    return 1;
}

int bu_hip_k_uastc_rdo(bu_hip_context* ctx, void* d_blocks, const void* d_px, uint32_t n_blocks, const bu_uastc_rdo_params* params, uint32_t flags,
                       uint32_t total_jobs, uint32_t out_stats[4]) {
    if (!ctx) return 0;
    if (!d_blocks || !d_px || !params) { set_error(ctx, "uastc_rdo: null pointer"); return 0; }
    // uastc_rdo's asserts (uastc_enc.cpp:4097-4099) as errors
    if (!(params->m_max_allowed_rms_increase_ratio > 1.0f) || !params->m_lz_dict_size || !(params->m_lambda > 0.0f)) {
        set_error(ctx, "uastc_rdo: need max_allowed_rms_increase_ratio > 1, lz_dict_size > 0, lambda > 0");
        return 0;
    }
    device_guard g(ctx->device);
    if (out_stats) out_stats[0] = out_stats[1] = out_stats[2] = 0, out_stats[3] = bu::uastc_rdo_strips(n_blocks, total_jobs);
    if (!n_blocks) return 1;
    const float fp[5] = { params->m_lambda, params->m_max_allowed_rms_increase_ratio, params->m_skip_block_rms_thresh, params->m_max_smooth_block_std_dev,
                          params->m_smooth_block_max_error_scale };
    const uint32_t up[3] = { params->m_lz_dict_size, params->m_lz_literal_cost, params->m_endpoint_refinement };
    arena& ws = ctx->scratch[5];
    BU_TRY(ctx, ws.reserve(bu::uastc_rdo_workspace_bytes(n_blocks, total_jobs)));
    {
        prof_scope ps(ctx, "uastc_rdo_prepare");
        BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->stream, 0, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p));
    }
    {
        prof_scope ps(ctx, "uastc_rdo_strips");   // both walks: the scope ends behind the join
        if (!uastc_rdo_walks(ctx, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p)) return 0;
    }
    // how many blocks each strip modified: sizes the finish launch (a 16-byte copy per 4 strips; the walk has to be over anyway)
    std::vector<uint32_t> per_strip(bu::uastc_rdo_strips(n_blocks, total_jobs));
    if (!fetch(ctx, per_strip.data(), bu::uastc_rdo_strip_counts(ws.p, n_blocks, total_jobs), per_strip.size() * 4)) return 0;
    uint32_t longest = 0;
    for (uint32_t c : per_strip) longest = c > longest ? c : longest;
    {
        prof_scope ps(ctx, "uastc_rdo_finish");
        BU_TRY(ctx, bu::launch_uastc_rdo_finish(ctx->stream, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p, longest));
    }
    uint32_t counters[4] = { 0, 0, 0, 0 };
    if (!fetch(ctx, counters, bu::uastc_rdo_counters(ws.p, n_blocks, total_jobs), sizeof(counters))) return 0;
#ifdef RDO_PROFILE
    {
        unsigned long long prof[16];
        hipMemcpy(prof, static_cast<const char*>(bu::uastc_rdo_counters(ws.p, n_blocks, total_jobs)) + 64, sizeof(prof), hipMemcpyDeviceToHost);
        fprintf(stderr, "rdo strip 0 cycles by phase:");
        for (int k = 0; k < 16; k++) fprintf(stderr, " %llu", prof[k]);
        fprintf(stderr, "\n");
    }
#endif
    if (counters[1]) { set_error(ctx, "uastc_rdo: a block does not unpack as UASTC"); return 0; }
    if (out_stats) { out_stats[0] = counters[0]; out_stats[1] = counters[2]; out_stats[2] = counters[3]; }
    return 1;
}

// ---------------------------------------------------------------- UASTC (+ RDO) over a stream of images: several in flight (BASELINE configs[4])
//
// uastc_rdo's walk is a serial chain per strip (uastc_enc.cpp:3824-4100): one workgroup per strip, ~3 us per block, so the strips of one batch of images occupy a
// fraction of the chip for ~20 ms whatever the batch holds (96 strips of the Kodak batch: 96 of 256 CUs). Nothing in one batch can fill the rest -- the next batch can:
// its encode / prepare kernels (and the previous batch's finish) run on the idle CUs while this batch's strips walk. The pipeline owns `lanes` private contexts (stream +
// workspaces each); a submission is ENQUEUED on the next lane without any host synchronisation -- the finish kernel is launched for the longest list a strip can have
// instead of waiting for the walk to learn the real one -- and completes behind an event. Results are those of bu_hip_k_encode_uastc_blocks + bu_hip_k_uastc_rdo.
} // extern "C" (reopened below)
struct bu_uastc_pipeline {
    struct lane { bu_hip_context* ctx = nullptr; hipEvent_t done = nullptr, input = nullptr; uint32_t* stats = nullptr; bool busy = false, with_rdo = false; uint64_t ticket = 0; uint32_t strips = 0; };
    bu_hip_context* parent = nullptr;
    std::vector<lane> lanes;
    uint64_t next_ticket = 1;
};
extern "C" {

static int uastc_rdo_enqueue(bu_hip_context* ctx, void* d_blocks, const void* d_px, uint32_t n_blocks, const bu_uastc_rdo_params* params, uint32_t flags, uint32_t total_jobs,
                             uint32_t* h_pinned_counters) {
    if (!(params->m_max_allowed_rms_increase_ratio > 1.0f) || !params->m_lz_dict_size || !(params->m_lambda > 0.0f)) {
        set_error(ctx, "uastc_rdo: need max_allowed_rms_increase_ratio > 1, lz_dict_size > 0, lambda > 0");
        return 0;
    }
    const float fp[5] = { params->m_lambda, params->m_max_allowed_rms_increase_ratio, params->m_skip_block_rms_thresh, params->m_max_smooth_block_std_dev,
                          params->m_smooth_block_max_error_scale };
    const uint32_t up[3] = { params->m_lz_dict_size, params->m_lz_literal_cost, params->m_endpoint_refinement };
    arena& ws = ctx->scratch[5];
    BU_TRY(ctx, ws.reserve(bu::uastc_rdo_workspace_bytes(n_blocks, total_jobs)));
    BU_TRY(ctx, bu::launch_uastc_rdo_phase(ctx->stream, 0, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p));
    if (!uastc_rdo_walks(ctx, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p)) return 0;
    // the longest list a strip can have (every block of it modified): the launch does not wait for the walk to know better, surplus workgroups leave at once
    const uint32_t strips = bu::uastc_rdo_strips(n_blocks, total_jobs);
    const uint32_t longest = strips > 1 ? (total_jobs ? n_blocks / total_jobs : n_blocks) : n_blocks;
    BU_TRY(ctx, bu::launch_uastc_rdo_finish(ctx->stream, d_blocks, d_px, n_blocks, fp, up, flags, total_jobs, ws.p, longest));
    BU_TRY(ctx, hipMemcpyAsync(h_pinned_counters, bu::uastc_rdo_counters(ws.p, n_blocks, total_jobs), 16, hipMemcpyDeviceToHost, ctx->stream));
    return 1;
}

bu_uastc_pipeline* bu_hip_uastc_pipeline_create(bu_hip_context* ctx, uint32_t lanes, uint32_t max_blocks, uint32_t flags, uint32_t max_total_jobs) {
    if (!ctx) return nullptr;
    if (lanes < 1 || lanes > 8 || !max_blocks) { set_error(ctx, "uastc_pipeline_create: 1..8 lanes, max_blocks > 0"); return nullptr; }
    device_guard g(ctx->device);
    bu_uastc_pipeline* p = new (std::nothrow) bu_uastc_pipeline();
    if (!p) return nullptr;
    p->parent = ctx;
    p->lanes.resize(lanes);
    // every workspace at its final size now: growing one later would free it under the kernels of an earlier submission
    const size_t ws_bytes = std::max(bu::uastc_workspace_bytes(max_blocks, flags), bu::uastc_rdo_workspace_bytes(max_blocks, max_total_jobs));
    for (auto& l : p->lanes) {
        l.ctx = create_context_kind(ctx->device, true);
        if (l.ctx) {
            l.ctx->tuning = ctx->tuning;   // the lanes take the paths their parent context is set to
            // The runtime maps ordinary streams onto its few shared hardware queues (GPU_MAX_HW_QUEUES) by how many streams each queue already carries -- history, as far
            // as a library can tell -- and two lanes whose streams share a queue run one after the other. A stream with a CU mask gets a hardware queue of its OWN: the
            // lanes' streams are made with one that enables every CU.
            const uint32_t walk_cus = ctx->tuning.uastc_walk_cus;
            if (!l.ctx->dedicated_queue || l.ctx->walk_cus != walk_cus) {
                hipStream_t fresh = make_dedicated_stream(l.ctx->device, walk_cus, false);
                if (fresh) {
                    (void)hipStreamSynchronize(l.ctx->own_stream);
                    const bool own = l.ctx->stream == l.ctx->own_stream;
                    (void)hipStreamDestroy(l.ctx->own_stream);
                    l.ctx->own_stream = fresh; l.ctx->dedicated_queue = true;
                    if (own) l.ctx->stream = fresh;
                    // the walks' streams: on the reserved CUs (or gone, when nothing is reserved)
                    if (l.ctx->walk_stream) { (void)hipStreamSynchronize(l.ctx->walk_stream); (void)hipStreamDestroy(l.ctx->walk_stream); l.ctx->walk_stream = nullptr; }
                    if (l.ctx->side_stream) { (void)hipStreamSynchronize(l.ctx->side_stream); (void)hipStreamDestroy(l.ctx->side_stream); l.ctx->side_stream = nullptr; }
                    l.ctx->walk_cus = 0;
                    if (walk_cus) {
                        l.ctx->walk_stream = make_dedicated_stream(l.ctx->device, walk_cus, true);
                        l.ctx->side_stream = make_dedicated_stream(l.ctx->device, walk_cus, true);
                        bool ok = l.ctx->walk_stream && l.ctx->side_stream;
                        if (ok && !l.ctx->walk_join) ok = hipEventCreateWithFlags(&l.ctx->walk_join, hipEventDisableTiming) == hipSuccess;
                        if (ok && !l.ctx->side_fork) ok = hipEventCreateWithFlags(&l.ctx->side_fork, hipEventDisableTiming) == hipSuccess;
                        if (ok && !l.ctx->side_join) ok = hipEventCreateWithFlags(&l.ctx->side_join, hipEventDisableTiming) == hipSuccess;
                        if (ok) l.ctx->walk_cus = walk_cus;
                        else {   // fall back to the unreserved form rather than fail: the lane's own stream keeps its (restricted) mask, the walks share it
                            (void)hipGetLastError();
                            if (l.ctx->walk_stream) { (void)hipStreamDestroy(l.ctx->walk_stream); l.ctx->walk_stream = nullptr; }
                            if (l.ctx->side_stream) { (void)hipStreamDestroy(l.ctx->side_stream); l.ctx->side_stream = nullptr; }
                        }
                    }
                }
            }
        }
        if (!l.ctx || hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&l.input, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&l.stats), 64, hipHostMallocDefault) != hipSuccess || l.ctx->scratch[5].reserve(ws_bytes) != hipSuccess) {
            set_error(ctx, "uastc_pipeline_create: lane set-up failed (%s)", l.ctx ? bu_hip_last_error(l.ctx) : "no context");
            bu_hip_uastc_pipeline_destroy(p);
            return nullptr;
        }
    }
    return p;
}

static int uastc_pipeline_collect(bu_uastc_pipeline* p, bu_uastc_pipeline::lane& l, uint32_t out_stats[4]) {
    if (!l.busy) return 1;
    if (hipEventSynchronize(l.done) != hipSuccess) { set_error(p->parent, "uastc_pipeline: a submission failed on the device"); l.busy = false; return 0; }
    l.busy = false;
    if (out_stats) { out_stats[0] = l.with_rdo ? l.stats[0] : 0; out_stats[1] = l.with_rdo ? l.stats[2] : 0; out_stats[2] = l.with_rdo ? l.stats[3] : 0; out_stats[3] = l.strips; }
    if (l.with_rdo && l.stats[1]) { set_error(p->parent, "uastc_rdo: a block does not unpack as UASTC"); return 0; }
    return 1;
}

int bu_hip_uastc_pipeline_submit(bu_uastc_pipeline* p, const void* d_px, uint32_t n_blocks, void* d_out, const bu_uastc_rdo_params* rdo, uint32_t flags, uint32_t total_jobs,
                                 uint64_t* out_ticket) {
    if (!p) return 0;
    bu_hip_context* ctx = p->parent;
    if (!d_px || !d_out || !n_blocks) { set_error(ctx, "uastc_pipeline_submit: null pointer / no blocks"); return 0; }
    device_guard g(ctx->device);
    const uint64_t ticket = p->next_ticket;
    bu_uastc_pipeline::lane& l = p->lanes[(size_t)(ticket % p->lanes.size())];
    if (!uastc_pipeline_collect(p, l, nullptr)) return 0;   // the lane's previous submission (its results are complete from here on; nobody asked for its statistics)
    const size_t need = std::max(bu::uastc_workspace_bytes(n_blocks, flags), rdo ? bu::uastc_rdo_workspace_bytes(n_blocks, total_jobs) : (size_t)0);
    if (need > l.ctx->scratch[5].cap) { set_error(ctx, "uastc_pipeline_submit: %u blocks / %u jobs exceed what the pipeline was created for", n_blocks, total_jobs); return 0; }
    // the input tiles may still be being produced on the caller's stream
    BU_TRY(ctx, hipEventRecord(l.input, ctx->stream));
    BU_TRY(ctx, hipStreamWaitEvent(l.ctx->stream, l.input, 0));
    if (!bu_hip_k_encode_uastc_blocks(l.ctx, d_px, n_blocks, flags, d_out)) { set_error(ctx, "uastc_pipeline_submit: %s", bu_hip_last_error(l.ctx)); return 0; }
    l.with_rdo = rdo != nullptr;
    l.strips = rdo ? bu::uastc_rdo_strips(n_blocks, total_jobs) : 0;
    if (rdo && !uastc_rdo_enqueue(l.ctx, d_out, d_px, n_blocks, rdo, flags, total_jobs, l.stats)) { set_error(ctx, "uastc_pipeline_submit: %s", bu_hip_last_error(l.ctx)); return 0; }
    BU_TRY(ctx, hipEventRecord(l.done, l.ctx->stream));
    l.busy = true; l.ticket = ticket;
    p->next_ticket++;
    if (out_ticket) *out_ticket = ticket;
    return 1;
}

int bu_hip_uastc_pipeline_wait(bu_uastc_pipeline* p, uint64_t ticket, uint32_t out_stats[4]) {
    if (!p) return 0;
    if (out_stats) out_stats[0] = out_stats[1] = out_stats[2] = out_stats[3] = 0;
    device_guard g(p->parent->device);
    int ok = 1;
    for (auto& l : p->lanes)
        if (l.busy && (ticket == 0 || l.ticket == ticket)) ok &= uastc_pipeline_collect(p, l, ticket ? out_stats : nullptr);
    return ok;
}

void bu_hip_uastc_pipeline_destroy(bu_uastc_pipeline* p) {
    if (!p) return;
    for (auto& l : p->lanes) {
        if (l.ctx) { device_guard g(l.ctx->device); (void)hipStreamSynchronize(l.ctx->stream); }
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.input) (void)hipEventDestroy(l.input);
        if (l.stats) (void)hipHostFree(l.stats);
        if (l.ctx) bu_hip_destroy_context(l.ctx);
    }
    delete p;
}

int bu_hip_uastc_rdo(bu_hip_context* ctx, bu_uastc_block* blocks, const bu_uastc_rdo_params* params, uint32_t flags, uint32_t total_jobs, uint32_t out_stats[4]) {
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    if (!blocks) { set_error(ctx, "uastc_rdo: null blocks"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    arena& o = ctx->scratch[0];
    BU_TRY(ctx, o.reserve((size_t)n * 16));
    BU_TRY(ctx, h2d(ctx, o.p, blocks, (size_t)n * 16));
    if (!bu_hip_k_uastc_rdo(ctx, o.p, ctx->d_pixel_blocks, n, params, flags, total_jobs, out_stats)) return 0;
    if (!fetch(ctx, blocks, o.p, (size_t)n * 16)) return 0;
    return 1;
}

int bu_hip_encode_etc1s_blocks(bu_hip_context* ctx, bu_etc_block* out, int perceptual, uint32_t total_perms) {
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    arena& o = ctx->scratch[0];
    BU_TRY(ctx, o.reserve((size_t)n * 8));
    BU_TRY(ctx, bu::launch_encode_etc1s_blocks(ctx->stream, ctx->d_pixel_blocks, n, quality_from_perms(total_perms), perceptual != 0, o.p));
    if (!fetch(ctx, out, o.p, (size_t)n * 8)) return 0;
    return 1;
}

int bu_hip_determine_selectors(bu_hip_context* ctx, const bu_color_rgba* color5_inten, bu_etc_block* out, int perceptual) {
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    arena &in = ctx->scratch[0], &o = ctx->scratch[1];
    BU_TRY(ctx, in.reserve((size_t)n * 4));
    BU_TRY(ctx, o.reserve((size_t)n * 8));
    BU_TRY(ctx, h2d(ctx, in.p, color5_inten, (size_t)n * 4));
    BU_TRY(ctx, bu::launch_determine_selectors(ctx->stream, ctx->d_pixel_blocks, n, static_cast<const uint8_t*>(in.p), nullptr, perceptual != 0, o.p));
    if (!fetch(ctx, out, o.p, (size_t)n * 8)) return 0;
    return 1;
}

int bu_hip_refine_endpoint_clusterization(bu_hip_context* ctx, const bu_block_info* info, uint32_t total_clusters, const bu_endpoint_cluster* clusters,
                                          const uint32_t* /*sorted_block_indices*/, uint32_t* out, int perceptual) {
    // The reference seam passes, per block, a window [first_cluster_ofs, first_cluster_ofs+num_clusters) into a flat list of
    // {unscaled colour, inten, cluster index} (frontend.cpp:1684-1750). We translate that to the device layer's form:
    // a parameter table addressed by POSITION in the flat list, one "parent" per distinct window. The block's current
    // cluster is identified by its index value; the kernel's tie rule compares against the candidate's position, so the
    // position of the current cluster inside the window is looked up here.
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    std::vector<uint32_t> params(total_clusters);
    for (uint32_t i = 0; i < total_clusters; i++)
        params[i] = clusters[i].m_unscaled_color.r | (clusters[i].m_unscaled_color.g << 8) | (clusters[i].m_unscaled_color.b << 16) | ((uint32_t)clusters[i].m_etc_inten << 24);
    // windows -> parents
    std::vector<uint32_t> win_first, win_count, cand_offsets(1, 0), cand_indices;
    std::vector<uint8_t> block_parent8;
    std::vector<uint32_t> block_parent(n), block_cur(n);
    std::vector<int32_t> first_to_parent(65536, -1);
    for (uint32_t b = 0; b < n; b++) {
        const uint32_t f = info[b].m_first_cluster_ofs, c = info[b].m_num_clusters;
        int32_t p = first_to_parent[f];
        if (p < 0 || win_count[p] != c) {
            p = (int32_t)win_first.size();
            first_to_parent[f] = p;
            win_first.push_back(f); win_count.push_back(c);
            for (uint32_t k = 0; k < c; k++) cand_indices.push_back(f + k);
            cand_offsets.push_back((uint32_t)cand_indices.size());
        }
        block_parent[b] = (uint32_t)p;
        // position of the block's current cluster inside its window (it is always present, frontend.cpp:971-996)
        uint32_t pos = f;
        for (uint32_t k = 0; k < c; k++)
            if (clusters[f + k].m_cluster_index == info[b].m_cur_cluster_index) { pos = f + k; break; }
        block_cur[b] = pos;
    }
    if (win_first.size() > 255) { set_error(ctx, "refine: more than 255 distinct candidate windows"); return 0; }
    block_parent8.resize(n);
    for (uint32_t b = 0; b < n; b++) block_parent8[b] = (uint8_t)block_parent[b];

    arena &a_par = ctx->scratch[0], &a_cur = ctx->scratch[1], &a_off = ctx->scratch[2], &a_idx = ctx->scratch[3], &a_out = ctx->scratch[4];
    arena& a_bp = ctx->scratch[5];
    BU_TRY(ctx, a_par.reserve(total_clusters * 4ull)); BU_TRY(ctx, a_cur.reserve(n * 4ull)); BU_TRY(ctx, a_off.reserve(cand_offsets.size() * 4ull));
    BU_TRY(ctx, a_idx.reserve(cand_indices.size() * 4ull + 4)); BU_TRY(ctx, a_out.reserve(n * 4ull)); BU_TRY(ctx, a_bp.reserve(n));
    BU_TRY(ctx, h2d(ctx, a_par.p, params.data(), total_clusters * 4ull));
    BU_TRY(ctx, h2d(ctx, a_cur.p, block_cur.data(), n * 4ull));
    BU_TRY(ctx, h2d(ctx, a_off.p, cand_offsets.data(), cand_offsets.size() * 4ull));
    if (!cand_indices.empty()) BU_TRY(ctx, h2d(ctx, a_idx.p, cand_indices.data(), cand_indices.size() * 4ull));
    BU_TRY(ctx, h2d(ctx, a_bp.p, block_parent8.data(), n));
    void* work = nullptr;
    if (const size_t wb = bu::refine_workspace_bytes(total_clusters, (uint32_t)win_first.size())) { BU_TRY(ctx, ctx->refine_lists.reserve(wb)); work = ctx->refine_lists.p; }
    BU_TRY(ctx, bu::launch_refine_endpoint_clusterization(ctx->stream, ctx->d_pixel_blocks, n, static_cast<const uint32_t*>(a_cur.p),
                                                          static_cast<const uint8_t*>(a_par.p), total_clusters, (uint32_t)win_first.size(),
                                                          static_cast<const uint32_t*>(a_off.p), static_cast<const uint32_t*>(a_idx.p),
                                                          static_cast<const uint8_t*>(a_bp.p), perceptual != 0, static_cast<uint32_t*>(a_out.p), work));
    std::vector<uint32_t> pos(n);
    if (!fetch(ctx, pos.data(), a_out.p, n * 4ull)) return 0;
    for (uint32_t b = 0; b < n; b++) out[b] = clusters[pos[b]].m_cluster_index; // positions -> cluster indices (.cl:1150)
    return 1;
}

int bu_hip_find_optimal_selector_clusters_for_each_block(bu_hip_context* ctx, const bu_fosc_block* info, uint32_t total_input_selectors,
                                                         const bu_fosc_selector* selectors, const uint32_t* selector_cluster_indices, uint32_t* out, int perceptual) {
    if (!ctx || !ctx->d_pixel_blocks) { if (ctx) set_error(ctx, "no pixel blocks set"); return 0; }
    device_guard g(ctx->device);
    const uint32_t n = (uint32_t)ctx->total_blocks;
    // packed 2-bit selectors [p*2] (frontend.cpp:2462-2464) -> etc_block selector bytes, addressed by position in the flat list
    std::vector<uint64_t> sel_blocks(total_input_selectors);
    for (uint32_t i = 0; i < total_input_selectors; i++) {
        uint32_t bits = 0;
        for (uint32_t p = 0; p < 16; p++) {
            const uint32_t s = (selectors[i].m_packed_selectors >> (p * 2)) & 3u, x = p & 3u, y = p >> 2;
            const uint32_t raw = (0x4Bu >> (s * 2)) & 3u, bit = x * 4 + y;
            bits |= ((raw & 1u) << bit) | ((raw >> 1) << (16 + bit));
        }
        sel_blocks[i] = __builtin_bswap64((uint64_t)bits);
    }
    std::vector<uint64_t> enc(n);
    std::vector<uint32_t> win_first, win_count, cand_offsets(1, 0), cand_indices, block_parent(n);
    std::vector<uint8_t> bp8(n);
    for (uint32_t b = 0; b < n; b++) {
        const bu_color_rgba c = info[b].m_etc_color5_inten;
        const uint64_t v = ((uint64_t)c.r << 59) | ((uint64_t)c.g << 51) | ((uint64_t)c.b << 43) | ((uint64_t)c.a << 37) | ((uint64_t)c.a << 34) | (3ull << 32);
        enc[b] = __builtin_bswap64(v);
        const uint32_t f = info[b].m_first_selector, cnt = info[b].m_num_selectors;
        int32_t p = -1;
        for (size_t w = 0; w < win_first.size(); w++) if (win_first[w] == f && win_count[w] == cnt) { p = (int32_t)w; break; }
        if (p < 0) {
            p = (int32_t)win_first.size();
            win_first.push_back(f); win_count.push_back(cnt);
            for (uint32_t k = 0; k < cnt; k++) cand_indices.push_back(f + k);
            cand_offsets.push_back((uint32_t)cand_indices.size());
        }
        block_parent[b] = (uint32_t)p;
    }
    if (win_first.size() > 255) { set_error(ctx, "fosc: more than 255 distinct candidate windows"); return 0; }
    for (uint32_t b = 0; b < n; b++) bp8[b] = (uint8_t)block_parent[b];

    arena &a_sel = ctx->scratch[0], &a_enc = ctx->scratch[1], &a_off = ctx->scratch[2], &a_idx = ctx->scratch[3], &a_tmp = ctx->scratch[4], &a_bp = ctx->scratch[5];
    BU_TRY(ctx, a_sel.reserve(total_input_selectors * 8ull + 8)); BU_TRY(ctx, a_enc.reserve(n * 8ull + n * 4ull)); BU_TRY(ctx, a_off.reserve(cand_offsets.size() * 4ull));
    BU_TRY(ctx, a_idx.reserve(cand_indices.size() * 4ull + 4)); BU_TRY(ctx, a_tmp.reserve(n * 4ull)); BU_TRY(ctx, a_bp.reserve(n));
    uint32_t* d_out = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(a_enc.p) + n * 8ull);
    if (total_input_selectors) BU_TRY(ctx, h2d(ctx, a_sel.p, sel_blocks.data(), total_input_selectors * 8ull));
    BU_TRY(ctx, h2d(ctx, a_enc.p, enc.data(), n * 8ull));
    BU_TRY(ctx, h2d(ctx, a_off.p, cand_offsets.data(), cand_offsets.size() * 4ull));
    if (!cand_indices.empty()) BU_TRY(ctx, h2d(ctx, a_idx.p, cand_indices.data(), cand_indices.size() * 4ull));
    BU_TRY(ctx, h2d(ctx, a_bp.p, bp8.data(), n));
    // chunk = 0: the OpenCL seam has no "same tile as previous block" shortcut (ocl_kernels.cl:1159-1225)
    BU_TRY(ctx, bu::launch_find_optimal_selector_clusters(ctx->stream, ctx->d_pixel_blocks, a_enc.p, n, a_sel.p, total_input_selectors, (uint32_t)win_first.size(),
                                                          static_cast<const uint32_t*>(a_off.p), static_cast<const uint32_t*>(a_idx.p), static_cast<const uint8_t*>(a_bp.p),
                                                          perceptual != 0, 0, static_cast<uint32_t*>(a_tmp.p), d_out, nullptr, 0));
    std::vector<uint32_t> pos(n);
    if (!fetch(ctx, pos.data(), d_out, n * 4ull)) return 0;
    for (uint32_t b = 0; b < n; b++) out[b] = selector_cluster_indices[pos[b]];
    return 1;
}

int bu_hip_encode_etc1s_pixel_clusters(bu_hip_context* ctx, bu_etc_block* out, uint32_t total_clusters, const bu_pixel_cluster* clusters,
                                       uint64_t total_pixels, const bu_color_rgba* pixels, const uint32_t* weights, int perceptual, uint32_t total_perms) {
    // The reference seam hands over de-duplicated colours with multiplicities. The device layer works on unweighted pixel lists
    // (bu_hip_k_generate_endpoint_codebook, which is what our own frontend uses and what INTEGRATION.md binds). For the legacy
    // call we expand the multiplicities into a temporary tile array laid out as "training vectors" of 8 pixels; clusters whose
    // expanded size is not a multiple of 8 cannot be expressed that way, so the expansion pads by REPEATING the whole colour
    // list k times (k = 8 / gcd(n, 8)): errors scale by k, the float mean and min/max are unchanged while sums stay < 2^24,
    // and the argmin over (colour, table) is invariant under a uniform positive scaling of all errors.
    if (!ctx) return 0;
    device_guard g(ctx->device);
    if (!total_clusters) return 1;
    std::vector<uint32_t> offsets(total_clusters + 1, 0), expanded;
    std::vector<uint32_t> words;
    words.reserve((size_t)total_pixels * 2);
    for (uint32_t c = 0; c < total_clusters; c++) {
        const uint64_t first = clusters[c].m_first_pixel_index, cnt = clusters[c].m_total_pixels;
        if (first + cnt > total_pixels) { set_error(ctx, "pixel cluster out of range"); return 0; }
        uint64_t n = 0;
        for (uint64_t i = 0; i < cnt; i++) n += weights[first + i];
        if (!n) { set_error(ctx, "empty pixel cluster"); return 0; }
        uint32_t gcd = 8; while (n % gcd) gcd >>= 1;
        const uint32_t reps = 8 / gcd;
        if (n * reps > 0x7FFFFFFFull) { set_error(ctx, "pixel cluster too large"); return 0; }
        const size_t base = words.size();
        for (uint32_t r = 0; r < reps; r++)
            for (uint64_t i = 0; i < cnt; i++) {
                uint32_t w; memcpy(&w, &pixels[first + i], 4);
                words.insert(words.end(), weights[first + i], w);
            }
        const uint32_t tv_first = (uint32_t)(base / 8), tv_cnt = (uint32_t)((words.size() - base) / 8);
        offsets[c + 1] = offsets[c] + tv_cnt;
        for (uint32_t t = 0; t < tv_cnt; t++) expanded.push_back(tv_first + t);
    }
    words.resize((words.size() + 15) / 16 * 16, 0);
    arena &a_px = ctx->scratch[0], &a_off = ctx->scratch[1], &a_idx = ctx->scratch[2], &a_par = ctx->scratch[3];
    const size_t params_bytes = ((total_clusters * 4ull + 7) / 8) * 8;
    BU_TRY(ctx, a_px.reserve(words.size() * 4ull)); BU_TRY(ctx, a_off.reserve(offsets.size() * 4ull)); BU_TRY(ctx, a_idx.reserve(expanded.size() * 4ull + 4));
    BU_TRY(ctx, a_par.reserve(params_bytes + total_clusters * 8ull + total_clusters));
    BU_TRY(ctx, h2d(ctx, a_px.p, words.data(), words.size() * 4ull));
    BU_TRY(ctx, h2d(ctx, a_off.p, offsets.data(), offsets.size() * 4ull));
    BU_TRY(ctx, h2d(ctx, a_idx.p, expanded.data(), expanded.size() * 4ull));
    uint8_t* d_params = static_cast<uint8_t*>(a_par.p);
    uint64_t* d_err = reinterpret_cast<uint64_t*>(d_params + params_bytes);
    uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_err + total_clusters);
    if (!bu_hip_k_generate_endpoint_codebook(ctx, a_px.p, total_clusters, offsets.data(), static_cast<const uint32_t*>(a_off.p), static_cast<const uint32_t*>(a_idx.p),
                                             std::max(quality_from_perms(total_perms), (int)bu::BU_Q_MEDIUM), perceptual, 0, d_params, d_err, d_valid))
        return 0;
    std::vector<uint8_t> params(total_clusters * 4ull);
    if (!fetch(ctx, params.data(), d_params, params.size())) return 0;
    for (uint32_t c = 0; c < total_clusters; c++) {
        const uint64_t v = ((uint64_t)params[c * 4] << 59) | ((uint64_t)params[c * 4 + 1] << 51) | ((uint64_t)params[c * 4 + 2] << 43) |
                           ((uint64_t)params[c * 4 + 3] << 37) | ((uint64_t)params[c * 4 + 3] << 34) | (3ull << 32);
        const uint64_t m = __builtin_bswap64(v);
        memcpy(&out[c], &m, 8);
    }
    return 1;
}

} // extern "C"
