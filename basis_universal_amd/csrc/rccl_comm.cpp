// rccl_comm.cpp -- libbasisu_rccl.so: bu_comm on RCCL (include/basisu_hip_comm.h). The reference has no communication layer at all
// (single process, std::thread pool); this is the MI355X-native equivalent for the sharded frontend: in-place all-gather of
// block-row slabs and in-place u64 sum all-reduce (exact: integer accumulators and disjoint per-rank results), enqueued on the
// context's HIP stream behind the kernels that produced the data, over xGMI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/basisu_hip_comm.h"

static_assert(BU_RCCL_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

// The communicators one bu_rccl_comm_init_all call made live in ONE process. A collective only completes once every rank has enqueued its part, so a host thread
// that walks the ranks of ONE collective in turn waits (inside RCCL, or at its next stream synchronisation) for a part it has not issued yet. The ranks of a group
// issue the same sequence of collectives, so the rule is stated on that sequence: a thread that has issued collective number e (or a later one) for rank j may not
// issue number e for another rank i -- it gets an error instead of a hang. Nothing is bound for longer than that: any thread may issue a rank's NEXT collective
// (executor pools), and threads are told apart by a token that is never handed out twice (std::thread::id values are reused once a thread has exited).
struct comm_group {
    std::mutex lock;
    std::vector<uint64_t> seq;         // collectives let through so far, per rank
    std::vector<uint64_t> last_token;  // who issued the last one (0 = nobody yet)
    uint32_t alive = 0;
};
struct bu_rccl_comm {
    ncclComm_t comm = nullptr;
    bu_hip_context* ctx = nullptr;
    uint32_t rank = 0, world = 1;
    comm_group* group = nullptr;   // null for communicators made by bu_rccl_comm_create (one per process)
};

namespace {

std::mutex g_lock;
std::string g_error;

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> g(g_lock);
    g_error = buf;
    return 0;
}

std::atomic<uint64_t> g_next_token{1};
uint64_t thread_token() { static thread_local uint64_t t = g_next_token.fetch_add(1, std::memory_order_relaxed); return t; }

// 1 = this thread may issue rank c->rank's next collective; 0 = it has already issued that collective (or a later one) for another rank of the group
int thread_rule(bu_rccl_comm* c, const char* what) {
    if (!c->group) return 1;
    comm_group& g = *c->group;
    const uint64_t me = thread_token();
    std::lock_guard<std::mutex> lk(g.lock);
    const uint64_t e = g.seq[c->rank];
    for (uint32_t j = 0; j < g.seq.size(); j++)
        if (j != c->rank && g.last_token[j] == me && g.seq[j] > e)
            return fail("%s: this host thread has already issued collective #%llu for rank %u of the same bu_rccl_comm_init_all group and now issues #%llu for rank %u; a "
                        "collective completes only when every rank has enqueued its part, so the ranks of one collective need a thread of their own each "
                        "(include/basisu_hip_comm.h)", what, (unsigned long long)(g.seq[j] - 1), j, (unsigned long long)e, c->rank);
    g.last_token[c->rank] = me; g.seq[c->rank] = e + 1;
    return 1;
}

int all_gather(void* user, void* d_buf, uint64_t bytes_per_rank) {
    bu_rccl_comm* c = static_cast<bu_rccl_comm*>(user);
    if (!c) return fail("all_gather: no communicator");
    if (!thread_rule(c, "all_gather")) return 0;
    if (!c->comm) return fail("all_gather: no communicator");
    hipStream_t st = static_cast<hipStream_t>(bu_hip_get_stream(c->ctx));
    if (hipSetDevice(bu_hip_context_device(c->ctx)) != hipSuccess) return fail("all_gather: hipSetDevice failed");
    char* base = static_cast<char*>(d_buf);
    const ncclResult_t r = ncclAllGather(base + (size_t)c->rank * bytes_per_rank, base, (size_t)bytes_per_rank, ncclUint8, c->comm, st);
    if (r != ncclSuccess) return fail("ncclAllGather: %s", ncclGetErrorString(r));
    return 1;   // stream-ordered (bu_comm::stream_ordered = 1): whoever reads the result on the host synchronises there
}

int all_reduce_u64(void* user, void* d_buf, uint64_t count) {
    bu_rccl_comm* c = static_cast<bu_rccl_comm*>(user);
    if (!c) return fail("all_reduce: no communicator");
    if (!thread_rule(c, "all_reduce")) return 0;
    if (!c->comm) return fail("all_reduce: no communicator");
    hipStream_t st = static_cast<hipStream_t>(bu_hip_get_stream(c->ctx));
    if (hipSetDevice(bu_hip_context_device(c->ctx)) != hipSuccess) return fail("all_reduce: hipSetDevice failed");
    const ncclResult_t r = ncclAllReduce(d_buf, d_buf, (size_t)count, ncclUint64, ncclSum, c->comm, st);
    if (r != ncclSuccess) return fail("ncclAllReduce: %s", ncclGetErrorString(r));
    return 1;
}

} // namespace

extern "C" {

const char* bu_rccl_last_error(void) { std::lock_guard<std::mutex> g(g_lock); static thread_local std::string copy; copy = g_error; return copy.c_str(); }

int bu_rccl_get_unique_id(void* out_id) {
    if (!out_id) return fail("get_unique_id: null pointer");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId: %s", ncclGetErrorString(r));
    std::memcpy(out_id, &id, sizeof(id));
    return 1;
}

bu_rccl_comm* bu_rccl_comm_create(bu_hip_context* ctx, const void* id_bytes, uint32_t rank, uint32_t world) {
    if (!ctx || !id_bytes || !world || rank >= world) { fail("comm_create: bad arguments"); return nullptr; }
    if (hipSetDevice(bu_hip_context_device(ctx)) != hipSuccess) { fail("comm_create: hipSetDevice failed"); return nullptr; }
    bu_rccl_comm* c = new (std::nothrow) bu_rccl_comm();
    if (!c) return nullptr;
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) { fail("ncclCommInitRank: %s", ncclGetErrorString(r)); delete c; return nullptr; }
    return c;
}

int bu_rccl_comm_init_all(bu_hip_context* const* ctxs, uint32_t n, bu_rccl_comm** out) {
    if (!ctxs || !out || !n) return fail("comm_init_all: bad arguments");
    std::vector<int> devs(n);
    for (uint32_t i = 0; i < n; i++) { if (!ctxs[i]) return fail("comm_init_all: null context"); devs[i] = bu_hip_context_device(ctxs[i]); }
    std::vector<ncclComm_t> comms(n, nullptr);
    const ncclResult_t r = ncclCommInitAll(comms.data(), (int)n, devs.data());
    if (r != ncclSuccess) return fail("ncclCommInitAll: %s", ncclGetErrorString(r));
    comm_group* grp = new (std::nothrow) comm_group();
    if (!grp) { for (uint32_t j = 0; j < n; j++) (void)ncclCommDestroy(comms[j]); return fail("comm_init_all: out of memory"); }
    grp->seq.assign(n, 0); grp->last_token.assign(n, 0); grp->alive = n;
    for (uint32_t i = 0; i < n; i++) {
        bu_rccl_comm* c = new (std::nothrow) bu_rccl_comm();
        if (!c) { for (uint32_t j = 0; j < n; j++) { if (j < i) { delete out[j]; } (void)ncclCommDestroy(comms[j]); } delete grp; return fail("comm_init_all: out of memory"); }
        c->comm = comms[i]; c->ctx = ctxs[i]; c->rank = i; c->world = n; c->group = grp;
        out[i] = c;
    }
    return 1;
}

// Test hook (tests/test_host_logic.py, no GPU needed): a group of n communicators with NO RCCL communicator behind them -- a collective on one of them fails with
// "no communicator" when the calling thread may drive it and with the thread rule's message when it may not, which is all the rule's test needs to tell apart.
int bu_rccl_debug_unconnected_group(uint32_t n, bu_rccl_comm** out) {
    if (!out || !n) return fail("debug_unconnected_group: bad arguments");
    comm_group* grp = new (std::nothrow) comm_group();
    if (!grp) return fail("debug_unconnected_group: out of memory");
    grp->seq.assign(n, 0); grp->last_token.assign(n, 0); grp->alive = n;
    for (uint32_t i = 0; i < n; i++) {
        bu_rccl_comm* c = new (std::nothrow) bu_rccl_comm();
        if (!c) return fail("debug_unconnected_group: out of memory");
        c->rank = i; c->world = n; c->group = grp;
        out[i] = c;
    }
    return 1;
}

void bu_rccl_comm_destroy(bu_rccl_comm* c) {
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->group) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->group->lock); c->group->last_token[c->rank] = 0; last = --c->group->alive == 0; }
        if (last) delete c->group;
    }
    delete c;
}

int bu_rccl_comm_fill(bu_rccl_comm* c, bu_comm* out) {
    if (!c || !out) return fail("comm_fill: null pointer");
    out->rank = c->rank; out->world = c->world; out->user = c;
    out->all_gather = all_gather; out->all_reduce_u64 = all_reduce_u64;
    out->stream_ordered = 1; out->reserved = 0;
    return 1;
}

} // extern "C"
