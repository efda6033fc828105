// block_pool.cpp -- operator new / delete of libbasisu_frontend.so: large host blocks are recycled instead of being mapped and unmapped per image.
//
// The frontend and the backend allocate and free tens of multi-megabyte index arrays per image (~300 MB at 4096^2), the same sizes for every image. With glibc's
// defaults each of them is a fresh mmap whose pages fault in one by one (~0.2 ms per MB) and is unmapped again on free; with many images in flight those faults
// serialise on the process's address-space lock (16 compressors: ~75,000 faults per image). Earlier rounds changed the PROCESS's malloc policy from a static
// constructor (mallopt) -- a side effect a plug-in library must not have. This file keeps the cure inside the library instead: blocks of kLarge bytes and more
// come from a recycling pool of page-aligned mappings (advised to use huge pages), everything smaller goes to malloc untouched. The operators are LOCAL to
// this library (the link hides them: csrc/host/exports.map + -Bsymbolic), so nobody else's allocations change; the library's C ABI never hands out or takes
// ownership of C++ objects, so every block is freed by the operator delete that sits next to the operator new it came from.
// INVARIANT the local operators rest on: a block is freed by the module that allocated it. That holds for everything this library allocates large -- std::vector and
// plain arrays are header-only code, instantiated here on both ends. It would NOT hold for a std::string (or stream buffer) of a megabyte and more: libstdc++ keeps
// out-of-line instantiations of those (regrowing one there would free a pooled block with glibc's free). The library builds no such string -- its strings are error
// texts and names -- and must not start to.
// What the pool keeps is visible as resident memory of the host process until it is given back: bu_host_pool_trim() unmaps every cached block (a host that encodes in
// bursts calls it between them); the cap bounds it at any time.
// Environment, read once: BU_HOST_POOL_MB = most megabytes kept cached (default 6144; 0 switches the pool off).
#include <sys/mman.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

namespace {

const size_t kLarge = (size_t)1 << 20;        // blocks below this go to malloc
const size_t kPage = 4096, kHuge = (size_t)2 << 20;
const int kClasses = 8 * 24;                  // eight size classes per power of two from 1 MiB up to 2^44

struct free_block { free_block* next; size_t mapped; int cls; };

struct pool {
    std::mutex m;
    free_block* free_list[kClasses] = {};
    size_t cached_bytes = 0, cap_bytes = (size_t)6144 << 20;
    // every block handed out, by address: open addressing, grows by doubling (its own storage comes from malloc)
    struct slot { uintptr_t p; size_t mapped; int cls; };
    slot* table = nullptr; size_t table_cap = 0, table_used = 0;
    std::atomic<int> state{0};                // 0 = not configured, 1 = on, 2 = off
    uint64_t n_maps = 0, n_reuses = 0;

    void configure() {
        int s = state.load(std::memory_order_acquire);
        if (s) return;
        std::lock_guard<std::mutex> g(m);
        if (state.load(std::memory_order_relaxed)) return;
        if (const char* e = std::getenv("BU_HOST_POOL_MB")) { const long v = std::atol(e); cap_bytes = v <= 0 ? 0 : (size_t)v << 20; }
        state.store(cap_bytes ? 1 : 2, std::memory_order_release);
    }

    static int size_class(size_t bytes, size_t* rounded) {
        // classes: 2^k * (8 + j) / 8, j = 0..7  (at most 12.5 % slack)
        int k = 63 - __builtin_clzll(bytes);
        size_t step = (size_t)1 << (k - 3);
        size_t r = (bytes + step - 1) & ~(step - 1);
        if (r == ((size_t)2 << k)) { k++; step <<= 1; }
        const int j = (int)((r >> (k - 3)) - 8);
        *rounded = r;
        const int c = (k - 20) * 8 + j;
        return c < kClasses ? c : -1;
    }

    bool table_put(uintptr_t p, size_t mapped, int cls) {   // false: the table could not grow -- the caller must not hand the block out (delete would not know it)
        if ((table_used + 1) * 2 > table_cap) {
            const size_t ncap = table_cap ? table_cap * 2 : 1024;
            slot* nt = static_cast<slot*>(std::calloc(ncap, sizeof(slot)));
            if (!nt) return false;
            for (size_t i = 0; i < table_cap; i++) if (table[i].p) { size_t h = (table[i].p >> 12) * 0x9E3779B97F4A7C15ull % ncap; while (nt[h].p) h = (h + 1) % ncap; nt[h] = table[i]; }
            std::free(table); table = nt; table_cap = ncap;
        }
        size_t h = (p >> 12) * 0x9E3779B97F4A7C15ull % table_cap;
        while (table[h].p) h = (h + 1) % table_cap;
        table[h] = slot{p, mapped, cls}; table_used++;
        return true;
    }
    size_t table_take(uintptr_t p, int* cls) {   // removes p, returns its mapped size (0 = not ours)
        if (!table_cap) return 0;
        size_t h = (p >> 12) * 0x9E3779B97F4A7C15ull % table_cap;
        while (table[h].p && table[h].p != p) h = (h + 1) % table_cap;
        if (!table[h].p) return 0;
        const size_t mapped = table[h].mapped;
        *cls = table[h].cls;
        // backward-shift deletion keeps the probe sequences intact
        size_t i = h;
        for (;;) {
            size_t j = (i + 1) % table_cap;
            for (;; j = (j + 1) % table_cap) {
                if (!table[j].p) { table[i].p = 0; table_used--; return mapped; }
                const size_t home = (table[j].p >> 12) * 0x9E3779B97F4A7C15ull % table_cap;
                const bool between = i <= j ? (home > i && home <= j) : (home > i || home <= j);
                if (!between) break;
            }
            table[i] = table[j]; i = j;
        }
    }

    void* take(size_t bytes) {
        size_t rounded;
        const int c = size_class(bytes, &rounded);
        if (c < 0) return nullptr;
        {
            std::lock_guard<std::mutex> g(m);
            if (free_block* b = free_list[c]) {
                if (!table_put(reinterpret_cast<uintptr_t>(b), b->mapped, c)) return nullptr;   // (stays cached; the caller falls back to malloc)
                free_list[c] = b->next; cached_bytes -= b->mapped; n_reuses++;
                return b;
            }
        }
        const size_t mapped = (rounded + kHuge - 1) & ~(kHuge - 1);
        void* p = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        (void)madvise(p, mapped, MADV_HUGEPAGE);
        std::lock_guard<std::mutex> g(m);
        if (!table_put(reinterpret_cast<uintptr_t>(p), mapped, c)) { munmap(p, mapped); return nullptr; }
        n_maps++;
        return p;
    }

    bool give(void* p) {   // false = not a pool block
        size_t mapped; int c = 0;
        {
            std::lock_guard<std::mutex> g(m);
            mapped = table_take(reinterpret_cast<uintptr_t>(p), &c);
            if (!mapped) return false;
            if (cached_bytes + mapped <= cap_bytes) {
                free_block* b = static_cast<free_block*>(p);
                b->mapped = mapped; b->cls = c; b->next = free_list[c]; free_list[c] = b; cached_bytes += mapped;
                return true;
            }
        }
        munmap(p, mapped);
        return true;
    }
};

pool& the_pool() { static pool* p = new (std::malloc(sizeof(pool))) pool(); return *p; }   // never destroyed: blocks may be freed during process exit

inline void* pool_new(size_t bytes, size_t align) {
    if (bytes >= kLarge && align <= kPage) {
        pool& P = the_pool();
        P.configure();
        if (P.state.load(std::memory_order_relaxed) == 1) if (void* p = P.take(bytes)) return p;
    }
    if (align > alignof(std::max_align_t)) { void* p = nullptr; return posix_memalign(&p, align, bytes ? bytes : 1) == 0 ? p : nullptr; }
    return std::malloc(bytes ? bytes : 1);
}
inline void pool_delete(void* p) {
    if (!p) return;
    if ((reinterpret_cast<uintptr_t>(p) & (kPage - 1)) == 0 && the_pool().give(p)) return;   // malloc never returns a page-aligned pointer for a block it mapped itself (header in front)
    std::free(p);
}

} // namespace

// Gives every cached block back to the kernel (blocks in use are not touched; later frees are cached again up to the cap). Returns the bytes released.
extern "C" __attribute__((visibility("default"))) uint64_t bu_host_pool_trim(void) {
    pool& P = the_pool();
    free_block* chain = nullptr;
    uint64_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(P.m);
        for (int c = 0; c < kClasses; c++) {
            while (free_block* b = P.free_list[c]) { P.free_list[c] = b->next; b->next = chain; chain = b; }
        }
        bytes = P.cached_bytes;
        P.cached_bytes = 0;
    }
    while (chain) { free_block* b = chain; chain = b->next; munmap(b, b->mapped); }   // (outside the lock: unmapping gigabytes takes a while)
    return bytes;
}

extern "C" __attribute__((visibility("default"))) void bu_host_pool_stats(uint64_t out[4]) {
    pool& P = the_pool();
    std::lock_guard<std::mutex> g(P.m);
    out[0] = P.n_maps; out[1] = P.n_reuses; out[2] = P.cached_bytes; out[3] = P.cap_bytes;
}

void* operator new(size_t n) { if (void* p = pool_new(n, 1)) return p; throw std::bad_alloc(); }
void* operator new[](size_t n) { if (void* p = pool_new(n, 1)) return p; throw std::bad_alloc(); }
void* operator new(size_t n, const std::nothrow_t&) noexcept { return pool_new(n, 1); }
void* operator new[](size_t n, const std::nothrow_t&) noexcept { return pool_new(n, 1); }
void* operator new(size_t n, std::align_val_t a) { if (void* p = pool_new(n, (size_t)a)) return p; throw std::bad_alloc(); }
void* operator new[](size_t n, std::align_val_t a) { if (void* p = pool_new(n, (size_t)a)) return p; throw std::bad_alloc(); }
void* operator new(size_t n, std::align_val_t a, const std::nothrow_t&) noexcept { return pool_new(n, (size_t)a); }
void* operator new[](size_t n, std::align_val_t a, const std::nothrow_t&) noexcept { return pool_new(n, (size_t)a); }
void operator delete(void* p) noexcept { pool_delete(p); }
void operator delete[](void* p) noexcept { pool_delete(p); }
void operator delete(void* p, size_t) noexcept { pool_delete(p); }
void operator delete[](void* p, size_t) noexcept { pool_delete(p); }
void operator delete(void* p, const std::nothrow_t&) noexcept { pool_delete(p); }
void operator delete[](void* p, const std::nothrow_t&) noexcept { pool_delete(p); }
void operator delete(void* p, std::align_val_t) noexcept { pool_delete(p); }
void operator delete[](void* p, std::align_val_t) noexcept { pool_delete(p); }
void operator delete(void* p, size_t, std::align_val_t) noexcept { pool_delete(p); }
void operator delete[](void* p, size_t, std::align_val_t) noexcept { pool_delete(p); }
void operator delete(void* p, std::align_val_t, const std::nothrow_t&) noexcept { pool_delete(p); }
void operator delete[](void* p, std::align_val_t, const std::nothrow_t&) noexcept { pool_delete(p); }
