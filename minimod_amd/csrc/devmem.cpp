// the library's block-keeping allocator (csrc/devmem.h says why)
#include "devmem.h"
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace mmdev {
namespace {
struct Key { int device; unsigned int flags; size_t bytes; bool operator<(const Key& o) const { return device != o.device ? device < o.device : flags != o.flags ? flags < o.flags : bytes < o.bytes; } };
struct Pool {
    std::mutex mu;
    std::map<Key, std::vector<void*>> kept;
    std::unordered_map<void*, Key> held;
    int64_t held_bytes = 0, kept_bytes = 0;
};
Pool g_dev, g_pin;
std::atomic<int64_t> g_hits{0}, g_misses{0}, g_returned{0};
const bool g_off = std::getenv("MM_DEVMEM_NO_KEEP") != nullptr;       // the driver's own malloc / free, as rounds 1 - 5 had it (tools/cli_stress.py shows the defect with it)
const int g_poison = std::getenv("MM_DEVMEM_POISON") ? std::atoi(std::getenv("MM_DEVMEM_POISON")) : -1;   // fill every block handed out: nothing may count on fresh memory being zero

// what a request is rounded up to: a power of two up to 64 KB, then four steps per doubling (at most a quarter more than asked for)
size_t size_class(size_t b) {
    if (b < 256) return 256;
    size_t p = 256;
    while (p < b) p <<= 1;
    if (p <= (size_t)64 << 10) return p;
    const size_t q = p >> 3;   // p/2 < b <= p: steps of p/8 between them
    return (b + q - 1) / q * q;
}
int64_t give_back(Pool& pool, bool pinned) {   // (lock held)
    int64_t n = 0;
    for (auto& kv : pool.kept) for (void* p : kv.second) { (void)(pinned ? hipHostFree(p) : hipFree(p)); n += (int64_t)kv.first.bytes; g_returned++; }
    pool.kept.clear();
    pool.kept_bytes = 0;
    return n;
}
hipError_t take(Pool& pool, bool pinned, void** p, size_t bytes, unsigned int flags) {
    if (!p) return hipErrorInvalidValue;
    *p = nullptr;
    int device = 0;
    (void)hipGetDevice(&device);
    const Key k{device, flags, g_off ? (bytes ? bytes : 16) : size_class(bytes)};   // (pinned blocks by device too: they were made with that device current)
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.kept.find(k);
    if (it != pool.kept.end() && !it->second.empty()) {
        *p = it->second.back(); it->second.pop_back();
        pool.kept_bytes -= (int64_t)k.bytes; g_hits++;
    } else {
        hipError_t e = pinned ? hipHostMalloc(p, k.bytes, flags) : hipMalloc(p, k.bytes);
        if (e != hipSuccess && pool.kept_bytes > 0) {   // the driver has nothing left: what is kept goes back (the one case in which an address may come round again)
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            give_back(pool, pinned);
            e = pinned ? hipHostMalloc(p, k.bytes, flags) : hipMalloc(p, k.bytes);
        }
        if (e != hipSuccess) { *p = nullptr; return e; }
        g_misses++;
    }
    pool.held[*p] = k;
    pool.held_bytes += (int64_t)k.bytes;
    if (g_poison >= 0) { if (pinned) std::memset(*p, g_poison, k.bytes); else { (void)hipMemset(*p, g_poison, k.bytes); (void)hipDeviceSynchronize(); } }   // (waited for: the caller's own initialisation may run on a stream the NULL stream does not order)
    return hipSuccess;
}
hipError_t put(Pool& pool, bool pinned, void* p) {
    if (!p) return hipSuccess;
    int owner = -1;
    { std::lock_guard<std::mutex> lk(pool.mu); auto it = pool.held.find(p); if (it != pool.held.end()) owner = it->second.device; }
    // hipFree's and hipHostFree's meaning: nothing that is queued or running uses the block once this returns -- on the device the block
    // was made for (a process with handles on two GPUs frees one's memory while the other is current)
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (owner >= 0 && owner != cur) (void)hipSetDevice(owner);
    const hipError_t se = hipDeviceSynchronize();
    if (owner >= 0 && owner != cur) (void)hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.held.find(p);
    if (it == pool.held.end()) return pinned ? hipHostFree(p) : hipFree(p);   // (not one of ours)
    const Key k = it->second;
    pool.held.erase(it);
    pool.held_bytes -= (int64_t)k.bytes;
    if (g_off) { g_returned++; return pinned ? hipHostFree(p) : hipFree(p); }
    pool.kept[k].push_back(p);
    pool.kept_bytes += (int64_t)k.bytes;
    return se;
}
}  // namespace

hipError_t dmalloc(void** p, size_t bytes) { return take(g_dev, false, p, bytes, 0u); }
hipError_t dfree(void* p) { return put(g_dev, false, p); }
hipError_t hmalloc(void** p, size_t bytes, unsigned int flags) { return take(g_pin, true, p, bytes, flags); }
hipError_t hfree(void* p) { return put(g_pin, true, p); }
}  // namespace mmdev

extern "C" void mm_devmem_stats(int64_t out[7]) {
    using namespace mmdev;
    std::lock_guard<std::mutex> a(g_dev.mu);
    std::lock_guard<std::mutex> b(g_pin.mu);
    out[0] = g_dev.held_bytes; out[1] = g_dev.kept_bytes; out[2] = g_pin.held_bytes; out[3] = g_pin.kept_bytes; out[4] = g_hits; out[5] = g_misses; out[6] = g_returned;
}
extern "C" int64_t mm_devmem_trim(void) {
    using namespace mmdev;
    (void)hipDeviceSynchronize();
    int64_t n = 0;
    { std::lock_guard<std::mutex> a(g_dev.mu); n += give_back(g_dev, false); }
    { std::lock_guard<std::mutex> b(g_pin.mu); n += give_back(g_pin, true); }
    return n;
}
