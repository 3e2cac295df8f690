// bgzf_api.hip -- host side of the device BGZF inflate (include/minimod_bgzf.h): slots of pinned staging + device buffers, one
// stream per slot so that one slot's copies run beside another's kernels.
#include <hip/hip_runtime.h>
#include "devmem.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "bgzf_kernels.hip.h"
#include "minimod_bgzf.h"

using namespace mmbgzf;

// Workgroups of k_bgzf_inflate a CU is to hold (MM_INFLATE_WGS = 4 .. 7; a workgroup is four wavefronts of 5.6 KB of LDS each): the kernel is
// bound by how many wavefronts' chains of dependent steps a SIMD interleaves, not by issue slots or memory, so more of
// them decode more -- up to where nothing else fits beside them: the record framing and k_stream_reads (22.6 KB of LDS, 72 registers) run
// beside the inflate in a `minimod freq --gpu-ingest`.  The launch asks for dynamic LDS it never touches to hold a CU at that many.
static int inflate_wgs_per_cu() {
    static int w = 0;
    if (!w) { const char* e = std::getenv("MM_INFLATE_WGS"); w = e ? std::atoi(e) : 6; if (w < 1) w = 1; if (w > 7) w = 7; }
    return w;
}
static unsigned inflate_lds_pad() {
    const int w = inflate_wgs_per_cu();
    const size_t fixed = sizeof(WaveLds) * kWaves, want = w >= 7 ? fixed : ((size_t)163840 / (size_t)w) / 1024 * 1024;
    return want > fixed ? (unsigned)(want - fixed) : 0u;
}
static_assert(sizeof(mm_bgzf_block_t) == sizeof(Block), "the ABI's block record is the kernels'");

namespace {
struct BSlot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t* h_c = nullptr;           // pinned
    mm_bgzf_block_t* h_blocks = nullptr;
    int32_t* h_status = nullptr;
    uint8_t* d_c = nullptr;
    uint8_t* d_out = nullptr;
    Block* d_blocks = nullptr;
    int32_t* d_status = nullptr;
    int n_blocks = 0;
    bool busy = false;
};
}  // namespace

struct mm_bgzf {
    int device = 0, n_cu = 0;
    int max_blocks = 0;
    size_t max_cbytes = 0, max_obytes = 0;
    std::vector<BSlot> slots;
};

#define BCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::snprintf(err ? err : dummy, err ? err_len : sizeof dummy, "%s: %s", #x, hipGetErrorString(e_)); mm_bgzf_destroy(h); return nullptr; } } while (0)

extern "C" {

mm_bgzf_t* mm_bgzf_create(int32_t device, int32_t slots, int32_t max_blocks, size_t max_cbytes, size_t max_obytes, char* err, size_t err_len) {
    char dummy[8];
    if (err && err_len) err[0] = 0;
    if (slots < 1 || slots > 16 || max_blocks < 1 || max_cbytes == 0 || max_obytes == 0 || max_cbytes >= 0xFFFF0000ull || max_obytes >= 0xFFFF0000ull) {
        if (err) std::snprintf(err, err_len, "mm_bgzf_create: arguments out of range");
        return nullptr;
    }
    mm_bgzf* h = new mm_bgzf();
    h->device = device; h->max_blocks = max_blocks; h->max_cbytes = max_cbytes; h->max_obytes = max_obytes;
    BCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    BCHK(hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount;
    h->slots.resize((size_t)slots);
    // the lowest stream priority there is: the inflate's workgroups run for milliseconds, and what shares the device with them
    // (the freq path's launches and copies) should get the CUs they leave first
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (BSlot& s : h->slots) {
        BCHK(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio_least));
        for (auto& e : s.ev) BCHK(hipEventCreate(&e));
        BCHK(mmdev::hmalloc((void**)&s.h_c, max_cbytes + 64, hipHostMallocDefault));
        BCHK(mmdev::hmalloc((void**)&s.h_blocks, sizeof(mm_bgzf_block_t) * (size_t)max_blocks, hipHostMallocDefault));
        BCHK(mmdev::hmalloc((void**)&s.h_status, sizeof(int32_t) * (size_t)max_blocks, hipHostMallocDefault));
        BCHK(mmdev::dmalloc((void**)&s.d_c, max_cbytes + kPad));
        BCHK(hipMemset(s.d_c, 0, max_cbytes + kPad));
        BCHK(mmdev::dmalloc((void**)&s.d_out, max_obytes + 64));
        BCHK(mmdev::dmalloc((void**)&s.d_blocks, sizeof(Block) * (size_t)max_blocks));
        BCHK(mmdev::dmalloc((void**)&s.d_status, sizeof(int32_t) * (size_t)max_blocks));
    }
    return h;
}

void mm_bgzf_destroy(mm_bgzf_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (BSlot& s : h->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        for (auto& e : s.ev) if (e) (void)hipEventDestroy(e);
        if (s.h_c) (void)mmdev::hfree(s.h_c);
        if (s.h_blocks) (void)mmdev::hfree(s.h_blocks);
        if (s.h_status) (void)mmdev::hfree(s.h_status);
        void* ds[] = {s.d_c, s.d_out, s.d_blocks, s.d_status};
        for (void* p : ds) if (p) (void)mmdev::dfree(p);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete h;
}

void* mm_bgzf_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (mmdev::hmalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void mm_bgzf_host_free(void* p) { if (p) (void)mmdev::hfree(p); }

uint8_t* mm_bgzf_staging(mm_bgzf_t* h, int32_t slot) { return (h && slot >= 0 && (size_t)slot < h->slots.size()) ? h->slots[(size_t)slot].h_c : nullptr; }
mm_bgzf_block_t* mm_bgzf_blocks(mm_bgzf_t* h, int32_t slot) { return (h && slot >= 0 && (size_t)slot < h->slots.size()) ? h->slots[(size_t)slot].h_blocks : nullptr; }

int32_t mm_bgzf_submit(mm_bgzf_t* h, int32_t slot, int32_t n_blocks, size_t cbytes, size_t obytes, uint8_t* out_host) {
    if (!h || slot < 0 || (size_t)slot >= h->slots.size() || n_blocks < 0 || n_blocks > h->max_blocks || cbytes > h->max_cbytes || obytes > h->max_obytes || (!out_host && obytes)) return -1;
    BSlot& s = h->slots[(size_t)slot];
    if (s.busy) return -2;
    // the kernels trust the block records no further than the slot's buffers
    for (int i = 0; i < n_blocks; i++) {
        const mm_bgzf_block_t& b = s.h_blocks[i];
        if ((size_t)b.c_off + b.c_len > cbytes || (size_t)b.o_off + b.isize > obytes || b.isize > 65536u) return -3;
    }
    if (hipSetDevice(h->device) != hipSuccess) return -4;
    s.n_blocks = n_blocks;
    if (n_blocks == 0) { s.busy = true; return hipEventRecord(s.ev[4], s.stream) == hipSuccess ? 0 : -4; }
#define SCHK(x) do { if ((x) != hipSuccess) return -4; } while (0)
    SCHK(hipEventRecord(s.ev[0], s.stream));
    SCHK(hipMemcpyAsync(s.d_c, s.h_c, cbytes, hipMemcpyHostToDevice, s.stream));
    SCHK(hipMemcpyAsync(s.d_blocks, s.h_blocks, sizeof(Block) * (size_t)n_blocks, hipMemcpyHostToDevice, s.stream));
    SCHK(hipEventRecord(s.ev[1], s.stream));
    const int wgs = std::max(1, std::min(h->n_cu * inflate_wgs_per_cu(), (n_blocks + kWaves - 1) / kWaves));
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(wgs), dim3(64 * kWaves), inflate_lds_pad(), s.stream, s.d_c, s.d_blocks, n_blocks, s.d_out, s.d_status);
    SCHK(hipEventRecord(s.ev[2], s.stream));
    hipLaunchKernelGGL(k_bgzf_crc, dim3(std::max(1, std::min(h->n_cu * 8, (n_blocks + 3) / 4))), dim3(256), 0, s.stream, s.d_out, s.d_blocks, n_blocks, s.d_status);
    SCHK(hipEventRecord(s.ev[3], s.stream));
    SCHK(hipMemcpyAsync(out_host, s.d_out, obytes, hipMemcpyDeviceToHost, s.stream));
    SCHK(hipMemcpyAsync(s.h_status, s.d_status, sizeof(int32_t) * (size_t)n_blocks, hipMemcpyDeviceToHost, s.stream));
    SCHK(hipEventRecord(s.ev[4], s.stream));
    SCHK(hipGetLastError());
#undef SCHK
    s.busy = true;
    return 0;
}

#ifndef MM_SOURCE_HASH
#define MM_SOURCE_HASH "unstamped"
#endif
// (the marker in front is what minimod_amd/build.py looks for in the file's bytes: the stamp is read without loading the library)
static const char kSourceStamp[] = "MMSRCHASH=" MM_SOURCE_HASH;
const char* mm_build_source_hash(void) { return kSourceStamp + 10; }
// The runtime's start AND what it puts off until first use -- a queue, a copy engine, this library's code object: one stream, one
// four-byte copy, one empty launch.  Once per device; later callers wait for the first and return.
__global__ void k_warm(int* p) { if (p && threadIdx.x == 12345u) *p = 0; }
int32_t mm_hip_warm(int32_t device) {
    static std::mutex mu;
    static int done[64];
    std::lock_guard<std::mutex> g(mu);
    if (device >= 0 && device < 64 && done[device]) return done[device] > 0 ? 0 : -4;
    bool ok = hipSetDevice(device) == hipSuccess && mmdev::dfree(nullptr) == hipSuccess;
    if (ok) {
        hipStream_t st = nullptr;
        int* d = nullptr;
        int v = 0;
        // (on the null stream, whose queue the handles' synchronous copies need anyway: a stream of its own here was one more hardware queue --
        // 8 - 12 ms to make, 173 MB of host memory for its waves' saved state, tools/exit_probe_streams.hip)
        ok = (!std::getenv("MM_WARM_OWN_STREAM") || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) && mmdev::dmalloc((void**)&d, 64) == hipSuccess &&
             hipMemcpyAsync(d, &v, sizeof v, hipMemcpyHostToDevice, st) == hipSuccess;
        if (ok) { hipLaunchKernelGGL(k_warm, dim3(1), dim3(64), 0, st, d); ok = hipStreamSynchronize(st) == hipSuccess; }
        if (d) (void)mmdev::dfree(d);
        if (st) (void)hipStreamDestroy(st);
    }
    if (device >= 0 && device < 64) done[device] = ok ? 1 : -1;
    return ok ? 0 : -4;
}

int32_t mm_bgzf_inflate_device(int32_t device, void* stream, const uint8_t* d_c, const mm_bgzf_block_t* d_blocks, int32_t n_blocks, uint8_t* d_out, int32_t* d_status,
                               void* between_event) {
    static int n_cu_of[64];   // (0: not asked yet)
    if (device < 0 || device >= 64 || n_blocks < 0) return -1;
    if (!n_cu_of[device]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -4;
        n_cu_of[device] = prop.multiProcessorCount;
    }
    const int n_cu = n_cu_of[device];
    hipStream_t st = (hipStream_t)stream;
    if (n_blocks) hipLaunchKernelGGL(k_bgzf_inflate, dim3(std::max(1, std::min(n_cu * inflate_wgs_per_cu(), (n_blocks + kWaves - 1) / kWaves))), dim3(64 * kWaves), inflate_lds_pad(), st, d_c,
                                     reinterpret_cast<const Block*>(d_blocks), n_blocks, d_out, d_status);
    if (between_event && hipEventRecord((hipEvent_t)between_event, st) != hipSuccess) return -4;
    if (n_blocks) hipLaunchKernelGGL(k_bgzf_crc, dim3(std::max(1, std::min(n_cu * 8, (n_blocks + 3) / 4))), dim3(256), 0, st, d_out, reinterpret_cast<const Block*>(d_blocks), n_blocks, d_status);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int32_t mm_bgzf_wait(mm_bgzf_t* h, int32_t slot, const int32_t** status) {
    if (!h || slot < 0 || (size_t)slot >= h->slots.size()) return -1;
    BSlot& s = h->slots[(size_t)slot];
    if (!s.busy) return -2;
    if (hipEventSynchronize(s.ev[4]) != hipSuccess) return -4;
    s.busy = false;
    if (status) *status = s.h_status;
    return 0;
}

int32_t mm_bgzf_times(mm_bgzf_t* h, int32_t slot, float ms[4]) {
    if (!h || slot < 0 || (size_t)slot >= h->slots.size() || !ms) return -1;
    BSlot& s = h->slots[(size_t)slot];
    if (s.busy || s.n_blocks == 0) return -2;
    for (int i = 0; i < 4; i++) if (hipEventElapsedTime(&ms[i], s.ev[i], s.ev[i + 1]) != hipSuccess) return -4;
    return 0;
}

}  // extern "C"
