// ingest_api.hip -- host side of the device-side BAM ingestion (include/minimod_ingest.h): group slots (pinned staging, device
// buffers, a stream each for the copies and the inflate), arenas (a batch's pools), one chain stream for framing + flattening.
#include <hip/hip_runtime.h>
#include "devmem.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "ingest_kernels.hip.h"
#include "minimod_ingest.h"

using namespace mmingest;

namespace {
struct GSlot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // copy start, inflate start, CRC start, inflated
    hipEvent_t ev_f0 = nullptr, ev_done = nullptr;             // around frame + flatten
    uint8_t* h_c = nullptr; mm_bgzf_block_t* h_blocks = nullptr; int32_t* h_status = nullptr; Result* h_result = nullptr;   // pinned
    uint8_t *d_c = nullptr, *d_out = nullptr;
    Block* d_blocks = nullptr; int32_t* d_status = nullptr;
    uint32_t *d_tab = nullptr;       // six tables of max_blocks + 1 words
    uint32_t *d_rec_off = nullptr, *d_acc = nullptr, *d_info = nullptr;
    Desc* d_desc = nullptr;
    Result* d_result = nullptr;
    int n_blocks = 0; size_t cbytes = 0, obytes = 0;
    uint64_t seq = 0;                // the group's number
    bool inflating = false, flattening = false, timed = false;
    std::atomic<int> ready{0};       // 0: its buffers are still being made (the slots behind the first: on a thread of their own), 1: made, -1: could not be
};
struct Arena {
    mm_read_t* reads = nullptr; uint8_t *cigar = nullptr, *seq = nullptr, *mm = nullptr, *ml = nullptr;
    uint64_t cap_reads = 0, cap_cigar = 0, cap_seq = 0, cap_mm = 0, cap_ml = 0;
    uint8_t* names = nullptr; uint64_t* name_off = nullptr; uint64_t cap_names = 0;   // opts.names only
};
}  // namespace

struct mm_ingest {
    mm_ingest_opts_t o;
    int device = 0, n_cu = 0;
    uint32_t H = 0, max_obytes = 0, max_records = 0;
    hipStream_t chain = nullptr;
    std::vector<GSlot> slots;
    std::vector<Arena> arenas;
    Carry* d_carry = nullptr;      // [2]
    Cursor* d_cursor = nullptr;    // [2]
    uint8_t* d_tail[2] = {nullptr, nullptr};
    uint64_t next_seq = 0;         // groups numbered by mm_ingest_inflate
    uint64_t flat_seq = 0;         // the next group to be flattened for the first time
    int prio_least = 0;
    std::vector<std::thread> slot_makers;   // make the slots behind the first while the caller's reader stages its first group (a thread a slot: the pinning, the queue and the allocations of one slot are mostly the kernel's work and run beside another's)
    CodeTab* d_codes = nullptr;    // mm_ingest_batch_codes (made at its first call)
    CodeTab* h_codes = nullptr;    // pinned
};

static hipError_t slot_alloc(mm_ingest* h, GSlot& s) {
#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)
    const size_t nb1 = (size_t)h->o.max_blocks + 1;
    SCHK(hipSetDevice(h->device));
    // the lowest priority: the inflate's workgroups run for milliseconds; the chain's and the freq path's kernels get the CUs they leave first
    // (a stream with a CU mask that keeps the inflate off a few CUs -- hipExtStreamCreateWithCUMask -- hung the first launch on this
    // pool's boxes: not used.  Instead the inflate's workgroups are sized so that four of them leave room on every CU, below.)
    if (std::getenv("MM_INGEST_PLAIN_STREAMS")) SCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    else SCHK(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, h->prio_least));
    for (auto& e : s.ev) SCHK(hipEventCreate(&e));
    SCHK(hipEventCreate(&s.ev_f0)); SCHK(hipEventCreate(&s.ev_done));
    SCHK(mmdev::hmalloc((void**)&s.h_c, h->o.max_cbytes + 64, hipHostMallocDefault));
    SCHK(mmdev::hmalloc((void**)&s.h_blocks, sizeof(mm_bgzf_block_t) * (size_t)h->o.max_blocks, hipHostMallocDefault));
    SCHK(mmdev::hmalloc((void**)&s.h_status, sizeof(int32_t) * (size_t)h->o.max_blocks, hipHostMallocDefault));
    SCHK(mmdev::hmalloc((void**)&s.h_result, sizeof(Result), hipHostMallocDefault));
    SCHK(mmdev::dmalloc((void**)&s.d_c, h->o.max_cbytes + 4096));   // (readable bytes behind the payloads: the inflate's window runs ahead)
    SCHK(hipMemset(s.d_c, 0, h->o.max_cbytes + 4096));
    SCHK(mmdev::dmalloc((void**)&s.d_out, (size_t)h->H + h->max_obytes + 256));
    SCHK(hipMemset(s.d_out + (size_t)h->H + h->max_obytes, 0, 256));
    SCHK(mmdev::dmalloc((void**)&s.d_blocks, sizeof(Block) * (size_t)h->o.max_blocks));
    SCHK(mmdev::dmalloc((void**)&s.d_status, sizeof(int32_t) * (size_t)h->o.max_blocks));
    SCHK(mmdev::dmalloc((void**)&s.d_tab, 6 * nb1 * sizeof(uint32_t)));
    SCHK(mmdev::dmalloc((void**)&s.d_rec_off, sizeof(uint32_t) * (size_t)h->max_records));
    SCHK(mmdev::dmalloc((void**)&s.d_acc, sizeof(uint32_t) * (size_t)h->max_records));
    SCHK(mmdev::dmalloc((void**)&s.d_info, sizeof(uint32_t) * (size_t)h->max_records));
    SCHK(mmdev::dmalloc((void**)&s.d_desc, sizeof(Desc) * (size_t)h->max_records));
    SCHK(mmdev::dmalloc((void**)&s.d_result, sizeof(Result)));
    return hipSuccess;
#undef SCHK
}

// A process that leaves through exit() while a handle's slot maker is still inside the HIP runtime would tear the runtime down under it:
// the makers of the live handles are waited for first (an atexit handler registered behind the runtime's own runs in front of it).
static std::mutex g_live_mu;
static std::vector<mm_ingest*> g_live;
static void join_slot_makers() {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (mm_ingest* h : g_live) for (std::thread& t : h->slot_makers) if (t.joinable()) t.join();
}
static void live_add(mm_ingest* h) {
    static bool registered = false;
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!registered) { std::atexit(join_slot_makers); registered = true; }
    g_live.push_back(h);
}
static void live_remove(mm_ingest* h) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), h), g_live.end());
}

#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::snprintf(err ? err : dummy, err ? err_len : sizeof dummy, "%s: %s", #x, hipGetErrorString(e_)); mm_ingest_destroy(h); return nullptr; } } while (0)
#define RCHK(x) do { if ((x) != hipSuccess) return -MM_INGEST_E_HIP; } while (0)

extern "C" {

const char* mm_ingest_strerror(int32_t code) {
    switch (code < 0 ? -code : code) {
        case MM_INGEST_OK: return "ok";
        case MM_INGEST_E_RECORD: return "a BAM record the reader refuses (truncated or corrupt file)";
        case MM_INGEST_E_ARENA: return "the batch's pools are full";
        case MM_INGEST_E_TAIL: return "a BAM record longer than the device reader's head room";
        case MM_INGEST_E_RECORDS: return "more records in a group of blocks than the device reader's tables hold";
        case MM_INGEST_E_HEADER: return "the BAM header does not end inside the first group of blocks";
        case MM_INGEST_E_CODES: return "modification codes the device-side census does not take (longer than 8 characters, or more than 1024 of them)";
        case MM_INGEST_E_ARG: return "bad argument";
        case MM_INGEST_E_HIP: return "HIP runtime error";
        case MM_INGEST_E_ORDER: return "groups out of order";
        default: return "unknown error";
    }
}

mm_ingest_t* mm_ingest_create(const mm_ingest_opts_t* opts, char* err, size_t err_len) {
    char dummy[8];
    if (err && err_len) err[0] = 0;
    if (!opts) return nullptr;
    mm_ingest* h = new mm_ingest();
    h->o = *opts;
    if (h->o.group_slots <= 0) h->o.group_slots = 4;
    if (h->o.max_blocks <= 0) h->o.max_blocks = 2048;
    if (h->o.arenas <= 0) h->o.arenas = 3;
    if (h->o.max_cbytes == 0) h->o.max_cbytes = (uint64_t)48 << 20;
    if (h->o.arena_bytes == 0) h->o.arena_bytes = (uint64_t)1 << 30;
    if (h->o.head_room == 0) h->o.head_room = (uint64_t)32 << 20;
    h->o.head_room = (h->o.head_room + 255) & ~(uint64_t)255;
    const uint64_t max_ob = (uint64_t)h->o.max_blocks * 65536ull;
    if (h->o.group_slots > 16 || h->o.arenas > 16 || h->o.max_blocks > (1 << 15) || h->o.head_room + max_ob >= 0xF0000000ull || h->o.max_cbytes >= 0xF0000000ull ||
        h->o.arena_bytes < max_ob + h->o.head_room) {
        if (err) std::snprintf(err, err_len, "mm_ingest_create: arguments out of range");
        delete h;
        return nullptr;
    }
    h->device = h->o.device; h->H = (uint32_t)h->o.head_room; h->max_obytes = (uint32_t)max_ob;
    h->max_records = (uint32_t)((h->o.head_room + max_ob) / 48);
    const bool tl = std::getenv("MM_TIMELINE") != nullptr;
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    auto big_maps = [&](const char* where) {   // (MM_TIMELINE=2: the resident mappings of 64 MB or more)
        if (!tl || std::atoi(std::getenv("MM_TIMELINE")) < 2) return;
        int n = 0; double mb = 0; char line[512]; unsigned long kb;
        if (FILE* f = std::fopen("/proc/self/smaps", "r")) { while (std::fgets(line, sizeof line, f)) if (std::sscanf(line, "Rss: %lu kB", &kb) == 1 && kb >= 65536) { n++; mb += (double)kb / 1024.0; } std::fclose(f); }
        std::fprintf(stderr, "[timeline]          mm_ingest_create, %s: %d big mappings, %.0f MB resident in them\n", where, n, mb);
    };
    const double t0 = now();
    big_maps("start");
    ICHK(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    ICHK(hipGetDeviceProperties(&prop, h->device));
    h->n_cu = prop.multiProcessorCount;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    ICHK(hipStreamCreateWithFlags(&h->chain, hipStreamNonBlocking));
    ICHK(mmdev::dmalloc((void**)&h->d_carry, 2 * sizeof(Carry)));
    ICHK(hipMemset(h->d_carry, 0, 2 * sizeof(Carry)));
    ICHK(mmdev::dmalloc((void**)&h->d_cursor, 2 * sizeof(Cursor)));
    ICHK(hipMemset(h->d_cursor, 0, 2 * sizeof(Cursor)));
    for (int k = 0; k < 2; k++) ICHK(mmdev::dmalloc((void**)&h->d_tail[k], h->H));
    const double t1 = now();
    big_maps("runtime + streams");
    h->slots = std::vector<GSlot>((size_t)h->o.group_slots);
    h->prio_least = prio_least;
    // the FIRST slot here; the others (48 MiB of pinned staging and ~230 MB of device memory each: 25 - 40 ms) on a thread of their own, beside the
    // arenas below, the caller's handle set-up and the reader's first group -- mm_ingest_staging waits for a slot that is not made yet
    ICHK(slot_alloc(h, h->slots[0]));
    h->slots[0].ready.store(1);
    const double t2 = now();
    big_maps("first slot");
    h->arenas.resize((size_t)h->o.arenas);
    for (Arena& a : h->arenas) {
        // a batch is made of about arena_bytes of decoded stream, and a BAM's records are mostly sequence (a nibble a base) and
        // qualities (not kept): the pools get fixed shares of that -- sequence a half, CIGARs and MM text a quarter each, ML an eighth
        // (what is allocated is also what the process's end has to unmap) -- and a group that does not fit is answered with
        // MM_INGEST_E_ARENA: the caller closes the batch and runs the group again into an empty arena
        const uint64_t D = h->o.arena_bytes;
        a.cap_reads = D / 512 + 4096;
        // (no pool smaller than what ONE group can decode to, head room included: whatever a group's bytes are made of -- all MM text, all
        // ML -- it fits an empty arena, so the "run the group again into an empty arena" answer to MM_INGEST_E_ARENA always ends)
        const uint64_t one = max_ob + h->o.head_room + (1 << 20);
        a.cap_cigar = std::max<uint64_t>(D / 4 + (1 << 20), one); a.cap_seq = std::max<uint64_t>(D / 2 + (1 << 20), one);
        a.cap_mm = std::min<uint64_t>(std::max<uint64_t>(D / 4 + (1 << 20), one), 0xFFFFF000ull); a.cap_ml = std::max<uint64_t>(D / 8 + (1 << 20), one);
        a.cap_reads = std::max<uint64_t>(a.cap_reads, one / 36 + 4096);   // (a record is at least 36 bytes of stream)
        ICHK(mmdev::dmalloc((void**)&a.reads, sizeof(mm_read_t) * a.cap_reads));
        ICHK(mmdev::dmalloc((void**)&a.cigar, a.cap_cigar)); ICHK(mmdev::dmalloc((void**)&a.seq, a.cap_seq));
        ICHK(mmdev::dmalloc((void**)&a.mm, a.cap_mm)); ICHK(mmdev::dmalloc((void**)&a.ml, a.cap_ml));
        if (h->o.names) {   // a name is at most 255 of its record's bytes; an eighth of the stream holds the names of any BAM a sequencer's pipeline writes,
            a.cap_names = std::max<uint64_t>(D / 8 + (1 << 20), one);   // and never less than what one group can hold (the answer to "full" is an empty arena)
            ICHK(mmdev::dmalloc((void**)&a.names, a.cap_names));
            ICHK(mmdev::dmalloc((void**)&a.name_off, sizeof(uint64_t) * a.cap_reads));
        }
    }
    big_maps("arenas");
    live_add(h);
    if (std::getenv("MM_INGEST_SERIAL_SLOTS")) {
        if (h->slots.size() > 1) h->slot_makers.emplace_back([h]() { for (size_t i = 1; i < h->slots.size(); i++) h->slots[i].ready.store(slot_alloc(h, h->slots[i]) == hipSuccess ? 1 : -1); });
    } else {
        for (size_t i = 1; i < h->slots.size(); i++) h->slot_makers.emplace_back([h, i]() { h->slots[i].ready.store(slot_alloc(h, h->slots[i]) == hipSuccess ? 1 : -1); });
    }
    ICHK(hipDeviceSynchronize());
    if (tl) std::fprintf(stderr, "[timeline] mm_ingest_create: runtime + streams %.3f s, group slots %.3f s, arenas %.3f s\n", t1 - t0, t2 - t1, now() - t2);
    return h;
}

void mm_ingest_destroy(mm_ingest_t* h) {
    if (!h) return;
    live_remove(h);
    for (std::thread& t : h->slot_makers) if (t.joinable()) t.join();
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (GSlot& s : h->slots) {
        for (auto& e : s.ev) if (e) (void)hipEventDestroy(e);
        if (s.ev_f0) (void)hipEventDestroy(s.ev_f0);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        void* hs[] = {s.h_c, s.h_blocks, s.h_status, s.h_result};
        for (void* p : hs) if (p) (void)mmdev::hfree(p);
        void* ds[] = {s.d_c, s.d_out, s.d_blocks, s.d_status, s.d_tab, s.d_rec_off, s.d_acc, s.d_info, s.d_desc, s.d_result};
        for (void* p : ds) if (p) (void)mmdev::dfree(p);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    for (Arena& a : h->arenas) { void* ds[] = {a.reads, a.cigar, a.seq, a.mm, a.ml, a.names, a.name_off}; for (void* p : ds) if (p) (void)mmdev::dfree(p); }
    if (h->d_codes) (void)mmdev::dfree(h->d_codes);
    if (h->h_codes) (void)mmdev::hfree(h->h_codes);
    void* ds[] = {h->d_carry, h->d_cursor, h->d_tail[0], h->d_tail[1]};
    for (void* p : ds) if (p) (void)mmdev::dfree(p);
    if (h->chain) (void)hipStreamDestroy(h->chain);
    delete h;
}

int32_t mm_ingest_group_slots(const mm_ingest_t* h) { return h ? h->o.group_slots : 0; }
int32_t mm_ingest_max_blocks(const mm_ingest_t* h) { return h ? h->o.max_blocks : 0; }
uint64_t mm_ingest_max_cbytes(const mm_ingest_t* h) { return h ? h->o.max_cbytes : 0; }
uint64_t mm_ingest_arena_bytes(const mm_ingest_t* h) { return h ? h->o.arena_bytes : 0; }
static GSlot* slot_of(mm_ingest_t* h, int32_t slot) {
    if (!h || slot < 0 || (size_t)slot >= h->slots.size()) return nullptr;
    GSlot* s = &h->slots[(size_t)slot];
    while (s->ready.load() == 0) std::this_thread::yield();   // (a slot behind the first is still being made: tens of milliseconds at most)
    return s->ready.load() > 0 ? s : nullptr;
}
uint8_t* mm_ingest_staging(mm_ingest_t* h, int32_t slot) { GSlot* s = slot_of(h, slot); return s ? s->h_c : nullptr; }
mm_bgzf_block_t* mm_ingest_blocks(mm_ingest_t* h, int32_t slot) { GSlot* s = slot_of(h, slot); return s ? s->h_blocks : nullptr; }
void* mm_ingest_stream(mm_ingest_t* h) { return h ? (void*)h->chain : nullptr; }

int32_t mm_ingest_inflate(mm_ingest_t* h, int32_t slot, int32_t n_blocks, size_t cbytes, size_t obytes) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || n_blocks < 0 || n_blocks > h->o.max_blocks || cbytes > h->o.max_cbytes || obytes > h->max_obytes) return -MM_INGEST_E_ARG;
    GSlot& s = *sp;
    if (s.inflating || s.flattening) return -MM_INGEST_E_ORDER;
    size_t at = 0;
    for (int i = 0; i < n_blocks; i++) {   // the kernels trust the block records no further than the slot's buffers; the blocks' outputs lie one behind the other
        const mm_bgzf_block_t& b = s.h_blocks[i];
        if ((size_t)b.c_off + b.c_len > cbytes || b.isize > 65536u || b.o_off != at || at + b.isize > obytes) return -MM_INGEST_E_ARG;
        at += b.isize;
    }
    if (at != obytes) return -MM_INGEST_E_ARG;
    RCHK(hipSetDevice(h->device));
    s.n_blocks = n_blocks; s.cbytes = cbytes; s.obytes = obytes; s.seq = h->next_seq++;
    RCHK(hipEventRecord(s.ev[0], s.stream));
    if (cbytes) RCHK(hipMemcpyAsync(s.d_c, s.h_c, cbytes, hipMemcpyHostToDevice, s.stream));
    if (n_blocks) RCHK(hipMemcpyAsync(s.d_blocks, s.h_blocks, sizeof(Block) * (size_t)n_blocks, hipMemcpyHostToDevice, s.stream));
    RCHK(hipEventRecord(s.ev[1], s.stream));
    if (n_blocks && mm_bgzf_inflate_device(h->device, s.stream, s.d_c, s.d_blocks, n_blocks, s.d_out + h->H, s.d_status, s.ev[2]) != 0) return -MM_INGEST_E_HIP;
    if (!n_blocks) RCHK(hipEventRecord(s.ev[2], s.stream));
    RCHK(hipEventRecord(s.ev[3], s.stream));
    RCHK(hipGetLastError());
    s.inflating = true; s.timed = false;
    return 0;
}

int32_t mm_ingest_flatten(mm_ingest_t* h, int32_t slot, int32_t arena, int32_t new_arena, uint64_t first_skip) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || arena < 0 || (size_t)arena >= h->arenas.size() || first_skip >= 0xFFFFFFFFull) return -MM_INGEST_E_ARG;
    GSlot& s = *sp;
    if (!s.inflating) return -MM_INGEST_E_ORDER;
    if (s.flattening) return -MM_INGEST_E_ORDER;                          // its chain is queued already: mm_ingest_result first
    const bool first_time = s.seq == h->flat_seq;                         // (counted once the launches below have been queued)
    if (!first_time && s.seq + 1 != h->flat_seq) return -MM_INGEST_E_ORDER;   // again is fine as long as no later group has been flattened
    if (first_skip > s.obytes + (uint64_t)h->H) return -MM_INGEST_E_ARG;  // (k_frame_chain adds it to the head room in 32 bits)
    RCHK(hipSetDevice(h->device));
    const Arena& a = h->arenas[(size_t)arena];
    const size_t nb1 = (size_t)h->o.max_blocks + 1;
    const int par = (int)(s.seq & 1u);
    Params P;
    std::memset(&P, 0, sizeof P);
    P.out = s.d_out; P.blocks = s.d_blocks; P.status = s.d_status;
    P.n_blocks = (uint32_t)s.n_blocks; P.H = h->H; P.obytes = (uint32_t)s.obytes; P.n_ref = h->o.n_targets;
    P.first_skip = (uint32_t)first_skip; P.is_first = s.seq == 0 ? 1 : 0;
    P.cand = s.d_tab; P.exit_ = s.d_tab + nb1; P.cnt = s.d_tab + 2 * nb1; P.entry = s.d_tab + 3 * nb1; P.nrec = s.d_tab + 4 * nb1; P.base = s.d_tab + 5 * nb1;
    P.rec_off = s.d_rec_off; P.max_records = h->max_records; P.desc = s.d_desc; P.acc_rec = s.d_acc; P.info = s.d_info;
    P.carry_in = h->d_carry + par; P.carry_out = h->d_carry + (par ^ 1);
    P.cursor_in = h->d_cursor + par; P.cursor_out = h->d_cursor + (par ^ 1);
    P.tail_out = h->d_tail[par ^ 1]; P.result = s.d_result;
    P.allow_secondary = h->o.allow_secondary; P.skip_supplementary = h->o.skip_supplementary;
    P.ranged = h->o.ranged; P.first = h->o.first; P.last = h->o.last; P.lo_tid = h->o.lo_tid; P.hi_tid = h->o.hi_tid; P.lo_pos = h->o.lo_pos; P.hi_pos = h->o.hi_pos;
    P.reads = a.reads; P.cigar = a.cigar; P.seq = a.seq; P.mm = a.mm; P.ml = a.ml;
    P.cap_reads = a.cap_reads; P.cap_cigar = a.cap_cigar; P.cap_seq = a.cap_seq; P.cap_mm = a.cap_mm; P.cap_ml = a.cap_ml;
    P.new_arena = new_arena ? 1 : 0;
    P.names = a.names; P.name_off = a.name_off; P.cap_names = a.cap_names;
    hipStream_t st = h->chain;
    RCHK(hipStreamWaitEvent(st, s.ev[3], 0));
    RCHK(hipEventRecord(s.ev_f0, st));
    hipLaunchKernelGGL(k_tail_in, dim3(64), dim3(256), 0, st, P, (const uint8_t*)h->d_tail[par]);
    if (s.n_blocks) hipLaunchKernelGGL(k_frame_spec, dim3((unsigned)((s.n_blocks + 3) / 4)), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_frame_chain, dim3(1), dim3(64), 0, st, P);
    hipLaunchKernelGGL(k_frame_fill, dim3((unsigned)((s.n_blocks + 1 + 255) / 256)), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_rec_parse, dim3((unsigned)(h->n_cu * 8)), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_rec_scan, dim3(1), dim3(kScanThreads), 0, st, P);
    hipLaunchKernelGGL(k_rec_copy, dim3((unsigned)(h->n_cu * 8)), dim3(256), 0, st, P);
    RCHK(hipGetLastError());
    RCHK(hipMemcpyAsync(s.h_result, s.d_result, sizeof(Result), hipMemcpyDeviceToHost, st));
    if (s.n_blocks) RCHK(hipMemcpyAsync(s.h_status, s.d_status, sizeof(int32_t) * (size_t)s.n_blocks, hipMemcpyDeviceToHost, st));
    RCHK(hipEventRecord(s.ev_done, st));
    s.flattening = true;
    if (first_time) h->flat_seq++;
    return 0;
}

int32_t mm_ingest_result(mm_ingest_t* h, int32_t slot, mm_ingest_result_t* out) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || !out) return -MM_INGEST_E_ARG;
    GSlot& s = *sp;
    if (!s.flattening) return -MM_INGEST_E_ORDER;
    RCHK(hipSetDevice(h->device));
    RCHK(hipEventSynchronize(s.ev_done));
    s.flattening = false; s.timed = true;
    const Result& R = *s.h_result;
    std::memset(out, 0, sizeof *out);
    out->err = R.carry.err; out->err_record = R.carry.err_at;
    out->n_bad_blocks = (int32_t)R.n_bad_blocks; out->status = s.h_status;
    out->n_records = R.n_records; out->n_accepted = R.n_accepted; out->n_slow_blocks = R.n_slow_blocks; out->done = R.carry.done;
    out->total_reads = R.total_reads; out->total_bytes = R.total_bytes; out->processed_bytes = R.processed_bytes;
    out->tail_len = R.carry.tail_len;
    out->batch_reads = R.cursor.n_reads; out->batch_bases = R.cursor.bases;
    out->cigar_bytes = R.cursor.cigar_bytes; out->seq_bytes = R.cursor.seq_bytes; out->mm_bytes = R.cursor.mm_bytes; out->ml_bytes = R.cursor.ml_bytes;
    out->max_n_cigar = R.cursor.max_n_cigar; out->max_l_qseq = R.cursor.max_l_qseq;
    out->qname_bytes = R.cursor.qname_bytes;
    if (R.n_bad_blocks == 0 && R.carry.err != IE_ARENA) s.inflating = false;   // the slot is the caller's again
    return 0;
}

int32_t mm_ingest_group_info(mm_ingest_t* h, int32_t slot, uint32_t* dst, uint32_t n) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || (!dst && n) || n > h->max_records) return -MM_INGEST_E_ARG;
    if (!n) return 0;
    RCHK(hipSetDevice(h->device));
    RCHK(hipMemcpy(dst, sp->d_info, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

int32_t mm_ingest_patch_block(mm_ingest_t* h, int32_t slot, int32_t block, const uint8_t* decoded, size_t n) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || block < 0 || block >= sp->n_blocks || !sp->inflating || sp->flattening || n != sp->h_blocks[block].isize || (!decoded && n)) return -MM_INGEST_E_ARG;
    RCHK(hipSetDevice(h->device));
    if (n) RCHK(hipMemcpy(sp->d_out + h->H + sp->h_blocks[block].o_off, decoded, n, hipMemcpyHostToDevice));
    const int32_t zero = 0;
    RCHK(hipMemcpy(sp->d_status + block, &zero, sizeof zero, hipMemcpyHostToDevice));
    return 0;
}

int32_t mm_ingest_arena_batch(mm_ingest_t* h, int32_t arena, const mm_ingest_result_t* r, mm_batch_t* out) {
    if (!h || arena < 0 || (size_t)arena >= h->arenas.size() || !r || !out) return -MM_INGEST_E_ARG;
    const Arena& a = h->arenas[(size_t)arena];
    std::memset(out, 0, sizeof *out);
    out->reads = a.reads; out->cigar = (const uint32_t*)a.cigar; out->seq = a.seq; out->mm = a.mm; out->ml = a.ml;
    out->n_reads = (int32_t)r->batch_reads;
    out->n_cigar_words = (r->cigar_bytes + 64) / 4; out->n_seq_bytes = r->seq_bytes + 64; out->n_mm_bytes = r->mm_bytes + 64; out->n_ml_bytes = r->ml_bytes + 64;
    out->max_n_cigar = r->max_n_cigar; out->max_l_qseq = r->max_l_qseq;
    return 0;
}

int32_t mm_ingest_arena_names(mm_ingest_t* h, int32_t arena, const uint8_t** names, const uint64_t** name_off) {
    if (!h || arena < 0 || (size_t)arena >= h->arenas.size() || !names || !name_off) return -MM_INGEST_E_ARG;
    const Arena& a = h->arenas[(size_t)arena];
    if (!a.names) return -MM_INGEST_E_ARG;   // the handle was made without opts.names
    *names = a.names; *name_off = a.name_off;
    return 0;
}

int32_t mm_ingest_batch_codes(mm_ingest_t* h, const mm_batch_t* b, char* codes, int32_t max_codes) {
    if (!h || !b || b->n_reads < 0 || (!codes && max_codes > 0) || max_codes < 0) return -MM_INGEST_E_ARG;
    RCHK(hipSetDevice(h->device));
    if (!h->d_codes) {
        RCHK(mmdev::dmalloc((void**)&h->d_codes, sizeof(CodeTab)));
        RCHK(mmdev::hmalloc((void**)&h->h_codes, sizeof(CodeTab), hipHostMallocDefault));
    }
    RCHK(hipMemsetAsync(h->d_codes, 0, sizeof(CodeTab), h->chain));
    RCHK(hipMemsetAsync(h->d_codes->stamp, 0xFF, sizeof(h->d_codes->stamp), h->chain));
    if (b->n_reads > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)b->n_reads + 3) / 4, (int64_t)h->n_cu * 16);
        hipLaunchKernelGGL(k_batch_codes, dim3(grid), dim3(256), 0, h->chain, b->reads, b->mm, (uint32_t)b->n_reads, h->d_codes);
        RCHK(hipGetLastError());
    }
    RCHK(hipMemcpyAsync(h->h_codes, h->d_codes, sizeof(CodeTab), hipMemcpyDeviceToHost, h->chain));
    RCHK(hipStreamSynchronize(h->chain));
    const CodeTab& T = *h->h_codes;
    if (T.flags) return -MM_INGEST_E_CODES;
    std::vector<std::pair<unsigned long long, unsigned long long>> found;   // (stamp, key)
    for (uint32_t s = 0; s < kCodeSlots; s++) if (T.key[s]) found.emplace_back(T.stamp[s], T.key[s]);
    std::sort(found.begin(), found.end());
    int32_t n = 0;
    for (const auto& f : found) {
        if (n >= max_codes) break;
        char* dst = codes + (size_t)n * MM_CODE_LEN;
        std::memset(dst, 0, MM_CODE_LEN);
        std::memcpy(dst, &f.second, 8);
        n++;
    }
    return n;
}

int32_t mm_ingest_copy_to_host(mm_ingest_t* h, void* dst, const void* src, size_t n) {
    if (!h || (!dst && n) || (!src && n)) return -MM_INGEST_E_ARG;
    RCHK(hipSetDevice(h->device));
    RCHK(hipStreamSynchronize(h->chain));
    if (n) RCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return 0;
}

int32_t mm_ingest_times(mm_ingest_t* h, int32_t slot, float ms[4]) {
    GSlot* sp = slot_of(h, slot);
    if (!sp || !ms || !sp->timed) return -MM_INGEST_E_ARG;
    for (int i = 0; i < 3; i++) if (hipEventElapsedTime(&ms[i], sp->ev[i], sp->ev[i + 1]) != hipSuccess) return -MM_INGEST_E_HIP;
    if (hipEventElapsedTime(&ms[3], sp->ev_f0, sp->ev_done) != hipSuccess) return -MM_INGEST_E_HIP;
    return 0;
}

}  // extern "C"
