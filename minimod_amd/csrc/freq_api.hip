// freq_api.hip -- the C ABI of include/minimod_hip.h on top of the gfx950 kernels in freq_kernels.hip.h.
//
// Host-side life cycle (reference seams in parentheses, paths under /root/reference):
//   mm_freq_create   upload reference, K0, allocate counter planes   (load_ref/load_ref_contexts src/ref.c:46-229,
//                                                                     init_core src/minimod.c:51-137)
//   mm_freq_submit*  one -K/-B batch -> K1                            (process_db src/minimod.c:344-350)
//   mm_freq_wait     per-read errors                                  (the ERROR()+exit paths of src/mod.c)
//   mm_freq_finalize K2 + side list + ordering                        (print_freq_output src/mod.c:644-728)
// There is no CPU fallback: every entry point needs a HIP device and fails with MM_E_HIP otherwise.
#include "freq_kinds.h"   // (first: this copy's names)
#include <hip/hip_runtime.h>
#include "devmem.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "freq_kernels.hip.h"
#include "freq_tiles.hip.h"
#include "freq_stream.hip.h"
#include "view_kernels.hip.h"
#include "sort_kernels.hip.h"
#include "minimod_hip.h"

using namespace mmhip;

namespace {

constexpr int kSlots = 4;
constexpr int kQueueWords = 192 * kQueueStride;   // per set: 64 tile-queue counters, 64 scan-queue counters, 64 stream-queue counters, 128 bytes apart

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_done = nullptr;
    hipEvent_t ev_wait = nullptr;   // what wait() synchronises on: ev_stop, or ev_done when copies follow the kernels (view)
    bool busy = false, timed = false;
    // staging for host batches
    void* d_reads = nullptr; size_t cap_reads = 0;
    void* d_cigar = nullptr; size_t cap_cigar = 0;
    void* d_seq = nullptr;   size_t cap_seq = 0;
    void* d_mm = nullptr;    size_t cap_mm = 0;
    void* d_ml = nullptr;    size_t cap_ml = 0;
    void* d_order = nullptr; size_t cap_order = 0;
    // host batches gathered in the staging buffers above (opts.coalesce): what is filled so far
    size_t fill_reads = 0, fill_cigar = 0, fill_seq = 0, fill_mm = 0, fill_ml = 0;   // reads / words / bytes
    hipEvent_t ev_copied = nullptr;   // behind the last host -> device copy of the slot's batches
    bool copy_pending = false;
    int32_t* d_status = nullptr; size_t cap_status = 0;
    uint32_t* d_spill = nullptr; size_t cap_spill = 0;
    unsigned int* d_ctl = nullptr;   // two sets of 128 words, used alternately: [0] read queue, [1] err_summary, [4] fb_count,
                                     // [5] fb_queue, [8..72) tile_count per region.  A launch's last kernel resets the other set.
    int ctl_set = 0;
    unsigned int* d_err_word = nullptr;   // err_summary of the slot's last launch
    bool tiles_deferred = false;          // every read of the launch was a stream item: the tile kernels run when the host waits, and only
    TileParams tile_params;               // if k_stream_reads handed a read on (h_ctl[131])
    int tile_ga = 1, tile_gs = 1, tile_gc = 1;
    bool fb_deferred = false;             // the fused kernel for the fallback list was not launched with the batch:
    DevParams fb_params;                  // it runs when the host waits, and only if some read went on the list (h_ctl[130])
    unsigned int* d_tq = nullptr;    // two sets of 64 tile-queue + 64 scan-queue counters, 128 bytes apart, alternating with the control sets
    uint32_t* d_gcq = nullptr; size_t cap_gcq = 0;
    uint32_t* d_gcr = nullptr; size_t cap_gcr = 0;
    uint32_t* d_gdir = nullptr; size_t cap_gdir = 0;
    uint32_t* d_gqtot = nullptr; size_t cap_gqtot = 0;
    uint32_t* d_gnb = nullptr; size_t cap_gnb = 0;
    uint32_t* d_gqdir = nullptr; size_t cap_gqdir = 0;
    uint32_t* d_grdir = nullptr; size_t cap_grdir = 0;
    uint2* d_gsum = nullptr; size_t cap_gsum = 0;
    uint32_t* d_gtok = nullptr; size_t cap_gtok = 0;
    TileRec* d_tiles = nullptr; size_t cap_tiles = 0;
    int32_t* d_fb = nullptr; size_t cap_fb = 0;
    int32_t* d_plan = nullptr; size_t cap_plan = 0;                        // work items planned on the device (k_plan_items)
    int32_t* d_plan_stream = nullptr; size_t cap_plan_stream = 0;          // ... and the reads k_stream_reads takes (one item each)
    PlanState* d_plan_state = nullptr; unsigned int plan_serial = 0;      // what the planning workgroups share
    unsigned int* h_ctl = nullptr;   // pinned copy
    int32_t n_reads = 0;
    int32_t members = 1;         // submits gathered into the slot's launch
    // view mode: regional record buffers filled by the call kernels, then the ordering pipeline's buffers
    unsigned long long* d_vkeys = nullptr; size_t cap_vkeys = 0;
    unsigned long long* d_vvals = nullptr; size_t cap_vvals = 0;
    unsigned int* d_vseq = nullptr; size_t cap_vseq = 0;     // per record: its place among its read's records, or 0xFFFFFFFF
    uint32_t* d_vtile = nullptr; size_t cap_vtile = 0;        // tile sums of the offsets scan
    unsigned int* d_vcount = nullptr;      // [kViewRegions * kViewCountStride] + [1] selected rows
    unsigned int* h_vcount = nullptr;      // pinned copy
    unsigned int view_cap = 0;             // records per region
    unsigned long long* d_ka = nullptr; size_t cap_ka = 0;   // records grouped by read, then sorted inside each read
    unsigned long long* d_va = nullptr; size_t cap_va = 0;
    ViewRow* d_vrows = nullptr; size_t cap_vrows = 0;        // rows in print order (dropped duplicates still in place)
    ViewRow* d_vout = nullptr; size_t cap_vout = 0;          // rows after compaction (only batches with duplicate keys)
    unsigned int* d_vreadcount = nullptr; size_t cap_vreadcount = 0;   // per read: records
    unsigned int* d_voff = nullptr; size_t cap_voff = 0;               // per read: segment start (+ total at [n_reads])
    unsigned int* d_vcursor = nullptr; size_t cap_vcursor = 0;
    unsigned int* d_vkept = nullptr; size_t cap_vkept = 0;             // per read: rows kept
    unsigned int* d_vnewoff = nullptr; size_t cap_vnewoff = 0;
    const ViewRow* view_dev_rows = nullptr;                             // where the ticket's final rows are
    ViewRow* h_vrows = nullptr; size_t cap_hrows = 0;   // pinned
    mm_batch_t last_batch;       // device-side batch of the ticket (view mode may have to run it again)
    hipStream_t last_stream = nullptr;
    int64_t view_rows = -1;      // rows of the ticket once ordered (-1 = not yet)
    bool view_on_host = false;   // ... and copied to h_vrows
};

}  // namespace

struct mm_freq {
    mm_freq_opts_t opts;
    int device = 0;
    int n_cu = 0, blocks_per_cu = 1, scan_blocks_per_cu = 8, call_blocks_per_cu = 4, stream_blocks_per_cu = 6, stream_blocks_per_cu_dot = 6, stream_blocks_per_cu_dot_ins = 6;
    bool stream_dot = false; // a read with a '.' group has been seen: k_stream_reads' '.'-capable instantiation from now on
    bool use_tiles = true;   // opts.force_fused: the fused one-wave-per-read kernel for every read
    int ref_kind = 1;   // reference words: 0 four bits a base (one mod, RefNib), 1 16-bit (up to 5 mods), 2 32-bit
    int n_contigs = 0;
    std::vector<std::string> names;
    std::vector<int64_t> ctg_len, ref_base, seg_begin, seg_len, cnt_base;
    std::vector<int> ctg_rank;  // rank of contig names in strcmp order
    int64_t ref_total = 0, plane_len = 0;
    int n_code_planes = 0, n_hp = 1, wildcard = -1;
    void* d_refw = nullptr;
    int64_t *d_ref_base = nullptr, *d_ctg_len = nullptr, *d_seg_begin = nullptr, *d_seg_len = nullptr, *d_cnt_base = nullptr;
    unsigned long long* d_counters = nullptr;
    int64_t n_counter_words = 0;
    // context classes and their site index (freq_kernels.hip.h, DevClass)
    int n_classes = 0;
    std::vector<int32_t> cls_of_mod;
    std::vector<int> first_mod;               // per class: the first entry with its context string
    std::vector<DevClass> classes;
    std::vector<int64_t> adj;                 // [(tid * n_classes + class) * 2 + strand]
    std::vector<int> plane_cls, plane_slot;   // per code plane
    DevClass* d_classes = nullptr; int32_t* d_cls_of_mod = nullptr; int64_t* d_adj = nullptr;
    std::vector<void*> d_site_arrays;
    std::vector<void*> ipc_open;              // peers' slab buffers this handle has mapped (mm_freq_slab_add_ipc): closed when the handle goes
    unsigned int* d_slab_flag = nullptr;
    void* d_ipc_slab = nullptr;   // the slab another process reads through an IPC handle (mm_freq_slab_export_ipc): kept until the next one or the end
    DevMod* d_mods = nullptr;
    DevMod* d_ctx_mods = nullptr;             // [n_classes]: the first entry of every context class (what the reference words' bits are built from)
    DevCode* d_codes = nullptr;
    std::vector<DevCode> codes;
    bool codes_dirty = false;
    SideRec* d_side = nullptr;
    unsigned long long* d_side_count = nullptr;
    int64_t side_cap = 0;
    // the side table: updates that do not fit the dense planes, counted per 64-bit key (side_insert); [0] occupied slots
    unsigned long long *d_stab = nullptr, *d_scount = nullptr;
    unsigned long long stab_slots = 0;     // records per region of the side lists
    unsigned int* d_scur = nullptr;        // the regions' cursors
    unsigned long long *d_base_k = nullptr, *d_base_v = nullptr; size_t n_base = 0, cap_base = 0;   // what earlier compactions left: unique keys, ordered
    bool side_possible = false;            // this handle's runs can produce side updates at all
    unsigned int launches_since_side_check = 0;
    unsigned long long *d_sort_k[2] = {nullptr, nullptr}, *d_sort_v[2] = {nullptr, nullptr}; size_t cap_sort = 0;
    uint32_t* d_sort_hist = nullptr; size_t cap_sort_hist = 0;
    unsigned long long* d_stats = nullptr;
    bool stats_on = false;
    Slot slots[kSlots];
    int next_slot = 0;
    hipStream_t stream = nullptr;  // set-up / finalize stream
    std::vector<mm_row_t> rows;
    int64_t device_bytes = 0;
    // an error of a batch whose ticket was never waited for (its slot was recycled), or of a deferred launch that failed:
    // reported by the next submit / wait / finalize instead of being lost
    int sticky_err = 0, sticky_read = -1, sticky_ticket = -1;   // (sticky_ticket: the slot whose batch sticky_read counts in; -1: none)
    // a group of consecutive windows of one resident read set, gathered but not launched yet (opts.coalesce)
    int pending_slot = -1, pending_members = 0;
    mm_batch_t pending_batch;
    hipStream_t pending_stream = nullptr;
    uint64_t n_launches = 0, n_stream_launches = 0, n_submits = 0, n_reads_submitted = 0;   // mm_freq_launch_counts
    bool pending_host = false;        // ... of host batches staged one behind the other (mm_freq_submit) instead
    uint64_t pending_bases = 0;       // bases of the gathered reads when known (host batches), else 0
    // finalize scratch
    uint32_t* d_tile_counts = nullptr; unsigned long long* d_tile_offsets = nullptr; size_t cap_tiles = 0;
    mm_row_t* d_rows = nullptr; size_t cap_rows = 0;   // (rows)
};

namespace {

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            std::fprintf(stderr, "[minimod_hip] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                                          \
            return -MM_E_HIP;                                                                          \
        }                                                                                              \
    } while (0)

// a slot's own stream (launches of host batches, and of device batches whose caller names none): made at its first use
static hipStream_t slot_stream(mm_freq* h, Slot& s) {
    if (!s.stream && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { s.stream = nullptr; return h->stream; }
    return s.stream;
}

int dev_alloc(mm_freq* h, void** p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = mmdev::dmalloc(p, bytes);
    if (e != hipSuccess) return -MM_E_NOMEM;
    h->device_bytes += (int64_t)bytes;
    return 0;
}

static double g_grow_seconds = 0;   // (MM_TIMELINE: what the allocations inside a launch took)
int grow(mm_freq* h, void** p, size_t* cap, size_t need) {
    if (need <= *cap) return 0;
    static const bool tl = std::getenv("MM_TIMELINE") != nullptr;
    struct Timer { bool on; timespec a; Timer(bool o) : on(o) { if (on) clock_gettime(CLOCK_MONOTONIC, &a); }
                   ~Timer() { if (on) { timespec b; clock_gettime(CLOCK_MONOTONIC, &b); g_grow_seconds += (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec); } } } timer(tl);
    size_t ncap = std::max(need + need / 4, (size_t)4096);
    if (*p) { (void)mmdev::dfree(*p); h->device_bytes -= (int64_t)*cap; }
    *p = nullptr; *cap = 0;
    int r = dev_alloc(h, p, ncap);
    if (r) return r;
    *cap = ncap;
    return 0;
}

// all regions empty, nothing compacted
int side_table_clear(mm_freq* h) {
    if (hipMemset(h->d_scur, 0, sizeof(unsigned int) * kSideRegions * kSideCurStride) != hipSuccess) return -MM_E_HIP;
    h->n_base = 0;
    return 0;
}

int ensure_sort_buffers(mm_freq* h, size_t n) {
    if (n > h->cap_sort) {
        for (int i = 0; i < 2; i++) { if (h->d_sort_k[i]) (void)mmdev::dfree(h->d_sort_k[i]); if (h->d_sort_v[i]) (void)mmdev::dfree(h->d_sort_v[i]); h->d_sort_k[i] = h->d_sort_v[i] = nullptr; }
        h->cap_sort = 0;
        size_t cap = n + n / 8 + 1024;
        for (int i = 0; i < 2; i++)
            if (dev_alloc(h, (void**)&h->d_sort_k[i], 8 * cap) || dev_alloc(h, (void**)&h->d_sort_v[i], 8 * cap)) return -MM_E_NOMEM;
        h->cap_sort = cap;
    }
    const size_t nblk = (n + kSortTile - 1) / kSortTile;
    if (256 * nblk + 256 > h->cap_sort_hist) {
        if (h->d_sort_hist) (void)mmdev::dfree(h->d_sort_hist);
        h->d_sort_hist = nullptr; h->cap_sort_hist = 0;
        if (dev_alloc(h, (void**)&h->d_sort_hist, 4 * (256 * nblk + 1024))) return -MM_E_NOMEM;
        h->cap_sort_hist = 256 * nblk + 1024;
    }
    return 0;
}

// The side lists compacted (every launch complete): the regions' records and what the last compaction left are gathered,
// ordered by key and reduced to one (key, counts) pair per key -- the new base, ordered; the regions are empty again.
// `fills` (optional): the regions' cursors as just read, to spare the copy.
int side_compact(mm_freq* h) {
    std::vector<unsigned int> cur(kSideRegions * kSideCurStride);
    HIPCHK(hipMemcpy(cur.data(), h->d_scur, sizeof(unsigned int) * cur.size(), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> off(kSideRegions + 1, 0);
    for (uint32_t r = 0; r < kSideRegions; r++) {
        if ((unsigned long long)cur[r * kSideCurStride] > h->stab_slots) return -MM_E_SIDEFULL;   // the region refused updates
        off[r + 1] = off[r] + cur[r * kSideCurStride];
    }
    const size_t n_new = (size_t)off[kSideRegions], n = n_new + h->n_base;
    if (n_new == 0) return 0;
    { int r = ensure_sort_buffers(h, n); if (r) return r; }
    unsigned long long* d_off = nullptr;
    if (mmdev::dmalloc((void**)&d_off, 8 * off.size()) != hipSuccess) return -MM_E_NOMEM;
    int result = 0;
    do {
        if (hipMemcpyAsync(d_off, off.data(), 8 * off.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
        hipLaunchKernelGGL(k_side_gather, dim3((unsigned)std::max(1, h->n_cu / 4), kSideRegions), dim3(256), 0, h->stream, h->d_stab, h->stab_slots, d_off, h->d_sort_k[0], h->d_sort_v[0]);
        if (h->n_base) {
            if (hipMemcpyAsync(h->d_sort_k[0] + n_new, h->d_base_k, 8 * h->n_base, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipMemcpyAsync(h->d_sort_v[0] + n_new, h->d_base_v, 8 * h->n_base, hipMemcpyDeviceToDevice, h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
        }
        const uint32_t nblk = (uint32_t)((n + kSortTile - 1) / kSortTile);
        int cb = 0;
        for (int shift = 0; shift < 64; shift += 8) {   // LSD radix sort on the whole key (63 bits used)
            hipLaunchKernelGGL(k_radix_hist, dim3(nblk), dim3(64), 0, h->stream, h->d_sort_k[cb], (unsigned long long)n, shift, h->d_sort_hist, nblk);
            hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, h->stream, h->d_sort_hist, (unsigned long long)256 * nblk);
            hipLaunchKernelGGL(k_radix_scatter, dim3(nblk), dim3(64), 0, h->stream, h->d_sort_k[cb], h->d_sort_v[cb], (unsigned long long)n, shift, h->d_sort_hist, nblk,
                               h->d_sort_k[cb ^ 1], h->d_sort_v[cb ^ 1]);
            cb ^= 1;
        }
        // one pair per key
        const uint32_t ntile = (uint32_t)((n + kReduceTile - 1) / kReduceTile);
        if (hipMemsetAsync(h->d_sort_hist + ntile, 0, 4, h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
        hipLaunchKernelGGL(k_reduce_count, dim3(ntile), dim3(256), 0, h->stream, h->d_sort_k[cb], (unsigned long long)n, h->d_sort_hist);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, h->stream, h->d_sort_hist, (unsigned long long)ntile + 1ull);
        uint32_t n_unique = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&n_unique, h->d_sort_hist + ntile, 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
        if ((size_t)n_unique > h->cap_base) {
            if (h->d_base_k) { (void)mmdev::dfree(h->d_base_k); (void)mmdev::dfree(h->d_base_v); h->device_bytes -= (int64_t)(16 * h->cap_base); }
            h->d_base_k = h->d_base_v = nullptr; h->cap_base = 0;
            const size_t cap = (size_t)n_unique + (size_t)n_unique / 4 + 1024;
            if (dev_alloc(h, (void**)&h->d_base_k, 8 * cap) || dev_alloc(h, (void**)&h->d_base_v, 8 * cap)) { result = -MM_E_NOMEM; break; }
            h->cap_base = cap;
        }
        hipLaunchKernelGGL(k_reduce_emit, dim3(ntile), dim3(256), 0, h->stream, h->d_sort_k[cb], h->d_sort_v[cb], (unsigned long long)n, h->d_sort_hist, h->d_base_k, h->d_base_v);
        if (hipGetLastError() != hipSuccess || hipMemsetAsync(h->d_scur, 0, sizeof(unsigned int) * kSideRegions * kSideCurStride, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
        h->n_base = n_unique;
    } while (0);
    (void)mmdev::dfree(d_off);
    return result;
}

// a launch has completed: should the side lists be compacted before the next one?  (runs that can fill them only; the regions'
// cursors are 8 KB to look at)
int side_check(mm_freq* h) {
    if (!h->side_possible) return 0;
    std::vector<unsigned int> cur(kSideRegions * kSideCurStride);
    HIPCHK(hipMemcpy(cur.data(), h->d_scur, sizeof(unsigned int) * cur.size(), hipMemcpyDeviceToHost));
    unsigned long long worst = 0;
    for (uint32_t r = 0; r < kSideRegions; r++) worst = std::max<unsigned long long>(worst, cur[r * kSideCurStride]);
    return worst > h->stab_slots / 2 ? 1 : 0;
}

int complement(int c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'U': return 'A'; case 'N': return 'N'; default: return 0;
    }
}

int upload_codes(mm_freq* h, hipStream_t s) {
    if (!h->codes_dirty) return 0;
    HIPCHK(hipMemcpyAsync(h->d_codes, h->codes.data(), sizeof(DevCode) * h->codes.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    h->codes_dirty = false;
    return 0;
}

DevParams base_params(mm_freq* h) {
    DevParams p;
    std::memset(&p, 0, sizeof(p));
    p.refw = h->d_refw;
    p.ref_base = h->d_ref_base; p.ctg_len = h->d_ctg_len;
    p.seg_begin = h->d_seg_begin; p.seg_len = h->d_seg_len; p.cnt_base = h->d_cnt_base;
    p.n_contigs = h->n_contigs;
    p.counters = h->d_counters; p.plane_len = h->plane_len; p.n_hp = h->n_hp;
    p.n_classes = h->n_classes; p.classes = h->d_classes; p.cls_of_mod = h->d_cls_of_mod; p.adj = h->d_adj;
    p.n_mods = h->opts.n_mods; p.n_codes = (int)h->codes.size();
    p.insertions = h->opts.insertions; p.haplotypes = h->opts.haplotypes; p.wildcard = h->wildcard >= 0;
    p.mods = h->d_mods; p.codes = h->d_codes;
    p.side = h->d_side; p.side_count = h->d_side_count; p.side_cap = (unsigned long long)h->side_cap;
    p.stab = h->d_stab; p.smask = h->stab_slots; p.scur = h->d_scur;
    p.stats = h->stats_on ? h->d_stats : nullptr;
#ifdef MM_STREAM_TIMING
    p.stats = h->d_stats;   // diagnostic build: phase times of every launch (mm_freq_stats_get reads and clears them)
#endif
    return p;
}

K2Params k2_params(mm_freq* h) {
    K2Params k;
    std::memset(&k, 0, sizeof k);
    k.cnt = h->d_counters; k.classes = h->d_classes; k.adj = h->d_adj; k.ref_base = h->d_ref_base;
    k.n_classes = h->n_classes; k.n_hp = h->n_hp; k.n_planes = h->n_code_planes; k.haplotypes = h->opts.haplotypes;
    for (int pl = 0; pl < h->n_code_planes && pl < MM_MAX_CODES; pl++) { k.plane_cls[pl] = (int8_t)h->plane_cls[(size_t)pl]; k.plane_slot[pl] = (int8_t)h->plane_slot[(size_t)pl]; }
    return k;
}

// order the rows of a view batch (view_kernels.hip.h): counting sort by read, then one small sort per read.  Everything
// takes its sizes from device memory, so nothing here waits for the call kernels.
int enqueue_view_ordering(mm_freq* h, Slot& s, const mm_batch_t* b, hipStream_t st) {
    unsigned int* tail = s.d_vcount + kViewRegions * kViewCountStride;   // [0] rows dropped as duplicates
    const uint32_t nr = (uint32_t)b->n_reads;
    if (nr <= 8192u) {
        hipLaunchKernelGGL(k_view_offsets, dim3(1), dim3(256), 0, st, s.d_vreadcount, nr, s.d_voff, s.d_vcursor);
    } else {   // a gathered launch: the scan in tiles
        const uint32_t nt = (nr + kScanTile - 1) / kScanTile;
        { int r = grow(h, (void**)&s.d_vtile, &s.cap_vtile, 4 * ((size_t)nt + 2)); if (r) return r; }
        HIPCHK(hipMemsetAsync(s.d_vtile + nt, 0, 4, st));
        hipLaunchKernelGGL(k_scan_tile_sums, dim3(nt), dim3(256), 0, st, s.d_vreadcount, (int64_t)nr, s.d_vtile);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, st, s.d_vtile, (unsigned long long)nt + 1ull);
        hipLaunchKernelGGL(k_view_offsets_apply, dim3(nt), dim3(256), 0, st, s.d_vreadcount, nr, s.d_vtile, nt, s.d_voff, s.d_vcursor);
    }
    hipLaunchKernelGGL(k_view_scatter, dim3(std::max(1, h->n_cu / 8), kViewRegions), dim3(256), 0, st, s.d_vkeys, s.d_vvals, s.d_vseq, s.d_vcount,
                       s.view_cap, 0xFFFFFFu, s.d_voff, s.d_vcursor, s.d_ka, s.d_va, b->reads, s.d_vrows, h->opts.view == 2 ? 1u : 0u);
    const uint32_t small_blocks = std::min<uint32_t>((nr + kWavesPerBlock - 1) / kWavesPerBlock, (uint32_t)h->n_cu * 4);
    const uint32_t big_blocks = std::min<uint32_t>(nr, (uint32_t)h->n_cu);
    hipLaunchKernelGGL(k_view_sort, dim3(big_blocks + small_blocks), dim3(256), 0, st, s.d_ka, s.d_va, s.d_voff, nr, big_blocks, b->reads,
                       s.d_vrows, s.d_vkept, tail, h->opts.view == 2 ? 1u : 0u, s.d_vcursor);
    HIPCHK(hipGetLastError());
    return 0;
}

// the reference word type of a handle: RW inside the statement
#if defined(MM_KIND) && MM_KIND == 0   // this copy's kind only (freq_kinds.h): the other kinds' instantiations are in the other copies
#define MM_REF_DISPATCH(h, ...) do { using RW = RefNib; __VA_ARGS__; } while (0)
#elif defined(MM_KIND) && MM_KIND == 1
#define MM_REF_DISPATCH(h, ...) do { using RW = uint16_t; __VA_ARGS__; } while (0)
#elif defined(MM_KIND) && MM_KIND == 2
#define MM_REF_DISPATCH(h, ...) do { using RW = uint32_t; __VA_ARGS__; } while (0)
#else
#define MM_REF_DISPATCH(h, ...) do { \
        if ((h)->ref_kind == 2) { using RW = uint32_t; __VA_ARGS__; } \
        else if ((h)->ref_kind == 1) { using RW = uint16_t; __VA_ARGS__; } \
        else { using RW = RefNib; __VA_ARGS__; } } while (0)
#endif

void launch_tile_kernels(mm_freq* h, const TileParams& tp, int ga, int gs, int gc, hipStream_t st) {
    const DevParams& p = tp.d;
    const bool plain = !p.insertions && !p.haplotypes;
    MM_REF_DISPATCH(h,
        hipLaunchKernelGGL(k_scan_reads<RW>, dim3(ga), dim3(256), 0, st, tp);
        hipLaunchKernelGGL(k_sum_tiles<RW>, dim3(gs), dim3(256), 0, st, tp);
        if (p.view) { if (plain) hipLaunchKernelGGL((k_call_tiles<RW, true, true>), dim3(gc), dim3(256), 0, st, tp);
                      else hipLaunchKernelGGL((k_call_tiles<RW, true, false>), dim3(gc), dim3(256), 0, st, tp); }
        else { if (plain) hipLaunchKernelGGL((k_call_tiles<RW, false, true>), dim3(gc), dim3(256), 0, st, tp);
               else hipLaunchKernelGGL((k_call_tiles<RW, false, false>), dim3(gc), dim3(256), 0, st, tp); });
}

// a host batch appended to a staging area: its reads' pool offsets move by where its pools went
__global__ void k_rebase_reads(mm_read_t* reads, int32_t n, uint64_t cigar_words, uint64_t seq_bytes, uint64_t mm_bytes, uint64_t ml_bytes) {
    const int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    reads[i].cigar_off += cigar_words; reads[i].seq_off += seq_bytes; reads[i].mm_off += mm_bytes; reads[i].ml_off += ml_bytes;
}

// bases_hint: the bases of the launch's reads when the caller knows them (host batches), else 0
static int launch_k1_body(mm_freq* h, Slot& s, const mm_batch_t* b, hipStream_t st, uint64_t bases_hint);
int launch_k1(mm_freq* h, Slot& s, const mm_batch_t* b, hipStream_t st, uint64_t bases_hint = 0) {
    static const bool tl = std::getenv("MM_TIMELINE") != nullptr;
    static int calls = 0;
    if (!tl || calls >= 6) return launch_k1_body(h, s, b, st, bases_hint);
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    const double t0 = now(), g0 = g_grow_seconds;
    const int r = launch_k1_body(h, s, b, st, bases_hint);
    std::fprintf(stderr, "[timeline] launch %d of the handle (%d reads): %.4f s on the host, %.4f of them in device allocations\n", ++calls, b->n_reads, now() - t0, g_grow_seconds - g0);
    return r;
}
static int launch_k1_body(mm_freq* h, Slot& s, const mm_batch_t* b, hipStream_t st, uint64_t bases_hint) {
    DevParams p = base_params(h);
    p.reads = b->reads; p.cigar = b->cigar; p.seq = b->seq; p.mm = b->mm; p.ml = b->ml; p.order = b->order;
    p.n_reads = b->n_reads;
    p.n_items = b->order ? b->n_order : b->n_reads;
    p.n_items_dev = nullptr;
    const int full_grid = h->n_cu * h->blocks_per_cu;
    int blocks = std::min((p.n_items + kWavesPerBlock - 1) / kWavesPerBlock, full_grid);
    if (blocks < 1) blocks = 1;
    if (h->use_tiles) blocks = std::min(full_grid, 128);   // the fused kernel then only runs the (usually empty) fallback list
    uint32_t spill_cig = b->max_n_cigar > (uint32_t)kCigCap ? b->max_n_cigar - kCigCap : 0;
    uint32_t max_blk = (b->max_l_qseq + 31u) / 32u;
    uint32_t spill_blk = max_blk > (uint32_t)kDirCap ? max_blk - kDirCap : 0;
    size_t per_wave = 2u * (size_t)spill_cig + spill_blk;
    size_t need = per_wave * (size_t)blocks * kWavesPerBlock * sizeof(uint32_t);
    int r = grow(h, (void**)&s.d_spill, &s.cap_spill, need);
    if (r) return r;
    r = grow(h, (void**)&s.d_status, &s.cap_status, sizeof(int32_t) * (size_t)std::max(b->n_reads, 1));
    if (r) return r;
    p.spill = s.d_spill; p.spill_cig = spill_cig; p.spill_blk = spill_blk;
    p.status = s.d_status;
    HIPCHK(hipMemsetAsync(s.d_status, 0, sizeof(int32_t) * (size_t)std::max(b->n_reads, 1), st));   // (-1 = handed to the fused kernel: the tile kernels read it)
    if (h->opts.view) {
        // every explicit row consumes one ML byte, so the ML pool bounds them; implicit rows ('.' groups) are not bounded
        // by anything cheap: mm_view_fetch grows the buffer and runs the batch again when a region overflows
        size_t want = (size_t)(b->n_ml_bytes / kViewRegions) * 3 / 2 + 4096;
        if (h->opts.view_cap > 0) want = (size_t)h->opts.view_cap;   // tests: force the overflow path
        if (want > s.view_cap) {
            if ((r = grow(h, (void**)&s.d_vkeys, &s.cap_vkeys, 8 * want * kViewRegions)) ||
                (r = grow(h, (void**)&s.d_vvals, &s.cap_vvals, 8 * want * kViewRegions)) ||
                (r = grow(h, (void**)&s.d_vseq, &s.cap_vseq, 4 * want * kViewRegions)))
                return r;
            s.view_cap = (unsigned int)std::min<size_t>(std::min(std::min(s.cap_vkeys, s.cap_vvals) / 8, s.cap_vseq / 4) / kViewRegions, 0xFFFFFFFFu / kViewRegions);
        }
        // the ordering pass works on at most what the regions can hold
        const size_t cap_total = (size_t)s.view_cap * kViewRegions, nr = (size_t)std::max(b->n_reads, 1);
        if ((r = grow(h, (void**)&s.d_ka, &s.cap_ka, 8 * cap_total)) || (r = grow(h, (void**)&s.d_va, &s.cap_va, 8 * cap_total)) ||
            (r = grow(h, (void**)&s.d_vrows, &s.cap_vrows, sizeof(ViewRow) * cap_total)) ||
            (r = grow(h, (void**)&s.d_vreadcount, &s.cap_vreadcount, 4 * nr)) || (r = grow(h, (void**)&s.d_voff, &s.cap_voff, 4 * (nr + 1))) ||
            (r = grow(h, (void**)&s.d_vcursor, &s.cap_vcursor, 4 * nr)) || (r = grow(h, (void**)&s.d_vkept, &s.cap_vkept, 4 * nr)) ||
            (r = grow(h, (void**)&s.d_vnewoff, &s.cap_vnewoff, 4 * (nr + 1))))
            return r;
        p.view = 1; p.view_cap = s.view_cap; p.view_keys = s.d_vkeys; p.view_vals = s.d_vvals; p.view_count = s.d_vcount;
        p.view_read_count = s.d_vreadcount; p.view_seq = s.d_vseq;
        HIPCHK(hipMemsetAsync(s.d_vcount, 0, sizeof(unsigned int) * (kViewRegions * kViewCountStride + 2), st));
        HIPCHK(hipMemsetAsync(s.d_vreadcount, 0, 4 * nr, st));
    }
    unsigned int* const ctl = s.d_ctl + kCtlSetWords * s.ctl_set;
    unsigned int* const ctl_other = s.d_ctl + kCtlSetWords * (s.ctl_set ^ 1);
    p.queue = ctl; p.err_summary = ctl + 1;
    s.h_ctl[129] = 0xFFFFFFFFu;   // host_flag: a failing read stores 0 here
    p.host_flag = s.h_ctl + 129;
    s.d_err_word = ctl + 1;
    p.ctl_next = ctl_other;
    p.queue_next = s.d_tq + kQueueWords * (s.ctl_set ^ 1);
    TileParams tp;
    std::memset(&tp, 0, sizeof(tp));
    if (h->use_tiles) {
        size_t tile_cap = (size_t)(b->n_mm_bytes / 4 + b->n_seq_bytes / 1024 + 16 * (size_t)std::max(b->n_reads, 1));
        if ((r = grow(h, (void**)&s.d_gcq, &s.cap_gcq, 4 * (size_t)b->n_cigar_words)) ||
            (r = grow(h, (void**)&s.d_gcr, &s.cap_gcr, 4 * (size_t)b->n_cigar_words)) ||
            (r = grow(h, (void**)&s.d_gdir, &s.cap_gdir, 4 * ((size_t)b->n_seq_bytes / 16 + 16))) ||
            (r = grow(h, (void**)&s.d_gqtot, &s.cap_gqtot, 4 * (size_t)std::max(b->n_reads, 1))) ||
            (r = grow(h, (void**)&s.d_gnb, &s.cap_gnb, 4 * (size_t)std::max(b->n_reads, 1))) ||
            (r = grow(h, (void**)&s.d_gqdir, &s.cap_gqdir, 4 * ((size_t)b->n_seq_bytes / 128 + 2 * (size_t)b->n_reads + 16))) ||
            (r = grow(h, (void**)&s.d_grdir, &s.cap_grdir, 4 * ((size_t)b->n_seq_bytes / 32 + 2 * (size_t)b->n_reads + 16))) ||
            (r = grow(h, (void**)&s.d_gtok, &s.cap_gtok, 4 * ((size_t)b->n_mm_bytes / 2 + 256))) ||
            (r = grow(h, (void**)&s.d_gsum, &s.cap_gsum, sizeof(uint2) * (tile_cap / kTileRegions + 64) * kTileRegions)) ||
            (r = grow(h, (void**)&s.d_tiles, &s.cap_tiles, sizeof(TileRec) * (tile_cap / kTileRegions + 64) * kTileRegions)) ||
            (r = grow(h, (void**)&s.d_fb, &s.cap_fb, 4 * (size_t)std::max(b->n_reads, 1))))
            return r;
        tp.g_cq = s.d_gcq; tp.g_cr = s.d_gcr; tp.g_dir = s.d_gdir; tp.g_qtot = s.d_gqtot; tp.g_nb = s.d_gnb; tp.g_sum = s.d_gsum; tp.g_tok = s.d_gtok; tp.g_qdir = s.d_gqdir; tp.g_rdir = s.d_grdir;
        tp.tiles = s.d_tiles; tp.tile_cap = (unsigned int)std::min<size_t>(tile_cap / kTileRegions + 64, 0x3FFFFFFu);
        tp.tile_count = ctl + 8; tp.tile_queue = s.d_tq + kQueueWords * s.ctl_set;
        tp.scan_queue = tp.tile_queue + 64 * kQueueStride;
        tp.fb_list = s.d_fb; tp.fb_count = ctl + 4;
        s.h_ctl[130] = 0u;
        tp.host_fb_flag = s.h_ctl + 130;
        tp.reset_in_call = 1;
    }
    if (b->n_reads <= 0) {   // no kernel will run: the set the next launch uses is reset from the host
        for (int i = 0; i < kCtlWords; i++) s.h_ctl[i] = 0u;
        s.h_ctl[1] = 0xFFFFFFFFu;
        HIPCHK(hipMemcpyAsync(ctl_other, s.h_ctl, kCtlWords * sizeof(unsigned int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(s.d_tq + kQueueWords * (s.ctl_set ^ 1), 0, kQueueWords * sizeof(unsigned int), st));
    }
    HIPCHK(hipEventRecord(s.ev_start, st));
    if (b->n_reads > 0) h->n_launches++;
    if (b->n_reads > 0) {
        if (h->use_tiles) {
            bool stream = false, all_stream = false;
            s.tiles_deferred = false;
            if (!b->order) {
                // no plan from the caller: the work items are made on the device (long reads cut into parts, costliest
                // first), their number stays in device memory (control word 6)
                const uint32_t split = h->opts.split_bases >= 1024 ? (uint32_t)h->opts.split_bases : kSplitBases;
                const size_t max_items = (size_t)b->n_reads + 2 * (size_t)b->n_seq_bytes / split + 64;
                if ((r = grow(h, (void**)&s.d_plan, &s.cap_plan, 4 * max_items))) return r;
                // plain freq runs: k_stream_reads takes the reads it can hide in the launch (and hands back what it does not do).
                // A read is one wavefront's work from start to end there, so the longest one bounds the launch from below:
                // measured on MI355X (gathered launches of 4 to 32 batches) a wavefront spends 4.6 ns per base of its read
                // while a full launch gets through a base in 0.75 ps, so a read of launch_bases / 6000 bases lasts as long as
                // the launch itself.  Reads up to about that long go (the costliest start first; one that long costs the launch
                // less than the tile kernels' three near-empty launches would), longer ones are cut into parts for the tile
                // pipeline.  The launch's bases are not known here for device batches -- windows of a resident set share pool
                // sizes -- so its reads are taken as 12 kb each; a launch too small to hide reads of `split` bases (a single
                // -K 4096 batch: 181 against 110 us) leaves everything to the tiles.
                const uint64_t hide = (bases_hint ? bases_hint : 12000ull * (uint64_t)b->n_reads) / 4500;
                const int mode = h->opts.stream_mode;
                // (view as well: the records go where the tile kernels' go; --insertions / --haplotypes: freq only, the kIns instantiation)
                const bool general = p.insertions || p.haplotypes;
                stream = mode != 1 && !(general && p.view) && (mode >= 2 || hide >= split);
                const uint32_t stream_max = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(split, hide), 0x00FFFFFFu);
                if (stream && (r = grow(h, (void**)&s.d_plan_stream, &s.cap_plan_stream, 4 * (size_t)b->n_reads))) return r;
                // every read is a stream item: k_stream_reads is the launch's last kernel (not in view mode: the ordering pass is
                // enqueued behind the call kernels and would have to run again behind tile kernels launched at wait time)
                all_stream = stream && !p.view && b->max_l_qseq <= stream_max;
                s.h_ctl[131] = 0u;
                tp.host_tile_flag = s.h_ctl + 131;
                tp.reset_in_stream = all_stream ? 1 : 0;
                if (all_stream) tp.reset_in_call = 0;
                if (!s.d_plan_state) {
                    if (dev_alloc(h, (void**)&s.d_plan_state, sizeof(PlanState))) return -MM_E_NOMEM;
                    HIPCHK(hipMemsetAsync(s.d_plan_state, 0, sizeof(PlanState), st));
                }
                const int pb = std::max(1, std::min(64, (b->n_reads + kPlanReadsPerBlock - 1) / kPlanReadsPerBlock));
                // sliced hand-out (one position slice per XCD, freq_tiles.hip.h stream_bucket): launches of thousands of reads
                uint32_t long_cut = 0u;   // (the launch's mean read length; 0 = not sliced)
                if (stream && b->n_reads >= 8192 && h->opts.stream_slices != 1)
                    long_cut = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((bases_hint ? bases_hint : 12000ull * (uint64_t)b->n_reads) / (uint64_t)b->n_reads, 256), 0x00FFFFFFu);
                hipLaunchKernelGGL(k_plan_items, dim3(pb), dim3(kPlanThreads), 0, st, b->reads, b->n_reads, split, s.d_plan, ctl + 6,
                                   s.d_plan_state, ++s.plan_serial, p.err_summary, p.host_flag, stream ? stream_max : 0u, s.d_plan_stream, ctl + 7,
                                   all_stream ? s.h_ctl + 131 : nullptr, long_cut);
                tp.stream_slices = long_cut ? s.d_plan_state->slices : nullptr;
                p.order = s.d_plan;
                p.n_items = (int32_t)std::min<size_t>(max_items, (size_t)0x7FFFFFFF);   // an upper bound: sizes the grid
                tp.plan_count = ctl + 6;
                tp.stream_items = s.d_plan_stream; tp.stream_count = ctl + 7;
                tp.stream_queue = tp.tile_queue + 128 * kQueueStride;
                tp.tile_items = s.d_plan; tp.tile_plan_count = ctl + 6;
            }
            tp.d = p;
            int ga = std::min((3 * p.n_items + kWavesPerBlock - 1) / kWavesPerBlock, h->n_cu * h->scan_blocks_per_cu);
            int gc = h->n_cu * h->call_blocks_per_cu;
            int gs = h->n_cu * 6;   // measured: 8 waves per SIMD 16.0 us, 6 14.8 us, 4 16.9 us
            if (ga < 1) ga = 1;
            if (stream) {
                h->n_stream_launches++;
                int gf = h->n_cu * h->stream_blocks_per_cu;
                // which instantiation: the lean one until a read with a '.' group has shown up (the flag of the slot's last
                // launch is looked at here: a file's reads carry one flag or the other), stream_mode 3 = the '.'-capable one at once
                for (auto& sl : h->slots) if (sl.h_ctl && sl.h_ctl[132] != 0u) h->stream_dot = true;   // (any slot's finished launch)
                s.h_ctl[132] = 0u;
                tp.host_dot_flag = s.h_ctl + 132;
                const bool kd = h->stream_dot || h->opts.stream_mode == 3;
                if (kd) gf = h->n_cu * ((p.insertions || p.haplotypes) ? h->stream_blocks_per_cu_dot_ins : h->stream_blocks_per_cu_dot);
#define MM_LAUNCH_STREAM(T, ST, DT) do { if (p.view) hipLaunchKernelGGL((k_stream_reads<T, ST, DT, true, false>), dim3(gf), dim3(256), 0, st, tp); \
                                           else hipLaunchKernelGGL((k_stream_reads<T, ST, DT, false, false>), dim3(gf), dim3(256), 0, st, tp); } while (0)
                MM_REF_DISPATCH(h,
                    if (p.insertions || p.haplotypes) {   // (round 4: '.' groups too -- only a reverse read of more than 512 ops under --insertions goes on to the tile pipeline)
                        if (p.stats) { if (kd) hipLaunchKernelGGL((k_stream_reads<RW, true, true, false, true>), dim3(gf), dim3(256), 0, st, tp);
                                       else hipLaunchKernelGGL((k_stream_reads<RW, true, false, false, true>), dim3(gf), dim3(256), 0, st, tp); }
                        else { if (kd) hipLaunchKernelGGL((k_stream_reads<RW, false, true, false, true>), dim3(gf), dim3(256), 0, st, tp);
                               else hipLaunchKernelGGL((k_stream_reads<RW, false, false, false, true>), dim3(gf), dim3(256), 0, st, tp); }
                    }
                    else if (p.stats) { if (kd) MM_LAUNCH_STREAM(RW, true, true); else MM_LAUNCH_STREAM(RW, true, false); }
                    else { if (kd) MM_LAUNCH_STREAM(RW, false, true); else MM_LAUNCH_STREAM(RW, false, false); });
#undef MM_LAUNCH_STREAM
            }
            if (all_stream) {
                // nothing is planned for the tile kernels: they run at wait time, and only if k_stream_reads handed a read on
                s.tiles_deferred = true;
                s.tile_params = tp; s.tile_params.reset_in_call = 0;
                s.tile_ga = ga; s.tile_gs = gs; s.tile_gc = gc;
            } else {
                launch_tile_kernels(h, tp, ga, gs, gc, st);
            }
            HIPCHK(hipGetLastError());
            // reads the tile form does not cover: the fused kernel over the fallback list.  The list is nearly always
            // empty, so that launch is not made now: a read that goes on the list raises a flag in pinned host memory and
            // mm_freq_wait / mm_view_fetch run the kernel then (finish_deferred; view orders its rows again afterwards).
            p.order = s.d_fb; p.n_items = 0; p.n_items_dev = ctl + 4; p.queue = ctl + 5;
            s.fb_deferred = true;
            s.fb_params = p;
        }
        if (!(h->use_tiles && s.fb_deferred)) {
        MM_REF_DISPATCH(h,
            if (p.view) hipLaunchKernelGGL((k_freq_reads<RW, true>), dim3(blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_freq_reads<RW, false>), dim3(blocks), dim3(256), 0, st, p));
        }
        HIPCHK(hipGetLastError());
    }
    if (h->opts.view && b->n_reads > 0) { int rv = enqueue_view_ordering(h, s, b, st); if (rv) return rv; }
    HIPCHK(hipEventRecord(s.ev_stop, st));
    // no copy back on the good path: kernels flag a failure in pinned host memory (DevParams.host_flag)
    s.ctl_set ^= 1;
    if (h->opts.view)
        HIPCHK(hipMemcpyAsync(s.h_vcount, s.d_vcount, sizeof(unsigned int) * (kViewRegions * kViewCountStride + 2), hipMemcpyDeviceToHost, st));
    if (h->opts.view) { HIPCHK(hipEventRecord(s.ev_done, st)); s.ev_wait = s.ev_done; }
    else s.ev_wait = s.ev_stop;   // nothing follows the kernels: one event less per batch
    s.busy = true; s.timed = true; s.n_reads = b->n_reads; s.members = 1;
    s.last_batch = *b; s.last_stream = st; s.view_rows = -1; s.view_on_host = false;
    return 0;
}

// The batch's kernels are complete (its event has been waited for): if a read went on the fallback list, run the fused
// kernel for the list now.
int finish_deferred(mm_freq* h, Slot& s) {
    int ran = 0;
    if (s.tiles_deferred) {
        s.tiles_deferred = false;
        if (s.h_ctl[131] != 0u) {   // k_stream_reads handed reads to the tile pipeline: its kernels run now
            launch_tile_kernels(h, s.tile_params, s.tile_ga, s.tile_gs, s.tile_gc, s.last_stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s.last_stream));   // (they may put reads on the fallback list)
            ran = 1;
        }
    }
    if (!s.fb_deferred) return ran;
    s.fb_deferred = false;
    if (s.h_ctl[130] == 0u) return ran;
    const DevParams& p = s.fb_params;
    const int blocks = std::min(h->n_cu * h->blocks_per_cu, 128);
    MM_REF_DISPATCH(h,
        if (p.view) hipLaunchKernelGGL((k_freq_reads<RW, true>), dim3(blocks), dim3(256), 0, s.last_stream, p);
        else hipLaunchKernelGGL((k_freq_reads<RW, false>), dim3(blocks), dim3(256), 0, s.last_stream, p));
    HIPCHK(hipGetLastError());
    if (p.view) {
        // the fused kernel appended records: order the batch's rows again (the ordering pass only reads the regional
        // buffers and the per-read counts, so running it twice is harmless) and fetch the counts it works from
        HIPCHK(hipMemsetAsync(s.d_vcount + kViewRegions * kViewCountStride, 0, 2 * sizeof(unsigned int), s.last_stream));
        int rv = enqueue_view_ordering(h, s, &s.last_batch, s.last_stream);
        if (rv) return rv;
        HIPCHK(hipMemcpyAsync(s.h_vcount, s.d_vcount, sizeof(unsigned int) * (kViewRegions * kViewCountStride + 2), hipMemcpyDeviceToHost, s.last_stream));
    }
    HIPCHK(hipStreamSynchronize(s.last_stream));
    return 1;
}

// batches that were never waited for may still owe their fallback list (slab functions run on a caller's stream and do not
// synchronise the device; this only blocks when such a batch exists)
int flush_pending(mm_freq* h);
int settle(mm_freq* h) {
    { int rf = flush_pending(h); if (rf) return rf; }
    for (auto& s : h->slots) {
        if (!s.fb_deferred) continue;
        HIPCHK(hipEventSynchronize(s.ev_wait));
        int r = finish_deferred(h, s);
        if (r < 0) return r;
    }
    return 0;
}

int slot_status(mm_freq* h, Slot& s, int32_t* bad_read);

int drain(mm_freq* h);
// before a launch: when the side lists are filling up (looked at every few launches; the look waits for the launches under
// way), everything is brought to an end and the lists are compacted
int side_room(mm_freq* h) {
    if (!h->side_possible || (++h->launches_since_side_check & 3u) != 0u) return 0;
    HIPCHK(hipDeviceSynchronize());
    const int need = side_check(h);
    if (need < 0) return need;
    if (!need) return 0;
    { int r = drain(h); if (r) return r; }
    return side_compact(h);
}

// launch the gathered group, if there is one
int flush_pending(mm_freq* h) {
    if (h->pending_slot < 0) return 0;
    Slot& s = h->slots[h->pending_slot];
    const int members = h->pending_members;
    const uint64_t bases = h->pending_bases;
    h->pending_slot = -1; h->pending_members = 0; h->pending_host = false; h->pending_bases = 0;
    int r = launch_k1(h, s, &h->pending_batch, h->pending_stream, bases);
    s.members = members;
    if (r) { s.busy = false; if (!h->sticky_err) h->sticky_err = -r; }
    return r;
}

// every batch submitted so far is complete, fallback lists included (before counters are read or changed)
int drain(mm_freq* h) {
    { int rf = flush_pending(h); if (rf) return rf; }
    HIPCHK(hipDeviceSynchronize());
    for (auto& s : h->slots) {
        int r = finish_deferred(h, s);
        if (r < 0) return r;
        if (s.busy) {   // never waited for: its reads' errors still count
            int32_t bad = -1;
            int e = slot_status(h, s, &bad);
            if (e && !h->sticky_err) { h->sticky_err = e; h->sticky_read = bad; h->sticky_ticket = (int)(&s - h->slots); }
            s.busy = false;
        }
    }
    return 0;
}

// status of a slot whose kernels are complete: 0, or the first failing read's code (its batch index in *bad_read)
int slot_status(mm_freq* h, Slot& s, int32_t* bad_read) {
    if (s.h_ctl[129] == 0xFFFFFFFFu) return 0;
    unsigned int sum = 0xFFFFFFFFu;
    if (hipMemcpy(&sum, s.d_err_word, sizeof sum, hipMemcpyDeviceToHost) != hipSuccess) return MM_E_HIP;
    if (bad_read) *bad_read = (int32_t)(sum >> 8);
    return (int)(sum & 0xFFu);
}

int acquire_slot(mm_freq* h) {
    int i = h->next_slot;
    h->next_slot = (h->next_slot + 1) % kSlots;
    Slot& s = h->slots[i];
    if (s.busy) {
        // the ticket was never waited for: its outcome must not vanish with the slot
        int e = hipEventSynchronize(s.ev_wait) == hipSuccess ? 0 : MM_E_HIP;
        if (!e && finish_deferred(h, s) < 0) e = MM_E_HIP;
        int32_t bad = -1;
        if (!e) e = slot_status(h, s, &bad);
        if (e && !h->sticky_err) { h->sticky_err = e; h->sticky_read = bad; h->sticky_ticket = i; }
        s.busy = false;
    } else if (finish_deferred(h, s) < 0 && !h->sticky_err) {
        h->sticky_err = MM_E_HIP;
    }
    return i;
}

}  // namespace

extern "C" {

int32_t mm_abi_version(void) { return MM_ABI_VERSION; }

const char* mm_strerror(int32_t code) {
    switch (code < 0 ? -code : code) {
        case MM_OK: return "ok";
        case MM_E_HARDCLIP: return "hard clipping found (not supported)";
        case MM_E_CIGAROP: return "unhandled CIGAR operation";
        case MM_E_MMBASE: return "invalid base in MM tag";
        case MM_E_MMSTRAND: return "invalid strand in MM tag";
        case MM_E_MMCODE: return "invalid base modification code in MM tag";
        case MM_E_MMEMPTY: return "empty modification codes in MM tag";
        case MM_E_MMMIXED: return "modification codes both numeric and alphabetic";
        case MM_E_SKIPLEN: return "skip count longer than 9 characters";
        case MM_E_SKIPVAL: return "invalid skip count";
        case MM_E_READPOS: return "read position exceeds sequence length";
        case MM_E_MLIDX: return "mod prob index mismatch (ML shorter than MM)";
        case MM_E_NOCONTIG: return "contig not found in reference provided";
        case MM_E_REFPOS: return "reference position outside contig";
        case MM_E_QOVER: return "CIGAR longer than the sequence";
        case MM_E_SIDEFULL: return "sparse side list full";
        case MM_E_ARG: return "invalid argument";
        case MM_E_HIP: return "HIP runtime error";
        case MM_E_NOMEM: return "out of device memory";
        case MM_E_TOOMANY: return "too many modification codes (or, in view mode, more than 2048 MM groups in one read)";
        case MM_E_NOCODE: return "modification code not interned";
        case MM_E_OVERFLOW: return "n_called overflowed (more than 4294967295 calls on one key)";
        default: return "unknown error";
    }
}

void mm_freq_destroy(mm_freq_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_ipc_slab) (void)mmdev::dfree(h->d_ipc_slab);
    (void)flush_pending(h);
    (void)hipDeviceSynchronize();
    for (auto& s : h->slots) {
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.ev_start) (void)hipEventDestroy(s.ev_start);
        if (s.ev_stop) (void)hipEventDestroy(s.ev_stop);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_copied) (void)hipEventDestroy(s.ev_copied);
        void* ps[] = {s.d_reads, s.d_cigar, s.d_seq, s.d_mm, s.d_ml, s.d_order, s.d_status, s.d_spill, s.d_ctl, s.d_tq,
                      s.d_gcq, s.d_gcr, s.d_gdir, s.d_gqtot, s.d_gnb, s.d_gqdir, s.d_grdir, s.d_gsum, s.d_gtok, s.d_tiles, s.d_fb,
                      s.d_plan, s.d_plan_stream, s.d_plan_state,
                      s.d_vkeys, s.d_vvals, s.d_vseq, s.d_vtile, s.d_vcount, s.d_ka, s.d_va, s.d_vrows, s.d_vout, s.d_vreadcount, s.d_voff, s.d_vcursor,
                      s.d_vkept, s.d_vnewoff};
        for (void* p : ps) if (p) (void)mmdev::dfree(p);
        if (s.h_ctl) (void)mmdev::hfree(s.h_ctl);
        if (s.h_vcount) (void)mmdev::hfree(s.h_vcount);
        if (s.h_vrows) (void)mmdev::hfree(s.h_vrows);
    }
    void* ps[] = {h->d_refw, h->d_ref_base, h->d_ctg_len, h->d_seg_begin, h->d_seg_len, h->d_cnt_base, h->d_counters,
                  h->d_mods, h->d_codes, h->d_side, h->d_side_count, h->d_stats, h->d_tile_counts, h->d_tile_offsets, h->d_rows,
                  h->d_stab, h->d_scount, h->d_scur, h->d_base_k, h->d_base_v, h->d_sort_k[0], h->d_sort_k[1], h->d_sort_v[0], h->d_sort_v[1], h->d_sort_hist};
    for (void* p : ps) if (p) (void)mmdev::dfree(p);
    for (void* p : h->ipc_open) if (p) (void)hipIpcCloseMemHandle(p);
    for (void* p : h->d_site_arrays) if (p) (void)mmdev::dfree(p);
    if (h->d_ctx_mods) (void)mmdev::dfree(h->d_ctx_mods);
    if (h->d_classes) (void)mmdev::dfree(h->d_classes);
    if (h->d_cls_of_mod) (void)mmdev::dfree(h->d_cls_of_mod);
    if (h->d_adj) (void)mmdev::dfree(h->d_adj);
    if (h->d_slab_flag) (void)mmdev::dfree(h->d_slab_flag);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// the site index failed its check (k_site_check): what the device's pieces hold, set against what the host computes from the same counts, and what the
// workgroups of every XCD see of them -- the lines that named round 5's defect (profiles/r6_site_index_root_cause.txt)
// what every XCD sees of the index: res[0..7] blocks that break the chain by plain loads, res[8..15] by agent-scope loads, per XCC_ID of the observer; res[16 + x] tiles observed by XCC x
static __global__ __launch_bounds__(256) void k_site_views(const uint2* __restrict__ site, int stride, const uint32_t* __restrict__ cnt, int64_t n_blocks, uint32_t* __restrict__ res) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    if (threadIdx.x == 0) atomicAdd(&res[16 + xcc], 1u);
    for (int64_t b = (int64_t)blockIdx.x * kScanTile + threadIdx.x; b + 1 < n_blocks && b < (int64_t)(blockIdx.x + 1) * kScanTile; b += 256) {
        const uint2 w = site[b * stride]; const uint32_t nx = site[(b + 1) * stride].y, c = cnt[b];
        const uint32_t* q = reinterpret_cast<const uint32_t*>(site);
        const uint32_t ax = __hip_atomic_load(&q[(b * stride) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ay = __hip_atomic_load(&q[(b * stride) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t an = __hip_atomic_load(&q[((b + 1) * stride) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ac = __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w.y + (uint32_t)__popc(w.x) != nx || c != (uint32_t)__popc(w.x)) atomicAdd(&res[xcc], 1u);
        if (ay + (uint32_t)__popc(ax) != an || ac != (uint32_t)__popc(ax)) atomicAdd(&res[8 + xcc], 1u);
    }
}
static void site_index_diag(int sd, uint32_t bad, const uint2* site, int stride, const uint32_t* cnt, int64_t n_blocks, const uint32_t* tsum, int64_t n_tiles, uint32_t total) {
    const std::vector<uint32_t> prev;
    {
        uint32_t* d_res = nullptr; uint32_t hr[24] = {0};
        if (mmdev::dmalloc((void**)&d_res, sizeof hr) == hipSuccess && hipMemset(d_res, 0, sizeof hr) == hipSuccess) {
            hipLaunchKernelGGL(k_site_views, dim3((unsigned)n_tiles), dim3(256), 0, 0, site, stride, cnt, n_blocks, d_res);
            (void)hipMemcpy(hr, d_res, sizeof hr, hipMemcpyDeviceToHost);
            std::fprintf(stderr, "[site-diag]   seen again by a kernel, per XCD of the observer (tiles / bad blocks by plain loads / by agent-scope loads):");
            for (int x = 0; x < 8; x++) std::fprintf(stderr, " %u/%u/%u", hr[16 + x], hr[x], hr[8 + x]);
            std::fprintf(stderr, "\n");
            (void)mmdev::dfree(d_res);
        }
        std::fprintf(stderr, "[site-diag]   addresses: site words %p + %zu; counts %p + %zu; tile sums %p\n", (const void*)site, sizeof(uint2) * (size_t)n_blocks * (size_t)stride, (const void*)cnt, 4 * (size_t)n_blocks, (const void*)tsum);
    }
    std::vector<uint32_t> c((size_t)n_blocks), ts((size_t)n_tiles + 1);
    std::vector<uint2> sw((size_t)n_blocks * (size_t)stride);
    (void)hipMemcpy(c.data(), cnt, 4 * c.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(ts.data(), tsum, 4 * ts.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(sw.data(), site, sizeof(uint2) * sw.size(), hipMemcpyDeviceToHost);
    std::vector<uint32_t> S((size_t)n_tiles + 1, 0), O((size_t)n_tiles + 1, 0);
    for (int64_t b = 0; b < n_blocks; b++) S[(size_t)(b / kScanTile)] += c[(size_t)b];
    for (int64_t t = 0; t < n_tiles; t++) O[(size_t)t + 1] = O[(size_t)t] + S[(size_t)t];
    int64_t ts_bad = 0, ts_is_sum = 0, first_bad = 0, first_is_zero = 0, first_is_sum = 0, first_is_prev = 0, intile_bad = 0, cnt_bad = 0;
    for (int64_t t = 0; t <= n_tiles; t++) { ts_bad += ts[(size_t)t] != O[(size_t)t]; ts_is_sum += ts[(size_t)t] == S[(size_t)t]; }
    for (int64_t t = 0; t < n_tiles; t++) {
        const uint32_t got = sw[(size_t)(t * kScanTile) * (size_t)stride].y;
        if (got != O[(size_t)t]) { first_bad++; first_is_zero += got == 0u; first_is_sum += got == S[(size_t)t]; first_is_prev += !prev.empty() && got == prev[(size_t)t]; }
    }
    for (int64_t b = 0; b < n_blocks; b++) {
        cnt_bad += c[(size_t)b] != (uint32_t)__builtin_popcount(sw[(size_t)b * (size_t)stride].x);
        if (b + 1 < n_blocks && (b + 1) % kScanTile) intile_bad += sw[(size_t)(b + 1) * (size_t)stride].y - sw[(size_t)b * (size_t)stride].y != c[(size_t)b];
    }
    std::fprintf(stderr, "[site-diag] strand %d: check counted %u bad blocks of %lld (%lld tiles); total seen %u, expected %u\n", sd, bad, (long long)n_blocks, (long long)n_tiles, total, O[(size_t)n_tiles]);
    std::fprintf(stderr, "[site-diag]   tile offsets in memory now: %lld of %lld differ from the expected offsets (%lld equal the unscanned sums)\n", (long long)ts_bad, (long long)n_tiles + 1, (long long)ts_is_sum);
    std::fprintf(stderr, "[site-diag]   tiles whose first rank is not the expected offset: %lld (of these: %lld zero, %lld the tile's own sum, %lld the other strand's offset)\n", (long long)first_bad, (long long)first_is_zero, (long long)first_is_sum, (long long)first_is_prev);
    std::fprintf(stderr, "[site-diag]   ranks wrong inside a tile: %lld; counts that are not their bits' popcount: %lld\n", (long long)intile_bad, (long long)cnt_bad);
    int shown = 0;
    for (int64_t t = 0; t < n_tiles && shown < 6; t++) {
        const uint32_t got = sw[(size_t)(t * kScanTile) * (size_t)stride].y;
        if (got != O[(size_t)t] || ts[(size_t)t] != O[(size_t)t]) { shown++; std::fprintf(stderr, "[site-diag]   tile %lld: first rank %u, offset in memory %u, expected %u, tile sum %u, other strand's offset %u\n", (long long)t, got, ts[(size_t)t], O[(size_t)t], S[(size_t)t], prev.empty() ? 0u : prev[(size_t)t]); }
    }
}

mm_freq_t* mm_freq_create(const mm_freq_opts_t* opts, int32_t n_contigs, const mm_contig_t* contigs,
                          int32_t n_intervals, const mm_interval_t* intervals, char* err, size_t err_len) {
    auto fail = [&](mm_freq* h, const char* msg) -> mm_freq_t* {
        if (err && err_len) std::snprintf(err, err_len, "%s", msg);
        if (h) mm_freq_destroy(h);
        return nullptr;
    };
    if (!opts || opts->abi_version != MM_ABI_VERSION) return fail(nullptr, "ABI version mismatch");
    if (opts->n_mods < 1 || opts->n_mods > MM_MAX_MODS) return fail(nullptr, "n_mods out of range (1..32)");
    if (n_contigs < 0 || (n_contigs > 0 && !contigs)) return fail(nullptr, "bad contig table");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "no HIP device (the HIP path has no CPU fallback)");
    if (opts->device < 0 || opts->device >= ndev) return fail(nullptr, "bad device ordinal");
    mm_freq* h = new mm_freq();
    h->opts = *opts;
    if (h->opts.coalesce <= 0) h->opts.coalesce = 32;   // the library's default (ABI 5); 1 = every submit its own launch
    h->device = opts->device;
    if (hipSetDevice(h->device) != hipSuccess) return fail(h, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) return fail(h, "hipGetDeviceProperties failed");
    h->n_cu = prop.multiProcessorCount;
    const bool tl_on = std::getenv("MM_TIMELINE") != nullptr;
    auto tl_now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    const double tl_a = tl_now();
    // four bits a position need a context made of A C G T (or `*`): only then does no other reference letter ever sit in a match
    auto plain_context = [&]() {
        const char* c = opts->mods[0].context;
        if (std::strcmp(c, "*") == 0) return true;
        for (size_t j = 0; j < MM_CODE_LEN && c[j]; j++) if (!std::strchr("ACGT", c[j])) return false;
        return true;
    };
    // context classes (ABI 6): entries with one context string share a class -- its two bits of the reference word, its site index, its
    // counters side by side per site.  The word's width goes by the classes, so thirty-two entries over a handful of contexts fit
    {
        h->cls_of_mod.assign(opts->n_mods, 0);
        h->first_mod.clear();
        for (int i = 0; i < opts->n_mods; i++) {
            int c = -1;
            for (size_t k = 0; k < h->first_mod.size(); k++) if (std::strcmp(opts->mods[h->first_mod[k]].context, opts->mods[i].context) == 0) c = (int)k;
            if (c < 0) { c = (int)h->first_mod.size(); h->first_mod.push_back(i); }
            h->cls_of_mod[i] = c;
        }
        h->n_classes = (int)h->first_mod.size();
        // (round 6: more than MM_MAX_CONTEXTS = 13 different contexts -- what a 32-bit reference word holds -- are built in passes of 13: the site index of a
        // class comes from the pass that carries its bits, and the kernels that test a position's context in the reference word take classes 13 and up from
        // the site word instead.  The reference has no limit of its own, src/mod.c:204-326)
    }
    h->ref_kind = h->n_classes > 5 ? 2 : (opts->n_mods == 1 && plain_context() ? 0 : 1);
#ifdef MM_KIND
    if (h->ref_kind != MM_KIND) return fail(h, "the handle's reference-word kind is another copy's (freq_dispatch.cpp picks the copy by the same rule)");
#endif
    const double tl_a0 = tl_now();
    double tl_a1 = tl_a0;
    double tl_last = tl_a0;
    auto big_maps = []() {   // (MM_TIMELINE: the mappings of 64 MB or more that are resident, counted from /proc/self/smaps)
        int n = 0; double mb = 0; char line[512]; unsigned long kb;
        if (FILE* f = std::fopen("/proc/self/smaps", "r")) { while (std::fgets(line, sizeof line, f)) if (std::sscanf(line, "Rss: %lu kB", &kb) == 1 && kb >= 65536) { n++; mb += (double)kb / 1024.0; } std::fclose(f); }
        std::fprintf(stderr, "[timeline]          %d big mappings, %.0f MB resident in them\n", n, mb);
    };
    auto tl_step = [&](const char* what) { if (tl_on) { const double t = tl_now(); std::fprintf(stderr, "[timeline] mm_freq_create:   %-44s %.3f s\n", what, t - tl_last); if (std::atoi(std::getenv("MM_TIMELINE")) >= 2) big_maps(); tl_last = tl_now(); } };
    {
        int nb = 0;
        hipError_t e = hipSuccess;
        MM_REF_DISPATCH(h, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (k_freq_reads<RW, false>), 256, 0));
        h->blocks_per_cu = (e == hipSuccess && nb > 0) ? nb : 2;
        int na = 0, nc = 0;
        const bool plain = !opts->insertions && !opts->haplotypes;
        MM_REF_DISPATCH(h,
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&na, k_scan_reads<RW>, 256, 0);
            if (opts->view) { if (plain) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, (k_call_tiles<RW, true, true>), 256, 0);
                              else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, (k_call_tiles<RW, true, false>), 256, 0); }
            else { if (plain) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, (k_call_tiles<RW, false, true>), 256, 0);
                   else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nc, (k_call_tiles<RW, false, false>), 256, 0); });
        h->scan_blocks_per_cu = na > 0 ? std::min(na, 8) : 4;
        h->call_blocks_per_cu = nc > 0 ? std::min(nc, 8) : 4;
        int nf = 0;
        MM_REF_DISPATCH(h,
            if (!plain) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, (k_stream_reads<RW, false, false, false, true>), 256, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, (k_stream_reads<RW, false, false, false, false>), 256, 0));
        int nfd = 0;   // the '.'-capable instantiation keeps more registers and one wavefront per SIMD fewer
        MM_REF_DISPATCH(h,
            if (!plain) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nfd, (k_stream_reads<RW, false, true, false, true>), 256, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nfd, (k_stream_reads<RW, false, true, false, false>), 256, 0));
        if (getenv("MM_DEBUG_OCC")) std::fprintf(stderr, "[minimod_hip] k_stream_reads: %d workgroups per CU (%d with '.' groups)\n", nf, nfd);
        h->stream_blocks_per_cu = nf > 0 ? std::min(nf, 8) : 4;
        h->stream_blocks_per_cu_dot = nfd > 0 ? std::min(nfd, 8) : 4;
        h->stream_blocks_per_cu_dot_ins = h->stream_blocks_per_cu_dot;   // (one of the two is this handle's: `plain` says which)
        tl_a1 = tl_now();
        tl_step("occupancy queries");
#ifdef MM_STREAM_GRID_BLOCKS   // experiment: fewer resident workgroups per CU
        h->stream_blocks_per_cu = std::min(h->stream_blocks_per_cu, MM_STREAM_GRID_BLOCKS);
        h->stream_blocks_per_cu_dot = std::min(h->stream_blocks_per_cu_dot, MM_STREAM_GRID_BLOCKS);
#endif
        h->use_tiles = opts->force_fused == 0;
    }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(h, "stream create failed");
    for (auto& s : h->slots) {
        // (s.stream: made when a launch first asks for it -- slot_stream(); a caller that brings its own stream, like the device-side reader's
        // chain stream, never does, and a stream is 8 - 10 ms of the process's start)
        if (hipEventCreate(&s.ev_start) != hipSuccess || hipEventCreate(&s.ev_stop) != hipSuccess || hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming) != hipSuccess) return fail(h, "event create failed");
        if (dev_alloc(h, (void**)&s.d_ctl, 2 * kCtlSetWords * sizeof(unsigned int))) return fail(h, "alloc failed");
        {
            unsigned int init[2 * kCtlSetWords];
            for (int i = 0; i < 2 * kCtlSetWords; i++) init[i] = (i % kCtlSetWords) == 1 ? 0xFFFFFFFFu : 0u;
            if (hipMemcpy(s.d_ctl, init, sizeof init, hipMemcpyHostToDevice) != hipSuccess) return fail(h, "control word init failed");
            if (dev_alloc(h, (void**)&s.d_tq, 2 * kQueueWords * sizeof(unsigned int))) return fail(h, "alloc failed");
            if (hipMemset(s.d_tq, 0, 2 * kQueueWords * sizeof(unsigned int)) != hipSuccess) return fail(h, "control word init failed");
        }
        if (mmdev::hmalloc((void**)&s.h_ctl, 512 * sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) return fail(h, "pinned alloc failed");
        std::memset(s.h_ctl, 0, 512 * sizeof(unsigned int));   // (every slot's word 132 is looked at, used or not: a kept block is not a zeroed one)
        if (opts->view) {
            if (dev_alloc(h, (void**)&s.d_vcount, sizeof(unsigned int) * (kViewRegions * kViewCountStride + 2))) return fail(h, "alloc failed");
            if (mmdev::hmalloc((void**)&s.h_vcount, sizeof(unsigned int) * (kViewRegions * kViewCountStride + 2), hipHostMallocDefault) != hipSuccess)
                return fail(h, "pinned alloc failed");
        }
    }
    tl_step("streams, events, control words of the slots");
    // ---- mods / codes
    std::vector<DevMod> mods(opts->n_mods);
    h->wildcard = -1;
    for (int i = 0; i < opts->n_mods; i++) {
        const mm_mod_t& m = opts->mods[i];
        DevMod& d = mods[i];
        std::memset(&d, 0, sizeof(d));
        std::memcpy(d.klass, m.klass, 256);
        d.t_hi = 256; d.t_lo = -1;
        for (int x = 255; x >= 0; x--) if (m.klass[x] == 3) d.t_hi = x; else break;
        for (int x = 0; x < 256; x++) if (m.klass[x] == 1) d.t_lo = x; else break;
        for (int x = 0; x < 256; x++) {
            int want = x >= d.t_hi ? 3 : (x <= d.t_lo ? 1 : 0);
            if (m.klass[x] != want) return fail(h, "klass table must be monotone (called | ambiguous | modified)");
        }
        size_t cl = strnlen(m.context, MM_CODE_LEN);
        if (cl >= MM_CODE_LEN) return fail(h, "context too long");
        d.ctx_is_star = std::strcmp(m.context, "*") == 0;
        d.ctx_len = (int)cl;
        for (size_t j = 0; j < cl; j++) {
            d.ctx_fwd[j] = m.context[j];
            d.ctx_rev[j] = (char)complement((unsigned char)m.context[cl - 1 - j]);  // ref.c:183-193
        }
        if (strnlen(m.code, MM_CODE_LEN) >= MM_CODE_LEN) return fail(h, "code too long");
        if (std::strcmp(m.code, "*") == 0) h->wildcard = i;
    }
    if (h->wildcard >= 0) {
        h->n_code_planes = opts->n_wild_planes > 0 ? opts->n_wild_planes : 4;
    } else {
        h->n_code_planes = opts->n_mods;
        for (int i = 0; i < opts->n_mods; i++) {
            DevCode c;
            std::memset(&c, 0, sizeof(c));
            std::snprintf(c.str, MM_CODE_LEN, "%s", opts->mods[i].code);
            c.len = (int16_t)std::strlen(c.str); c.req = (int16_t)i; c.plane = (int16_t)i; c.slot = 0;   // (slot: set with the context classes below)
            h->codes.push_back(c);
        }
    }
    h->n_hp = opts->haplotypes ? (opts->n_hp_planes > 0 ? std::min(opts->n_hp_planes, MM_MAX_HP_PLANES) : 4) : 1;
    // opts.side_capacity sizes the side LISTS (records of 16 bytes in all, shared out to kSideRegions regions; when a region is half
    // full the lists are compacted to one record per key); the list behind them only takes what has no 64-bit key (haplotype
    // tags above 29, positions past 2^35)
    h->side_cap = opts->view ? 16 : (int64_t)(1 << 16);
    {
        unsigned long long want = opts->view ? 1024ull : (unsigned long long)(opts->side_capacity > 0 ? opts->side_capacity : (int64_t)(16 << 20));
        h->stab_slots = std::max<unsigned long long>((want + kSideRegions - 1) / kSideRegions, 16ull);
        if (dev_alloc(h, (void**)&h->d_stab, 16 * (size_t)h->stab_slots * kSideRegions) || dev_alloc(h, (void**)&h->d_scount, 8) ||
            dev_alloc(h, (void**)&h->d_scur, sizeof(unsigned int) * kSideRegions * kSideCurStride)) return fail(h, "side list alloc failed");
        if (side_table_clear(h) != 0) return fail(h, "side list init failed");
        h->side_possible = !opts->view && (opts->insertions || opts->haplotypes || h->wildcard >= 0 || n_intervals > 0);
    }
    tl_step("side lists (alloc + clear)");
    if (dev_alloc(h, (void**)&h->d_mods, sizeof(DevMod) * mods.size())) return fail(h, "alloc failed");
    if (dev_alloc(h, (void**)&h->d_codes, sizeof(DevCode) * MM_MAX_CODES)) return fail(h, "alloc failed");
    if (dev_alloc(h, (void**)&h->d_side, sizeof(SideRec) * (size_t)h->side_cap)) return fail(h, "side list alloc failed");
    if (dev_alloc(h, (void**)&h->d_side_count, sizeof(unsigned long long))) return fail(h, "alloc failed");
    (void)hipMemcpy(h->d_mods, mods.data(), sizeof(DevMod) * mods.size(), hipMemcpyHostToDevice);
    {
        std::vector<DevMod> cm((size_t)h->n_classes);
        for (int c = 0; c < h->n_classes; c++) cm[(size_t)c] = mods[(size_t)h->first_mod[(size_t)c]];
        if (dev_alloc(h, (void**)&h->d_ctx_mods, sizeof(DevMod) * cm.size())) return fail(h, "alloc failed");
        (void)hipMemcpy(h->d_ctx_mods, cm.data(), sizeof(DevMod) * cm.size(), hipMemcpyHostToDevice);
    }
    (void)hipMemset(h->d_side_count, 0, sizeof(unsigned long long));
    if (dev_alloc(h, (void**)&h->d_stats, (16 + 4 * (size_t)kStatSlots) * sizeof(unsigned long long))) return fail(h, "alloc failed");
    (void)hipMemset(h->d_stats, 0, (16 + 4 * (size_t)kStatSlots) * sizeof(unsigned long long));
    h->codes_dirty = !h->codes.empty();
    tl_step("mods, codes, stats");
    // ---- contigs: reference words for every contig that has a sequence; counter segments
    h->n_contigs = n_contigs;
    h->names.resize(n_contigs);
    h->ctg_len.assign(n_contigs, 0); h->ref_base.assign(n_contigs, -1);
    h->seg_begin.assign(n_contigs, 0); h->seg_len.assign(n_contigs, 0); h->cnt_base.assign(n_contigs, 0);
    int64_t ref_total = 0;
    for (int t = 0; t < n_contigs; t++) {
        h->names[t] = contigs[t].name ? contigs[t].name : "";
        h->ctg_len[t] = contigs[t].length;
        if (contigs[t].seq) {
            if (contigs[t].seq_length != contigs[t].length) {   // mod.c:861 ref_len == target_len
                char msg[256];
                std::snprintf(msg, sizeof msg, "ref_len:%lld target_len:%lld for contig %s", (long long)contigs[t].seq_length,
                              (long long)contigs[t].length, h->names[t].c_str());
                return fail(h, msg);
            }
            h->ref_base[t] = ref_total;
            ref_total += (contigs[t].length + 63) / 64 * 64;
        }
    }
    h->ref_total = ref_total;
    int64_t plane_len = 0;
    if (n_intervals > 0) {
        for (int i = 0; i < n_intervals; i++) {
            const mm_interval_t& iv = intervals[i];
            if (iv.tid < 0 || iv.tid >= n_contigs || h->ref_base[iv.tid] < 0) return fail(h, "interval on unknown contig");
            int64_t b = std::max<int64_t>(0, iv.begin);
            int64_t e = std::min<int64_t>(h->ctg_len[iv.tid], iv.end + iv.halo);
            if (e <= b || h->seg_len[iv.tid] != 0) return fail(h, "bad or duplicate interval");
            h->seg_begin[iv.tid] = b; h->seg_len[iv.tid] = e - b; h->cnt_base[iv.tid] = plane_len;
            plane_len += (e - b + 63) / 64 * 64;
        }
    } else {
        for (int t = 0; t < n_contigs; t++) {
            if (h->ref_base[t] < 0) continue;
            h->seg_begin[t] = 0; h->seg_len[t] = h->ctg_len[t]; h->cnt_base[t] = plane_len;
            plane_len += (h->ctg_len[t] + 63) / 64 * 64;
        }
    }
    h->plane_len = plane_len;
    // contig order of the output: strcmp on names (cmp_key_fast, mod.c:59-76)
    {
        std::vector<int> idx(n_contigs);
        for (int t = 0; t < n_contigs; t++) idx[t] = t;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::strcmp(h->names[a].c_str(), h->names[b].c_str()) < 0; });
        h->ctg_rank.assign(n_contigs, 0);
        for (int r = 0; r < n_contigs; r++) h->ctg_rank[idx[r]] = r;
    }
    const size_t ref_bytes = h->ref_kind == 0 ? (size_t)std::max<int64_t>(ref_total, 64) / 2 : (size_t)std::max<int64_t>(ref_total, 64) * (h->ref_kind == 2 ? 4 : 2);
    if (dev_alloc(h, &h->d_refw, ref_bytes)) return fail(h, "reference alloc failed");
    // (the padding behind every contig: no sites there.  On the stream the context kernels run on: a plain hipMemset is not
    // ordered with a non-blocking stream and could land on top of their words)
    if (hipMemsetAsync(h->d_refw, 0, ref_bytes, h->stream) != hipSuccess) return fail(h, "reference memset failed");
    size_t tb = sizeof(int64_t) * (size_t)std::max(n_contigs, 1);
    if (dev_alloc(h, (void**)&h->d_ref_base, tb) || dev_alloc(h, (void**)&h->d_ctg_len, tb) || dev_alloc(h, (void**)&h->d_seg_begin, tb) ||
        dev_alloc(h, (void**)&h->d_seg_len, tb) || dev_alloc(h, (void**)&h->d_cnt_base, tb)) return fail(h, "alloc failed");
    if (n_contigs > 0) {
        (void)hipMemcpy(h->d_ref_base, h->ref_base.data(), tb, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_ctg_len, h->ctg_len.data(), tb, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_seg_begin, h->seg_begin.data(), tb, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_seg_len, h->seg_len.data(), tb, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_cnt_base, h->cnt_base.data(), tb, hipMemcpyHostToDevice);
    }
    tl_step("reference words' buffer + contig tables");
    const double tl_b = tl_now();
    // K0 per contig through a staging buffer.  A PASS carries the context bits of up to MM_MAX_CONTEXTS classes (pass q: classes 13 q ... 13 q + 12 in
    // bits 5 ... 30); runs with more classes come back for the later passes while the site indices are built, and once more for pass 0, whose bits stay
    auto build_ref_words = [&](int pass) -> bool {
        int64_t maxlen = 0;
        for (int t = 0; t < n_contigs; t++) if (h->ref_base[t] >= 0) maxlen = std::max(maxlen, h->ctg_len[t]);
        uint8_t* d_raw = nullptr;
        if (maxlen <= 0) return true;
        if (mmdev::dmalloc((void**)&d_raw, (size_t)maxlen + 64) != hipSuccess) return false;   // (k_build_refnibs loads whole 16-byte pieces)
        const DevMod* pass_mods = h->d_ctx_mods + (size_t)pass * MM_MAX_CONTEXTS;
        const int pass_n = std::min(MM_MAX_CONTEXTS, h->n_classes - pass * MM_MAX_CONTEXTS);
        bool ok = true;
        for (int t = 0; t < n_contigs && ok; t++) {
            if (h->ref_base[t] < 0 || h->ctg_len[t] == 0) continue;
            int64_t len = h->ctg_len[t];
            if (hipMemcpy(d_raw, contigs[t].seq, (size_t)len, hipMemcpyHostToDevice) != hipSuccess) { ok = false; break; }
            int blocks = (int)std::min<int64_t>((len + 255) / 256, (int64_t)h->n_cu * 16);
            // (bits 5 + 2c / 6 + 2c: class c's context, described by the class's first entry)
            if (h->ref_kind == 2) hipLaunchKernelGGL(k_build_refwords<uint32_t>, dim3(blocks), dim3(256), 0, h->stream, d_raw, len,
                                                     (uint32_t*)h->d_refw + h->ref_base[t], pass_mods, pass_n);
            else if (h->ref_kind == 1) hipLaunchKernelGGL(k_build_refwords<uint16_t>, dim3(blocks), dim3(256), 0, h->stream, d_raw, len,
                                                          (uint16_t*)h->d_refw + h->ref_base[t], pass_mods, pass_n);
            else if (std::strlen(opts->mods[0].context) <= 4)
                hipLaunchKernelGGL(k_build_refnibs<4>, dim3((unsigned)std::min<int64_t>((len / 16 + 256) / 256, (int64_t)h->n_cu * 16)), dim3(256), 0, h->stream, d_raw, len,
                                   (uint8_t*)h->d_refw + h->ref_base[t] / 2, h->d_mods);   // (ref_base: a multiple of 64)
            else hipLaunchKernelGGL(k_build_refnibs<15>, dim3((unsigned)std::min<int64_t>((len / 16 + 256) / 256, (int64_t)h->n_cu * 16)), dim3(256), 0, h->stream, d_raw, len,
                                    (uint8_t*)h->d_refw + h->ref_base[t] / 2, h->d_mods);
            ok = hipStreamSynchronize(h->stream) == hipSuccess;
        }
        (void)mmdev::dfree(d_raw);
        return ok;
    };
    if (!build_ref_words(0)) return fail(h, "reference upload or context kernel failed");
    const double tl_c = tl_now();
    // ---- context classes: mods with one context string share a class (their counters lie side by side per site); every position
    // is a site of a DENSE class: the context `*`, and any class under --insertions, where no context is looked at (mod.c:1167-1172)
    {
        const std::vector<int>& first_mod = h->first_mod;   // (the classes were made in front of the reference words: their bits are per class)
        h->classes.assign(h->n_classes, DevClass{});
        // code planes -> (class, slot): plane i belongs to mod i, or, with -c '*', every plane to the one `*` mod
        h->plane_cls.assign(h->n_code_planes, 0); h->plane_slot.assign(h->n_code_planes, 0);
        std::vector<int> np(h->n_classes, 0);
        for (int pl = 0; pl < h->n_code_planes; pl++) {
            const int mod = h->wildcard >= 0 ? h->wildcard : pl;
            const int c = h->cls_of_mod[mod];
            h->plane_cls[pl] = c; h->plane_slot[pl] = np[c]++;
        }
        for (auto& dc : h->codes) if (dc.plane >= 0) dc.slot = (int16_t)h->plane_slot[(size_t)dc.plane];
        const int64_t n_blocks = (ref_total + 31) / 32;
        h->adj.assign((size_t)std::max(n_contigs, 1) * h->n_classes * 2, 0);
        int64_t words = 0;
        for (int c = 0; c < h->n_classes; c++) {
            if (c > 0 && c % MM_MAX_CONTEXTS == 0 && !build_ref_words(c / MM_MAX_CONTEXTS)) return fail(h, "reference upload or context kernel failed");   // the next thirteen classes' bits
            DevClass& k = h->classes[c];
            k.np = std::max(np[c], 1);
            k.dense = (opts->insertions || std::strcmp(opts->mods[first_mod[c]].context, "*") == 0) ? 1 : 0;
            k.site[0] = k.site[1] = nullptr;
            k.stride = h->ref_kind == 0 ? 2 : 1;   // four-bit reference words: the site words carry the bases too
            k.pad = 0;
            k.base = words;
            if (k.dense) {
                k.nsites = plane_len;
                for (int t = 0; t < n_contigs; t++) for (int sd = 0; sd < 2; sd++)
                    h->adj[((size_t)t * h->n_classes + c) * 2 + sd] = h->ref_base[t] >= 0 ? h->cnt_base[t] - h->ref_base[t] - h->seg_begin[t] : 0;
            } else {
                // the class's site index over the whole reference-word space: bits, then ranks (an exclusive scan in tiles)
                uint2* site[2] = {nullptr, nullptr};
                uint32_t* cnt[2] = {nullptr, nullptr};
                uint32_t* tsum[2] = {nullptr, nullptr};   // (one a strand: no address of the sequence changes its meaning between two kernels)
                uint32_t* d_bad = nullptr;
                const int64_t n_tiles = (n_blocks + kScanTile - 1) / kScanTile;
                for (int sd = 0; sd < 2; sd++) {
                    if (dev_alloc(h, (void**)&site[sd], sizeof(uint2) * (size_t)k.stride * (size_t)std::max<int64_t>(n_blocks, 1))) return fail(h, "site index alloc failed");
                    h->d_site_arrays.push_back(site[sd]);
                    if (mmdev::dmalloc((void**)&cnt[sd], 4 * (size_t)std::max<int64_t>(n_blocks, 1)) != hipSuccess || mmdev::dmalloc((void**)&tsum[sd], 4 * (size_t)(n_tiles + 1)) != hipSuccess)
                        return fail(h, "site index alloc failed");
                }
                if (mmdev::dmalloc((void**)&d_bad, 8) != hipSuccess) return fail(h, "site index alloc failed");
                uint32_t total[2] = {0, 0};
                bool ok = true;
                // Both strands' kernels are queued behind each other and waited for once.  Behind them the guard (k_site_check): every block's rank plus its own
                // sites is the next block's rank, the last block ends at the strand's total.  An index that fails it is built once more and then REFUSED --
                // nothing counts into a wrong index (round 5's defect: one XCD read the counts through a freed buffer's translation, csrc/devmem.h).
                const int site_fault = std::getenv("MM_SITE_FAULT") ? std::atoi(std::getenv("MM_SITE_FAULT")) : 0;
                for (int attempt = 0; n_blocks > 0 && ok && attempt < 2; attempt++) {
                    const int blocks = (int)std::min<int64_t>((n_blocks + 255) / 256, (int64_t)h->n_cu * 16);
                    uint32_t bad[2] = {0, 0};
                    (void)hipMemsetAsync(d_bad, 0, 8, h->stream);
                    MM_REF_DISPATCH(h, hipLaunchKernelGGL(k_site_bits<RW>, dim3(blocks), dim3(256), 0, h->stream, h->d_refw, n_blocks, c % MM_MAX_CONTEXTS, (int)k.stride, site[0], site[1], cnt[0], cnt[1]));
                    for (int sd = 0; sd < 2; sd++) {
                        (void)hipMemsetAsync(tsum[sd] + n_tiles, 0, 4, h->stream);
                        hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(256), 0, h->stream, cnt[sd], n_blocks, tsum[sd]);
                        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, h->stream, tsum[sd], (unsigned long long)n_tiles + 1ull);
                        if (sd == 1 && attempt < site_fault) (void)hipMemsetAsync(cnt[1] + n_blocks / 2, 0x01, 4, h->stream);   // (the tests' way to trip the guard: MM_SITE_FAULT=1 once, =2 twice)
                        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles), dim3(256), 0, h->stream, cnt[sd], n_blocks, tsum[sd], site[sd], (int)k.stride);
                        hipLaunchKernelGGL(k_site_check, dim3(blocks), dim3(256), 0, h->stream, site[sd], (int)k.stride, cnt[sd], n_blocks, tsum[sd] + n_tiles, d_bad + sd);
                    }
                    ok = hipGetLastError() == hipSuccess;
                    for (int sd = 0; sd < 2 && ok; sd++)
                        ok = hipMemcpyAsync(&total[sd], tsum[sd] + n_tiles, 4, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipMemcpyAsync(&bad[sd], d_bad + sd, 4, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
                    ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
                    if (!ok || !(bad[0] | bad[1])) break;
                    for (int sd = 0; sd < 2; sd++) if (bad[sd]) site_index_diag(sd, bad[sd], site[sd], (int)k.stride, cnt[sd], n_blocks, tsum[sd], n_tiles, total[sd]);
                    std::fprintf(stderr, "[minimod_hip] the site index of context class %d failed its check (%u + %u blocks): %s\n", c, bad[0], bad[1], attempt == 0 ? "building it once more" : "refused");
                    if (attempt == 1) ok = false;
                }
                // the contigs' segments in the class's site numbering: ranks at every segment's two ends
                std::vector<int64_t> gq;
                std::vector<int> gt;
                for (int t = 0; t < n_contigs; t++) if (h->ref_base[t] >= 0 && h->seg_len[t] > 0) { gt.push_back(t); gq.push_back(h->ref_base[t] + h->seg_begin[t]); gq.push_back(h->ref_base[t] + h->seg_begin[t] + h->seg_len[t]); }
                std::vector<uint32_t> rk[2];
                if (ok && !gq.empty()) {
                    int64_t* d_g = nullptr; uint32_t* d_o = nullptr;
                    ok = mmdev::dmalloc((void**)&d_g, 8 * gq.size()) == hipSuccess && mmdev::dmalloc((void**)&d_o, 4 * gq.size()) == hipSuccess &&
                         hipMemcpy(d_g, gq.data(), 8 * gq.size(), hipMemcpyHostToDevice) == hipSuccess;
                    for (int sd = 0; sd < 2 && ok; sd++) {
                        rk[sd].resize(gq.size());
                        hipLaunchKernelGGL(k_rank_at, dim3((unsigned)((gq.size() + 255) / 256)), dim3(256), 0, h->stream, site[sd], (int)k.stride, d_g, (int)gq.size(), n_blocks, total[sd], d_o);
                        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(rk[sd].data(), d_o, 4 * gq.size(), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                             hipStreamSynchronize(h->stream) == hipSuccess;
                    }
                    if (d_g) (void)mmdev::dfree(d_g);
                    if (d_o) (void)mmdev::dfree(d_o);
                }
                for (int sd = 0; sd < 2; sd++) { (void)mmdev::dfree(cnt[sd]); (void)mmdev::dfree(tsum[sd]); }
                (void)mmdev::dfree(d_bad);
                if (!ok) return fail(h, "site index kernels failed, or the index failed its check twice");
                int64_t ns[2] = {0, 0};
                for (size_t i = 0; i < gt.size(); i++)
                    for (int sd = 0; sd < 2; sd++) {
                        h->adj[((size_t)gt[i] * h->n_classes + c) * 2 + sd] = ns[sd] - (int64_t)rk[sd][2 * i];
                        ns[sd] += (int64_t)rk[sd][2 * i + 1] - (int64_t)rk[sd][2 * i];
                    }
                k.nsites = std::max<int64_t>(std::max(ns[0], ns[1]), 1);
                k.site[0] = site[0]; k.site[1] = site[1];
            }
            words += (int64_t)h->n_hp * 2 * k.nsites * k.np;
        }
        if (h->n_classes > MM_MAX_CONTEXTS && !build_ref_words(0)) return fail(h, "reference upload or context kernel failed");   // (the words keep the first thirteen classes' bits)
        h->n_counter_words = opts->view ? 0 : words;   // view keeps no counters
        const size_t nadj = h->adj.size();
        if (dev_alloc(h, (void**)&h->d_classes, sizeof(DevClass) * (size_t)h->n_classes) || dev_alloc(h, (void**)&h->d_cls_of_mod, 4 * (size_t)opts->n_mods) ||
            dev_alloc(h, (void**)&h->d_adj, 8 * nadj) || dev_alloc(h, (void**)&h->d_slab_flag, 4)) return fail(h, "alloc failed");
        (void)hipMemcpy(h->d_classes, h->classes.data(), sizeof(DevClass) * (size_t)h->n_classes, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_cls_of_mod, h->cls_of_mod.data(), 4 * (size_t)opts->n_mods, hipMemcpyHostToDevice);
        (void)hipMemcpy(h->d_adj, h->adj.data(), 8 * nadj, hipMemcpyHostToDevice);
        (void)hipMemset(h->d_slab_flag, 0, 4);
        h->codes_dirty = !h->codes.empty();
    }
    const double tl_d = tl_now();
    if (dev_alloc(h, (void**)&h->d_counters, sizeof(unsigned long long) * (size_t)std::max<int64_t>(h->n_counter_words, 1)))
        return fail(h, "counter plane alloc failed");
    if (hipMemset(h->d_counters, 0, sizeof(unsigned long long) * (size_t)std::max<int64_t>(h->n_counter_words, 1)) != hipSuccess)
        return fail(h, "counter memset failed");
    if (hipDeviceSynchronize() != hipSuccess) return fail(h, "device sync failed");
    if (tl_on) std::fprintf(stderr, "[timeline] mm_freq_create: occupancy queries (the code object is loaded here) %.3f s\n", tl_a1 - tl_a0);
    if (tl_on) std::fprintf(stderr, "[timeline] mm_freq_create: tables + allocations %.3f s, reference upload + context kernels %.3f s, site index %.3f s, counters %.3f s (the runtime's start lies in front of these)\n",
                            tl_b - tl_a, tl_c - tl_b, tl_d - tl_c, tl_now() - tl_d);
    return h;
}

int32_t mm_freq_plan_batch(const mm_read_t* reads, int32_t n, int32_t* items, int32_t cap) {
    if (n < 0 || (n > 0 && (!reads || !items)) || n >= (1 << 24)) return -MM_E_ARG;
    const int split = 24576;   // measured on C2: 8192 479, 16384 507, 24576 519, 49152 517, 131072 456 Gbases/s
    // counting sort on estimated cost (bases per part, 256-base buckets), costliest first
    enum { NB = 4096 };
    std::vector<uint32_t> cnt(NB + 1, 0);
    auto parts_of = [&](uint32_t L) { uint32_t w = (L + (uint32_t)split - 1) / (uint32_t)split; return w < 1 ? 1u : (w > 16 ? 16u : w); };
    auto bucket = [&](uint32_t L, uint32_t w) { uint32_t k = (L / w) >> 8; if (k >= NB) k = NB - 1; return (uint32_t)(NB - 1 - k); };
    int64_t total = 0;
    for (int32_t i = 0; i < n; i++) { uint32_t w = parts_of(reads[i].l_qseq); cnt[bucket(reads[i].l_qseq, w) + 1] += w; total += w; }
    if (total > cap) return -MM_E_ARG;
    for (int k = 0; k < NB; k++) cnt[k + 1] += cnt[k];
    for (int32_t i = 0; i < n; i++) {
        uint32_t w = parts_of(reads[i].l_qseq), b = bucket(reads[i].l_qseq, w);
        for (uint32_t j = 0; j < w; j++) items[cnt[b]++] = (int32_t)((uint32_t)i | (j << 24) | ((w - 1) << 28));
    }
    return (int32_t)total;
}

int32_t mm_freq_intern_code(mm_freq_t* h, const char* code) {
    if (!h || !code) return -MM_E_ARG;
    size_t L = std::strlen(code);
    if (L == 0 || L >= MM_CODE_LEN) return -MM_E_ARG;
    for (size_t i = 0; i < h->codes.size(); i++)
        if (std::strcmp(h->codes[i].str, code) == 0) return (int32_t)i;
    if (h->wildcard < 0) return -MM_E_ARG;
    if (h->codes.size() >= MM_MAX_CODES) return -MM_E_TOOMANY;
    DevCode c;
    std::memset(&c, 0, sizeof(c));
    std::snprintf(c.str, MM_CODE_LEN, "%s", code);
    c.len = (int16_t)L; c.req = (int16_t)h->wildcard;
    c.plane = (int16_t)((int)h->codes.size() < h->n_code_planes ? (int)h->codes.size() : -1);
    c.slot = (int16_t)(c.plane >= 0 ? h->plane_slot[(size_t)c.plane] : 0);
    h->codes.push_back(c);
    h->codes_dirty = true;
    return (int32_t)h->codes.size() - 1;
}
int32_t mm_freq_n_codes(const mm_freq_t* h) { return h ? (int32_t)h->codes.size() : 0; }
const char* mm_freq_code_name(const mm_freq_t* h, int32_t code) {
    if (!h || code < 0 || code >= (int32_t)h->codes.size()) return "";
    return h->codes[code].str;
}

int32_t mm_freq_submit_device(mm_freq_t* h, const mm_batch_t* b, void* hip_stream) {
    if (!h || !b || b->n_reads < 0 || b->n_reads >= (1 << 24) || b->n_mm_bytes >= 0xFFFFF000ull) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->opts.view == 2 && b->n_reads >= (1 << 21)) return -MM_E_ARG;
    { int rs = side_room(h); if (rs) return rs; }
    h->n_submits++; h->n_reads_submitted += (uint64_t)b->n_reads;
    const bool gather = h->opts.coalesce > 1 && h->use_tiles && !b->order && b->n_reads > 0;   // (view too: a ticket's rows are then those of the group, `read` counted from its first read)
    if (gather && h->pending_slot >= 0 && !h->pending_host && !h->codes_dirty) {   // (a code interned since the group began: the table is uploaded before a launch's FIRST window, so the group ends here)
        // does this submit continue the gathered group?  (windows of one resident read set, one after the other)
        const mm_batch_t& g = h->pending_batch;
        hipStream_t st = hip_stream ? (hipStream_t)hip_stream : slot_stream(h, h->slots[h->pending_slot]);
        if (b->cigar == g.cigar && b->seq == g.seq && b->mm == g.mm && b->ml == g.ml && b->reads == g.reads + g.n_reads &&
            b->n_cigar_words == g.n_cigar_words && b->n_seq_bytes == g.n_seq_bytes && b->n_mm_bytes == g.n_mm_bytes &&
            b->n_ml_bytes == g.n_ml_bytes && st == h->pending_stream && (int64_t)g.n_reads + b->n_reads < (1 << 24)) {
            h->pending_batch.n_reads += b->n_reads;
            h->pending_batch.max_n_cigar = std::max(g.max_n_cigar, b->max_n_cigar);
            h->pending_batch.max_l_qseq = std::max(g.max_l_qseq, b->max_l_qseq);
            const int si = h->pending_slot;
            if (++h->pending_members >= h->opts.coalesce) { int r = flush_pending(h); if (r) return r; }
            return si;
        }
    }
    { int r = flush_pending(h); if (r) return r; }
    int si = acquire_slot(h);
    if (h->sticky_err) return -h->sticky_err;
    Slot& s = h->slots[si];
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : slot_stream(h, s);
    // all slots of the handle share the code table; make sure it is current (sync only when it changed)
    int r = upload_codes(h, st);
    if (r) return r;
    if (gather) {   // the first window of a group: launched when the group is full or somebody needs it
        h->pending_slot = si; h->pending_members = 1; h->pending_batch = *b; h->pending_stream = st;
        h->pending_host = false; h->pending_bases = 0;
        s.busy = true; s.timed = false; s.members = 1;
        return si;
    }
    r = launch_k1(h, s, b, st);
    s.members = 1;
    return r ? r : si;
}

int32_t mm_freq_submit_device_now(mm_freq_t* h, const mm_batch_t* b, void* hip_stream, uint64_t bases) {
    if (!h || !b || b->n_reads < 0 || b->n_reads >= (1 << 24) || b->n_mm_bytes >= 0xFFFFF000ull) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->opts.view == 2 && b->n_reads >= (1 << 21)) return -MM_E_ARG;
    { int rs = side_room(h); if (rs) return rs; }
    h->n_submits++; h->n_reads_submitted += (uint64_t)b->n_reads;
    { int r = flush_pending(h); if (r) return r; }
    int si = acquire_slot(h);
    if (h->sticky_err) return -h->sticky_err;
    Slot& s = h->slots[si];
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : slot_stream(h, s);
    int r = upload_codes(h, st);
    if (r) return r;
    r = launch_k1(h, s, b, st, bases);
    s.members = 1;
    return r ? r : si;
}

int32_t mm_freq_ticket_batches(mm_freq_t* h, int32_t ticket) {
    if (!h || ticket < 0 || ticket >= kSlots) return -MM_E_ARG;
    return ticket == h->pending_slot ? h->pending_members : h->slots[ticket].members;
}

// bases of a host batch, near enough for the size of its launch: two a byte of the sequence pool, less the pool's slack and
// the reads' alignment padding
static uint64_t batch_bases(const mm_batch_t* b) {
    const uint64_t pad = 64 + 8ull * (uint64_t)b->n_reads;
    return b->n_seq_bytes > pad ? 2 * (b->n_seq_bytes - pad) : 0;
}

static int copy_host_batch(mm_freq* h, Slot& s, const mm_batch_t* hb, hipStream_t st, size_t at_reads, size_t at_cigar, size_t at_seq, size_t at_mm, size_t at_ml) {
    if (hb->n_reads) HIPCHK(hipMemcpyAsync((char*)s.d_reads + sizeof(mm_read_t) * at_reads, hb->reads, sizeof(mm_read_t) * (size_t)hb->n_reads, hipMemcpyHostToDevice, st));
    if (hb->n_cigar_words) HIPCHK(hipMemcpyAsync((char*)s.d_cigar + 4 * at_cigar, hb->cigar, 4 * hb->n_cigar_words, hipMemcpyHostToDevice, st));
    if (hb->n_seq_bytes) HIPCHK(hipMemcpyAsync((char*)s.d_seq + at_seq, hb->seq, hb->n_seq_bytes, hipMemcpyHostToDevice, st));
    if (hb->n_mm_bytes) HIPCHK(hipMemcpyAsync((char*)s.d_mm + at_mm, hb->mm, hb->n_mm_bytes, hipMemcpyHostToDevice, st));
    if (hb->n_ml_bytes) HIPCHK(hipMemcpyAsync((char*)s.d_ml + at_ml, hb->ml, hb->n_ml_bytes, hipMemcpyHostToDevice, st));
    return 0;
}

// mm_freq_submit with opts.coalesce > 1: the batch goes behind the ones already staged in the group's slot; the group is
// launched when it is full (batches, staging room) or when somebody needs it
static int32_t submit_host_gathered(mm_freq* h, const mm_batch_t* hb) {
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    if (h->pending_slot >= 0) {
        Slot& s = h->slots[h->pending_slot];
        const bool fits = h->pending_host && !h->codes_dirty && h->pending_members < h->opts.coalesce &&
                          sizeof(mm_read_t) * (s.fill_reads + (size_t)hb->n_reads) <= s.cap_reads && 4 * (s.fill_cigar + hb->n_cigar_words) <= s.cap_cigar &&
                          s.fill_seq + hb->n_seq_bytes <= s.cap_seq && s.fill_mm + hb->n_mm_bytes <= s.cap_mm && s.fill_ml + hb->n_ml_bytes <= s.cap_ml &&
                          s.fill_mm + hb->n_mm_bytes < 0xFFFFF000ull && s.fill_reads + (size_t)hb->n_reads < ((size_t)1 << (h->opts.view == 2 ? 21 : 24));   // (view == 2: the rows number a launch's reads with 21 bits)
        if (!fits) { int r = flush_pending(h); if (r) return r; }
    }
    if (h->pending_slot < 0) {
        const int si = acquire_slot(h);
        if (h->sticky_err) return -h->sticky_err;
        Slot& s = h->slots[si];
        int r = upload_codes(h, slot_stream(h, s));
        if (r) return r;
        // staging for as many batches like this one as may be gathered, within the budget
        const size_t budget = (size_t)(h->opts.gather_mb > 0 ? h->opts.gather_mb : 1024) << 20;
        const size_t one = sizeof(mm_read_t) * (size_t)hb->n_reads + 4 * hb->n_cigar_words + hb->n_seq_bytes + hb->n_mm_bytes + hb->n_ml_bytes + 80;
        const size_t f = std::max<size_t>(1, std::min<size_t>((size_t)h->opts.coalesce, budget / one));
        if ((r = grow(h, &s.d_reads, &s.cap_reads, f * sizeof(mm_read_t) * (size_t)hb->n_reads)) || (r = grow(h, &s.d_cigar, &s.cap_cigar, f * (4 * hb->n_cigar_words + 16))) ||
            (r = grow(h, &s.d_seq, &s.cap_seq, f * (hb->n_seq_bytes + 16))) || (r = grow(h, &s.d_mm, &s.cap_mm, f * (hb->n_mm_bytes + 16))) ||
            (r = grow(h, &s.d_ml, &s.cap_ml, f * (hb->n_ml_bytes + 16))))
            return r;
        s.fill_reads = s.fill_cigar = s.fill_seq = s.fill_mm = s.fill_ml = 0;
        h->pending_slot = si; h->pending_members = 0; h->pending_stream = slot_stream(h, s); h->pending_host = true; h->pending_bases = 0;
        std::memset(&h->pending_batch, 0, sizeof h->pending_batch);
        s.busy = true; s.timed = false; s.members = 1;
    }
    const int si = h->pending_slot;
    Slot& s = h->slots[si];
    hipStream_t st = slot_stream(h, s);
    { int r = copy_host_batch(h, s, hb, st, s.fill_reads, s.fill_cigar, s.fill_seq, s.fill_mm, s.fill_ml); if (r) return r; }
    if (s.fill_reads) {   // (the group's first batch lies at offset 0)
        hipLaunchKernelGGL(k_rebase_reads, dim3((unsigned)((hb->n_reads + 255) / 256)), dim3(256), 0, st, (mm_read_t*)s.d_reads + s.fill_reads, hb->n_reads,
                           (uint64_t)s.fill_cigar, (uint64_t)s.fill_seq, (uint64_t)s.fill_mm, (uint64_t)s.fill_ml);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(s.ev_copied, st));
    s.copy_pending = true;
    mm_batch_t& g = h->pending_batch;
    // the group as one batch: its pools end where this batch's end (every batch brings its own zero slack); the next one
    // starts on a 16-byte boundary behind it
    g.reads = (const mm_read_t*)s.d_reads; g.cigar = (const uint32_t*)s.d_cigar; g.seq = (const uint8_t*)s.d_seq;
    g.mm = (const uint8_t*)s.d_mm; g.ml = (const uint8_t*)s.d_ml; g.order = nullptr; g.n_order = 0;
    g.n_reads = (int32_t)(s.fill_reads + (size_t)hb->n_reads);
    g.n_cigar_words = s.fill_cigar + hb->n_cigar_words; g.n_seq_bytes = s.fill_seq + hb->n_seq_bytes;
    g.n_mm_bytes = s.fill_mm + hb->n_mm_bytes; g.n_ml_bytes = s.fill_ml + hb->n_ml_bytes;
    g.max_n_cigar = std::max(g.max_n_cigar, hb->max_n_cigar); g.max_l_qseq = std::max(g.max_l_qseq, hb->max_l_qseq);
    s.fill_reads += (size_t)hb->n_reads;
    s.fill_cigar = (s.fill_cigar + hb->n_cigar_words + 3) & ~(size_t)3; s.fill_seq = up16(s.fill_seq + hb->n_seq_bytes);
    s.fill_mm = up16(s.fill_mm + hb->n_mm_bytes); s.fill_ml = up16(s.fill_ml + hb->n_ml_bytes);
    h->pending_bases += batch_bases(hb);
    if (++h->pending_members >= h->opts.coalesce) { int r = flush_pending(h); if (r) return r; }
    return si;
}

int32_t mm_freq_submit(mm_freq_t* h, const mm_batch_t* hb) {
    if (!h || !hb || hb->n_reads < 0 || hb->n_reads >= (1 << 24) || hb->n_mm_bytes >= 0xFFFFF000ull) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->opts.view == 2 && hb->n_reads >= (1 << 21)) return -MM_E_ARG;   // (the group ordinal shares mm_view_row_t.read with the read's index)
    { int rs = side_room(h); if (rs) return rs; }
    h->n_submits++; h->n_reads_submitted += (uint64_t)hb->n_reads;
    if (h->opts.coalesce > 1 && h->use_tiles && !hb->order && hb->n_reads > 0) return submit_host_gathered(h, hb);
    { int rf = flush_pending(h); if (rf) return rf; }
    int si = acquire_slot(h);
    if (h->sticky_err) return -h->sticky_err;
    Slot& s = h->slots[si];
    hipStream_t st = slot_stream(h, s);
    int r = upload_codes(h, st);
    if (r) return r;
    size_t nr = sizeof(mm_read_t) * (size_t)hb->n_reads;
    if ((r = grow(h, &s.d_reads, &s.cap_reads, nr)) || (r = grow(h, &s.d_cigar, &s.cap_cigar, 4 * hb->n_cigar_words)) ||
        (r = grow(h, &s.d_seq, &s.cap_seq, hb->n_seq_bytes)) || (r = grow(h, &s.d_mm, &s.cap_mm, hb->n_mm_bytes)) ||
        (r = grow(h, &s.d_ml, &s.cap_ml, hb->n_ml_bytes)))
        return r;
    mm_batch_t db = *hb;
    if ((r = copy_host_batch(h, s, hb, st, 0, 0, 0, 0, 0))) return r;
    db.reads = (const mm_read_t*)s.d_reads; db.cigar = (const uint32_t*)s.d_cigar; db.seq = (const uint8_t*)s.d_seq;
    db.mm = (const uint8_t*)s.d_mm; db.ml = (const uint8_t*)s.d_ml; db.order = nullptr;
    if (hb->n_reads && hb->order) {
        // the caller's work items go along with the batch (without a plan the items are made on the device, k_plan_items)
        if ((r = grow(h, &s.d_order, &s.cap_order, sizeof(int32_t) * (size_t)hb->n_order))) return r;
        HIPCHK(hipMemcpyAsync(s.d_order, hb->order, sizeof(int32_t) * (size_t)hb->n_order, hipMemcpyHostToDevice, st));
        db.order = (const int32_t*)s.d_order;
        db.n_order = hb->n_order;
    }
    HIPCHK(hipEventRecord(s.ev_copied, st));
    s.copy_pending = true;
    r = launch_k1(h, s, &db, st, batch_bases(hb));
    return r ? r : si;
}

int32_t mm_freq_host_done(mm_freq_t* h, int32_t ticket) {
    if (!h || ticket < 0 || ticket >= kSlots) return MM_E_ARG;
    Slot& s = h->slots[ticket];
    if (!s.copy_pending) return MM_OK;
    if (hipSetDevice(h->device) != hipSuccess || hipEventSynchronize(s.ev_copied) != hipSuccess) return MM_E_HIP;
    s.copy_pending = false;
    return MM_OK;
}

int32_t mm_freq_read_record(mm_freq_t* h, int32_t ticket, int32_t index, mm_read_t* out) {
    if (!h || !out || ticket < 0 || ticket >= kSlots) return MM_E_ARG;
    const mm_batch_t& b = ticket == h->pending_slot ? h->pending_batch : h->slots[ticket].last_batch;
    if (index < 0 || index >= b.n_reads || !b.reads) return MM_E_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return MM_E_HIP;
    if (hipMemcpy(out, b.reads + index, sizeof(mm_read_t), hipMemcpyDeviceToHost) != hipSuccess) return MM_E_HIP;
    return MM_OK;
}

int32_t mm_freq_ticket_batch(mm_freq_t* h, int32_t ticket, mm_batch_t* out) {
    if (!h || !out || ticket < 0 || ticket >= kSlots) return MM_E_ARG;
    if (ticket == h->pending_slot) { if (hipSetDevice(h->device) != hipSuccess) return MM_E_HIP; (void)flush_pending(h); }
    *out = h->slots[ticket].last_batch;
    return out->reads ? MM_OK : MM_E_ARG;
}

int32_t mm_freq_wait(mm_freq_t* h, int32_t ticket, int32_t* bad_read) {
    if (!h || ticket < 0 || ticket >= kSlots) return MM_E_ARG;
    Slot& s = h->slots[ticket];
    if (hipSetDevice(h->device) != hipSuccess) return MM_E_HIP;
    if (ticket == h->pending_slot) (void)flush_pending(h);   // a failed launch is the sticky error reported next
    // (an error of ANOTHER ticket's batch -- found when its slot was recycled or the device was drained -- is reported here too, but the
    // read it names is no read of this ticket's batch: the caller gets the code and -1)
    if (h->sticky_err) { if (bad_read) *bad_read = h->sticky_ticket == ticket ? h->sticky_read : -1; return h->sticky_err; }
    if (!s.busy) return MM_OK;
    if (hipEventSynchronize(s.ev_wait) != hipSuccess) return MM_E_HIP;
    s.busy = false;
    if (finish_deferred(h, s) < 0) return MM_E_HIP;
    return slot_status(h, s, bad_read);
}

float mm_freq_last_kernel_ms(mm_freq_t* h, int32_t ticket) {
    if (!h || ticket < 0 || ticket >= kSlots || !h->slots[ticket].timed) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->slots[ticket].ev_start, h->slots[ticket].ev_stop) != hipSuccess) return -1.f;
    return ms;
}

int32_t mm_freq_stats_enable(mm_freq_t* h, int32_t enable) {
    if (!h) return -MM_E_ARG;
    h->stats_on = enable != 0;
    return 0;
}
int32_t mm_freq_stats_get(mm_freq_t* h, uint64_t out[16]) {
    if (!h || !out) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int r = drain(h); if (r) return r; }
    std::vector<unsigned long long> all(16 + 4 * (size_t)kStatSlots);
    HIPCHK(hipMemcpy(all.data(), h->d_stats, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_stats, 0, all.size() * sizeof(unsigned long long)));
    HIPCHK(hipDeviceSynchronize());   // (launches go to non-blocking streams: not ordered behind a plain memset)
    for (int i = 0; i < 16; i++) out[i] = all[i];
    for (int i = 0; i < 4; i++) out[i] = 0;
    for (size_t w = 0; w < kStatSlots; w++) for (int i = 0; i < 4; i++) out[i] += all[16 + 4 * w + i];
    return 0;
}

int64_t mm_freq_device_bytes(const mm_freq_t* h) { return h ? h->device_bytes : 0; }

int32_t mm_freq_launch_counts(const mm_freq_t* h, uint64_t out[4]) {
    if (!h || !out) return -MM_E_ARG;
    out[0] = h->n_launches; out[1] = h->n_stream_launches; out[2] = h->n_submits; out[3] = h->n_reads_submitted;
    return 0;
}

void mm_freq_reset_counters(mm_freq_t* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)drain(h);
    (void)hipMemset(h->d_counters, 0, sizeof(unsigned long long) * (size_t)std::max<int64_t>(h->n_counter_words, 1));
    (void)hipMemset(h->d_side_count, 0, sizeof(unsigned long long));
    (void)side_table_clear(h);
    (void)hipDeviceSynchronize();
    h->sticky_err = 0; h->sticky_read = -1; h->sticky_ticket = -1;
}

}  // extern "C"
namespace mmhip {   // (mmhip is the kind's own namespace, freq_kinds.h)
__global__ __launch_bounds__(256) void k_rows_overflow(const mm_row_t* __restrict__ rows, unsigned long long n, unsigned int* __restrict__ flag) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i < n && rows[i].n_mod > rows[i].n_called) *flag = 1u;
}
}  // namespace mmhip
extern "C" {
static int64_t finalize_impl(mm_freq_t* h, const mm_row_t** out_rows, const mm_row_t** out_dev);
int64_t mm_freq_finalize(mm_freq_t* h, const mm_row_t** out_rows) { return finalize_impl(h, out_rows, nullptr); }
int64_t mm_freq_finalize_device(mm_freq_t* h, const mm_row_t** out_rows, const mm_row_t** out_dev) {
    if (!out_rows || !out_dev) return -MM_E_ARG;
    *out_rows = nullptr; *out_dev = nullptr;
    return finalize_impl(h, out_rows, out_dev);
}
// out_dev non-null: the dense rows stay on the device unless side rows have to be merged into them on the host
static int64_t finalize_impl(mm_freq_t* h, const mm_row_t** out_rows, const mm_row_t** out_dev) {
    if (!h || h->opts.view) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    { int r = drain(h); if (r) return r; }
    if (h->sticky_err) return -h->sticky_err;
    {   // a slab that carried counts for positions that are no sites here: the two handles were not built from one reference
        unsigned int sf = 0;
        HIPCHK(hipMemcpy(&sf, h->d_slab_flag, 4, hipMemcpyDeviceToHost));
        if (sf) return -MM_E_ARG;
    }
    std::vector<mm_row_t>& rows = h->rows;
    rows.clear();
    // n_called is the low half of a packed 64-bit counter: a carry out of it lands in n_mod, which can then exceed it
    // (never otherwise: every modified call is a call).  The reference exits on such an overflow (src/mod.c:900-904).
    auto overflowed = [&]() { for (const mm_row_t& r : rows) if (r.n_mod > r.n_called) return true; return false; };
    // ---- the dense counters: the device walks the positions and writes finished rows in output order (k_site_count /
    // k_site_emit: contig by name, position, strand, code plane, haplotype plane); the host copies them
    size_t n_dense = 0, dense_total = 0;
    bool dense_on_host = true;
    auto dense_to_host = [&]() -> int {   // the dense rows were left on the device and are wanted here after all
        if (dense_on_host) return 0;
        dense_on_host = true;
        rows.resize(dense_total);
        if (hipMemcpyAsync(rows.data(), h->d_rows, sizeof(mm_row_t) * dense_total, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -MM_E_HIP;
        return 0;
    };
    if (h->n_counter_words > 0) {
        std::vector<int> order;
        for (int t = 0; t < h->n_contigs; t++) if (h->seg_len[t] > 0) order.push_back(t);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return h->ctg_rank[a] < h->ctg_rank[b]; });
        std::vector<SiteSeg> segs(order.size());
        int64_t tiles = 0;
        for (size_t k = 0; k < order.size(); k++) {
            int t = order[k];
            segs[k].seg_begin = h->seg_begin[t]; segs[k].seg_len = h->seg_len[t];
            segs[k].tile_start = tiles; segs[k].tid = t; segs[k].pad = 0;
            tiles += (h->seg_len[t] + kTile - 1) / kTile;
        }
        if (tiles >= (int64_t)0x7FFFFFFF) return -MM_E_ARG;
        if (tiles > 0) {
            if ((size_t)tiles > h->cap_tiles) {
                if (h->d_tile_counts) (void)mmdev::dfree(h->d_tile_counts);
                if (h->d_tile_offsets) (void)mmdev::dfree(h->d_tile_offsets);
                h->d_tile_counts = nullptr; h->d_tile_offsets = nullptr; h->cap_tiles = 0;
                if (dev_alloc(h, (void**)&h->d_tile_counts, sizeof(uint32_t) * (size_t)tiles) ||
                    dev_alloc(h, (void**)&h->d_tile_offsets, sizeof(unsigned long long) * (size_t)tiles)) return -MM_E_NOMEM;
                h->cap_tiles = (size_t)tiles;
            }
            const K2Params kp = k2_params(h);
            SiteSeg* d_segs = nullptr;
            if (dev_alloc(h, (void**)&d_segs, sizeof(SiteSeg) * segs.size())) return -MM_E_NOMEM;
            int64_t result = 0;
            do {
                if (hipMemcpyAsync(d_segs, segs.data(), sizeof(SiteSeg) * segs.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
                hipLaunchKernelGGL(k_site_count, dim3((unsigned)tiles), dim3(256), 0, h->stream, kp, d_segs, (int)segs.size(), h->d_tile_counts);
                if (hipGetLastError() != hipSuccess) { result = -MM_E_HIP; break; }
                std::vector<uint32_t> tc((size_t)tiles);
                if (hipMemcpyAsync(tc.data(), h->d_tile_counts, sizeof(uint32_t) * (size_t)tiles, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                    hipStreamSynchronize(h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
                std::vector<unsigned long long> to((size_t)tiles);
                unsigned long long total = 0;
                for (size_t i = 0; i < (size_t)tiles; i++) { to[i] = total; total += tc[i]; }
                if (total > 0) {
                    if ((size_t)total > h->cap_rows) {
                        if (h->d_rows) (void)mmdev::dfree(h->d_rows);
                        h->d_rows = nullptr; h->cap_rows = 0;
                        size_t cap = (size_t)total + (size_t)total / 8 + 1024;
                        if (dev_alloc(h, (void**)&h->d_rows, sizeof(mm_row_t) * cap)) { result = -MM_E_NOMEM; break; }
                        h->cap_rows = cap;
                    }
                    if (hipMemcpyAsync(h->d_tile_offsets, to.data(), sizeof(unsigned long long) * (size_t)tiles, hipMemcpyHostToDevice, h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
                    hipLaunchKernelGGL(k_site_emit, dim3((unsigned)tiles), dim3(256), 0, h->stream, kp, d_segs, (int)segs.size(), h->d_tile_offsets, h->d_rows);
                    if (hipGetLastError() != hipSuccess) { result = -MM_E_HIP; break; }
                    dense_total = (size_t)total;
                    if (out_dev) { dense_on_host = false; if (hipStreamSynchronize(h->stream) != hipSuccess) { result = -MM_E_HIP; break; } }   // (brought over below if side rows turn up)
                    else {
                        rows.resize((size_t)total);
                        if (hipMemcpyAsync(rows.data(), h->d_rows, sizeof(mm_row_t) * (size_t)total, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                            hipStreamSynchronize(h->stream) != hipSuccess) { result = -MM_E_HIP; break; }
                    }
                }
            } while (0);
            (void)mmdev::dfree(d_segs); h->device_bytes -= (int64_t)std::max<size_t>(sizeof(SiteSeg) * segs.size(), 16);
            if (result < 0) return result;
        }
    }
    // every row a dense one (no haplotype planes, nothing on the side table or list): done
    if (!h->opts.haplotypes && !h->opts.finalize_by_runs) {
        unsigned long long ns0 = 0;
        HIPCHK(hipMemcpy(&ns0, h->d_side_count, sizeof(ns0), hipMemcpyDeviceToHost));
        { int rc = side_compact(h); if (rc) return rc; }
        if (ns0 == 0 && h->n_base == 0) {
            if (!dense_on_host) {   // the caller takes them from the device: the overflow check (below, on the host) as a kernel
                unsigned int* d_flag = nullptr; unsigned int hflag = 0;
                if (mmdev::dmalloc((void**)&d_flag, 4) != hipSuccess) return -MM_E_NOMEM;
                bool ok = hipMemsetAsync(d_flag, 0, 4, h->stream) == hipSuccess;
                if (ok) { hipLaunchKernelGGL(k_rows_overflow, dim3((unsigned)((dense_total + 255) / 256)), dim3(256), 0, h->stream, (const mm_row_t*)h->d_rows, (unsigned long long)dense_total, d_flag);
                          ok = hipMemcpyAsync(&hflag, d_flag, 4, hipMemcpyDeviceToHost, h->stream) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess; }
                (void)mmdev::dfree(d_flag);
                if (!ok) return -MM_E_HIP;
                if (hflag) return -MM_E_OVERFLOW;
                *out_dev = (const mm_row_t*)h->d_rows;
                return (int64_t)dense_total;
            }
            if (overflowed()) return -MM_E_OVERFLOW;
            if (out_rows) *out_rows = rows.data();
            return (int64_t)rows.size();
        }
    }
    { int rc = dense_to_host(); if (rc) return rc; }
    n_dense = rows.size();
    // ---- side table (K3): unique keys with their counts, compacted and ordered on the device
    size_t n_side_sorted = 0;   // rows.size() up to which the side rows are known to be in output order
    {
        { int rc = side_compact(h); if (rc) return rc; }
        const unsigned long long nu = h->n_base;
        if (nu) {
            std::vector<unsigned long long> sk((size_t)nu), sv((size_t)nu);
            HIPCHK(hipMemcpyAsync(sk.data(), h->d_base_k, 8 * (size_t)nu, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(sv.data(), h->d_base_v, 8 * (size_t)nu, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            // keys are ordered by (position in the reference-word space, strand, code, ins_offset, haplotype): inside a contig
            // that is the output order; the contigs' runs are put in name order
            std::vector<int> seq_tids;
            for (int t = 0; t < h->n_contigs; t++) if (h->ref_base[t] >= 0) seq_tids.push_back(t);   // ref_base rises with tid
            struct Run { int rank; size_t lo, hi; };
            std::vector<Run> runs;
            std::vector<mm_row_t> srows((size_t)nu);
            size_t cursor = 0;
            for (size_t i = 0; i < (size_t)nu; i++) {
                const unsigned long long k = sk[i];
                const int64_t rpos = (int64_t)(k >> 28);
                while (cursor + 1 < seq_tids.size() && h->ref_base[seq_tids[cursor + 1]] <= rpos) cursor++;
                const int t = seq_tids.empty() ? 0 : seq_tids[cursor];
                mm_row_t r;
                std::memset(&r, 0, sizeof(r));
                r.tid = t; r.pos = (int32_t)(rpos - h->ref_base[t]); r.strand = (uint8_t)((k >> 27) & 1u); r.code = (int16_t)((k >> 21) & 63u);
                r.ins_offset = (uint16_t)((k >> 5) & 0xFFFFu);
                const int h5 = (int)(k & 31u);
                r.hp = (int16_t)((h->opts.haplotypes && h5 != 31) ? h5 : -1);
                r.n_called = (uint32_t)sv[i]; r.n_mod = (uint32_t)(sv[i] >> 32);
                srows[i] = r;
                if (runs.empty() || runs.back().rank != h->ctg_rank[t]) runs.push_back({h->ctg_rank[t], i, i + 1});
                else runs.back().hi = i + 1;
            }
            std::stable_sort(runs.begin(), runs.end(), [](const Run& a, const Run& b) { return a.rank < b.rank; });
            rows.reserve(rows.size() + (size_t)nu);
            for (const Run& rn : runs) rows.insert(rows.end(), srows.begin() + (ptrdiff_t)rn.lo, srows.begin() + (ptrdiff_t)rn.hi);
            n_side_sorted = rows.size();
        }
    }
    // ---- side list (what has no side key: rare)
    unsigned long long ns = 0;
    HIPCHK(hipMemcpy(&ns, h->d_side_count, sizeof(ns), hipMemcpyDeviceToHost));
    if (ns > (unsigned long long)h->side_cap) return -MM_E_SIDEFULL;
    if (ns) {
        std::vector<SideRec> sr((size_t)ns);
        HIPCHK(hipMemcpy(sr.data(), h->d_side, sizeof(SideRec) * (size_t)ns, hipMemcpyDeviceToHost));
        rows.reserve(rows.size() + (size_t)ns);
        for (const SideRec& s : sr) {
            mm_row_t r;
            std::memset(&r, 0, sizeof(r));
            r.tid = s.tid; r.pos = s.pos; r.strand = s.strand; r.ins_offset = s.ins_off; r.code = s.code;
            r.hp = (int16_t)(h->opts.haplotypes ? s.hp : -1);
            r.n_called = 1; r.n_mod = s.is_mod;
            rows.push_back(r);
        }
    }
    // ---- order + merge equal keys (+ '*' aggregates with haplotypes, update_freq_map mod.c:906-928)
    auto hpkey = [](int hp) { return hp < 0 ? 100000 : hp; };
    auto less = [&](const mm_row_t& a, const mm_row_t& b) {
        if (a.tid != b.tid) {
            int ra = h->ctg_rank[a.tid], rb = h->ctg_rank[b.tid];
            return ra < rb;
        }
        if (a.pos != b.pos) return a.pos < b.pos;
        if (a.strand != b.strand) return a.strand < b.strand;
        if (a.code != b.code) return a.code < b.code;
        if (a.ins_offset != b.ins_offset) return a.ins_offset < b.ins_offset;
        return hpkey(a.hp) < hpkey(b.hp);
    };
    // the dense rows arrive in output order, the table's rows likewise; only the list's (rare) need sorting: three ordered runs merged
    {
        if (n_side_sorted < rows.size()) {
            std::sort(rows.begin() + (ptrdiff_t)std::max(n_side_sorted, n_dense), rows.end(), less);
            if (n_side_sorted > n_dense)
                std::inplace_merge(rows.begin() + (ptrdiff_t)n_dense, rows.begin() + (ptrdiff_t)n_side_sorted, rows.end(), less);
        }
        if (n_dense > 0 && n_dense < rows.size()) std::inplace_merge(rows.begin(), rows.begin() + (ptrdiff_t)n_dense, rows.end(), less);
    }
    auto same_site = [](const mm_row_t& a, const mm_row_t& b) {
        return a.tid == b.tid && a.pos == b.pos && a.strand == b.strand && a.code == b.code && a.ins_offset == b.ins_offset;
    };
    std::vector<mm_row_t> merged;
    merged.reserve(rows.size() + (h->opts.haplotypes ? rows.size() : 0));
    size_t i = 0;
    while (i < rows.size()) {
        size_t j = i;
        mm_row_t agg = rows[i];
        agg.hp = -1; agg.n_called = 0; agg.n_mod = 0;
        while (j < rows.size() && same_site(rows[i], rows[j])) {
            mm_row_t cur = rows[j];
            size_t k = j + 1;
            while (k < rows.size() && same_site(rows[i], rows[k]) && rows[k].hp == cur.hp) {
                cur.n_called += rows[k].n_called; cur.n_mod += rows[k].n_mod; k++;
            }
            agg.n_called += cur.n_called; agg.n_mod += cur.n_mod;
            merged.push_back(cur);
            j = k;
        }
        if (h->opts.haplotypes) merged.push_back(agg);
        i = j;
    }
    rows.swap(merged);
    if (overflowed()) return -MM_E_OVERFLOW;
    if (out_rows) *out_rows = rows.data();
    return (int64_t)rows.size();
}

// ---------------------------------------------------------------------------------- view rows
// print_view_output (src/mod.c:560-626) up to the fprintf: wait, order the records on the device, hand the rows over.
static int64_t view_finish(mm_freq_t* h, int32_t ticket, int32_t* bad_read, bool to_host) {
    if (!h || !h->opts.view || ticket < 0 || ticket >= kSlots) return -MM_E_ARG;
    Slot& s = h->slots[ticket];
    HIPCHK(hipSetDevice(h->device));
    if (ticket == h->pending_slot) { int rf = flush_pending(h); if (rf) return rf; }   // a gathered group is launched by whoever needs it
    if (!s.timed) return -MM_E_ARG;
    hipStream_t st = s.last_stream;
    auto copy_out = [&](size_t nsel) -> int {
        if (!to_host || s.view_on_host || nsel == 0) return 0;
        if (nsel > s.cap_hrows) {
            if (s.h_vrows) (void)mmdev::hfree(s.h_vrows);
            s.h_vrows = nullptr; s.cap_hrows = 0;
            size_t cap = nsel + nsel / 4 + 4096;
            if (mmdev::hmalloc((void**)&s.h_vrows, sizeof(ViewRow) * cap, hipHostMallocDefault) != hipSuccess) return -MM_E_NOMEM;
            s.cap_hrows = cap;
        }
        HIPCHK(hipMemcpyAsync(s.h_vrows, s.view_dev_rows, sizeof(ViewRow) * nsel, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        s.view_on_host = true;
        return 0;
    };
    if (s.view_rows >= 0) {   // already ordered (fetch after fetch_device, or twice)
        int r = copy_out((size_t)s.view_rows);
        return r ? r : s.view_rows;
    }
    unsigned long long n = 0;
    for (int attempt = 0;; attempt++) {
        HIPCHK(hipEventSynchronize(s.ev_wait));
        s.busy = false;
        { int rf = finish_deferred(h, s); if (rf < 0) return rf; }
        if (s.h_ctl[129] != 0xFFFFFFFFu) {
            unsigned int sum = 0xFFFFFFFFu;
            HIPCHK(hipMemcpy(&sum, s.d_err_word, sizeof sum, hipMemcpyDeviceToHost));
            if (bad_read) *bad_read = (int32_t)(sum >> 8);
            return -(int64_t)(sum & 0xFFu);
        }
        unsigned int worst = 0;
        n = 0;
        for (uint32_t r = 0; r < kViewRegions; r++) { unsigned int c = s.h_vcount[r * kViewCountStride]; worst = std::max(worst, c); n += c; }
        if (worst <= s.view_cap) break;
        if (attempt >= 2) return -MM_E_NOMEM;   // a deterministic re-run cannot overflow twice
        // a region overflowed: size every region for the fullest one (+25 %) and run the batch again
        size_t want = (size_t)worst + worst / 4 + 1024;
        int r;
        if ((r = grow(h, (void**)&s.d_vkeys, &s.cap_vkeys, 8 * want * kViewRegions)) ||
            (r = grow(h, (void**)&s.d_vvals, &s.cap_vvals, 8 * want * kViewRegions)) ||
            (r = grow(h, (void**)&s.d_vseq, &s.cap_vseq, 4 * want * kViewRegions)))
            return r;
        s.view_cap = (unsigned int)std::min<size_t>(std::min(std::min(s.cap_vkeys, s.cap_vvals) / 8, s.cap_vseq / 4) / kViewRegions, 0xFFFFFFFFu / kViewRegions);
        mm_batch_t again = s.last_batch;
        if ((r = launch_k1(h, s, &again, st))) return r;
    }
    s.view_rows = 0;
    s.view_dev_rows = s.d_vrows;
    if (n == 0) return 0;
    size_t nsel = (size_t)n;
    const unsigned int dropped = s.h_vcount[kViewRegions * kViewCountStride];
    if (dropped) {
        // some read carried two entries of one key (add_view_entry keeps the first): close the gaps
        int r;
        if ((r = grow(h, (void**)&s.d_vout, &s.cap_vout, sizeof(ViewRow) * (size_t)n))) return r;
        const uint32_t nr = (uint32_t)s.n_reads;
        hipLaunchKernelGGL(k_view_offsets, dim3(1), dim3(256), 0, st, s.d_vkept, nr, s.d_vnewoff, s.d_vcursor);
        hipLaunchKernelGGL(k_view_compact, dim3(std::min<uint32_t>(nr, (uint32_t)h->n_cu * 8)), dim3(256), 0, st, s.d_vrows, s.d_voff, s.d_vnewoff, nr,
                           s.d_vout);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        nsel -= dropped;
        s.view_dev_rows = s.d_vout;
    }
    s.view_rows = (int64_t)nsel;
    int r2 = copy_out(nsel);
    if (r2) return r2;
    return (int64_t)nsel;
}

static_assert(sizeof(mm_view_row_t) == sizeof(ViewRow), "mm_view_row_t layout");

int64_t mm_view_fetch(mm_freq_t* h, int32_t ticket, const mm_view_row_t** rows, int32_t* bad_read) {
    int64_t n = view_finish(h, ticket, bad_read, true);
    if (n >= 0 && rows) *rows = reinterpret_cast<const mm_view_row_t*>(h->slots[ticket].h_vrows);
    return n;
}
int64_t mm_view_fetch_device(mm_freq_t* h, int32_t ticket, const void** dev_rows, int32_t* bad_read) {
    int64_t n = view_finish(h, ticket, bad_read, false);
    if (n >= 0 && dev_rows) *dev_rows = h->slots[ticket].view_dev_rows;
    return n;
}

// ---------------------------------------------------------------------------------- halo slabs
int64_t mm_freq_slab_words(const mm_freq_t* h, int64_t len) {
    return h ? (int64_t)h->n_code_planes * h->n_hp * 2 * len : 0;
}
static int slab_op(mm_freq_t* h, int op, int32_t tid, int64_t begin, int64_t len, void* buf, void* st) {
    if (!h || tid < 0 || tid >= h->n_contigs || len < 0 || h->opts.view) return -MM_E_ARG;
    const int64_t rel = begin - h->seg_begin[tid];
    if (rel < 0 || rel + len > h->seg_len[tid]) return -MM_E_ARG;
    if (len == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    { int rs = settle(h); if (rs) return rs; }
    hipStream_t s = st ? (hipStream_t)st : h->stream;
    const int64_t total = (int64_t)h->n_code_planes * h->n_hp * 2 * len;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_slab_op, dim3(blocks), dim3(256), 0, s, k2_params(h), op, (int)tid, begin, len, h->d_counters, (unsigned long long*)buf, h->d_slab_flag);
    HIPCHK(hipGetLastError());
    if (!st) HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int32_t mm_freq_slab_export(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* dst, void* st) { return slab_op(h, 0, tid, begin, len, dst, st); }
int32_t mm_freq_slab_add(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* src, void* st) { return slab_op(h, 1, tid, begin, len, const_cast<void*>(src), st); }
int32_t mm_freq_slab_clear(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* st) { return slab_op(h, 2, tid, begin, len, nullptr, st); }
int32_t mm_freq_slab_export_host(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* dst_host) {
    if (!h || !dst_host || len < 0) return -MM_E_ARG;
    const size_t bytes = 8 * (size_t)mm_freq_slab_words(h, len);
    if (bytes == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    void* d = nullptr;
    if (mmdev::dmalloc(&d, bytes) != hipSuccess) return -MM_E_NOMEM;
    int r = slab_op(h, 0, tid, begin, len, d, nullptr);
    if (!r && hipMemcpy(dst_host, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) r = -MM_E_HIP;
    (void)mmdev::dfree(d);
    return r;
}
int32_t mm_freq_slab_add_host(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* src_host) {
    if (!h || !src_host || len < 0) return -MM_E_ARG;
    const size_t bytes = 8 * (size_t)mm_freq_slab_words(h, len);
    if (bytes == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    void* d = nullptr;
    if (mmdev::dmalloc(&d, bytes) != hipSuccess) return -MM_E_NOMEM;
    int r = hipMemcpy(d, src_host, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -MM_E_HIP;
    if (!r) r = slab_op(h, 1, tid, begin, len, d, nullptr);
    (void)mmdev::dfree(d);
    return r;
}

// The same between two PROCESSES that each own a GPU (the workers of `minimod freq --devices`): the slab is packed into a device
// buffer of its own, whose IPC handle goes to the neighbour (64 bytes through a socket); the neighbour opens it, copies device to
// device -- over xGMI when the two are different GPUs -- and adds.  Nothing of the slab passes through host memory.
static_assert(sizeof(hipIpcMemHandle_t) == MM_IPC_HANDLE_BYTES, "the ABI's handle size is HIP's");
int32_t mm_freq_slab_export_ipc(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* handle_out) {
    if (!h || !handle_out || len < 0) return -MM_E_ARG;
    const size_t bytes = 8 * (size_t)mm_freq_slab_words(h, len);
    if (bytes == 0) return -MM_E_ARG;
    HIPCHK(hipSetDevice(h->device));
    if (h->d_ipc_slab) { (void)mmdev::dfree(h->d_ipc_slab); h->d_ipc_slab = nullptr; }
    if (mmdev::dmalloc(&h->d_ipc_slab, bytes) != hipSuccess) return -MM_E_NOMEM;
    int r = slab_op(h, 0, tid, begin, len, h->d_ipc_slab, nullptr);
    if (r) return r;
    HIPCHK(hipDeviceSynchronize());   // the neighbour reads it from another process: complete, and written back
    hipIpcMemHandle_t hd;
    if (hipIpcGetMemHandle(&hd, h->d_ipc_slab) != hipSuccess) { (void)hipGetLastError(); return -MM_E_HIP; }
    std::memcpy(handle_out, &hd, sizeof hd);
    return 0;
}
int32_t mm_freq_slab_add_ipc(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* handle) {
    if (!h || !handle || len < 0) return -MM_E_ARG;
    const size_t bytes = 8 * (size_t)mm_freq_slab_words(h, len);
    if (bytes == 0) return 0;
    HIPCHK(hipSetDevice(h->device));
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, handle, sizeof hd);
    void* theirs = nullptr;
    if (hipIpcOpenMemHandle(&theirs, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return -MM_E_HIP; }
    void* d = nullptr;
    int r = mmdev::dmalloc(&d, bytes) == hipSuccess ? 0 : -MM_E_NOMEM;
    // the copy on the handle's own (non-blocking) stream and waited for: the kernel that adds the slab runs on that stream, and the mapping
    // is closed -- and the sender told it may free the buffer -- only once the bytes are here (a D2D hipMemcpy on the null stream promises neither)
    if (!r && (hipMemcpyAsync(d, theirs, bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)) r = -MM_E_HIP;
    // (the peer's mapping stays open until the handle goes: closing it here would unmap an address range in the middle of the run, which a later
    // allocation could be handed again -- what csrc/devmem.h is there to keep from happening; a worker of --devices ends a moment later anyway)
    h->ipc_open.push_back(theirs);
    if (!r) r = slab_op(h, 1, tid, begin, len, d, nullptr);
    if (d) (void)mmdev::dfree(d);
    return r;
}

}  // extern "C"
