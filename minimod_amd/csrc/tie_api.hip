// tie_api.hip -- host side of include/minimod_tie.h: buffers, launch sequences and fixpoint loops around csrc/tie_kernels.hip.h.
// The fixpoint loops (placement rounds, growth passes, sort levels) are driven from the host: a round is a kernel over all keys of the
// step, a flag in pinned host memory says whether anything moved.
#include <hip/hip_runtime.h>
#include "devmem.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "minimod_tie.h"
#include "tie_kernels.hip.h"

using namespace mmtie;

namespace {

uint64_t g_stats[8];

struct Bufs {   // device allocations of one call, freed together
    std::vector<void*> p;
    int64_t bytes = 0;
    template <class T> T* get(size_t n) {
        void* q = nullptr;
        if (mmdev::dmalloc(&q, sizeof(T) * (n ? n : 1)) != hipSuccess) return nullptr;
        p.push_back(q); bytes += (int64_t)(sizeof(T) * (n ? n : 1));
        return (T*)q;
    }
    ~Bufs() { for (void* q : p) (void)mmdev::dfree(q); }
};

inline unsigned blocks(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per ? (n + per - 1) / per : 1); }
inline uint32_t upper_of(uint64_t c) { return (uint32_t)(c * 0.77 + 0.5); }

#define LAUNCH(k, grid, block, st, ...) do { hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, st, __VA_ARGS__); g_stats[0]++; } while (0)

// inclusive scan of n 64-bit words in place (tile sums in `tiles`, >= n / 2048 + 2 words)
void scan64(u64* a, uint64_t n, u64* tiles, hipStream_t st) {
    const unsigned nt = blocks(n, kScanTile);
    LAUNCH(k_scan_reduce, nt, 256, st, (const u64*)a, (u64)n, tiles);
    LAUNCH(k_scan_spine, 1, 1024, st, tiles, (uint32_t)nt);
    LAUNCH(k_scan_apply, nt, 256, st, a, (u64)n, (const u64*)tiles);
}

// stable LSD radix sort of (key, value) pairs on the bits that differ anywhere; returns which of the two buffer pairs holds the result (0 / 1), -1 on error
int radix_sort(u64* k0, uint32_t* v0, u64* k1, uint32_t* v1, uint64_t n, uint32_t* hist, u64* d_or, volatile uint64_t* h_word, hipStream_t st) {
    if (n < 2) return 0;
    if (hipMemsetAsync(d_or, 0, 8, st) != hipSuccess) return -1;
    LAUNCH(k_or_diff, blocks(n), 256, st, (const u64*)k0, (u64)n, d_or);
    uint64_t diff = 0;
    if (hipMemcpyAsync((void*)h_word, d_or, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
    diff = *h_word;
    const uint32_t nblk = blocks(n, kSortTile);
    int cb = 0;
    u64* kk[2] = {k0, k1};
    uint32_t* vv[2] = {v0, v1};
    for (int shift = 0; shift < 64; shift += 8) {
        if (((diff >> shift) & 255ull) == 0) continue;
        LAUNCH(k_rx_hist, nblk, 64, st, (const u64*)kk[cb], (u64)n, shift, hist, nblk);
        LAUNCH(k_rx_scan, 1, 1024, st, hist, (u64)256 * nblk);
        LAUNCH(k_rx_scatter, nblk, 64, st, (const u64*)kk[cb], (const uint32_t*)vv[cb], (u64)n, shift, (const uint32_t*)hist, nblk, kk[cb ^ 1], vv[cb ^ 1]);
        cb ^= 1;
    }
    return cb;
}

// T3 + T4 on device arrays: hash / sortkey by first-insertion rank -> d_slot (may be null), d_final: ranks in slot / printing order
int core_and_sort(const uint32_t* d_hash, const long long* d_sortkey, uint64_t n, int put_after_last, uint32_t* d_slot, uint32_t* d_final, hipStream_t st) {
    if (n == 0) return 0;
    if (n >= 0xFFFFFFF0ull) return -MM_E_TOOMANY;
    const bool tl = std::getenv("MM_TIMELINE") != nullptr;   // (where the call's milliseconds go: a synchronisation at every mark, so only then)
    auto now = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; };
    double t_last = now();
    auto mark = [&](const char* what) { if (tl) { (void)hipStreamSynchronize(st); const double t = now(); std::fprintf(stderr, "[timeline] tie order: %-40s %.2f ms\n", what, 1e3 * (t - t_last)); t_last = t; } };
    Bufs B;
    volatile uint32_t* hflag = nullptr;   // [0] changed, [1] moved, [2] fail, [4..5] a 64-bit word, [6] a count
    if (mmdev::hmalloc((void**)&hflag, 64, hipHostMallocDefault) != hipSuccess) return -MM_E_NOMEM;
    struct HF { volatile uint32_t* p; ~HF() { (void)mmdev::hfree((void*)p); } } hf{hflag};
    for (int i = 0; i < 16; i++) hflag[i] = 0;
    uint32_t* f_changed = (uint32_t*)&hflag[0];
    uint32_t* f_moved = (uint32_t*)&hflag[1];
    uint32_t* f_fail = (uint32_t*)&hflag[2];

    // ---- T3: the capacity history is a function of the number of keys (kh_put grows a table that holds 0.77 of its buckets)
    uint64_t Cfin = 4;
    while (upper_of(Cfin) < n) Cfin *= 2;
    if (put_after_last && n >= upper_of(Cfin)) Cfin *= 2;
    const uint64_t half = Cfin / 2 > 4 ? Cfin / 2 : 4;
    uint32_t* tab[2] = {B.get<uint32_t>(Cfin), B.get<uint32_t>(Cfin)};
    uint32_t* cur = B.get<uint32_t>(std::max<uint64_t>(n, half));
    uint32_t* stp = B.get<uint32_t>(std::max<uint64_t>(n, half));
    uint32_t* land = B.get<uint32_t>(half);
    uint32_t* pred = B.get<uint32_t>(half);
    u64* word = B.get<u64>(half);
    u64* tw = B.get<u64>(Cfin);
    if (!tab[0] || !tab[1] || !cur || !stp || !land || !pred || !word || !tw) return -MM_E_NOMEM;
    mark("T3 buffers");
    int tcur = 0;
    uint64_t C = 4, done = 0;
    LAUNCH(k_fill32, blocks(C), 256, st, tab[0], (u64)C, kNone);
    auto grow = [&]() -> int {
        const uint32_t Cc = (uint32_t)C, mask2 = (uint32_t)(2 * C - 1);
        uint32_t* told = tab[tcur];
        uint32_t* tnew = tab[tcur ^ 1];
        g_stats[1]++;
        LAUNCH(k_grow_home, blocks(Cc), 256, st, (const uint32_t*)told, Cc, d_hash, mask2, land);
        for (int pass = 0;; pass++) {
            g_stats[2]++;
            LAUNCH(k_fill32, blocks(Cc), 256, st, pred, (u64)Cc, kNone);
            LAUNCH(k_grow_succ, blocks(Cc), 256, st, (const uint32_t*)told, Cc, (const uint32_t*)land, pred);
            LAUNCH(k_grow_prio, blocks(Cc), 256, st, (const uint32_t*)told, Cc, (const uint32_t*)pred, d_hash, mask2, word, cur, stp, f_fail);
            LAUNCH(k_fill64, blocks(2ull * Cc), 256, st, tw, (u64)(2ull * Cc), kNone64);
            for (int rounds = 0;; rounds++) {
                *f_changed = 0;
                for (int q = 0; q < 4; q++) { LAUNCH(k_grow_round, blocks(Cc), 256, st, (const uint32_t*)told, Cc, (const u64*)word, tw, mask2, cur, stp, f_changed); g_stats[3]++; }
                if (hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
                if (!*f_changed) break;
                if (rounds > 100000) return -MM_E_HIP;
            }
            if (*f_fail) return -MM_E_TOOMANY;
            *f_moved = 0;
            LAUNCH(k_grow_check, blocks(Cc), 256, st, (const uint32_t*)told, Cc, (const uint32_t*)cur, land, f_moved);
            if (hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
            if (!*f_moved) break;
            if (pass > 4096) return -MM_E_HIP;
        }
        LAUNCH(k_fill32, blocks(2ull * Cc), 256, st, tnew, (u64)(2ull * Cc), kNone);
        LAUNCH(k_grow_commit, blocks(Cc), 256, st, (const uint32_t*)told, Cc, (const uint32_t*)land, tnew);
        tcur ^= 1; C *= 2;
        return 0;
    };
    // the small tables (4 ... 8 192 buckets) in ONE workgroup (k_small_epochs: the same steps without a launch a round and a synchronisation a
    // fixpoint test); MM_TIE_SMALL=0: everything through the loop below, as rounds 5's first version did (the tests run both)
    bool finished = false;
    {
        const char* e = std::getenv("MM_TIE_SMALL");
        const uint64_t Cstop = e ? (uint64_t)std::strtoull(e, nullptr, 10) : 8192ull;
        if (Cstop >= 8) {
            SmallState* d_state = B.get<SmallState>(1);
            unsigned long long* d_counts = B.get<unsigned long long>(4);
            if (!d_state || !d_counts) return -MM_E_NOMEM;
            LAUNCH(k_small_epochs, 1, 1024, st, d_hash, (u64)n, put_after_last, (uint32_t)std::min<uint64_t>(Cstop, Cfin), tab[0], tab[1], cur, stp, land, pred, word, tw, d_state, d_counts);
            SmallState hs;
            unsigned long long hcnt[4] = {0, 0, 0, 0};
            if (hipMemcpyAsync(&hs, d_state, sizeof hs, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hcnt, d_counts, 24, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
            if (hs.fail) return -MM_E_TOOMANY;
            C = hs.C; done = hs.done; tcur = (int)hs.tcur; finished = hs.finished != 0;
            g_stats[1] += hcnt[0]; g_stats[2] += hcnt[1]; g_stats[3] += hcnt[2];
        }
    }
    mark("T3 small tables (one workgroup)");
    while (!finished) {
        const uint64_t U = upper_of(C), hi = std::min<uint64_t>(n, U);
        if (hi > done) {
            const uint32_t lo32 = (uint32_t)done, hi32 = (uint32_t)hi, mask = (uint32_t)(C - 1);
            LAUNCH(k_place_init, blocks(hi - done), 256, st, d_hash, lo32, hi32, mask, cur, stp);
            for (int rounds = 0;; rounds++) {
                *f_changed = 0;
                for (int q = 0; q < 4; q++) { LAUNCH(k_place_round, blocks(hi - done), 256, st, tab[tcur], mask, lo32, hi32, cur, stp, f_changed); g_stats[3]++; }
                if (hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
                if (!*f_changed) break;
                if (rounds > 100000) return -MM_E_HIP;
            }
            done = hi;
        }
        if (done == n) {
            if (put_after_last && n >= U) { const int r = grow(); if (r) return r; }
            break;
        }
        { const int r = grow(); if (r) return r; }
    }
    mark("T3 big tables (host loop)");
    // the keys in slot order
    u64* f = B.get<u64>(std::max<uint64_t>(C, n));
    u64* tiles = B.get<u64>(std::max<uint64_t>(C, n) / kScanTile + 4);
    long long* key = B.get<long long>(n);
    uint32_t* id = B.get<uint32_t>(n);
    if (!f || !tiles || !key || !id) return -MM_E_NOMEM;
    LAUNCH(k_slot_flags, blocks(C), 256, st, (const uint32_t*)tab[tcur], (u64)C, f);
    scan64(f, C, tiles, st);
    LAUNCH(k_slot_gather, blocks(C), 256, st, (const uint32_t*)tab[tcur], (u64)C, (const u64*)f, d_sortkey, key, id);
    if (d_slot && hipMemcpyAsync(d_slot, id, 4 * n, hipMemcpyDeviceToDevice, st) != hipSuccess) return -MM_E_HIP;

    mark("slot order");
    // ---- T4: ks_introsort's partitions, level by level
    if (n == 2) {   // (src/ksort.h: two elements are compared and that is all)
        long long hk[2]; uint32_t hi2[2];
        if (hipMemcpyAsync(hk, key, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hi2, id, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
        if (hk[1] < hk[0]) std::swap(hi2[0], hi2[1]);
        if (hipMemcpyAsync(d_final, hi2, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
        return 0;
    }
    if (n >= 3) {
        int d = 2;
        while ((1ull << d) < n) d++;
        d <<= 1;
        const uint64_t max_segs = n / kSmallSeg + 4, max_small = n / 16 + 16;
        Seg* segs[2] = {B.get<Seg>(max_segs), B.get<Seg>(max_segs)};
        Seg* small = B.get<Seg>(max_small);
        uint32_t* cnt = B.get<uint32_t>(4);   // [0] segments of the next level, [1] small segments
        long long* rp = B.get<long long>(max_segs);
        uint32_t* nswap = B.get<uint32_t>(max_segs);
        uint32_t* pivot_at = B.get<uint32_t>(max_segs);
        uint32_t* child = B.get<uint32_t>(2 * max_segs);
        uint32_t* segof = B.get<uint32_t>(n);
        uint32_t* lpos = B.get<uint32_t>(n + 2);
        uint32_t* rpos = B.get<uint32_t>(n + 2);
        if (!segs[0] || !segs[1] || !small || !cnt || !rp || !nswap || !pivot_at || !child || !segof || !lpos || !rpos) return -MM_E_NOMEM;
        Seg top; top.s = 0; top.t = (uint32_t)(n - 1); top.d = d; top.pad = 0;
        uint32_t n_seg = 0, n_small = 0;
        if (hipMemsetAsync(cnt, 0, 16, st) != hipSuccess) return -MM_E_HIP;
        if (n <= kSmallSeg) {
            if (hipMemcpyAsync(small, &top, sizeof top, hipMemcpyHostToDevice, st) != hipSuccess) return -MM_E_HIP;
            n_small = 1;
            const uint32_t one = 1;
            if (hipMemcpyAsync(cnt + 1, &one, 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
        } else {
            if (hipMemcpyAsync(segs[0], &top, sizeof top, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
            n_seg = 1;
            LAUNCH(k_fill32, blocks(n), 256, st, segof, (u64)n, 0u);
        }
        int sb = 0;
        while (n_seg) {
            g_stats[4]++;
            LAUNCH(k_fill32, blocks(n_seg), 256, st, nswap, (u64)n_seg, 0u);
            LAUNCH(k_qs_pivot, blocks(n_seg), 256, st, segs[sb], n_seg, key, id, rp, f_fail);
            LAUNCH(k_qs_flags, blocks(n), 256, st, (const uint32_t*)segof, (const Seg*)segs[sb], (const long long*)key, (const long long*)rp, (u64)n, f);
            scan64(f, n, tiles, st);
            LAUNCH(k_qs_scatter, blocks(n), 256, st, (const uint32_t*)segof, (const Seg*)segs[sb], (const long long*)key, (const long long*)rp, (u64)n, (const u64*)f, lpos, rpos);
            LAUNCH(k_qs_swap, blocks(n), 256, st, (const uint32_t*)segof, (const Seg*)segs[sb], (u64)n, (const u64*)f, (const uint32_t*)lpos, (const uint32_t*)rpos, key, id, nswap);
            if (hipMemsetAsync(cnt, 0, 4, st) != hipSuccess) return -MM_E_HIP;
            LAUNCH(k_qs_finish, blocks(n_seg), 256, st, (const Seg*)segs[sb], n_seg, (const uint32_t*)lpos, (const uint32_t*)rpos, (const uint32_t*)nswap, key, id, segs[sb ^ 1], cnt, small, cnt + 1, pivot_at, child);
            LAUNCH(k_qs_assign, blocks(n), 256, st, segof, (u64)n, (const uint32_t*)pivot_at, (const uint32_t*)child);
            uint32_t hc[2] = {0, 0};
            if (hipMemcpyAsync(hc, cnt, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
            if (*f_fail) return -MM_E_TOOMANY;
            n_seg = hc[0]; n_small = hc[1];
            sb ^= 1;
            if (n_seg > max_segs || n_small > max_small) return -MM_E_HIP;
        }
        if (n_small) {
            uint32_t hc[2] = {0, 0};
            if (hipMemcpyAsync(hc, cnt, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
            n_small = hc[1];
        }
        g_stats[5] = n_small;
        mark("T4 partition levels");
        if (n_small) LAUNCH(k_qs_small, blocks(n_small, 64), 64, st, (const Seg*)small, n_small, key, id);
    }
    mark("T4 small segments (a thread each)");
    // the insertion sort over everything that ends ks_introsort: a stable sort of what the partitions left
    u64* bk[2] = {B.get<u64>(n), B.get<u64>(n)};
    uint32_t* bv1 = B.get<uint32_t>(n);
    const uint32_t nblk = blocks(n, kSortTile);
    uint32_t* hist = B.get<uint32_t>((size_t)256 * nblk + 8);
    u64* d_or = B.get<u64>(2);
    if (!bk[0] || !bk[1] || !bv1 || !hist || !d_or) return -MM_E_NOMEM;
    LAUNCH(k_bias_keys, blocks(n), 256, st, (const long long*)key, (u64)n, bk[0]);
    const int w = radix_sort(bk[0], id, bk[1], bv1, n, hist, d_or, (volatile uint64_t*)&hflag[4], st);
    if (w < 0) return -MM_E_HIP;
    if (hipMemcpyAsync(d_final, w ? bv1 : id, 4 * n, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    mark("final stable sort");
    return 0;
}

}  // namespace

struct mm_tie {
    mm_tie_opts_t o;
    hipStream_t st = nullptr;
    int64_t device_bytes = 0;
    uint32_t failed = 0;
    std::vector<std::string> names;
    std::vector<int64_t> len;
    // tables on the device
    uint8_t* d_klass = nullptr; uint32_t* d_ctg_hash = nullptr; u64* d_ctg_base = nullptr; int32_t* d_ctg_rank = nullptr; uint2* d_mid = nullptr; char* d_codes = nullptr;
    int32_t n_codes = 0;
    // per-launch scratch (grown as needed)
    uint32_t *d_beg = nullptr, *d_end = nullptr; size_t cap_reads = 0;
    u64 *d_ska = nullptr, *d_skb = nullptr, *d_keys = nullptr; uint32_t* d_khash = nullptr; size_t cap_rows = 0;
    u64* d_tab = nullptr; size_t cap_tab = 0;
    uint32_t* d_rt = nullptr; size_t cap_rt = 0;
    uint32_t* d_rc = nullptr; size_t cap_rc = 0;
    // the stamps
    u64 *d_gkey = nullptr, *d_gstamp = nullptr; uint64_t gcap = 0, distinct = 0;
    u64* d_words = nullptr;   // [0] last put, [1] count scratch
    uint32_t* d_fail = nullptr;
    uint64_t serial = 0;
    volatile uint64_t* h_words = nullptr;   // pinned: [0..1] copies of d_words, [2] fail
};

namespace {
template <class T> int grow_buf(mm_tie* t, T** p, size_t* cap, size_t need, size_t elems_per = 1) {
    (void)elems_per;
    if (need <= *cap && *p) return 0;
    if (*p) { (void)mmdev::dfree(*p); t->device_bytes -= (int64_t)(sizeof(T) * *cap); *p = nullptr; *cap = 0; }
    const size_t nc = need + need / 4 + 1024;
    if (mmdev::dmalloc((void**)p, sizeof(T) * nc) != hipSuccess) { *p = nullptr; return -MM_E_NOMEM; }
    *cap = nc; t->device_bytes += (int64_t)(sizeof(T) * nc);
    return 0;
}
uint32_t x31_str(uint32_t h, const char* p, size_t n) { for (size_t i = 0; i < n; i++) h = (h << 5) - h + (uint32_t)(unsigned char)p[i]; return h; }

TieTables tables_of(const mm_tie* t) {
    TieTables T;
    T.klass = t->d_klass; T.ctg_hash = t->d_ctg_hash; T.ctg_base = t->d_ctg_base; T.ctg_rank = t->d_ctg_rank; T.mid = t->d_mid; T.codes = t->d_codes;
    T.n_codes = t->n_codes; T.n_contigs = t->o.n_contigs; T.insertions = t->o.insertions; T.haplotypes = t->o.haplotypes;
    return T;
}

int ensure_stamps(mm_tie* t, uint64_t incoming) {
    const uint64_t need = 2 * (t->distinct + incoming) + 1024;
    if (need <= t->gcap) return 0;
    uint64_t nc = t->gcap ? t->gcap : (1ull << 16);
    while (nc < need) nc *= 2;
    u64 *nk = nullptr, *ns = nullptr;
    if (mmdev::dmalloc((void**)&nk, 8 * nc) != hipSuccess) return -MM_E_NOMEM;
    if (mmdev::dmalloc((void**)&ns, 8 * nc) != hipSuccess) { (void)mmdev::dfree(nk); return -MM_E_NOMEM; }
    LAUNCH(k_fill64, blocks(nc), 256, t->st, nk, (u64)nc, kNone64);
    LAUNCH(k_fill64, blocks(nc), 256, t->st, ns, (u64)nc, kNone64);
    if (t->gcap) LAUNCH(k_stamp_rehash, blocks(t->gcap), 256, t->st, (const u64*)t->d_gkey, (const u64*)t->d_gstamp, (u64)t->gcap, nk, ns, (u64)(nc - 1));
    if (hipStreamSynchronize(t->st) != hipSuccess) { (void)mmdev::dfree(nk); (void)mmdev::dfree(ns); return -MM_E_HIP; }
    if (t->d_gkey) { (void)mmdev::dfree(t->d_gkey); (void)mmdev::dfree(t->d_gstamp); t->device_bytes -= (int64_t)(16 * t->gcap); }
    t->d_gkey = nk; t->d_gstamp = ns; t->gcap = nc; t->device_bytes += (int64_t)(16 * nc);
    return 0;
}

// the rows' stamps, hashes and comparator keys; order[] = row indices by stamp.  Device arrays in B.
struct RowSeq { uint32_t* order; uint32_t* hash_r; long long* sortkey_r; int put_after_last; mm_row_t* d_rows; };
int rows_sequence(mm_tie* t, Bufs& B, const mm_row_t* rows, uint64_t n, RowSeq* out) {
    hipStream_t st = t->st;
    mm_row_t* d_rows = B.get<mm_row_t>(n);
    u64* stamp[2] = {B.get<u64>(n), B.get<u64>(n)};
    uint32_t* idx[2] = {B.get<uint32_t>(n), B.get<uint32_t>(n)};
    uint32_t* hash = B.get<uint32_t>(n);
    long long* sortkey = B.get<long long>(n);
    uint32_t* hash_r = B.get<uint32_t>(n);
    long long* sortkey_r = B.get<long long>(n);
    const uint32_t nblk = blocks(n, kSortTile);
    uint32_t* hist = B.get<uint32_t>((size_t)256 * nblk + 8);
    u64* d_or = B.get<u64>(2);
    if (!d_rows || !stamp[0] || !stamp[1] || !idx[0] || !idx[1] || !hash || !sortkey || !hash_r || !sortkey_r || !hist || !d_or) return -MM_E_NOMEM;
    if (hipMemcpyAsync(d_rows, rows, sizeof(mm_row_t) * n, hipMemcpyHostToDevice, st) != hipSuccess) return -MM_E_HIP;
    if (hipMemsetAsync(t->d_fail, 0, 4, st) != hipSuccess) return -MM_E_HIP;
    if (!t->d_gkey) { const int r = ensure_stamps(t, 0); if (r) return r; }
    LAUNCH(k_tie_rows, blocks(n), 256, st, tables_of(t), (const mm_row_t*)d_rows, (u64)n, (const u64*)t->d_gkey, (const u64*)t->d_gstamp, (u64)(t->gcap - 1), stamp[0], hash, sortkey, t->d_fail);
    LAUNCH(k_iota32, blocks(n), 256, st, idx[0], (u64)n);
    uint32_t hfail = 0;
    if (hipMemcpyAsync(&hfail, t->d_fail, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    if (hfail) { t->failed |= hfail; return -MM_E_NOCODE; }
    const int w = radix_sort(stamp[0], idx[0], stamp[1], idx[1], n, hist, d_or, &t->h_words[3], st);
    if (w < 0) return -MM_E_HIP;
    LAUNCH(k_gather32, blocks(n), 256, st, (const uint32_t*)hash, (const uint32_t*)idx[w], (u64)n, hash_r);
    LAUNCH(k_gather64, blocks(n), 256, st, (const long long*)sortkey, (const uint32_t*)idx[w], (u64)n, sortkey_r);
    uint64_t top = 0, lastput = 0;
    if (hipMemcpyAsync(&top, stamp[w] + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&lastput, t->d_words, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    out->order = idx[w]; out->hash_r = hash_r; out->sortkey_r = sortkey_r; out->put_after_last = lastput > top ? 1 : 0; out->d_rows = d_rows;
    return 0;
}
}  // namespace

extern "C" {

mm_tie_t* mm_tie_create(const mm_tie_opts_t* opts, const char* const* contig_names, const int64_t* contig_len, char* err, size_t err_len) {
    auto fail = [&](const char* m) -> mm_tie_t* { if (err && err_len) snprintf(err, err_len, "%s", m); return nullptr; };
    if (!opts || opts->abi_version != MM_TIE_ABI_VERSION) return fail("mm_tie_create: ABI version mismatch");
    if (opts->n_contigs < 0 || (opts->n_contigs > 0 && (!contig_names || !contig_len))) return fail("mm_tie_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device (there is no CPU fallback in this library)");
    if (hipSetDevice(opts->device) != hipSuccess) return fail("hipSetDevice failed");
    mm_tie* t = new mm_tie();
    t->o = *opts;
    const int nc = opts->n_contigs;
    std::vector<uint32_t> ch((size_t)std::max(nc, 1));
    std::vector<u64> cb((size_t)std::max(nc, 1));
    std::vector<int32_t> rk((size_t)std::max(nc, 1));
    u64 run = 0;
    for (int i = 0; i < nc; i++) {
        t->names.emplace_back(contig_names[i] ? contig_names[i] : "");
        t->len.push_back(contig_len[i]);
        const std::string& s = t->names.back();
        if (s.empty()) { delete t; return fail("mm_tie_create: a contig without a name (khash's string hash stops at the first NUL: the host replay does those)"); }
        uint32_t h = (uint32_t)(unsigned char)s[0];
        h = x31_str(h, s.data() + 1, s.size() - 1);
        ch[(size_t)i] = (h << 5) - h + (uint32_t)'\t';
        cb[(size_t)i] = run;
        run += (u64)(contig_len[i] > 0 ? contig_len[i] : 0) + 1ull;
    }
    if (run >= (1ull << 35) - 1ull) { delete t; return fail("mm_tie_create: more than 2^35 reference positions"); }
    {   // rank of every name in strcmp order; equal names share a rank (cmp_key_fast cannot tell them apart)
        std::vector<int> idx((size_t)nc);
        for (int i = 0; i < nc; i++) idx[(size_t)i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return strcmp(t->names[(size_t)a].c_str(), t->names[(size_t)b].c_str()) < 0; });
        for (int r = 0, cur = -1; r < nc; r++) {
            if (r == 0 || t->names[(size_t)idx[(size_t)r]] != t->names[(size_t)idx[(size_t)r - 1]]) cur = r;
            rk[(size_t)idx[(size_t)r]] = cur;
        }
    }
    bool ok = hipStreamCreateWithFlags(&t->st, hipStreamNonBlocking) == hipSuccess;
    const size_t ncs = (size_t)std::max(nc, 1);
    ok = ok && mmdev::dmalloc((void**)&t->d_ctg_hash, 4 * ncs) == hipSuccess && mmdev::dmalloc((void**)&t->d_ctg_base, 8 * ncs) == hipSuccess && mmdev::dmalloc((void**)&t->d_ctg_rank, 4 * ncs) == hipSuccess;
    ok = ok && mmdev::dmalloc((void**)&t->d_klass, 256 * MM_MAX_CODES) == hipSuccess && mmdev::dmalloc((void**)&t->d_mid, sizeof(uint2) * 2 * MM_MAX_CODES) == hipSuccess &&
         mmdev::dmalloc((void**)&t->d_codes, MM_MAX_CODES * MM_CODE_LEN) == hipSuccess;
    ok = ok && mmdev::dmalloc((void**)&t->d_words, 32) == hipSuccess && mmdev::dmalloc((void**)&t->d_fail, 16) == hipSuccess;
    ok = ok && mmdev::hmalloc((void**)&t->h_words, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) {
        ok = hipMemcpy(t->d_ctg_hash, ch.data(), 4 * ncs, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(t->d_ctg_base, cb.data(), 8 * ncs, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(t->d_ctg_rank, rk.data(), 4 * ncs, hipMemcpyHostToDevice) == hipSuccess && hipMemset(t->d_words, 0, 32) == hipSuccess && hipMemset(t->d_fail, 0, 16) == hipSuccess &&
             hipMemset(t->d_klass, 0, 256 * MM_MAX_CODES) == hipSuccess && hipMemset(t->d_mid, 0, sizeof(uint2) * 2 * MM_MAX_CODES) == hipSuccess && hipMemset(t->d_codes, 0, MM_MAX_CODES * MM_CODE_LEN) == hipSuccess;
    }
    if (!ok) { mm_tie_destroy(t); return fail("mm_tie_create: device allocation failed"); }
    return t;
}

int32_t mm_tie_set_codes(mm_tie_t* t, int32_t n_codes, const char* const* codes, const uint8_t* const* klass_of_code) {
    if (!t || n_codes < 0 || n_codes > MM_MAX_CODES) return -MM_E_ARG;
    std::vector<uint8_t> kl((size_t)256 * MM_MAX_CODES, 0);
    std::vector<uint2> mid((size_t)2 * MM_MAX_CODES);
    std::vector<char> cs((size_t)MM_MAX_CODES * MM_CODE_LEN, 0);
    for (int c = 0; c < n_codes; c++) {
        memcpy(&kl[(size_t)256 * c], klass_of_code[c], 256);
        const size_t cl = strnlen(codes[c], MM_CODE_LEN - 1);
        memcpy(&cs[(size_t)c * MM_CODE_LEN], codes[c], cl);
        for (int s = 0; s < 2; s++) {   // "\t<strand>\t<code>\t" as h -> h * mul + add
            char tmp[MM_CODE_LEN + 4];
            size_t tl = 0;
            tmp[tl++] = '\t'; tmp[tl++] = s ? '-' : '+'; tmp[tl++] = '\t';
            memcpy(tmp + tl, codes[c], cl); tl += cl;
            tmp[tl++] = '\t';
            uint32_t mul = 1, add = 0;
            for (size_t i = 0; i < tl; i++) { add = (add << 5) - add + (uint32_t)(unsigned char)tmp[i]; mul *= 31u; }
            mid[(size_t)s * 64 + (size_t)c] = make_uint2(mul, add);
        }
    }
    if (hipSetDevice(t->o.device) != hipSuccess) return -MM_E_HIP;
    if (hipStreamSynchronize(t->st) != hipSuccess) return -MM_E_HIP;
    if (hipMemcpy(t->d_klass, kl.data(), kl.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(t->d_mid, mid.data(), sizeof(uint2) * mid.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->d_codes, cs.data(), cs.size(), hipMemcpyHostToDevice) != hipSuccess) return -MM_E_HIP;
    t->n_codes = n_codes;
    return 0;
}

// (the launch's work: whatever it returns, mm_tie_add_launch has counted the launch's reads -- a launch that is NOT taken leaves the replay marked failed:
// keys of an unstamped launch that come again later would get a later first-insertion stamp, and the order printed from that would be wrong without a word)
static int32_t add_launch(mm_tie_t* t, const mm_batch_t* b, const void* dev_view_rows, int64_t n_rows, const uint64_t serial0) {
    const uint32_t nr = (uint32_t)b->n_reads;
    if (b->n_reads >= (1 << 21) || n_rows >= 0xFFFFFFF0ll) { t->failed |= TIE_F_ROWS; return -MM_E_TOOMANY; }
    if (hipSetDevice(t->o.device) != hipSuccess) return -MM_E_HIP;
    if (n_rows == 0) return 0;
    const size_t per = t->o.haplotypes ? 2 : 1;
    size_t cap_beg = t->cap_reads, cap_end = t->cap_reads;
    if (grow_buf(t, &t->d_beg, &cap_beg, nr) || grow_buf(t, &t->d_end, &cap_end, nr)) return -MM_E_NOMEM;
    t->cap_reads = std::min(cap_beg, cap_end);
    size_t c1 = t->cap_rows, c2 = t->cap_rows, c3 = t->cap_rows * per, c4 = t->cap_rows * per;
    if (grow_buf(t, &t->d_ska, &c1, (size_t)n_rows) || grow_buf(t, &t->d_skb, &c2, (size_t)n_rows) || grow_buf(t, &t->d_keys, &c3, (size_t)n_rows * per) || grow_buf(t, &t->d_khash, &c4, (size_t)n_rows * per))
        return -MM_E_NOMEM;
    t->cap_rows = std::min(std::min(c1, c2), std::min(c3, c4) / per);
    const size_t tabn = 4 * per * (size_t)n_rows + 8 * (size_t)nr;
    if (grow_buf(t, &t->d_tab, &t->cap_tab, tabn) || grow_buf(t, &t->d_rt, &t->cap_rt, (size_t)nr) || grow_buf(t, &t->d_rc, &t->cap_rc, (size_t)nr)) return -MM_E_NOMEM;
    { const int r = ensure_stamps(t, (uint64_t)n_rows * per); if (r) return r; }
    hipStream_t st = t->st;
    if (hipMemsetAsync(t->d_beg, 0, 4 * (size_t)nr, st) != hipSuccess || hipMemsetAsync(t->d_end, 0, 4 * (size_t)nr, st) != hipSuccess ||
        hipMemsetAsync(t->d_fail, 0, 4, st) != hipSuccess || hipMemsetAsync(t->d_words + 1, 0, 8, st) != hipSuccess || hipMemsetAsync(t->d_rt, 0, 4 * (size_t)nr, st) != hipSuccess || hipMemsetAsync(t->d_rc, 0, 4 * (size_t)nr, st) != hipSuccess) return -MM_E_HIP;
    LAUNCH(k_tie_bounds, blocks((uint64_t)n_rows), 256, st, (const mm_view_row_t*)dev_view_rows, (uint32_t)n_rows, nr, t->d_beg, t->d_end, t->d_fail);
    TieLaunch L;
    L.reads = b->reads; L.mm = b->mm; L.rows = (const mm_view_row_t*)dev_view_rows; L.n_reads = nr; L.n_rows = (uint32_t)n_rows;
    L.beg = t->d_beg; L.end = t->d_end; L.serial0 = serial0;
    L.sk_a = t->d_ska; L.sk_b = t->d_skb; L.keys = t->d_keys; L.khash = t->d_khash;
    L.tab = t->d_tab; L.rt = t->d_rt; L.rc = t->d_rc;
    L.gkey = t->d_gkey; L.gstamp = t->d_gstamp; L.gmask = t->gcap - 1;
    L.last_put = t->d_words; L.fail = t->d_fail;
    LAUNCH(k_tie_reads, blocks(nr, 64), 64, st, tables_of(t), L);
    LAUNCH(k_tie_keys, blocks((uint64_t)n_rows), 256, st, tables_of(t), L);
    LAUNCH(k_tie_puts, blocks(nr, 64), 64, st, tables_of(t), L);
    LAUNCH(k_tie_stamps, blocks(nr, 4), 256, st, tables_of(t), L);
    LAUNCH(k_stamp_count, (unsigned)std::min<uint64_t>(blocks(t->gcap), 2048), 256, st, (const u64*)t->d_gkey, (u64)t->gcap, t->d_words + 1);
    uint64_t cnt = 0; uint32_t hfail = 0;
    if (hipMemcpyAsync(&cnt, t->d_words + 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&hfail, t->d_fail, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    t->distinct = cnt;
    if (hfail) { t->failed |= hfail; return -MM_E_NOCODE; }
    return 0;
}
int32_t mm_tie_add_launch(mm_tie_t* t, const mm_batch_t* b, const void* dev_view_rows, int64_t n_rows, void* hip_stream) {
    if (!t || !b || n_rows < 0) return -MM_E_ARG;
    (void)hip_stream;   // (the rows are complete when mm_view_fetch_device returns; the replay has a stream of its own)
    if (b->n_reads <= 0) return 0;
    const uint64_t serial0 = t->serial;
    t->serial += (uint64_t)b->n_reads;   // once, whatever becomes of the launch: later launches' reads keep their serials
    const int32_t r = add_launch(t, b, dev_view_rows, n_rows, serial0);
    if (r != 0 && !t->failed) t->failed |= TIE_F_INTERNAL;   // (sticky: mm_tie_order_rows / mm_tie_sequence refuse from here on)
    if (r != 0) (void)hipGetLastError();
    return r;
}

int32_t mm_tie_order_rows(mm_tie_t* t, const mm_row_t* rows, int64_t n, uint32_t* perm) { return mm_tie_order_rows2(t, rows, n, perm, nullptr); }

int32_t mm_tie_order_rows2(mm_tie_t* t, const mm_row_t* rows, int64_t n, uint32_t* perm, mm_row_t* ordered) {
    if (!t || n < 0 || (n > 0 && (!rows || (!perm && !ordered)))) return -MM_E_ARG;
    if (t->failed) return -MM_E_NOCODE;
    if (n == 0) return 0;
    if (hipSetDevice(t->o.device) != hipSuccess) return -MM_E_HIP;
    memset(g_stats, 0, sizeof g_stats);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, t->st);
    Bufs B;
    RowSeq q;
    int r = rows_sequence(t, B, rows, (uint64_t)n, &q);
    uint32_t* d_final = B.get<uint32_t>((size_t)n);
    uint32_t* d_perm = B.get<uint32_t>((size_t)n);
    if (!r && (!d_final || !d_perm)) r = -MM_E_NOMEM;
    if (!r) r = core_and_sort(q.hash_r, q.sortkey_r, (uint64_t)n, q.put_after_last, nullptr, d_final, t->st);
    if (!r) {
        LAUNCH(k_gather32, blocks((uint64_t)n), 256, t->st, (const uint32_t*)q.order, (const uint32_t*)d_final, (u64)n, d_perm);
        if (perm && hipMemcpyAsync(perm, d_perm, 4 * (size_t)n, hipMemcpyDeviceToHost, t->st) != hipSuccess) r = -MM_E_HIP;
        if (!r && ordered) {   // the rows themselves in that order (they are on the device already)
            mm_row_t* d_ord = B.get<mm_row_t>((size_t)n);
            if (!d_ord) r = -MM_E_NOMEM;
            else {
                LAUNCH(k_gather_rows, blocks((uint64_t)n), 256, t->st, (const mm_row_t*)q.d_rows, (const uint32_t*)d_perm, (u64)n, d_ord);
                if (hipMemcpyAsync(ordered, d_ord, sizeof(mm_row_t) * (size_t)n, hipMemcpyDeviceToHost, t->st) != hipSuccess) r = -MM_E_HIP;
            }
        }
    }
    (void)hipEventRecord(e1, t->st);
    if (hipStreamSynchronize(t->st) != hipSuccess) r = r ? r : -MM_E_HIP;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_stats[6] = (uint64_t)(ms * 1000.0f);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return r;
}

int64_t mm_tie_sequence_size(const mm_tie_t* t) { return t ? (int64_t)t->distinct : -MM_E_ARG; }

int64_t mm_tie_sequence(mm_tie_t* t, mm_row_t* keys, uint32_t* hash, int64_t cap, int32_t* put_after_last) {
    if (!t || cap < 0 || (cap > 0 && (!keys || !hash))) return -MM_E_ARG;
    if (t->failed) return -MM_E_NOCODE;
    if (put_after_last) *put_after_last = 0;
    const uint64_t n = t->distinct;
    if (n == 0) return 0;
    if ((uint64_t)cap < n) return -MM_E_ARG;
    if (hipSetDevice(t->o.device) != hipSuccess) return -MM_E_HIP;
    hipStream_t st = t->st;
    Bufs B;
    u64* f = B.get<u64>(t->gcap);
    u64* tiles = B.get<u64>(t->gcap / kScanTile + 4);
    u64* ck = B.get<u64>(n);
    u64* cs[2] = {B.get<u64>(n), B.get<u64>(n)};
    uint32_t* idx[2] = {B.get<uint32_t>(n), B.get<uint32_t>(n)};
    mm_row_t* d_rows = B.get<mm_row_t>(n);
    uint32_t* d_hash = B.get<uint32_t>(n);
    const uint32_t nblk = blocks(n, kSortTile);
    uint32_t* hist = B.get<uint32_t>((size_t)256 * nblk + 8);
    u64* d_or = B.get<u64>(2);
    if (!f || !tiles || !ck || !cs[0] || !cs[1] || !idx[0] || !idx[1] || !d_rows || !d_hash || !hist || !d_or) return -MM_E_NOMEM;
    LAUNCH(k_stamp_flags, blocks(t->gcap), 256, st, (const u64*)t->d_gkey, (u64)t->gcap, f);
    scan64(f, t->gcap, tiles, st);
    LAUNCH(k_stamp_gather, blocks(t->gcap), 256, st, (const u64*)t->d_gkey, (const u64*)t->d_gstamp, (u64)t->gcap, (const u64*)f, ck, cs[0]);
    LAUNCH(k_iota32, blocks(n), 256, st, idx[0], (u64)n);
    const int w = radix_sort(cs[0], idx[0], cs[1], idx[1], n, hist, d_or, &t->h_words[3], st);
    if (w < 0) return -MM_E_HIP;
    LAUNCH(k_key_decode, blocks(n), 256, st, tables_of(t), (const u64*)ck, (const uint32_t*)idx[w], (u64)n, d_rows, d_hash);
    uint64_t top = 0, lastput = 0;
    if (hipMemcpyAsync(keys, d_rows, sizeof(mm_row_t) * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(hash, d_hash, 4 * n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&top, cs[w] + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&lastput, t->d_words, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    if (put_after_last) *put_after_last = lastput > top ? 1 : 0;
    return (int64_t)n;
}

uint32_t mm_tie_failed(const mm_tie_t* t) { return t ? t->failed : 0u; }
int64_t mm_tie_device_bytes(const mm_tie_t* t) { return t ? t->device_bytes : 0; }

void mm_tie_destroy(mm_tie_t* t) {
    if (!t) return;
    (void)hipSetDevice(t->o.device);
    if (t->st) (void)hipStreamSynchronize(t->st);
    void* ps[] = {t->d_klass, t->d_ctg_hash, t->d_ctg_base, t->d_ctg_rank, t->d_mid, t->d_codes, t->d_beg, t->d_end, t->d_ska, t->d_skb, t->d_keys, t->d_khash,
                  t->d_tab, t->d_rt, t->d_rc, t->d_gkey, t->d_gstamp, t->d_words, t->d_fail};
    for (void* p : ps) if (p) (void)mmdev::dfree(p);
    if (t->h_words) (void)mmdev::hfree((void*)t->h_words);
    if (t->st) (void)hipStreamDestroy(t->st);
    delete t;
}

int32_t mm_tie_order_plain(int32_t device, const uint32_t* hash, const int64_t* sortkey, int64_t n, int32_t put_after_last, uint32_t* slot_order, uint32_t* final_order) {
    if (n < 0 || (n > 0 && (!hash || !sortkey || !final_order))) return -MM_E_ARG;
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipSetDevice(device) != hipSuccess) return -MM_E_HIP;
    memset(g_stats, 0, sizeof g_stats);
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -MM_E_HIP;
    int r = 0;
    {
        Bufs B;
        uint32_t* d_hash = B.get<uint32_t>((size_t)n);
        long long* d_key = B.get<long long>((size_t)n);
        uint32_t* d_slot = B.get<uint32_t>((size_t)n);
        uint32_t* d_final = B.get<uint32_t>((size_t)n);
        if (!d_hash || !d_key || !d_slot || !d_final) r = -MM_E_NOMEM;
        if (!r && (hipMemcpyAsync(d_hash, hash, 4 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_key, sortkey, 8 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)) r = -MM_E_HIP;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        if (!r) r = core_and_sort(d_hash, d_key, (uint64_t)n, put_after_last, d_slot, d_final, st);
        (void)hipEventRecord(e1, st);
        if (!r && slot_order && hipMemcpyAsync(slot_order, d_slot, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) r = -MM_E_HIP;
        if (!r && hipMemcpyAsync(final_order, d_final, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) r = -MM_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) r = r ? r : -MM_E_HIP;
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g_stats[6] = (uint64_t)(ms * 1000.0f);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipStreamDestroy(st);
    return r;
}

int32_t mm_tie_last_stats(uint64_t out[8]) { memcpy(out, g_stats, sizeof g_stats); return 0; }

}  // extern "C"

#include "fmt_api.hip.h"
#include "summary_api.hip.h"
