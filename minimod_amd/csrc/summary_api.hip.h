// summary_api.hip.h -- `minimod summary` as a census kernel (reference summary_single src/mod.c:1426-1555,
// print_summary_output src/mod.c:1376-1400): per read, the set of "<base>|<codes>|<flag>" of its MM groups that list at least one call,
// in the slot order of the read's own khash (make_key_summary / add_summary_entry, src/mod.c:1402-1424) -- a thread per read walks the
// MM text as the reference does, enters the keys into its table (the per-read table of tie_kernels.hip.h: kh_get, then kh_put for a
// key that is not there, growth with kick-outs) and writes the column's text.  The host prints "<read name>\t" in front of it.
// (included at the end of tie_api.hip: one translation unit holds the scan kernels this uses)
#pragma once
#include "minimod_summary.h"

namespace {

struct SumKey { uint32_t off, len; uint8_t base, flag, pad[2]; uint32_t hash; };   // codes = mm[off, off + len)

__device__ inline bool sum_valid_base(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'U' || c == 'N' || c == 'a' || c == 'c' || c == 'g' || c == 't' || c == 'u' || c == 'n'; }

// per read: how many groups its MM text can hold at most (its ';' + 1): sizes of its scratch and of its text
// a read's table of census keys: key numbers and khash's two flag arrays (kh_resize works in place with both)
struct SumTab { uint32_t* id; uint8_t* old; uint8_t* nw; uint32_t nb, size, upper; };
__global__ __launch_bounds__(256) void k_sum_bound(const mm_read_t* __restrict__ reads, const uint8_t* __restrict__ mm, uint32_t n, u64* __restrict__ groups) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint8_t* s = mm + reads[r].mm_off;
    const uint32_t len = reads[r].mm_len;
    uint32_t g = 1;
    for (uint32_t i = 0; i < len; i++) g += s[i] == ';';
    groups[r] = g;
}
__global__ __launch_bounds__(64) void k_sum_reads(const mm_read_t* __restrict__ reads, const uint8_t* __restrict__ mm, uint32_t n, const u64* __restrict__ gincl,
                                                  SumKey* __restrict__ keys, uint32_t* __restrict__ tab_id, uint8_t* __restrict__ tab_old, uint8_t* __restrict__ tab_new,
                                                  char* __restrict__ text, uint64_t* __restrict__ text_off, uint32_t* __restrict__ text_len, int32_t* __restrict__ status) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= n) return;
    const mm_read_t rd = reads[r];
    const uint8_t* s = mm + rd.mm_off;
    const uint32_t len = rd.mm_len;
    const u64 g0 = r ? gincl[r - 1] : 0ull;           // groups of the reads in front: this read's keys start there
    SumKey* K = keys + g0;
    const u64 tb = 4ull * g0 + 8ull * r;              // its table: up to 4 g + 8 buckets
    SumTab tab;
    tab.id = tab_id + tb; tab.old = tab_old + tb; tab.nw = tab_new + tb; tab.nb = 0; tab.size = 0; tab.upper = 0;
    const u64 t0 = rd.mm_off + 6ull * g0;             // its text: at most mm_len + 6 per group
    char* out = text + t0;
    text_off[r] = t0;
    int32_t err = 0;
    uint32_t i = 0, nk = 0;
    while (i < len && !err) {
        uint8_t flag = '.', base = 0;
        if (i < len) { if (!sum_valid_base(s[i])) { err = MM_E_MMBASE; break; } base = s[i] == 'U' ? 'T' : s[i]; i++; }
        if (i < len) { if (s[i] != '+' && s[i] != '-') { err = MM_E_MMSTRAND; break; } i++; }
        const uint32_t c0 = i;
        bool nums = false, alpha = false;
        while (i < len && s[i] != ',' && s[i] != ';' && s[i] != '?' && s[i] != '.') {
            const uint8_t c = s[i];
            if (c >= '0' && c <= '9') nums = true;
            else if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) alpha = true;
            else { err = MM_E_MMCODE; break; }
            i++;
        }
        if (err) break;
        const uint32_t clen = i - c0;
        if (clen == 0) { err = MM_E_MMEMPTY; break; }
        if (nums && alpha) { err = MM_E_MMMIXED; break; }
        if (i < len && (s[i] == '?' || s[i] == '.')) { flag = s[i]; i++; }
        uint32_t skips = 0;
        while (i < len && s[i] != ';') {
            if (s[i] == ',') { i++; continue; }
            uint32_t l = 0;
            while (i < len && s[i] != ',' && s[i] != ';') { i++; l++; if (l >= 10) { err = MM_E_SKIPLEN; break; } }
            if (err) break;
            skips++;
        }
        if (err) break;
        i++;
        if (skips == 0) continue;   // no calls listed: the group is not reported (src/mod.c:1549)
        // "<base>|<codes>|<flag>" and its X31 hash (src/khash.h:486-494)
        uint32_t h = base;
        h = x31_c(h, '|');
        for (uint32_t q = 0; q < clen; q++) h = x31_c(h, s[c0 + q]);
        h = x31_c(h, '|'); h = x31_c(h, flag);
        SumKey k; k.off = c0; k.len = clen; k.base = base; k.flag = flag; k.pad[0] = k.pad[1] = 0; k.hash = h;
        K[nk] = k;
        // kh_get, then kh_put for a key that is not there (add_summary_entry, src/mod.c:1412-1424)
        bool found = false;
        if (tab.nb) {
            const uint32_t mask = tab.nb - 1u;
            uint32_t p = h & mask, step = 0;
            while (tab.old[p]) {
                const SumKey o = K[tab.id[p]];
                if (o.hash == h && o.len == clen && o.base == base && o.flag == flag) {
                    bool same = true;
                    for (uint32_t q = 0; q < clen; q++) if (s[o.off + q] != s[c0 + q]) { same = false; break; }
                    if (same) { found = true; break; }
                }
                p = (p + (++step)) & mask;
            }
            if (!found && tab.size < tab.upper) { tab.old[p] = 1; tab.id[p] = nk; tab.size++; nk++; continue; }
        }
        if (found) continue;
        {   // grow (kh_resize with its kick-outs), then the free slot of the key's path
            uint32_t nb2 = tab.nb ? tab.nb * 2u : 4u;
            const uint32_t mask = nb2 - 1u;
            for (uint32_t q = 0; q < nb2; q++) tab.nw[q] = 0;
            for (uint32_t j = 0; j < tab.nb; j++) {
                if (!tab.old[j]) continue;
                uint32_t key = tab.id[j];
                tab.old[j] = 0;
                for (;;) {
                    uint32_t p = K[key].hash & mask, step = 0;
                    while (tab.nw[p]) p = (p + (++step)) & mask;
                    tab.nw[p] = 1;
                    if (p < tab.nb && tab.old[p]) { const uint32_t tmp = tab.id[p]; tab.id[p] = key; key = tmp; tab.old[p] = 0; }
                    else { tab.id[p] = key; break; }
                }
            }
            uint8_t* x = tab.old; tab.old = tab.nw; tab.nw = x;
            tab.nb = nb2; tab.upper = (uint32_t)(nb2 * 0.77 + 0.5);
            uint32_t p = h & mask, step = 0;
            while (tab.old[p]) p = (p + (++step)) & mask;
            tab.old[p] = 1; tab.id[p] = nk; tab.size++; nk++;
        }
    }
    uint32_t w = 0;
    if (!err)
        for (uint32_t p = 0; p < tab.nb; p++) {   // the keys in slot order, a blank behind each (print_summary_output, src/mod.c:1389-1394)
            if (!tab.old[p]) continue;
            const SumKey k = K[tab.id[p]];
            out[w++] = (char)k.base; out[w++] = '|';
            for (uint32_t q = 0; q < k.len; q++) out[w++] = (char)s[k.off + q];
            out[w++] = '|'; out[w++] = (char)k.flag; out[w++] = ' ';
        }
    text_len[r] = w;
    status[r] = err;
}

}  // namespace

struct mm_summary {
    int device = 0;
    hipStream_t st = nullptr;
    mm_read_t* d_reads = nullptr; size_t cap_reads = 0;
    uint8_t* d_mm = nullptr; size_t cap_mm = 0;
    u64 *d_groups = nullptr, *d_tiles = nullptr; uint64_t* d_toff = nullptr; uint32_t* d_tlen = nullptr; int32_t* d_status = nullptr;
    SumKey* d_keys = nullptr; uint32_t* d_tab = nullptr; uint8_t *d_old = nullptr, *d_new = nullptr; size_t cap_groups = 0, cap_tab = 0;
    char* d_text = nullptr; size_t cap_text = 0;
    std::vector<char> h_text; std::vector<uint64_t> h_off; std::vector<uint32_t> h_len; std::vector<int32_t> h_status;
};

extern "C" {

mm_summary_t* mm_summary_create(int32_t device, char* err, size_t err_len) {
    auto fail = [&](const char* m) -> mm_summary_t* { if (err && err_len) snprintf(err, err_len, "%s", m); return nullptr; };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device (the census kernel has no CPU fallback in this library)");
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    mm_summary* s = new mm_summary();
    s->device = device;
    if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) { delete s; return fail("stream creation failed"); }
    return s;
}

int32_t mm_summary_batch(mm_summary_t* s, const mm_batch_t* b, const char** text, const uint64_t** off, const uint32_t** len, int32_t* bad_read) {
    if (!s || !b || !text || !off || !len) return MM_E_ARG;
    if (bad_read) *bad_read = -1;
    const uint32_t n = (uint32_t)b->n_reads;
    if (b->n_reads <= 0) { *text = ""; *off = nullptr; *len = nullptr; return 0; }
    if (hipSetDevice(s->device) != hipSuccess) return MM_E_HIP;
    hipStream_t st = s->st;
    auto regrow = [&](void** p, size_t* cap, size_t need, size_t elem) -> int {
        if (*p && need <= *cap) return 0;
        if (*p) (void)mmdev::dfree(*p);
        *p = nullptr;
        const size_t nc = need + need / 4 + 1024;
        if (mmdev::dmalloc(p, nc * elem) != hipSuccess) return MM_E_NOMEM;
        *cap = nc;
        return 0;
    };
    if (n > s->cap_reads) {
        void* ps[] = {s->d_reads, s->d_groups, s->d_tiles, s->d_toff, s->d_tlen, s->d_status};
        for (void* p : ps) if (p) (void)mmdev::dfree(p);
        const size_t nc = (size_t)n + n / 4 + 1024;
        if (mmdev::dmalloc((void**)&s->d_reads, sizeof(mm_read_t) * nc) != hipSuccess || mmdev::dmalloc((void**)&s->d_groups, 8 * nc) != hipSuccess || mmdev::dmalloc((void**)&s->d_tiles, 8 * (nc / kScanTile + 4)) != hipSuccess ||
            mmdev::dmalloc((void**)&s->d_toff, 8 * nc) != hipSuccess || mmdev::dmalloc((void**)&s->d_tlen, 4 * nc) != hipSuccess || mmdev::dmalloc((void**)&s->d_status, 4 * nc) != hipSuccess) return MM_E_NOMEM;
        s->cap_reads = nc;
    }
    { size_t c = s->cap_mm; if (regrow((void**)&s->d_mm, &c, (size_t)b->n_mm_bytes + 64, 1)) return MM_E_NOMEM; s->cap_mm = c; }
    if (hipMemcpyAsync(s->d_reads, b->reads, sizeof(mm_read_t) * n, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(s->d_mm, b->mm, (size_t)b->n_mm_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return MM_E_HIP;
    hipLaunchKernelGGL(k_sum_bound, dim3(blocks(n)), dim3(256), 0, st, (const mm_read_t*)s->d_reads, (const uint8_t*)s->d_mm, n, s->d_groups);
    scan64(s->d_groups, n, s->d_tiles, st);
    u64 total_groups = 0;
    if (hipMemcpyAsync(&total_groups, s->d_groups + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return MM_E_HIP;
    const size_t tab_need = 4 * (size_t)total_groups + 8 * (size_t)n + 64;
    if (total_groups > s->cap_groups || !s->d_keys || tab_need > s->cap_tab) {
        void* ps[] = {s->d_keys, s->d_tab, s->d_old, s->d_new};
        for (void* p : ps) if (p) (void)mmdev::dfree(p);
        s->d_keys = nullptr; s->d_tab = nullptr; s->d_old = nullptr; s->d_new = nullptr; s->cap_groups = 0; s->cap_tab = 0;
        const size_t nc = (size_t)total_groups + (size_t)total_groups / 4 + 1024, tn = tab_need + tab_need / 4;
        if (mmdev::dmalloc((void**)&s->d_keys, sizeof(SumKey) * nc) != hipSuccess || mmdev::dmalloc((void**)&s->d_tab, 4 * tn) != hipSuccess || mmdev::dmalloc((void**)&s->d_old, tn) != hipSuccess || mmdev::dmalloc((void**)&s->d_new, tn) != hipSuccess) return MM_E_NOMEM;
        s->cap_groups = nc; s->cap_tab = tn;
    }
    const size_t text_cap = (size_t)b->n_mm_bytes + 6 * (size_t)total_groups + 64;
    { size_t c = s->cap_text; if (regrow((void**)&s->d_text, &c, text_cap, 1)) return MM_E_NOMEM; s->cap_text = c; }
    hipLaunchKernelGGL(k_sum_reads, dim3(blocks(n, 64)), dim3(64), 0, st, (const mm_read_t*)s->d_reads, (const uint8_t*)s->d_mm, n, (const u64*)s->d_groups, s->d_keys, s->d_tab, s->d_old, s->d_new,
                       s->d_text, s->d_toff, s->d_tlen, s->d_status);
    s->h_text.resize(text_cap); s->h_off.resize(n); s->h_len.resize(n); s->h_status.resize(n);
    if (hipMemcpyAsync(s->h_text.data(), s->d_text, text_cap, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(s->h_off.data(), s->d_toff, 8 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(s->h_len.data(), s->d_tlen, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(s->h_status.data(), s->d_status, 4 * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) return MM_E_HIP;
    *text = s->h_text.data(); *off = s->h_off.data(); *len = s->h_len.data();
    for (uint32_t r = 0; r < n; r++) if (s->h_status[r]) { if (bad_read) *bad_read = (int32_t)r; return s->h_status[r]; }
    return 0;
}

void mm_summary_destroy(mm_summary_t* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->st) (void)hipStreamSynchronize(s->st);
    void* ps[] = {s->d_reads, s->d_mm, s->d_groups, s->d_tiles, s->d_toff, s->d_tlen, s->d_status, s->d_keys, s->d_tab, s->d_old, s->d_new, s->d_text};
    for (void* p : ps) if (p) (void)mmdev::dfree(p);
    if (s->st) (void)hipStreamDestroy(s->st);
    delete s;
}

}  // extern "C"
