// fmt_api.hip.h -- print_freq_output's text made on the device (second half; reference src/mod.c:666-719):
// a row's length (k_fmt_len), the rows' places by a prefix sum, the characters (k_fmt_write: a thread per row, csrc/fmt_core.h's
// integer arithmetic for every "%d" and the "%f").  The host moves rows in and text out and writes it; nothing is formatted there.
// HBM-bound by construction (24 B of row in, ~25 - 60 B of text out); the byte stores of neighbouring threads fall into neighbouring
// lines, which the L2 merges.
// (included at the end of tie_api.hip: one translation unit holds the kernels of tie_kernels.hip.h, whose scan this uses)
#pragma once
#include "fmt_core.h"

namespace {

struct FmtTables {
    const char* names; const uint32_t* name_off; const uint32_t* name_len;   // contig names, one behind the other
    const char* codes;                                                       // [64][MM_CODE_LEN]
    const uint32_t* code_len;
    int32_t n_contigs, n_codes, bedmethyl, insertions, haplotypes;
};

__device__ inline void row_in(const FmtTables& T, const mm_row_t& w, mm_fmt_row_in_t* r) {
    const bool ct = w.tid >= 0 && w.tid < T.n_contigs;
    r->contig = ct ? T.names + T.name_off[w.tid] : "*"; r->contig_len = ct ? (int)T.name_len[w.tid] : 1;
    const bool cc = w.code >= 0 && w.code < T.n_codes;
    r->code = cc ? T.codes + (uint32_t)w.code * MM_CODE_LEN : ""; r->code_len = cc ? (int)T.code_len[w.code] : 0;
    r->pos = w.pos; r->n_called = w.n_called; r->n_mod = w.n_mod; r->strand = w.strand; r->ins_offset = w.ins_offset; r->hp = w.hp;
    r->bedmethyl = T.bedmethyl; r->insertions = T.insertions; r->haplotypes = T.haplotypes;
}
__global__ __launch_bounds__(256) void k_fmt_len(FmtTables T, const mm_row_t* __restrict__ rows, u64 n, u64* __restrict__ len) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    mm_fmt_row_in_t r;
    row_in(T, rows[i], &r);
    len[i] = (u64)mm_row_len(&r);
}
__global__ __launch_bounds__(256) void k_fmt_write(FmtTables T, const mm_row_t* __restrict__ rows, u64 n, const u64* __restrict__ incl, char* __restrict__ text) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    mm_fmt_row_in_t r;
    row_in(T, rows[i], &r);
    char* p = text + (i ? incl[i - 1] : 0ull);
    (void)mm_row_write(p, &r);
}

}  // namespace

struct mm_fmt {
    mm_fmt_opts_t o;
    hipStream_t st = nullptr;
    char* d_names = nullptr; uint32_t *d_name_off = nullptr, *d_name_len = nullptr; char* d_codes = nullptr; uint32_t* d_code_len = nullptr;
    mm_row_t* d_rows = nullptr; u64* d_len = nullptr; u64* d_tiles = nullptr; size_t cap_rows = 0, cap_own_rows = 0;
    char* d_text = nullptr; size_t cap_text = 0;
    char* h_text[2] = {nullptr, nullptr}; size_t cap_htext[2] = {0, 0}; int turn = 0; bool pinned[2] = {false, false};   // two host buffers taken in turn: pinned since the pieces are 128 k rows (~9 MB: 1.3 ms to pin, and the 446 MB of a 12-Gbase run come over at the link's speed instead of a pageable copy's)
    float last_ms = 0.f;
};

extern "C" {

mm_fmt_t* mm_fmt_create(const mm_fmt_opts_t* o, const char* const* contig_names, const char* const* codes, char* err, size_t err_len) {
    auto fail = [&](const char* m) -> mm_fmt_t* { if (err && err_len) snprintf(err, err_len, "%s", m); return nullptr; };
    if (!o || o->abi_version != MM_TIE_ABI_VERSION) return fail("mm_fmt_create: ABI version mismatch");
    if (o->n_contigs < 0 || o->n_codes < 0 || o->n_codes > MM_MAX_CODES) return fail("mm_fmt_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device (there is no CPU fallback in this library)");
    if (hipSetDevice(o->device) != hipSuccess) return fail("hipSetDevice failed");
    mm_fmt* f = new mm_fmt();
    f->o = *o;
    std::string names;
    std::vector<uint32_t> off((size_t)std::max(o->n_contigs, 1)), len((size_t)std::max(o->n_contigs, 1));
    for (int i = 0; i < o->n_contigs; i++) {
        const char* s = contig_names[i] ? contig_names[i] : "";
        off[(size_t)i] = (uint32_t)names.size(); len[(size_t)i] = (uint32_t)strlen(s);
        names.append(s);
    }
    names.push_back('\0');
    std::vector<char> cs((size_t)MM_MAX_CODES * MM_CODE_LEN, 0);
    std::vector<uint32_t> cl((size_t)MM_MAX_CODES, 0);
    for (int c = 0; c < o->n_codes; c++) {
        const size_t l = strnlen(codes[c], MM_CODE_LEN - 1);
        memcpy(&cs[(size_t)c * MM_CODE_LEN], codes[c], l); cl[(size_t)c] = (uint32_t)l;
    }
    // (the NULL stream: a stream of its own is one more hardware queue -- 8 - 12 ms to make at the one moment of a run when nothing else is under way,
    // 173 MB of host memory for its waves' saved state; MM_FMT_OWN_STREAM=1 makes one all the same)
    bool ok = !std::getenv("MM_FMT_OWN_STREAM") || hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking) == hipSuccess;
    ok = ok && mmdev::dmalloc((void**)&f->d_names, names.size()) == hipSuccess && mmdev::dmalloc((void**)&f->d_name_off, 4 * off.size()) == hipSuccess && mmdev::dmalloc((void**)&f->d_name_len, 4 * len.size()) == hipSuccess;
    ok = ok && mmdev::dmalloc((void**)&f->d_codes, cs.size()) == hipSuccess && mmdev::dmalloc((void**)&f->d_code_len, 4 * cl.size()) == hipSuccess;
    ok = ok && hipMemcpy(f->d_names, names.data(), names.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(f->d_name_off, off.data(), 4 * off.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(f->d_name_len, len.data(), 4 * len.size(), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(f->d_codes, cs.data(), cs.size(), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(f->d_code_len, cl.data(), 4 * cl.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mm_fmt_destroy(f); return fail("mm_fmt_create: device allocation failed"); }
    return f;
}

static int64_t fmt_rows_impl(mm_fmt_t* f, const mm_row_t* rows, bool on_device, int64_t n, const char** text);
int64_t mm_fmt_rows(mm_fmt_t* f, const mm_row_t* rows, int64_t n, const char** text) { return fmt_rows_impl(f, rows, false, n, text); }
int64_t mm_fmt_rows_device(mm_fmt_t* f, const mm_row_t* device_rows, int64_t n, const char** text) { return fmt_rows_impl(f, device_rows, true, n, text); }
static int64_t fmt_rows_impl(mm_fmt_t* f, const mm_row_t* rows, bool on_device, int64_t n, const char** text) {
    if (!f || n < 0 || (n > 0 && !rows) || !text) return -MM_E_ARG;
    *text = "";
    if (n == 0) return 0;
    if (hipSetDevice(f->o.device) != hipSuccess) return -MM_E_HIP;
    hipStream_t st = f->st;
    if ((size_t)n > f->cap_rows) {
        if (f->d_len) { (void)mmdev::dfree(f->d_len); (void)mmdev::dfree(f->d_tiles); f->d_len = nullptr; f->d_tiles = nullptr; f->cap_rows = 0; }
        const size_t cap = (size_t)n + (size_t)n / 8 + 1024;
        if (mmdev::dmalloc((void**)&f->d_len, 8 * cap) != hipSuccess || mmdev::dmalloc((void**)&f->d_tiles, 8 * (cap / kScanTile + 4)) != hipSuccess) return -MM_E_NOMEM;
        f->cap_rows = cap;
    }
    if (!on_device && (size_t)n > f->cap_own_rows) {   // (rows that are on the device already are read where they lie)
        if (f->d_rows) { (void)mmdev::dfree(f->d_rows); f->d_rows = nullptr; f->cap_own_rows = 0; }
        const size_t cap = (size_t)n + (size_t)n / 8 + 1024;
        if (mmdev::dmalloc((void**)&f->d_rows, sizeof(mm_row_t) * cap) != hipSuccess) return -MM_E_NOMEM;
        f->cap_own_rows = cap;
    }
    const mm_row_t* const src = on_device ? rows : (const mm_row_t*)f->d_rows;
    FmtTables T;
    T.names = f->d_names; T.name_off = f->d_name_off; T.name_len = f->d_name_len; T.codes = f->d_codes; T.code_len = f->d_code_len;
    T.n_contigs = f->o.n_contigs; T.n_codes = f->o.n_codes; T.bedmethyl = f->o.bedmethyl; T.insertions = f->o.insertions; T.haplotypes = f->o.haplotypes;
    struct Events {   // (made per call, gone on every way out)
        hipEvent_t a = nullptr, b = nullptr;
        Events() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    (void)hipGetLastError();
    if (!on_device && hipMemcpyAsync(f->d_rows, rows, sizeof(mm_row_t) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) return -MM_E_HIP;
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(k_fmt_len, dim3(blocks((uint64_t)n)), dim3(256), 0, st, T, src, (u64)n, f->d_len);
    const unsigned nt = blocks((uint64_t)n, kScanTile);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nt), dim3(256), 0, st, (const u64*)f->d_len, (u64)n, f->d_tiles);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, st, f->d_tiles, (uint32_t)nt);
    hipLaunchKernelGGL(k_scan_apply, dim3(nt), dim3(256), 0, st, f->d_len, (u64)n, (const u64*)f->d_tiles);
    uint64_t total = 0;
    if (hipGetLastError() != hipSuccess) return -MM_E_HIP;   // (a launch that did not start must not leave `total` to whatever d_len held)
    if (hipMemcpyAsync(&total, f->d_len + (n - 1), 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    if (total + 64 > f->cap_text) {
        if (f->d_text) (void)mmdev::dfree(f->d_text);
        f->d_text = nullptr; f->cap_text = 0;
        const size_t cap = (size_t)total + (size_t)total / 8 + 4096;
        if (mmdev::dmalloc((void**)&f->d_text, cap) != hipSuccess) return -MM_E_NOMEM;
        f->cap_text = cap;
    }
    const int tn = f->turn;
    f->turn ^= 1;
    if (total + 64 > f->cap_htext[tn]) {
        if (f->pinned[tn]) (void)mmdev::hfree(f->h_text[tn]); else free(f->h_text[tn]);
        f->h_text[tn] = nullptr; f->cap_htext[tn] = 0;
        const size_t cap = (size_t)total + (size_t)total / 8 + 4096;
        f->pinned[tn] = cap <= ((size_t)64 << 20) && !std::getenv("MM_FMT_PLAIN_TEXT") && mmdev::hmalloc((void**)&f->h_text[tn], cap, hipHostMallocDefault) == hipSuccess;
        if (!f->pinned[tn]) f->h_text[tn] = (char*)malloc(cap);
        if (!f->h_text[tn]) return -MM_E_NOMEM;
        f->cap_htext[tn] = cap;
    }
    hipLaunchKernelGGL(k_fmt_write, dim3(blocks((uint64_t)n)), dim3(256), 0, st, T, src, (u64)n, (const u64*)f->d_len, f->d_text);
    if (hipGetLastError() != hipSuccess) return -MM_E_HIP;
    (void)hipEventRecord(e1, st);
    if (hipMemcpyAsync(f->h_text[tn], f->d_text, (size_t)total, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -MM_E_HIP;
    (void)hipEventElapsedTime(&f->last_ms, e0, e1);
    *text = f->h_text[tn];
    return (int64_t)total;
}

float mm_fmt_last_kernel_ms(const mm_fmt_t* f) { return f ? f->last_ms : -1.f; }

void mm_fmt_destroy(mm_fmt_t* f) {
    if (!f) return;
    (void)hipSetDevice(f->o.device);
    (void)hipStreamSynchronize(f->st);
    void* ps[] = {f->d_names, f->d_name_off, f->d_name_len, f->d_codes, f->d_code_len, f->d_rows, f->d_len, f->d_tiles, f->d_text};
    for (void* p : ps) if (p) (void)mmdev::dfree(p);
    for (int i = 0; i < 2; i++) { if (f->pinned[i]) (void)mmdev::hfree(f->h_text[i]); else free(f->h_text[i]); }
    if (f->st) (void)hipStreamDestroy(f->st);
    delete f;
}

}  // extern "C"
