// freq_dispatch.cpp -- the public names of include/minimod_hip.h.  The freq / view path (freq_api.hip) is built once per
// reference-word kind (freq_kinds.h says why); mm_freq_create picks the kind its options need -- the rule of freq_api.hip:
// four bits a position for ONE requested mod whose context is made of A C G T (or is `*`), 16 bits up to five mods, 32 beyond --
// and every other call goes to the copy the handle was made by.  Plain C++: nothing here touches HIP.
#include <cstring>

#include "minimod_hip.h"

#define MM_DECL_KIND(K) \
    extern "C" { \
    mm_freq_t* mm_freq_create_k##K(const mm_freq_opts_t*, int32_t, const mm_contig_t*, int32_t, const mm_interval_t*, char*, size_t); \
    void mm_freq_destroy_k##K(mm_freq_t*); \
    int32_t mm_freq_submit_k##K(mm_freq_t*, const mm_batch_t*); \
    int32_t mm_freq_host_done_k##K(mm_freq_t*, int32_t); \
    int32_t mm_freq_read_record_k##K(mm_freq_t*, int32_t, int32_t, mm_read_t*); \
    int32_t mm_freq_ticket_batch_k##K(mm_freq_t*, int32_t, mm_batch_t*); \
    int32_t mm_freq_submit_device_k##K(mm_freq_t*, const mm_batch_t*, void*); \
    int32_t mm_freq_submit_device_now_k##K(mm_freq_t*, const mm_batch_t*, void*, uint64_t); \
    int32_t mm_freq_ticket_batches_k##K(mm_freq_t*, int32_t); \
    int32_t mm_freq_wait_k##K(mm_freq_t*, int32_t, int32_t*); \
    int64_t mm_view_fetch_k##K(mm_freq_t*, int32_t, const mm_view_row_t**, int32_t*); \
    int64_t mm_view_fetch_device_k##K(mm_freq_t*, int32_t, const void**, int32_t*); \
    int32_t mm_freq_intern_code_k##K(mm_freq_t*, const char*); \
    int32_t mm_freq_n_codes_k##K(const mm_freq_t*); \
    const char* mm_freq_code_name_k##K(const mm_freq_t*, int32_t); \
    int64_t mm_freq_finalize_k##K(mm_freq_t*, const mm_row_t**); \
    int64_t mm_freq_finalize_device_k##K(mm_freq_t*, const mm_row_t**, const mm_row_t**); \
    int64_t mm_freq_slab_words_k##K(const mm_freq_t*, int64_t); \
    int32_t mm_freq_slab_export_k##K(mm_freq_t*, int32_t, int64_t, int64_t, void*, void*); \
    int32_t mm_freq_slab_add_k##K(mm_freq_t*, int32_t, int64_t, int64_t, const void*, void*); \
    int32_t mm_freq_slab_clear_k##K(mm_freq_t*, int32_t, int64_t, int64_t, void*); \
    int32_t mm_freq_slab_export_host_k##K(mm_freq_t*, int32_t, int64_t, int64_t, void*); \
    int32_t mm_freq_slab_add_host_k##K(mm_freq_t*, int32_t, int64_t, int64_t, const void*); \
    int32_t mm_freq_slab_export_ipc_k##K(mm_freq_t*, int32_t, int64_t, int64_t, void*); \
    int32_t mm_freq_slab_add_ipc_k##K(mm_freq_t*, int32_t, int64_t, int64_t, const void*); \
    float mm_freq_last_kernel_ms_k##K(mm_freq_t*, int32_t); \
    int32_t mm_freq_stats_enable_k##K(mm_freq_t*, int32_t); \
    int32_t mm_freq_stats_get_k##K(mm_freq_t*, uint64_t*); \
    int64_t mm_freq_device_bytes_k##K(const mm_freq_t*); \
    int32_t mm_freq_launch_counts_k##K(const mm_freq_t*, uint64_t*); \
    void mm_freq_reset_counters_k##K(mm_freq_t*); \
    }
MM_DECL_KIND(0)
MM_DECL_KIND(1)
MM_DECL_KIND(2)
extern "C" {
int32_t mm_abi_version_k0(void);
const char* mm_strerror_k0(int32_t);
int32_t mm_freq_plan_batch_k0(const mm_read_t*, int32_t, int32_t*, int32_t);
}

struct mm_freq {   // the public handle: which copy made the real one
    int kind;
    mm_freq_t* impl;
};

// forward `call` to the handle's kind; `fail` when there is no handle
#define MM_FWD(h, fail, call) do { if (!(h)) return fail; switch ((h)->kind) { case 2: return call(2); case 1: return call(1); default: return call(0); } } while (0)

extern "C" {

int32_t mm_abi_version(void) { return mm_abi_version_k0(); }
const char* mm_strerror(int32_t code) { return mm_strerror_k0(code); }
int32_t mm_freq_plan_batch(const mm_read_t* reads, int32_t n, int32_t* items, int32_t cap) { return mm_freq_plan_batch_k0(reads, n, items, cap); }

mm_freq_t* mm_freq_create(const mm_freq_opts_t* opts, int32_t n_contigs, const mm_contig_t* contigs, int32_t n_intervals, const mm_interval_t* intervals, char* err, size_t err_len) {
    int kind = 1;
    if (opts) {
        bool plain = true;   // (freq_api.hip: four bits a position need a context made of A C G T, or `*`)
        const char* c = opts->mods[0].context;
        if (std::strcmp(c, "*") != 0) for (size_t j = 0; j < MM_CODE_LEN && c[j]; j++) if (!std::strchr("ACGT", c[j])) plain = false;
        // (the word's width goes by the DIFFERENT contexts: entries with one context string share its bits)
        int n_ctx = 0;
        const int nm = opts->n_mods < 0 ? 0 : (opts->n_mods > MM_MAX_MODS ? MM_MAX_MODS : opts->n_mods);
        for (int i = 0; i < nm; i++) {
            bool seen = false;
            for (int j = 0; j < i; j++) if (std::strcmp(opts->mods[j].context, opts->mods[i].context) == 0) seen = true;
            if (!seen) n_ctx++;
        }
        kind = n_ctx > 5 ? 2 : (opts->n_mods == 1 && plain ? 0 : 1);
    }
    mm_freq_t* impl = kind == 2 ? mm_freq_create_k2(opts, n_contigs, contigs, n_intervals, intervals, err, err_len)
                    : kind == 1 ? mm_freq_create_k1(opts, n_contigs, contigs, n_intervals, intervals, err, err_len)
                                : mm_freq_create_k0(opts, n_contigs, contigs, n_intervals, intervals, err, err_len);
    if (!impl) return nullptr;
    mm_freq* h = new mm_freq();
    h->kind = kind; h->impl = impl;
    return h;
}
void mm_freq_destroy(mm_freq_t* h) {
    if (!h) return;
    if (h->kind == 2) mm_freq_destroy_k2(h->impl); else if (h->kind == 1) mm_freq_destroy_k1(h->impl); else mm_freq_destroy_k0(h->impl);
    delete h;
}
#define C1(K) mm_freq_submit_k##K(h->impl, b)
int32_t mm_freq_submit(mm_freq_t* h, const mm_batch_t* b) { MM_FWD(h, -MM_E_ARG, C1); }
#define C2(K) mm_freq_host_done_k##K(h->impl, ticket)
int32_t mm_freq_host_done(mm_freq_t* h, int32_t ticket) { MM_FWD(h, MM_E_ARG, C2); }
#define C3(K) mm_freq_read_record_k##K(h->impl, ticket, index, out)
int32_t mm_freq_read_record(mm_freq_t* h, int32_t ticket, int32_t index, mm_read_t* out) { MM_FWD(h, MM_E_ARG, C3); }
#define C3b(K) mm_freq_ticket_batch_k##K(h->impl, ticket, out)
int32_t mm_freq_ticket_batch(mm_freq_t* h, int32_t ticket, mm_batch_t* out) { MM_FWD(h, MM_E_ARG, C3b); }
#define C4(K) mm_freq_submit_device_k##K(h->impl, b, st)
int32_t mm_freq_submit_device(mm_freq_t* h, const mm_batch_t* b, void* st) { MM_FWD(h, -MM_E_ARG, C4); }
#define C5(K) mm_freq_submit_device_now_k##K(h->impl, b, st, bases)
int32_t mm_freq_submit_device_now(mm_freq_t* h, const mm_batch_t* b, void* st, uint64_t bases) { MM_FWD(h, -MM_E_ARG, C5); }
#define C6(K) mm_freq_ticket_batches_k##K(h->impl, ticket)
int32_t mm_freq_ticket_batches(mm_freq_t* h, int32_t ticket) { MM_FWD(h, -MM_E_ARG, C6); }
#define C7(K) mm_freq_wait_k##K(h->impl, ticket, bad)
int32_t mm_freq_wait(mm_freq_t* h, int32_t ticket, int32_t* bad) { MM_FWD(h, MM_E_ARG, C7); }
#define C8(K) mm_view_fetch_k##K(h->impl, ticket, rows, bad)
int64_t mm_view_fetch(mm_freq_t* h, int32_t ticket, const mm_view_row_t** rows, int32_t* bad) { MM_FWD(h, -MM_E_ARG, C8); }
#define C9(K) mm_view_fetch_device_k##K(h->impl, ticket, rows, bad)
int64_t mm_view_fetch_device(mm_freq_t* h, int32_t ticket, const void** rows, int32_t* bad) { MM_FWD(h, -MM_E_ARG, C9); }
#define C10(K) mm_freq_intern_code_k##K(h->impl, code)
int32_t mm_freq_intern_code(mm_freq_t* h, const char* code) { MM_FWD(h, -MM_E_ARG, C10); }
#define C11(K) mm_freq_n_codes_k##K(h->impl)
int32_t mm_freq_n_codes(const mm_freq_t* h) { MM_FWD(h, 0, C11); }
#define C12(K) mm_freq_code_name_k##K(h->impl, code)
const char* mm_freq_code_name(const mm_freq_t* h, int32_t code) { MM_FWD(h, nullptr, C12); }
#define C13(K) mm_freq_finalize_k##K(h->impl, rows)
int64_t mm_freq_finalize(mm_freq_t* h, const mm_row_t** rows) { MM_FWD(h, -MM_E_ARG, C13); }
#define C13D(K) mm_freq_finalize_device_k##K(h->impl, rows, device_rows)
int64_t mm_freq_finalize_device(mm_freq_t* h, const mm_row_t** rows, const mm_row_t** device_rows) { MM_FWD(h, -MM_E_ARG, C13D); }
#define C14(K) mm_freq_slab_words_k##K(h->impl, len)
int64_t mm_freq_slab_words(const mm_freq_t* h, int64_t len) { MM_FWD(h, 0, C14); }
#define C15(K) mm_freq_slab_export_k##K(h->impl, tid, begin, len, dst, st)
int32_t mm_freq_slab_export(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* dst, void* st) { MM_FWD(h, -MM_E_ARG, C15); }
#define C16(K) mm_freq_slab_add_k##K(h->impl, tid, begin, len, src, st)
int32_t mm_freq_slab_add(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* src, void* st) { MM_FWD(h, -MM_E_ARG, C16); }
#define C17(K) mm_freq_slab_clear_k##K(h->impl, tid, begin, len, st)
int32_t mm_freq_slab_clear(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* st) { MM_FWD(h, -MM_E_ARG, C17); }
#define C18(K) mm_freq_slab_export_host_k##K(h->impl, tid, begin, len, dst)
int32_t mm_freq_slab_export_host(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* dst) { MM_FWD(h, -MM_E_ARG, C18); }
#define C19(K) mm_freq_slab_add_host_k##K(h->impl, tid, begin, len, src)
int32_t mm_freq_slab_add_host(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* src) { MM_FWD(h, -MM_E_ARG, C19); }
#define C20(K) mm_freq_slab_export_ipc_k##K(h->impl, tid, begin, len, hd)
int32_t mm_freq_slab_export_ipc(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, void* hd) { MM_FWD(h, -MM_E_ARG, C20); }
#define C21(K) mm_freq_slab_add_ipc_k##K(h->impl, tid, begin, len, hd)
int32_t mm_freq_slab_add_ipc(mm_freq_t* h, int32_t tid, int64_t begin, int64_t len, const void* hd) { MM_FWD(h, -MM_E_ARG, C21); }
#define C22(K) mm_freq_last_kernel_ms_k##K(h->impl, ticket)
float mm_freq_last_kernel_ms(mm_freq_t* h, int32_t ticket) { MM_FWD(h, -1.f, C22); }
#define C23(K) mm_freq_stats_enable_k##K(h->impl, enable)
int32_t mm_freq_stats_enable(mm_freq_t* h, int32_t enable) { MM_FWD(h, -MM_E_ARG, C23); }
#define C24(K) mm_freq_stats_get_k##K(h->impl, out)
int32_t mm_freq_stats_get(mm_freq_t* h, uint64_t out[16]) { MM_FWD(h, -MM_E_ARG, C24); }
#define C25(K) mm_freq_device_bytes_k##K(h->impl)
int64_t mm_freq_device_bytes(const mm_freq_t* h) { MM_FWD(h, 0, C25); }
#define C26(K) mm_freq_launch_counts_k##K(h->impl, out)
int32_t mm_freq_launch_counts(const mm_freq_t* h, uint64_t out[4]) { MM_FWD(h, -MM_E_ARG, C26); }
void mm_freq_reset_counters(mm_freq_t* h) {
    if (!h) return;
    if (h->kind == 2) mm_freq_reset_counters_k2(h->impl); else if (h->kind == 1) mm_freq_reset_counters_k1(h->impl); else mm_freq_reset_counters_k0(h->impl);
}

}  // extern "C"
