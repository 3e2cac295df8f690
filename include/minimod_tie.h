/*
 * minimod_tie.h -- C ABI of the device-side replay of the order in which the reference prints rows that tie on (contig, start)
 * (first half; the second half, the row text, is mm_fmt_* below).
 *
 * What it stands for (all paths under /root/reference):
 *   update_freq_map + the read's own khash       src/mod.c:883-929, src/khash.h:242-420      -> mm_tie_add_launch (k_tie_reads)
 *   merge_freq_maps (first insertion per key)    src/mod.c:743-774                          -> mm_tie_add_launch (stamps, atomicMin)
 *   the core table's slot order                  src/khash.h kh_put / kh_resize             -> mm_tie_order_rows (k_place_*, k_grow_*)
 *   ks_introsort under cmp_key_fast              src/ksort.h:180-230, src/mod.c:59-93,655-663  -> mm_tie_order_rows (k_qs_*)
 *   fprintf of a row, "%f" of the frequency      src/mod.c:666-719                          -> mm_fmt_rows (k_fmt_*)
 * The COUNTS never come from here; csrc/host/tieorder.c restates the same serially and is the checker (tests/test_hip_tie_gpu.py).
 *
 * Plain C: pointers, sizes, POD structs; a HIP stream is passed as void*.  Every call returns 0 / a count or -MM_E_*.
 */
#ifndef MINIMOD_TIE_H
#define MINIMOD_TIE_H

#include <stddef.h>
#include <stdint.h>

#include "minimod_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MM_TIE_ABI_VERSION 1

typedef struct mm_tie_opts {
    int32_t abi_version;
    int32_t device;
    int32_t insertions;      /* opt.insertions: ins_offset is part of the key (else 0, src/mod.c:1167-1172) */
    int32_t haplotypes;      /* opt.haplotypes: every call makes the key with the read's HP and the `-1` aggregate (src/mod.c:906-928) */
    int32_t n_contigs;
    int32_t rsvd;
} mm_tie_opts_t;

typedef struct mm_tie mm_tie_t;

/* contig names and lengths as in the BAM header (the key string begins with the name, src/mod.c:428-439; rows are compared by it,
 * src/mod.c:59-76).  NULL with a message in err on failure. */
mm_tie_t *mm_tie_create(const mm_tie_opts_t *opts, const char *const *contig_names, const int64_t *contig_len, char *err, size_t err_len);
/* the code strings rows and calls name by index (mm_freq_code_name) and, per code, the 256-entry class table of the mod it counts for
 * (mm_mod_t.klass: 0 = ambiguous, such a call never reaches update_freq_map).  May be called again when a wildcard run has interned more. */
int32_t mm_tie_set_codes(mm_tie_t *t, int32_t n_codes, const char *const *codes, const uint8_t *const *klass_of_code);
/* One launch's calls: `dev_batch` as submitted to a handle created with opts.view = 2 (device pointers; reads and MM text are looked
 * at), `dev_view_rows` / n_rows what mm_view_fetch_device returned for it.  Reads count on from the launches before (file order).
 * Returns 0 once the launch's keys are stamped (the call waits for its kernels). */
int32_t mm_tie_add_launch(mm_tie_t *t, const mm_batch_t *dev_batch, const void *dev_view_rows, int64_t n_rows, void *hip_stream);
/* rows (host memory, any order, one per key -- mm_freq_finalize's) -> perm[i] = index of the row print_freq_output prints i-th.
 * Returns 0, or -MM_E_* when the replay cannot be made for this input (a key without a stamp, a haplotype above 61, a read with more
 * than 2^18 calls...: mm_tie_failed says which); the caller then prints the canonical order, as it does when the host replay gives up. */
int32_t mm_tie_order_rows(mm_tie_t *t, const mm_row_t *rows, int64_t n, uint32_t *perm);
/* ... and / or the rows themselves in that order (either of perm, ordered may be NULL; `ordered` must not alias `rows`) */
int32_t mm_tie_order_rows2(mm_tie_t *t, const mm_row_t *rows, int64_t n, uint32_t *perm, mm_row_t *ordered);
/* Every key stamped so far in the order of its first insertion (a worker of `--devices` hands its own to the parent, which strings the
 * workers' sequences together: a key in a halo may have no row in this worker's output and still come first): keys[i] = the i-th key
 * entered as a row without counts, hash[i] = the reference's hash of its string, *put_after_last = whether any put followed the last
 * new key's.  mm_tie_sequence_size: how many there are; mm_tie_sequence returns that count (cap must hold it). */
int64_t mm_tie_sequence_size(const mm_tie_t *t);
int64_t mm_tie_sequence(mm_tie_t *t, mm_row_t *keys, uint32_t *hash, int64_t cap, int32_t *put_after_last);
uint32_t mm_tie_failed(const mm_tie_t *t);   /* 0, or why the replay gave up (bit set, csrc/tie_kernels.hip.h TIE_F_*; 64 = a launch was not taken: out of memory, a HIP
                                                * failure -- STICKY: from then on mm_tie_order_rows* and mm_tie_sequence refuse, and the caller hands the run to the host's replay) */
int64_t mm_tie_device_bytes(const mm_tie_t *t);
void mm_tie_destroy(mm_tie_t *t);

/* The core table and the sort alone: every distinct key's reference hash and comparator key (contig rank << 32 | start) in
 * first-insertion order -> slot_order (may be NULL): key numbers in the core table's slot order; final: in printing order.  The same
 * signature as the host's mmh_tie_order_plain, which it is tested against. */
int32_t mm_tie_order_plain(int32_t device, const uint32_t *hash, const int64_t *sortkey, int64_t n, int32_t put_after_last, uint32_t *slot_order, uint32_t *final_order);
/* kernel launches and fixpoint rounds of the last mm_tie_order_* call: [0] launches, [1] growths, [2] growth passes in all, [3] placement
 * rounds in all, [4] sort levels, [5] segments finished by one thread, [6] milliseconds on the device */
int32_t mm_tie_last_stats(uint64_t out[8]);

/* ---- the row text on the device (second half): print_freq_output's fprintf per row, src/mod.c:666-719 ----
 * One handle per output format.  mm_fmt_rows takes rows in printing order (host memory), makes their text on the device -- every
 * "%d" and the "%f" of the frequency by integer arithmetic, csrc/fmt_core.h, checked against snprintf in the CPU suite -- and hands back
 * the bytes in host memory the handle owns -- two buffers taken in turn: a call's text stays valid until the call after the next one, so a
 * caller can write one piece while the next is made.  The header line (print_freq_header) is the caller's.
 * A caller with more rows than it wants text for at once calls it piece by piece. */
typedef struct mm_fmt_opts {
    int32_t abi_version;     /* MM_TIE_ABI_VERSION */
    int32_t device;
    int32_t bedmethyl;       /* opt.bedmethyl_out: the bedMethyl columns, freq = n_mod * 100 / n_called (src/mod.c:671-690) */
    int32_t insertions;      /* TSV only: the ins_offset column */
    int32_t haplotypes;      /* TSV only: the haplotype column (`*` for -1) */
    int32_t n_contigs;
    int32_t n_codes;
    int32_t rsvd;
} mm_fmt_opts_t;
typedef struct mm_fmt mm_fmt_t;
mm_fmt_t *mm_fmt_create(const mm_fmt_opts_t *opts, const char *const *contig_names, const char *const *codes, char *err, size_t err_len);
int64_t mm_fmt_rows(mm_fmt_t *f, const mm_row_t *rows, int64_t n, const char **text);   /* bytes of text, or -MM_E_* */
/* ... of rows that are in GPU memory already (mm_freq_finalize_device's, include/minimod_hip.h; any part of them): read where they lie.  The rows must be
 * COMPLETE when this is called -- the stream that produced them synchronised (mm_freq_finalize_device returns that way): the formatter runs on the NULL
 * stream, which does not order itself behind the handles' non-blocking streams. */
int64_t mm_fmt_rows_device(mm_fmt_t *f, const mm_row_t *device_rows, int64_t n, const char **text);
float mm_fmt_last_kernel_ms(const mm_fmt_t *f);   /* device time of the last call's kernels (length, scan, write) */
void mm_fmt_destroy(mm_fmt_t *f);

#ifdef __cplusplus
}
#endif
#endif /* MINIMOD_TIE_H */
