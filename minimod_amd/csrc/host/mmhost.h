/* mmhost.h -- C host side of `minimod freq` and `minimod view` on the MI355X library (include/minimod_hip.h).
 * Mirrors the reference's driver layer: options (src/freq_main.c:46-64,182-296, src/mod.c:204-398), reference load
 * (src/ref.c:46-89), load_db (src/minimod.c:235-333) and print_freq_output (src/mod.c:628-728). */
#ifndef MMHOST_H
#define MMHOST_H
#include <stdint.h>
#include <stdio.h>

#include "bamio.h"
#include "minimod_hip.h"
#include "synth.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MMH_VERSION "0.1.0-mi355x (minimod v0.5.0 freq semantics)"

/* ---- logging in the reference's format (src/error.h:58-152) ---- */
extern int mmh_log_level;
#define MMH_INFO(msg, ...) do { if (mmh_log_level >= 3) fprintf(stderr, "[%s::INFO]\033[1;34m " msg "\033[0m\n", __func__, __VA_ARGS__); } while (0)
#define MMH_WARNING(msg, ...) do { if (mmh_log_level >= 2) fprintf(stderr, "[%s::WARNING]\033[1;33m " msg "\033[0m At %s:%d\n", __func__, __VA_ARGS__, __FILE__, __LINE__); } while (0)
#define MMH_ERROR(msg, ...) do { if (mmh_log_level >= 1) fprintf(stderr, "[%s::ERROR]\033[1;31m " msg "\033[0m At %s:%d\n", __func__, __VA_ARGS__, __FILE__, __LINE__); } while (0)

double mmh_realtime(void);
double mmh_cputime(void);
long mmh_peakrss(void);
int64_t mmh_parse_num(const char *str);   /* mm_parse_num, src/misc.c:74-87 */

/* ---- -c / -m ---- */
typedef struct mmh_mods {
    int n_mods;
    char code[MM_MAX_MODS][MM_CODE_LEN];
    char context[MM_MAX_MODS][MM_CODE_LEN];
    double thresh[MM_MAX_MODS];
} mmh_mods_t;
/* both return 0 or -1 with a message in err (the CLI prints it as the reference would and exits) */
int mmh_parse_mod_codes(const char *s, mmh_mods_t *out, char *err, size_t errlen);
int mmh_parse_mod_threshes(const char *s, mmh_mods_t *m, char *err, size_t errlen);
void mmh_klass_lut(double thresh, uint8_t lut[256]);   /* src/mod.c:56,1180-1191 */
void mmh_fill_opts(const mmh_mods_t *m, int insertions, int haplotypes, int device, mm_freq_opts_t *o);   /* o->view = 0 */

/* ---- reference ---- */
typedef struct mmh_ref {
    int n;
    char **name;
    uint8_t **seq;
    int64_t *len;
} mmh_ref_t;
mmh_ref_t *mmh_load_ref(const char *path);      /* load_ref: names up to the first whitespace, raw letters */
mmh_ref_t *mmh_load_ref_mt(const char *path, int threads);   /* ... a plain file in parallel passes over its mapping; threads <= 0: the stream parser */
int mmh_ref_find(const mmh_ref_t *r, const char *name);   /* last duplicate wins like kh_put (src/ref.c:81-82) */
void mmh_free_ref(mmh_ref_t *r);

/* ---- load_db ---- */
typedef struct mmh_loader {
    mm_bam_t *bam;
    int allow_secondary, skip_supplementary;
    int32_t K;
    int64_t B;
    /* statistics of the last batch / totals (db_t / core_t counters, src/minimod.h:147-150,190-194) */
    int32_t last_total_reads; int64_t last_total_bytes, last_processed_bytes;
    uint64_t total_reads, total_bytes, processed_reads, processed_bytes, processed_bases;
    void *priv;   /* pool sets and framing scratch (loader.c) */
} mmh_loader_t;
mmh_loader_t *mmh_loader_open(const char *bam_path, int threads, int32_t K, int64_t B, int allow_secondary, int skip_supplementary);
/* A worker of a sharded run (`minimod freq --devices`): the reader starts at `voffset` (from the .bai; UINT64_MAX = the
 * share is empty) and hands out the alignments that start inside [lo, hi) of the genome in (tid, pos) order; `first` /
 * `last` mark the workers that also take what lies in front of the first / behind the last share. */
mmh_loader_t *mmh_loader_open_share(const char *bam_path, int threads, int32_t K, int64_t B, int allow_secondary, int skip_supplementary,
                                    uint64_t voffset, int32_t lo_tid, int64_t lo_pos, int32_t hi_tid, int64_t hi_pos, int first, int last);
/* Fills `out` with the next batch (pointers into the loader's pools, valid until the next call with the same pool set).
 * Returns the number of accepted reads, or -1 on a read error.  *more = 0 when the reference's loop would stop
 * (src/freq_main.c:410). */
#define MMH_POOL_SETS 2   /* pool sets of a loader: the caller fills set k + 1 while the batches of sets k and k - 1 are still leaving for the device */
int32_t mmh_loader_next(mmh_loader_t *ld, int pool_set, mm_batch_t *out, int *more);
/* process-wide, before the loaders are opened: where the batches' device-bound arrays come from (pinned memory: the copies of
 * mm_freq_submit become DMA transfers); NULL, NULL = malloc */
void mmh_loader_set_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *));
/* read name of read `read` of the batch last loaded into `pool_set` */
const char *mmh_loader_qname(const mmh_loader_t *ld, int pool_set, int32_t read);
void mmh_loader_close(mmh_loader_t *ld);

/* ---- load_db with the decoded BAM kept in GPU memory (devloader.c on include/minimod_ingest.h) ---- */
typedef struct mmh_devloader mmh_devloader_t;
typedef struct mmh_devloader_opts {
    int device, n_targets, allow_secondary, skip_supplementary;
    uint64_t header_bytes;          /* where the first record begins in the decoded stream (mm_bam_peek_header2) */
    uint64_t voffset;               /* not 0: start at this virtual offset of a .bai instead (a worker of a sharded run) */
    int ranged, first, last, range_done_before_start;
    int32_t lo_tid, hi_tid; int64_t lo_pos, hi_pos;
    uint64_t target_bases;          /* a batch is handed out once it holds this many bases (0 = 600 M: what fills an MI355X several times over) */
    /* sizes, 0 = the library's defaults (tests make them small) */
    int group_slots, max_blocks, arenas;
    uint64_t max_cbytes, arena_bytes, head_room;
    int names;                      /* not 0: the batches carry their read names (view) */
} mmh_devloader_opts_t;
typedef struct mmh_devbatch {
    mm_batch_t batch;               /* DEVICE pointers: for mm_freq_submit_device_now on mmh_devloader_stream() */
    int arena;                      /* give it back with mmh_devloader_release once the batch's ticket has been waited for */
    uint64_t bases;
    uint64_t total_reads, total_bytes, processed_bytes;   /* of the records this batch was made from (db_t counters, src/minimod.h:147-150) */
    const uint8_t *names;           /* opts.names: DEVICE pointers -- read i's name at names + name_off[i], names_bytes in all */
    const uint64_t *name_off;
    uint64_t names_bytes;
} mmh_devbatch_t;
typedef struct mmh_devloader_stats {
    uint64_t total_reads, total_bytes, processed_reads, processed_bytes, processed_bases;   /* core_t counters, src/minimod.h:190-194 */
    uint64_t groups, slow_blocks, patched_blocks;
    double wait_seconds, stage_seconds;
    double stage_ms[4];             /* device milliseconds summed over the groups: host -> device copies, inflate, CRC32, frame + flatten (they overlap between groups) */
    int err;                        /* MM_INGEST_E_* of a failed run */
} mmh_devloader_stats_t;
mmh_devloader_t *mmh_devloader_open(const char *bam_path, mm_pool_t *pool, const mmh_devloader_opts_t *o, char *err, size_t err_len);
/* the next batch: the number of reads (0: none were accepted), -1 on a damaged file; *more = 0 behind the last one */
int32_t mmh_devloader_next(mmh_devloader_t *dl, mmh_devbatch_t *out, int *more);
void mmh_devloader_release(mmh_devloader_t *dl, int arena);
void *mmh_devloader_stream(mmh_devloader_t *dl);
int mmh_devloader_fetch(mmh_devloader_t *dl, void *dst_host, const void *src_dev, size_t n);   /* device bytes of a batch to the host (0 ok) */
/* the codes a batch's MM tags name, in the order a walk over the batch meets them (mm_ingest_batch_codes: -c '*' interns them before the
 * batch is submitted); codes: max_codes strings of MM_CODE_LEN bytes.  The number written, or < 0 (the caller walks the MM text itself) */
int mmh_devloader_codes(mmh_devloader_t *dl, const mm_batch_t *batch, char *codes, int max_codes);
const mmh_devloader_stats_t *mmh_devloader_stats(mmh_devloader_t *dl);
void mmh_devloader_close(mmh_devloader_t *dl);

/* ---- output ---- */
/* Rows are formatted on `pool` (NULL: on the calling thread) and written by a writer thread in row order; the text of
 * the rows is complete when a print call returns, the write may still be under way: call mmh_emit_flush() before
 * closing the file or writing to it otherwise (returns -1 if a write failed), mmh_emit_finish() when done with output.  `codes[c]` names mod code c. */
void mmh_print_freq_header(FILE *fp, int bedmethyl, int insertions, int haplotypes);
void mmh_print_freq_rows(FILE *fp, mm_pool_t *pool, const mm_row_t *rows, int64_t n, const mm_bam_hdr_t *hdr,
                         const char *const *codes, int n_codes, int bedmethyl, int insertions, int haplotypes);

/* print_view_header / print_view_output (src/mod.c:545-626) for one batch's rows */
void mmh_print_view_header(FILE *fp, int insertions, int haplotypes);
void mmh_print_view_rows(FILE *fp, mm_pool_t *pool, const mm_view_row_t *rows, int64_t n, const mm_batch_t *batch, const mmh_loader_t *ld, int pool_set,
                         const mm_bam_hdr_t *hdr, const char *const *codes, int n_codes, int insertions, int haplotypes);
void mmh_print_view_rows_of(FILE *fp, mm_pool_t *pool, const mm_view_row_t *rows, int64_t n, const mm_read_t *reads, const uint64_t *name_off, const char *names,
                            const mm_bam_hdr_t *hdr, const char *const *codes, int n_codes, int insertions, int haplotypes);   /* ... of a gathered launch */
int mmh_emit_flush(void);
int mmh_emit_finish(void);   /* flush, then stop the writer thread and free the recycled buffers */

/* ---- the reference's order of rows that tie on (contig, start) (tieorder.c) ---- */
typedef struct mmh_tie mmh_tie_t;
mmh_tie_t *mmh_tie_create(const mm_bam_hdr_t *hdr, int insertions, int haplotypes);
/* one batch's calls: the rows of a handle created with mm_freq_opts_t.view == 2; klass_of_code[c] = the threshold classes
 * (mmh_klass_lut) of the mod code c counts for */
int mmh_tie_add_batch(mmh_tie_t *t, mm_pool_t *pool, const mm_batch_t *batch, const mm_view_row_t *rows, int64_t n,
                      const uint8_t *const *klass_of_code, const char *const *codes, int n_codes);
/* mm_freq_finalize's rows, reordered in place into the order print_freq_output prints; -1 = no replay possible (rows untouched) */
int mmh_tie_order_rows(mmh_tie_t *t, mm_row_t *rows, int64_t n);
int mmh_tie_order_rows_mt(mmh_tie_t *t, mm_pool_t *pool, mm_row_t *rows, int64_t n);
/* the core table and the sort alone, from every distinct key's reference hash and comparator key in first-insertion order (the checker of
 * the device-side replay; what a --devices parent runs) */
int mmh_tie_order_plain(const uint32_t *hash, const int64_t *sortkey, int64_t n, int put_after_last, uint32_t *slot_order, uint32_t *final);
/* the first-insertion sequence of the keys so far (opaque 16-byte keys + hashes), and the same appended to another replay: the
 * workers of `--devices` replay their own reads, the parent strings the sequences together in file order */
int64_t mmh_tie_export(const mmh_tie_t *t, const void **keys, const uint32_t **hash);
int mmh_tie_import(mmh_tie_t *t, const void *keys, const uint32_t *hash, int64_t n);
int64_t mmh_tie_export2(const mmh_tie_t *t, const void **keys, const uint32_t **hash, int *put_after_last);
int mmh_tie_import2(mmh_tie_t *t, const void *keys, const uint32_t *hash, int64_t n, int put_after_last);
void mmh_tie_keys_from_rows(const mm_row_t *rows, const uint32_t *seq, int64_t n, void *keys16);
void mmh_tie_destroy(mmh_tie_t *t);

extern int mmh_gpu_in_use;   /* set once the HIP runtime is up in this process (exitpath.c) */
void mmh_leave_teardown_behind(void);   /* just before _exit(): the address space outlives the process in a helper that holds nothing else (exitpath.c) */
int mmh_freq_main(int argc, char **argv);
int mmh_view_main(int argc, char **argv);
int mmh_summary_main(int argc, char **argv);   /* host only */

#ifdef __cplusplus
}
#endif
#endif
