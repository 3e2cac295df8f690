/*
 * minimod_hip.h -- C ABI of the MI355X (gfx950) `freq` hot path.
 *
 * The reference (warp9seq/minimod v0.5.0) has no plugin or FFI layer; the seam this library sits behind is
 * the three internal calls freq_main() makes per batch plus the reference set-up (all paths under
 * /root/reference):
 *
 *   load_ref + load_ref_contexts + init_core   src/ref.c:46-89,177-229, src/minimod.c:51-137   -> mm_freq_create
 *   process_db -> work_db -> freq_view_single  src/minimod.c:344-350, src/thread.c:145-158,
 *                                              src/mod.c:948-1370 (get_aln :776-881, update_freq_map :883-929)
 *                                                                                             -> mm_freq_submit*
 *   merge_db -> merge_freq_maps                src/minimod.c:373-386, src/mod.c:743-774        -> (none: counters
 *                                              are global on the device; mm_freq_wait reports per-read errors)
 *   output_core -> print_freq_output           src/minimod.c:388-394, src/mod.c:644-728        -> mm_freq_finalize
 *   destroy_ref / free_core                    src/ref.c:241-259, src/minimod.c:140-161        -> mm_freq_destroy
 *
 * `minimod view` (src/view_main.c) runs the same per-read function with core->opt.subtool == VIEW: a handle created with
 * opts.view = 1 collects per-read rows instead of counters:
 *
 *   add_view_entry                              src/mod.c:931-946 (called at :1195, :1282, :1362)   -> the call kernels
 *   print_view_output (per batch)               src/mod.c:560-626                                  -> mm_view_fetch
 *
 * Everything is plain C: pointers, sizes, POD structs.  No torch / HIP types appear in signatures (a HIP
 * stream is passed as void*).  Errors: the reference prints and exit(1)s from inside the hot path
 * (src/error.h:98-152); here every entry point returns a code and the caller (minimod_amd/csrc/host) prints
 * the reference's message and exits, so the behaviour at the CLI is the same.
 */
#ifndef MINIMOD_HIP_H
#define MINIMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 6
#define MM_MAX_MODS 32    /* requested -c entries (ABI 6; 13 before: the reference has no limit of its own, src/minimod.h:114 counts them in a byte) */
#define MM_MAX_CONTEXTS 13 /* DIFFERENT context strings a PASS of the reference words holds: entries with one context share its two bits (5 base bits + 2 x 13 in 32).  Not a limit
                            * of the run since round 6: more contexts are built in further passes (mm_freq_create) and tested through their site words */
#define MM_MAX_CODES 64   /* code strings known to the device (wildcard -c '*' interns what reads carry) */
#define MM_CODE_LEN 16    /* bytes per code / context string incl. NUL */
#define MM_MAX_HP_PLANES 8

/* One read of a batch: what load_db keeps of a bam1_t (src/minimod.c:235-333) flattened to offsets into four
 * pools.  64 bytes.  seq_off/mm_off are 16-byte aligned, cigar_off (in uint32 units) 4-aligned, and every pool
 * carries >= 64 zero bytes of slack at its end. */
typedef struct mm_read {
    uint64_t cigar_off;  /* index into cigar pool (uint32 units)   bam_get_cigar */
    uint64_t seq_off;    /* byte offset into seq pool (4-bit packed, high nibble first)  bam_get_seq */
    uint64_t mm_off;     /* byte offset into mm pool (MM:Z text, not NUL-counted)  get_mm_tag_ptr */
    uint64_t ml_off;     /* byte offset into ml pool (ML:B:C bytes)  get_ml_tag */
    int32_t tid;         /* core.tid */
    int32_t pos;         /* core.pos */
    uint32_t l_qseq;     /* core.l_qseq */
    uint32_t n_cigar;    /* core.n_cigar */
    uint32_t mm_len;     /* strlen(MM) */
    uint32_t ml_len;     /* ML array length (0 when absent / not B:C) */
    uint16_t flag;       /* core.flag (0x10 = reverse strand) */
    uint8_t hp;          /* get_hp_tag: (uint8) HP value, 0 when absent */
    uint8_t rsvd;
    uint32_t rsvd2;
} mm_read_t;

/* A -K/-B sized batch (db_t, src/minimod.h:125-156).  Pointers are host pointers for mm_freq_submit and device
 * pointers for mm_freq_submit_device. */
typedef struct mm_batch {
    const mm_read_t *reads;
    const uint32_t *cigar;
    const uint8_t *seq;
    const uint8_t *mm;
    const uint8_t *ml;
    const int32_t *order;    /* optional work items from mm_freq_plan_batch (long reads split into parts, costliest
                              * first); NULL = one item per read in stored order (mm_freq_submit plans by itself) */
    int32_t n_reads;
    int32_t n_order;         /* entries of order[] (ignored when order is NULL) */
    uint64_t n_cigar_words;  /* pool sizes incl. slack (elements / bytes) */
    uint64_t n_seq_bytes;
    uint64_t n_mm_bytes;     /* below 4 GiB per batch (tiles address the MM pool with 32 bits; more is -MM_E_ARG) */
    uint64_t n_ml_bytes;
    uint32_t max_n_cigar;    /* largest n_cigar / l_qseq in the batch (sizes the spill scratch) */
    uint32_t max_l_qseq;
} mm_batch_t;

/* One requested modification (modcodem_t, src/minimod.h:60-64 + the key of modcodes_map).  klass is the
 * threshold rule of src/mod.c:1180-1191 tabulated by the host in double precision for all 256 ML values:
 * 0 ambiguous (skipped), 1 called unmodified, 3 called modified. */
typedef struct mm_mod {
    char code[MM_CODE_LEN];
    char context[MM_CODE_LEN];
    uint8_t klass[256];
} mm_mod_t;

typedef struct mm_freq_opts {
    int32_t abi_version;
    int32_t n_mods;
    int32_t insertions;      /* opt.insertions */
    int32_t haplotypes;      /* opt.haplotypes */
    int32_t device;          /* HIP device ordinal */
    int32_t n_hp_planes;     /* dense planes for HP 0..n-1 (others go to the sparse side list); 0 = default */
    int64_t side_capacity;   /* sparse side-list records (16 B each); 0 = default */
    int32_t n_wild_planes;   /* with -c '*': dense planes for the first n interned codes; 0 = default */
    int32_t view;            /* 1 = `minimod view`: per-read rows (mm_view_fetch) instead of counters (mm_freq_finalize);
                              * 2 = the same rows with the MM group's ordinal (<= 2047) in the top eleven bits of mm_view_row_t.read
                              * (batches of fewer than 2^21 reads), bit 31 of read_pos set for the implicit calls of '.' groups,
                              * and later entries of a key kept: what the host's replay of the reference's row order needs
                              * (csrc/host/tieorder.c) */
    /* test / diagnostic switches (0 = the product's behaviour) */
    int32_t force_fused;     /* 1: every read through the fused one-wavefront-per-read kernel instead of the tile pipeline */
    int32_t view_cap;        /* view: records per append region before the grow-and-rerun path (0 = sized from the ML pool) */
    int32_t finalize_by_runs;/* 1: mm_freq_finalize takes the per-run compaction + host merge even when all rows are dense */
    int32_t split_bases;     /* device planning: reads longer than this are cut into parts of about this many bases (0 = default) */
    int32_t coalesce;        /* up to this many consecutive submits share one launch: for mm_freq_submit_device consecutive windows of
                              * one resident read set, for mm_freq_submit host batches staged one behind the other in device memory
                              * (see both).  0 = the library's default (32: a -K batch fills a fraction of an MI355X, and a caller that
                              * sets nothing should not get a launch per batch); 1 = every submit is its own launch.  (ABI 4: 0 meant 1.) */
    int32_t stream_mode;     /* which reads take the streaming kernel (k_stream_reads: a whole read in one wavefront) instead of the
                              * tile pipeline, in plain runs (freq or view; no --insertions, no --haplotypes).  0 (default): by the
                              * size of the launch -- none in a launch of fewer than about 15 000 reads (a single -K 4096 batch:
                              * the longest read would be the launch), in bigger (gathered) launches every read short enough to
                              * hide; 1: none; 2: every read of up to split_bases bases whatever the launch (tests);
                              * 3: like 2, with the '.'-capable instantiation of the kernel from the first launch on (otherwise the
                              * leaner one runs until a read with a '.' group has shown up -- that read goes through the tile
                              * pipeline -- and the '.'-capable one from the next launch on: a file's reads carry one flag or the other).
                              * (A reserved field before: same layout.) */
    int32_t gather_mb;       /* mm_freq_submit with coalesce > 1: MiB of staging a launch may gather (0 = 1024) */
    int32_t stream_slices;   /* k_stream_reads in launches of 8192 reads and more: 0 (default) one POSITION slice of the launch per XCD, each
                              * worked through in file order (the slice's site words, bases and counters stay in that XCD's L2); 1 the
                              * launch's reads costliest first whatever their place (rounds 2 and 3) */
    int32_t rsvd_opts;
    mm_mod_t mods[MM_MAX_MODS];
} mm_freq_opts_t;

/* One BAM-header contig (bam_hdr_t target_name/target_len) with its FASTA sequence (ref_t.forward before
 * normalisation; NULL when the FASTA lacks the contig -- a read on it then fails like src/mod.c:793). */
typedef struct mm_contig {
    const char *name;
    int64_t length;        /* BAM header target_len */
    const uint8_t *seq;    /* host pointer, raw FASTA letters, or NULL */
    int64_t seq_length;    /* FASTA length; must equal `length` (src/mod.c:861) */
} mm_contig_t;

/* Optional: restrict the dense counter planes to one reference interval per contig (multi-GPU sharding,
 * SURVEY.md section 8e).  Updates outside [begin, end + halo) of every listed contig go to the side list. */
typedef struct mm_interval {
    int32_t tid;
    int32_t rsvd;
    int64_t begin, end;    /* owned interval [begin, end) */
    int64_t halo;          /* extra positions kept past `end` */
} mm_interval_t;

/* One output row = one key of the reference's freq map (src/mod.c:428-439) with its value. */
typedef struct mm_row {
    int32_t tid;
    int32_t pos;
    uint8_t strand;        /* 0 '+', 1 '-' */
    uint8_t rsvd;
    uint16_t ins_offset;
    int16_t code;          /* index for mm_freq_code_name */
    int16_t hp;            /* -1 = '*' (all haplotypes / haplotypes off) */
    uint32_t n_called;
    uint32_t n_mod;
} mm_row_t;

/* One row of `minimod view` = one entry of a read's view map (view_t + its key, src/mod.c:931-946) in the order
 * print_view_output prints it (src/mod.c:560-626): reads in batch order, a read's rows by reference position; rows of
 * one read on one position -- which the reference leaves in hash order -- by (code, ins_offset).  Strand, contig,
 * haplotype and read name are per-read values the caller already holds (reads[row.read]).  16 bytes. */
typedef struct mm_view_row {
    uint32_t read;         /* index of the read in the batch */
    int32_t pos;           /* ref_pos */
    uint32_t read_pos;     /* position in the read as sequenced (view_t.read_pos) */
    uint16_t ins_offset;   /* 0 unless --insertions */
    uint8_t code;          /* index for mm_freq_code_name */
    uint8_t prob;          /* ML byte (view_t.mod_prob); 0 for the implicit calls of a '.' group */
} mm_view_row_t;

/* per-read status codes (0 = ok); the reference's message for each is in INTEGRATION.md */
enum {
    MM_OK = 0, MM_E_HARDCLIP = 1, MM_E_CIGAROP = 2, MM_E_MMBASE = 3, MM_E_MMSTRAND = 4, MM_E_MMCODE = 5,
    MM_E_MMEMPTY = 6, MM_E_MMMIXED = 7, MM_E_SKIPLEN = 8, MM_E_SKIPVAL = 9, MM_E_READPOS = 10,
    MM_E_MLIDX = 11, MM_E_NOCONTIG = 12, MM_E_REFPOS = 13, MM_E_QOVER = 14,
    /* library-level */
    MM_E_SIDEFULL = 32, MM_E_ARG = 33, MM_E_HIP = 34, MM_E_NOMEM = 35, MM_E_TOOMANY = 36, MM_E_NOCODE = 37,
    MM_E_OVERFLOW = 38   /* a counter passed 2^32 - 1 calls (the reference exits on n_called overflow, src/mod.c:900-904) */
};

typedef struct mm_freq mm_freq_t;

/* set-up: upload the reference, build per-base context words (kernel K0), allocate counter planes.
 * Returns NULL on failure with a message in err. */
mm_freq_t *mm_freq_create(const mm_freq_opts_t *opts, int32_t n_contigs, const mm_contig_t *contigs,
                          int32_t n_intervals, const mm_interval_t *intervals, char *err, size_t err_len);

/* Process one batch from HOST memory: H2D on an internal stream, then the hot-path kernels.
 * Asynchronous; the batch memory must stay valid until mm_freq_host_done(ticket) or mm_freq_wait(ticket) returns.
 * Returns a ticket >= 0 or -MM_E_*.
 *
 * Gathering (opts.coalesce > 1): process_db is called once per -K batch (src/minimod.c:344-350), and a -K batch is a
 * fraction of what fills an MI355X.  Consecutive host batches are therefore copied one behind the other into one staging
 * area in device memory (offsets of the read records rebased by a kernel) and launched TOGETHER, up to opts.coalesce of
 * them or opts.gather_mb MiB, as one batch: they return the SAME ticket, as with mm_freq_submit_device.  The launch is made
 * when the group is full, when something waits for the ticket, or when any call needs the counters.  A per-read error is
 * reported with the read's index counted from the group's first read (mm_freq_read_record returns its record). */
int32_t mm_freq_submit(mm_freq_t *h, const mm_batch_t *host_batch);
/* the host memory of every batch submitted under this ticket so far has been copied: the caller may reuse it (the
 * batches themselves may not have been launched yet).  Returns 0 or MM_E_*. */
int32_t mm_freq_host_done(mm_freq_t *h, int32_t ticket);
/* the record of read `index` of the ticket's (gathered) batch as the device holds it: tid, pos, l_qseq, ... as submitted,
 * pool offsets those of the staging area.  For error messages (the reference prints contig and position).  0 or MM_E_*. */
int32_t mm_freq_read_record(mm_freq_t *h, int32_t ticket, int32_t index, mm_read_t *out);
/* the ticket's (gathered) batch as the device holds it -- DEVICE pointers: the staging area's for mm_freq_submit, the caller's own for
 * mm_freq_submit_device*; valid until the ticket's slot is used again (four launches later).  What the device-side replay of the
 * reference's row order reads a launch's read records and MM text from (include/minimod_tie.h).  Launches a gathered group that is
 * still open.  0 or MM_E_*. */
int32_t mm_freq_ticket_batch(mm_freq_t *h, int32_t ticket, mm_batch_t *out);

/* Process one batch already RESIDENT in device memory (all mm_batch_t pointers are device pointers) on the given
 * HIP stream (hipStream_t as void*, NULL = the handle's own stream).  Returns a ticket >= 0 or -MM_E_*.  The batch
 * must stay resident until mm_freq_wait(ticket) has returned: reads the tile kernels do not cover (more than four
 * codes in an MM group, groups on different canonical bases) are processed when the host waits for the batch.
 *
 * Coalescing (opts.coalesce > 1, freq mode): a -K batch of 4096 reads fills a quarter of an MI355X, so consecutive
 * submits that are WINDOWS of one resident read set -- the same four pool pointers and sizes, `reads` continuing where
 * the previous submit's reads ended, no caller's plan, the same stream -- are gathered and launched together, up to
 * opts.coalesce of them, as one batch.  The gathered submits return the SAME ticket; the launch is made when the group
 * is full, when something waits for the ticket, or when any call needs the counters (finalize, reset, slabs, a submit
 * that does not continue the group).  A per-read error of a group is reported with the read's index counted from the
 * group's first read. */
int32_t mm_freq_submit_device(mm_freq_t *h, const mm_batch_t *dev_batch, void *hip_stream);
/* The same for a batch that is a launch by itself (a whole group of -K batches flattened on the device, include/minimod_ingest.h):
 * launched at once whatever opts.coalesce says, and sized by `bases` (the sum of its reads' l_qseq; 0 = not known) where
 * mm_freq_submit_device has to guess a resident window's bases from its read count. */
int32_t mm_freq_submit_device_now(mm_freq_t *h, const mm_batch_t *dev_batch, void *hip_stream, uint64_t bases);
/* submits that went into the ticket's launch (1 without coalescing) */
int32_t mm_freq_ticket_batches(mm_freq_t *h, int32_t ticket);

/* Wait for a ticket.  Returns 0, or the first failing read's MM_E_* code with its batch index in *bad_read. */
int32_t mm_freq_wait(mm_freq_t *h, int32_t ticket, int32_t *bad_read);

/* View mode: wait for a ticket, order the batch's rows on the device and hand them over -- as a host array
 * (mm_view_fetch) or left in device memory (mm_view_fetch_device).  Returns the row count, or -MM_E_* with the failing
 * read's batch index in *bad_read.  The rows stay valid until the ticket's slot is used by a later submit (four
 * submits later).  A batch submitted with mm_freq_submit_device must stay resident until it has been fetched: when a
 * batch produces more rows than the record buffer was sized for, fetch grows the buffer and runs the batch again. */
int64_t mm_view_fetch(mm_freq_t *h, int32_t ticket, const mm_view_row_t **rows, int32_t *bad_read);
int64_t mm_view_fetch_device(mm_freq_t *h, int32_t ticket, const void **dev_rows, int32_t *bad_read);

/* Plan a batch: writes work items (read index | part << 24 | (parts-1) << 28) for reads[0..n), long reads split into
 * up to 16 parts, costliest first.  Returns the number of items (<= cap) or -MM_E_ARG when cap is too small
 * (cap >= 16*n always suffices).  Pure host function. */
int32_t mm_freq_plan_batch(const mm_read_t *reads, int32_t n, int32_t *items, int32_t cap);

/* Intern a code string seen in reads (only meaningful with -c '*'): returns its code index. */
int32_t mm_freq_intern_code(mm_freq_t *h, const char *code);
int32_t mm_freq_n_codes(const mm_freq_t *h);
const char *mm_freq_code_name(const mm_freq_t *h, int32_t code);

/* Finalize: compact non-zero counters (kernel K2), fold in the side list, order rows by (contig name in strcmp
 * order, pos) like cmp_key_fast (src/mod.c:59-87) with ties in (strand, code, ins_offset, haplotype, '*' last)
 * order.  *rows is owned by the handle and valid until the next finalize/destroy.  Returns the row count or
 * -MM_E_*.  Counters are left intact (more batches may follow). */
int64_t mm_freq_finalize(mm_freq_t *h, const mm_row_t **rows);
/* The same rows, left WHERE THEY ARE MADE when every one of them comes from the dense counters (no haplotype planes, nothing on the
 * side lists -- a plain freq run): *device_rows is then the handle's array in GPU memory (in output order, valid until the next
 * finalize / destroy) and *rows is NULL -- a caller that formats on the device (mm_fmt_rows_device, include/minimod_tie.h) never
 * brings the rows to the host (print_freq_output's walk over the map, src/mod.c:644-728, becomes two kernels and no copy).
 * Otherwise *device_rows is NULL and *rows is what mm_freq_finalize returns.  Returns the row count or -MM_E_*. */
int64_t mm_freq_finalize_device(mm_freq_t *h, const mm_row_t **rows, const mm_row_t **device_rows);

/* Multi-GPU halo exchange: device pointer and element count (uint64 each) of the
 * counter slab covering [begin, begin+len) of an interval, plane-major: for plane, for strand: len words.
 * mm_freq_slab_export packs it into `dst` (device), mm_freq_slab_add adds a packed slab into the planes. */
int64_t mm_freq_slab_words(const mm_freq_t *h, int64_t len);
int32_t mm_freq_slab_export(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, void *dst_dev, void *hip_stream);
int32_t mm_freq_slab_add(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, const void *src_dev, void *hip_stream);
int32_t mm_freq_slab_clear(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, void *hip_stream);
/* the same through HOST memory (mm_freq_slab_words(h, len) words), for callers that have no device buffers of their own:
 * the workers of `minimod freq --devices` pass a slab from process to process */
int32_t mm_freq_slab_export_host(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, void *dst_host);
int32_t mm_freq_slab_add_host(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, const void *src_host);

/* ... and from GPU to GPU between two processes (round 4; north_star: "reduce of boundary counter slabs over xGMI"): export packs the
 * slab into a device buffer the handle keeps (until its next export or its end) and writes that buffer's HIP IPC handle -- 64 bytes
 * the caller passes to the other process by whatever means it has; add opens the handle there, copies device to device (a peer copy
 * when the two processes own different GPUs) and adds.  0, or -MM_E_HIP when this platform gives or takes no IPC handle (the callers
 * then fall back to the host-memory pair above). */
#define MM_IPC_HANDLE_BYTES 64
int32_t mm_freq_slab_export_ipc(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, void *handle_out);
int32_t mm_freq_slab_add_ipc(mm_freq_t *h, int32_t tid, int64_t begin, int64_t len, const void *handle);

/* Measurement hook (bench.py): device time of a ticket's hot-path kernels in milliseconds (HIP events recorded on the
 * launch stream around them). */
float mm_freq_last_kernel_ms(mm_freq_t *h, int32_t ticket);
/* Work tallies for the algorithmic-bytes figure: enable!=0 makes the kernels count reference-word
 * lookups, ML bytes read, dense counter updates and side-list updates; get copies and clears the four totals. */
int32_t mm_freq_stats_enable(mm_freq_t *h, int32_t enable);
int32_t mm_freq_stats_get(mm_freq_t *h, uint64_t out[16]);   /* [4..6]: reads k_stream_reads did itself / handed to the tile pipeline /
                                                                * to the fused kernel; [7..15]: phase time sums in diagnostic builds, else 0 */
int64_t mm_freq_device_bytes(const mm_freq_t *h);
/* how the submits so far were launched: [0] hot-path launches, [1] of which with k_stream_reads (the whole read in one
 * wavefront), [2] submits, [3] reads submitted.  (A gathered group not yet launched is not counted in [0] and [1].) */
int32_t mm_freq_launch_counts(const mm_freq_t *h, uint64_t out[4]);

void mm_freq_reset_counters(mm_freq_t *h);
void mm_freq_destroy(mm_freq_t *h);

const char *mm_strerror(int32_t code);
int32_t mm_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MINIMOD_HIP_H */
