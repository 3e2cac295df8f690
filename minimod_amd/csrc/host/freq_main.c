/* freq_main.c -- `minimod freq` and `minimod view` with the hot path on the MI355X library.  Same options, defaults,
 * progress lines, output and exit behaviour as the reference's freq_main (src/freq_main.c:46-64 option table, :166-519
 * driver) and view_main (src/view_main.c:46-63, :166-470), with load(N+1) overlapping process(N) like their 3-stage
 * pipeline (freq_main.c:404-474): the batch is handed to mm_freq_submit (H2D + kernels, asynchronous) and the next
 * batch is decoded meanwhile.  view prints a batch's rows when the batch is retired (print_view_output per db_t,
 * src/view_main.c:142-160); freq prints once at the end. */
#include <errno.h>
#include <getopt.h>
#include <pthread.h>
#include <spawn.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <time.h>
#include <sys/stat.h>
#include <unistd.h>

#include "minimod_bgzf.h"
#include "minimod_ingest.h"
#include "minimod_tie.h"
#include "mmhost.h"

static struct option long_options[] = {
    {"bedmethyl", no_argument, 0, 'b'},            /* 0 */
    {"mod_codes", required_argument, 0, 'c'},      /* 1 */
    {"mod_thresh", required_argument, 0, 'm'},     /* 2 */
    {"threads", required_argument, 0, 't'},        /* 3 */
    {"batchsize", required_argument, 0, 'K'},      /* 4 */
    {"max-bytes", required_argument, 0, 'B'},      /* 5 */
    {"verbose", required_argument, 0, 'v'},        /* 6 */
    {"help", no_argument, 0, 'h'},                 /* 7 */
    {"version", no_argument, 0, 'V'},              /* 8 */
    {"prog-interval", required_argument, 0, 'p'},  /* 9 */
    {"debug-break", required_argument, 0, 0},      /* 10 */
    {"output", required_argument, 0, 'o'},         /* 11 */
    {"insertions", no_argument, 0, 0},             /* 12 */
    {"haplotypes", no_argument, 0, 0},             /* 13 */
    {"allow-secondary", no_argument, 0, 0},        /* 14 */
    {"include-non-ref", no_argument, 0, 0},        /* 15 */
    {"skip-supplementary", no_argument, 0, 0},     /* 16 */
    {"device", required_argument, 0, 0},           /* 17 (new: HIP device ordinal) */
    {"devices", required_argument, 0, 0},          /* 18 (new: one worker process per listed GPU, the genome cut into shares) */
    {"canonical-order", no_argument, 0, 0},        /* 19 (new: rows that tie on (contig, start) in a fixed order instead of the reference's hash order) */
    {"gather", required_argument, 0, 0},           /* 20 (new: -K batches that may share one kernel launch) */
    {"gpu-inflate", no_argument, 0, 0},            /* 21 (new: BGZF blocks inflated on the device, next to the host pool) */
    {"no-gpu-inflate", no_argument, 0, 0},         /* 22 */
    {"gpu-ingest", no_argument, 0, 0},             /* 23 (new: the decoded BAM stays in GPU memory: inflate, record framing and load_db's flattening on the device) */
    {"no-gpu-ingest", no_argument, 0, 0},          /* 24 */
    {"region", required_argument, 0, 0},           /* 26 (new: chr:from-to, through the .bai -- the reference's own region code is commented out, src/minimod.c:92-130) */
    {"host-replay", no_argument, 0, 0},            /* 25 (new: minimod's row order replayed by the host's serial restatement instead of on the device) */
    {0, 0, 0, 0}};

/* view takes neither -b nor -m (src/view_main.c:46-63); long options are matched by name below */
static struct option view_long_options[] = {
    {"mod_codes", required_argument, 0, 'c'},
    {"threads", required_argument, 0, 't'},
    {"batchsize", required_argument, 0, 'K'},
    {"max-bytes", required_argument, 0, 'B'},
    {"verbose", required_argument, 0, 'v'},
    {"help", no_argument, 0, 'h'},
    {"version", no_argument, 0, 'V'},
    {"prog-interval", required_argument, 0, 'p'},
    {"debug-break", required_argument, 0, 0},
    {"output", required_argument, 0, 'o'},
    {"insertions", no_argument, 0, 0},
    {"haplotypes", no_argument, 0, 0},
    {"allow-secondary", no_argument, 0, 0},
    {"include-non-ref", no_argument, 0, 0},
    {"skip-supplementary", no_argument, 0, 0},
    {"device", required_argument, 0, 0},
    {"devices", required_argument, 0, 0},
    {"gpu-inflate", no_argument, 0, 0},
    {"no-gpu-inflate", no_argument, 0, 0},
    {"gather", required_argument, 0, 0},
    {"gpu-ingest", no_argument, 0, 0},
    {"no-gpu-ingest", no_argument, 0, 0},
    {0, 0, 0, 0}};

typedef struct {
    int32_t K; int64_t B; int threads, debug_break, bedmethyl, insertions, haplotypes, allow_secondary, skip_supplementary;
    int progress_interval, device, view, canonical_order, gather, gpu_inflate, gpu_ingest, host_replay;
    const char *codes, *threshes, *out_path, *devices, *region;
    int32_t region_tid; int64_t region_beg, region_end;   /* --region resolved against the BAM header: rows of [beg, end) on that contig are printed */
    FILE *out;
} fopt_t;

/* one worker of `--devices`: its share of the genome and where its rows go */
#define MMH_MAX_DEVICES 64
#define MMH_SHARE_HALO ((int64_t)1 << 18)     /* dense counters kept past a cut inside a contig; the calls of a read that reaches further (>256 kb behind
                                               * the cut: spliced or ultra-long) go to the side table, which takes any position */
#define MMH_SHARE_ALIGN ((int64_t)1 << 16)
typedef struct {
    int sharded, first, last, fd;          /* fd: pipe to the parent (totals, then sections or rows) */
    int slab_in, slab_out;                 /* pipes from the left / to the right neighbour: the halo slab behind a cut inside a contig */
    int rows_in, rows_out;                 /* ... and the rows that lie in a share further right (side rows past the halo) */
    int tied;                              /* rows can tie on (contig, start) and the reference's order is wanted: the parent orders and formats */
    char part_path[512];                   /* the worker's own formatted text */
    int n_iv;
    mm_interval_t iv[64];                  /* dense counters: the share's intervals (+ halo behind a cut) */
    int32_t lo_tid, hi_tid; int64_t lo_pos, hi_pos;
    uint64_t voffset;
} wspec_t;
typedef struct {                           /* what a worker reports besides its rows */
    int64_t n_rows;                        /* rows sent to the parent (tied runs), else 0 */
    int64_t n_sections;                    /* sections of its part file (one per contig it has rows on), sent as wsection_t */
    int64_t n_tie_keys;                    /* tied runs: keys of its first-insertion sequence, sent behind the rows */
    int64_t tie_put_after;                 /* ... and whether any put followed the sequence's last new key */
    uint64_t total_reads, total_bytes, processed_reads, processed_bytes, processed_bases;
    double load_time, wait_time, sort_time;
} wtotals_t;

typedef struct { int32_t tid, pad; int64_t off, len; } wsection_t;   /* bytes [off, off + len) of the part file: the rows of contig tid */

static void print_help(FILE *fp, const fopt_t *o) {
    fprintf(fp, "Usage: minimod %s ref.fa reads.bam\n", o->view ? "view" : "freq");
    fprintf(fp, "\nbasic options:\n");
    if (!o->view) fprintf(fp, "   -b                         output in bedMethyl format [%s]\n", o->bedmethyl ? "yes" : "not set");
    fprintf(fp, "   -c STR                     modification code(s) (eg. m, h or mh or as ChEBI) [%s]\n", o->codes ? o->codes : (o->view ? "m" : "(null)"));
    if (!o->view) fprintf(fp, "   -m FLOAT                   min modification threshold(s). Comma separated values for each modification code given in -c [%s]\n", o->threshes ? o->threshes : "(null)");
    fprintf(fp, "   -t INT                     number of BAM decoding threads [%d]\n", o->threads);
    fprintf(fp, "   -K INT                     batch size (max number of reads loaded at once) [%d]\n", o->K);
    fprintf(fp, "   -B FLOAT[K/M/G]            max number of bases loaded at once [%.1fM]\n", o->B / (float)(1000 * 1000));
    fprintf(fp, "   -h                         help\n");
    fprintf(fp, "   -p INT                     print progress every INT seconds (0: per batch) [%d]\n", o->progress_interval);
    fprintf(fp, "   -o FILE                    output file [%s]\n", o->out_path == NULL ? "stdout" : o->out_path);
    fprintf(fp, "   --insertions               output modifications in insertions [%s]\n", o->insertions ? "yes" : "no");
    fprintf(fp, "   --haplotypes               output haplotypes [%s]\n", o->haplotypes ? "yes" : "no");
    fprintf(fp, "   --verbose INT              verbosity level [%d]\n", mmh_log_level);
    fprintf(fp, "   --version                  print version\n");
    fprintf(fp, "   --allow-secondary          allow secondary alignments [%s]\n", o->allow_secondary ? "yes" : "no");
    fprintf(fp, "   --skip-supplementary       skip supplementary alignments [%s]\n", o->skip_supplementary ? "yes" : "no");
    fprintf(fp, "\nadvanced options:\n");
    fprintf(fp, "   --debug-break INT          break after processing the specified no. of batches\n");
    fprintf(fp, "   --device INT               GPU to use [%d]\n", o->device);
    if (!o->view) fprintf(fp, "   --canonical-order          rows of one (contig, start) by strand, code, ins_offset, haplotype instead of the order\n"
                              "                              minimod's hash table leaves them in (skips the replay of that table) [%s]\n", o->canonical_order ? "yes" : "no");
    fprintf(fp, "   --gather INT               -K batches that may share one kernel launch at most (they are staged in GPU memory one behind the\n"
                "                              other, 1 GiB of them, and processed together; 1: every batch is its own launch) [%d]\n", o->gather);
    fprintf(fp, "   --gpu-inflate              inflate the BAM's BGZF blocks on the GPU as well (groups of 1024 blocks per launch, next to the\n"
                "   --no-gpu-inflate           -t host threads; blocks the device refuses are the host decoder's) [%s]\n",
            o->gpu_inflate < 0 ? "for a BAM file of 4 GiB or more per GPU" : (o->gpu_inflate ? "yes" : "no"));
    if (!o->view) fprintf(fp, "   --host-replay              replay minimod's row order (rows that tie on contig and start) with the host's serial restatement of its\n"
                              "                              hash table and sort instead of the device's parallel one (the checker; reads with the host threads) [%s]\n", o->host_replay ? "yes" : "no");
    fprintf(fp, "   --gpu-ingest               keep the decoded BAM in GPU memory: BGZF inflate, record framing and the read filters all run on the\n"
                "   --no-gpu-ingest            device and the host only moves compressed bytes (-K / -B then do not cut the batches; runs that\n"
                "                              --host-replay or --debug-break, and pipes, read with the host threads) [%s]\n",
            o->gpu_ingest < 0 ? (o->view ? "for a BAM file of 2 GiB or more per GPU" : "for a BAM file of 512 MiB or more per GPU") : (o->gpu_ingest ? "yes" : "no"));
    if (!o->view) fprintf(fp, "   --region STR               only the rows of chr:from-to (1-based, inclusive; chr alone: the whole contig): the reads that can reach it are\n"
                              "                              found through reads.bam.bai and counted, rows outside are dropped [%s]\n", o->region ? o->region : "whole file");
    fprintf(fp, "   --devices LIST             GPUs to share the genome between, e.g. 0,1,2,3 (one worker process each; needs reads.bam.bai)\n");
}

/* --gpu-inflate: the device inflater of include/minimod_bgzf.h behind the reader's backend table (bamio.h) */
static uint8_t *bz_staging(void *ctx, int slot) { return mm_bgzf_staging((mm_bgzf_t *)ctx, slot); }
static void *bz_blocks(void *ctx, int slot) { return mm_bgzf_blocks((mm_bgzf_t *)ctx, slot); }
static int bz_submit(void *ctx, int slot, int n, size_t cb, size_t ob, uint8_t *out) { return mm_bgzf_submit((mm_bgzf_t *)ctx, slot, n, cb, ob, out); }
static int bz_wait(void *ctx, int slot, const int32_t **st) { return mm_bgzf_wait((mm_bgzf_t *)ctx, slot, st); }
static mm_bgzf_backend_t bz_backend;
#define BZ_SLOTS 4
static mm_bgzf_t *bz_create(int device) {   /* (any thread: nothing here is seen by the readers yet) */
    char err[256];
    mm_bgzf_t *bz = mm_bgzf_create(device, BZ_SLOTS, 1024, ((size_t)41 << 20), ((size_t)64 << 20), err, sizeof err);
    if (!bz) MMH_WARNING("--gpu-inflate: %s; the host threads inflate alone", err);
    return bz;
}
static void bz_attach(mm_bgzf_t *bz) {      /* before the readers are opened */
    if (!bz) return;
    bz_backend.ctx = bz; bz_backend.slots = BZ_SLOTS; bz_backend.max_blocks = 1024;
    bz_backend.max_cbytes = (size_t)41 << 20; bz_backend.max_obytes = (size_t)64 << 20;
    bz_backend.host_alloc = mm_bgzf_host_alloc; bz_backend.host_free = mm_bgzf_host_free;
    bz_backend.staging = bz_staging; bz_backend.blocks = bz_blocks; bz_backend.submit = bz_submit; bz_backend.wait = bz_wait;
    mm_bam_set_backend(&bz_backend);
    mmh_loader_set_allocator(mm_bgzf_host_alloc, mm_bgzf_host_free);   /* (the batches leave by DMA as well) */
}
typedef struct { int device; mm_bgzf_t *bz; } bz_job_t;
static void *bz_create_main(void *arg) { bz_job_t *j = (bz_job_t *)arg; j->bz = bz_create(j->device); return NULL; }
static void bz_report(mm_bgzf_t *bz) {
    if (!bz) return;
    unsigned long long st[3];
    mm_bam_backend_stats(st);
    fprintf(stderr, "[gpu-inflate] %llu groups (%llu blocks) inflated on the device, %llu blocks again on the host\n", st[0], st[1], st[2]);
}
static void bz_stop(mm_bgzf_t *bz) {
    if (!bz) return;
    bz_report(bz);
    mm_bam_set_backend(NULL);
    mmh_loader_set_allocator(NULL, NULL);
    mm_bgzf_destroy(bz);
}

static void *hip_warm_main(void *arg) { (void)mm_hip_warm(*(int *)arg); return NULL; }
/* --gpu-ingest: the device loader's buffers (pinned staging, device memory) are made beside the freq handle's set-up */
typedef struct { const char *path; mm_pool_t *pool; mmh_devloader_opts_t o; mmh_devloader_t *dl; char err[256]; } dl_job_t;
static void *dl_open_main(void *arg) { dl_job_t *j = (dl_job_t *)arg; j->dl = mmh_devloader_open(j->path, j->pool, &j->o, j->err, sizeof j->err); return NULL; }
static void close_loaders(mmh_loader_t *ld, mmh_devloader_t *dl, mm_pool_t *own_pool) {
    if (ld) mmh_loader_close(ld);
    if (dl) mmh_devloader_close(dl);
    if (own_pool) mm_pool_destroy(own_pool);
}

/* the reference's message for a per-read device status (src/mod.c line in brackets), then exit(1) like it does */
static void die_read_record(int code, int32_t read, const mm_read_t *rd, const mm_bam_hdr_t *hdr);
/* `read`: the read's index in its -K batch (what the reference prints); rd: its record, or NULL */
static void die_read_record(int code, int32_t read, const mm_read_t *rd, const mm_bam_hdr_t *hdr) {
    (void)mmh_emit_flush();   /* rows of earlier batches reach the output as they did with stdio */
    const char *tname = (rd && rd->tid >= 0 && rd->tid < hdr->n_targets) ? hdr->target_name[rd->tid] : "*";
    switch (code) {
        case MM_E_HARDCLIP:   /* :843 */
            MMH_ERROR("Hard clipping found in read %d of the batch (contig %s, pos %d) and they are not supported.\nTry following workarounds.\n\t01. Filter out non-primary alignments\n\t\tsamtools view -h -F 2308 reads.bam -o primary_reads.bam\n\t02. Use minimap2 with -Y to use soft clipping for suplimentary alignments.\n", read, tname, rd ? rd->pos : -1);
            break;
        case MM_E_CIGAROP: MMH_ERROR("Unhandled CIGAR OPT in read %d (contig %s)\n", read, tname); break;                     /* :846 */
        case MM_E_MMBASE: MMH_ERROR("Assertion failed. Invalid base in the MM tag of read %d", read); break;                  /* :1005 */
        case MM_E_MMSTRAND: MMH_ERROR("Assertion failed. Invalid strand in the MM tag of read %d", read); break;              /* :1012 */
        case MM_E_MMCODE: MMH_ERROR("Invalid base modification code in read %d. Modification codes should be either numeric or alphabetic.\n", read); break; /* :1030 */
        case MM_E_MMEMPTY: MMH_ERROR("Assertion failed. Invalid modification codes in read %d. Modification codes cannot be empty.", read); break;       /* :1053 */
        case MM_E_MMMIXED: MMH_ERROR("Assertion failed. Invalid modification codes in read %d. Modification codes should be either numeric or alphabetic, not both.", read); break; /* :1054 */
        case MM_E_SKIPLEN: MMH_ERROR("Assertion failed. Skip count longer than 9 characters in read %d", read); break;        /* :1080 */
        case MM_E_SKIPVAL: MMH_ERROR("Assertion failed. Invalid skip count in read %d", read); break;                         /* :1083-1085 */
        case MM_E_READPOS: MMH_ERROR("Assertion failed. Read pos cannot exceed seq len. read %d seq_len: %u", read, rd ? rd->l_qseq : 0); break; /* :1116 */
        case MM_E_MLIDX: MMH_ERROR("Assertion failed. read %d mod prob index mismatch. ml_len:%u", read, rd ? rd->ml_len : 0); break;            /* :1174 */
        case MM_E_NOCONTIG: MMH_ERROR("Assertion failed. Contig %s not found in reference provided", tname); break;           /* :793 */
        case MM_E_REFPOS: MMH_ERROR("Assertion failed. ref_pos outside contig %s (read %d)", tname, read); break;             /* :860 */
        case MM_E_QOVER: MMH_ERROR("Assertion failed. read_pos exceeds seq_len in read %d", read); break;                     /* :853 */
        default: MMH_ERROR("GPU path failed: %s (read %d)", mm_strerror(code), read); break;
    }
    fprintf(stderr, "Exiting.\n");
    exit(EXIT_FAILURE);
}

/* -c '*': every code string reads carry must have an index before the kernel runs (src/mod.c:1146-1160) */
static void intern_batch_codes(mm_freq_t *h, const mm_batch_t *b) {
    for (int32_t i = 0; i < b->n_reads; i++) {
        const char *mm = (const char *)b->mm + b->reads[i].mm_off;
        size_t n = b->reads[i].mm_len, p = 0;
        while (p < n) {
            size_t s = p + 2, e = s;
            while (e < n && mm[e] != ',' && mm[e] != ';' && mm[e] != '?' && mm[e] != '.') e++;
            if (e > s && e - s < MM_CODE_LEN) {
                char code[MM_CODE_LEN];
                memcpy(code, mm + s, e - s); code[e - s] = 0;
                if (code[0] >= '0' && code[0] <= '9') (void)mm_freq_intern_code(h, code);
                else for (size_t m = 0; m < e - s; m++) (void)mm_freq_intern_code(h, code + m);
            }
            while (p < n && mm[p] != ';') p++;
            p++;
        }
    }
}

static int code_names(mm_freq_t *h, const char **codes) {
    int n = mm_freq_n_codes(h);
    if (n > MM_MAX_CODES) n = MM_MAX_CODES;
    for (int i = 0; i < n; i++) codes[i] = mm_freq_code_name(h, i);
    return n;
}

/* print_freq_output's rows as text (src/mod.c:666-719).  Round 5: outputs of 65 536 rows and more are formatted ON THE DEVICE
 * (include/minimod_tie.h mm_fmt_rows: every "%d" and the "%f" by integer arithmetic in a kernel, a piece of two million rows at a time;
 * the host moves rows in and text out and writes it); smaller ones by the worker pool (emit.c), which needs no launch.  MINIMOD_FMT=device
 * / host decides it by hand (the tests run both on the reference's goldens: the bytes are the same). */
static double fmt_device_ms; static int64_t fmt_device_rows;
typedef struct { FILE *fp; const char *p; size_t n; int err; } fmt_wjob_t;
static void *fmt_write_main(void *arg) { fmt_wjob_t *j = (fmt_wjob_t *)arg; if (j->n && fwrite(j->p, 1, j->n, j->fp) != j->n) j->err = 1; return NULL; }
/* (drows: the same rows in GPU memory, when they never came to the host -- then the device formats, whatever their number) */
static void print_freq_rows_any(FILE *fp, mm_pool_t *pool, const mm_row_t *rows, const mm_row_t *drows, int64_t n, const mm_bam_hdr_t *hdr, const char *const *codes, int n_codes,
                                int bedmethyl, int insertions, int haplotypes, int device) {
    const char *e = getenv("MINIMOD_FMT");
    const int use = drows ? 1 : (e ? strcmp(e, "device") == 0 : n >= 65536);
    if (use && n > 0) {
        static mm_fmt_t *f; static int f_key = -1;
        const int key = (bedmethyl ? 1 : 0) | (insertions ? 2 : 0) | (haplotypes ? 4 : 0) | (n_codes << 3);
        if (f && f_key != key) { mm_fmt_destroy(f); f = NULL; }
        if (!f) {
            mm_fmt_opts_t fo;
            memset(&fo, 0, sizeof fo);
            fo.abi_version = MM_TIE_ABI_VERSION; fo.device = device; fo.bedmethyl = bedmethyl; fo.insertions = insertions; fo.haplotypes = haplotypes;
            fo.n_contigs = hdr->n_targets; fo.n_codes = n_codes;
            char ferr[256];
            const double tc = mmh_realtime();
            f = mm_fmt_create(&fo, (const char *const *)hdr->target_name, codes, ferr, sizeof ferr);
            f_key = key;
            if (getenv("MM_TIMELINE")) fprintf(stderr, "[timeline] output: mm_fmt_create %.3f s\n", mmh_realtime() - tc);
            if (!f) MMH_WARNING("the device-side row formatter could not be set up (%s): the host threads format", ferr);
        }
        if (f) {
            if (mmh_emit_flush() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
            /* a piece's text is written by a thread of its own while the device makes the next piece's (the handle's two buffers take turns) */
            /* (pieces of 128 k rows, ~9 MB of text: the copy out of the device, the write of the piece before and the kernels of the piece behind
             * overlap from the second piece on -- with pieces of a million rows an output of 770 k rows was one piece and nothing overlapped) */
            const char *pe = getenv("MM_FMT_PIECE");
            const int64_t piece = pe && atoll(pe) > 0 ? atoll(pe) : (int64_t)1 << 17;
            const int tl = getenv("MM_TIMELINE") != NULL;
            double t_fmt = 0, t_join = 0;
            int ok = 1, writing = 0;
            pthread_t wt;
            fmt_wjob_t job;
            memset(&job, 0, sizeof job);
            for (int64_t i = 0; i < n && ok; i += piece) {
                const int64_t m = n - i < piece ? n - i : piece;
                const char *text = NULL;
                const double tf = mmh_realtime();
                const int64_t nb = drows ? mm_fmt_rows_device(f, drows + i, m, &text) : mm_fmt_rows(f, rows + i, m, &text);
                t_fmt += mmh_realtime() - tf;
                if (nb < 0) { ok = 0; if (i == 0) break; MMH_ERROR("the device-side row formatter failed: %s", mm_strerror((int32_t)nb)); exit(EXIT_FAILURE); }
                const double tj = mmh_realtime();
                if (writing) { pthread_join(wt, NULL); writing = 0; if (job.err) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); } }
                t_join += mmh_realtime() - tj;
                job.fp = fp; job.p = text; job.n = (size_t)nb; job.err = 0;
                if (i + piece < n && pthread_create(&wt, NULL, fmt_write_main, &job) == 0) writing = 1;
                else { (void)fmt_write_main(&job); if (job.err) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); } }
                fmt_device_ms += mm_fmt_last_kernel_ms(f); fmt_device_rows += m;
            }
            if (writing) { pthread_join(wt, NULL); if (job.err) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); } }
            if (tl) fprintf(stderr, "[timeline] output: %ld rows in pieces of %ld: mm_fmt_rows %.3f s, waiting for the writer %.3f s\n", (long)n, (long)piece, t_fmt, t_join);
            if (ok) return;
            MMH_WARNING("%s", "the device-side row formatter failed on its first piece: the host threads format");
        }
    }
    if (!rows && n > 0) { MMH_ERROR("%s", "the device-side row formatter could not be used and the rows are not in host memory"); exit(EXIT_FAILURE); }
    mmh_print_freq_rows(fp, pool, rows, n, hdr, codes, n_codes, bedmethyl, insertions, haplotypes);
}

static int write_all(int fd, const void *buf, size_t n) {
    const char *p = (const char *)buf;
    while (n) {
        ssize_t k = write(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n);
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        p += k; n -= (size_t)k;
    }
    return 0;
}
static int read_all(int fd, void *buf, size_t n) {
    char *p = (char *)buf;
    while (n) {
        ssize_t k = read(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n);
        if (k < 0) { if (errno == EINTR) continue; return -1; }
        if (k == 0) return -1;
        p += k; n -= (size_t)k;
    }
    return 0;
}

/* freq: the -K batches that share a ticket (mm_freq_submit gathers up to opts.coalesce of them into one launch) */
#define MMH_MAX_GATHER 2048
/* view (and the replay handle of a tied freq run): what the rows of a gathered launch are printed from -- the read records and the
 * read names of every batch that went into it, one behind the other (a row names its read by its index in the launch) */
typedef struct { mm_read_t *reads; size_t n, cap; char *names; size_t names_len, names_cap; uint64_t *name_off; size_t off_cap;
                 uint8_t *mm; size_t mm_len, mm_cap; } journal_t;   /* (mm: the replay's journal keeps the MM text instead of the names) */
typedef struct { int32_t ticket, n; int32_t n_reads[MMH_MAX_GATHER]; journal_t j;
                 int64_t dev_before; /* >= 0: a batch of the device-side reader -- accepted reads of the file in front of it */ } group_t;
static void journal_reset(journal_t *j) { j->n = 0; j->names_len = 0; j->mm_len = 0; }
static void journal_free(journal_t *j) { free(j->reads); free(j->names); free(j->name_off); free(j->mm); memset(j, 0, sizeof *j); }
/* the replay's journal: read records with their MM text (tieorder.c looks at a read's group headers), offsets moved behind the text so far */
static int journal_add_mm(journal_t *j, const mm_batch_t *b) {
    const size_t n = (size_t)b->n_reads;
    if (j->n + n > j->cap) {
        size_t nc = j->cap ? j->cap : 8192;
        while (nc < j->n + n) nc *= 2;
        mm_read_t *r = (mm_read_t *)realloc(j->reads, nc * sizeof(mm_read_t));
        if (!r) return -1;
        j->reads = r; j->cap = nc;
    }
    if (j->mm_len + b->n_mm_bytes > j->mm_cap) {
        size_t nc = j->mm_cap ? j->mm_cap : ((size_t)1 << 22);
        while (nc < j->mm_len + b->n_mm_bytes) nc *= 2;
        uint8_t *g = (uint8_t *)realloc(j->mm, nc);
        if (!g) return -1;
        j->mm = g; j->mm_cap = nc;
    }
    memcpy(j->mm + j->mm_len, b->mm, b->n_mm_bytes);
    memcpy(j->reads + j->n, b->reads, n * sizeof(mm_read_t));
    for (size_t i = 0; i < n; i++) j->reads[j->n + i].mm_off += j->mm_len;
    j->mm_len += b->n_mm_bytes;
    j->n += n;
    return 0;
}
static int journal_add(journal_t *j, const mm_batch_t *b, const mmh_loader_t *ld, int pool_set) {
    const size_t n = (size_t)b->n_reads;
    if (j->n + n > j->cap) {
        size_t nc = j->cap ? j->cap : 8192;
        while (nc < j->n + n) nc *= 2;
        mm_read_t *r = (mm_read_t *)realloc(j->reads, nc * sizeof(mm_read_t));
        uint64_t *o = (uint64_t *)realloc(j->name_off, nc * sizeof(uint64_t));
        if (r) j->reads = r;
        if (o) j->name_off = o;
        if (!r || !o) return -1;
        j->cap = nc;
    }
    memcpy(j->reads + j->n, b->reads, n * sizeof(mm_read_t));
    for (size_t i = 0; i < n; i++) {
        const char *q = mmh_loader_qname(ld, pool_set, (int32_t)i);
        const size_t l = strlen(q) + 1;
        if (j->names_len + l > j->names_cap) {
            size_t nc = j->names_cap ? j->names_cap * 2 : ((size_t)1 << 20);
            while (nc < j->names_len + l) nc *= 2;
            char *g = (char *)realloc(j->names, nc);
            if (!g) return -1;
            j->names = g; j->names_cap = nc;
        }
        memcpy(j->names + j->names_len, q, l);
        j->name_off[j->n + i] = j->names_len;
        j->names_len += l;
    }
    j->n += n;
    return 0;
}

/* the same for a batch of the device-side reader: its read records and names come over from GPU memory (the names were kept for this:
 * mmh_devloader_opts_t.names) */
static int journal_from_device(journal_t *j, mmh_devloader_t *dl, const mmh_devbatch_t *db) {
    const size_t n = (size_t)db->batch.n_reads;
    journal_reset(j);
    if (n > j->cap) {
        size_t nc = j->cap ? j->cap : 8192;
        while (nc < n) nc *= 2;
        mm_read_t *r = (mm_read_t *)realloc(j->reads, nc * sizeof(mm_read_t));
        uint64_t *o = (uint64_t *)realloc(j->name_off, nc * sizeof(uint64_t));
        if (r) j->reads = r;
        if (o) j->name_off = o;
        if (!r || !o) return -1;
        j->cap = nc;
    }
    if (db->names_bytes > j->names_cap) {
        size_t nc = j->names_cap ? j->names_cap : ((size_t)1 << 20);
        while (nc < db->names_bytes) nc *= 2;
        char *g = (char *)realloc(j->names, nc);
        if (!g) return -1;
        j->names = g; j->names_cap = nc;
    }
    if (!db->names || !db->name_off) return -1;
    if (mmh_devloader_fetch(dl, j->reads, db->batch.reads, n * sizeof(mm_read_t)) != 0 || mmh_devloader_fetch(dl, j->name_off, db->name_off, n * sizeof(uint64_t)) != 0 ||
        mmh_devloader_fetch(dl, j->names, db->names, (size_t)db->names_bytes) != 0) return -1;
    j->n = n; j->names_len = (size_t)db->names_bytes;
    return 0;
}

/* (a read that fails in a batch of the device-side reader is named by where the host reader's -K / -B batches would have it: below) */
static struct { const void *o; const char *bam; const void *ws; } err_ctx;
static int32_t batch_index_in_file_ctx(uint64_t ordinal, int32_t fallback);

/* wait for a group's launch; a failing read is named by its index in its own -K batch, like the reference does */
static void retire_group(mm_freq_t *h, group_t *g, const mm_bam_hdr_t *hdr, double *wait_time) {
    if (g->ticket < 0) return;
    double tw = mmh_realtime();
    int32_t bad = -1;
    int e = mm_freq_wait(h, g->ticket, &bad);
    *wait_time += mmh_realtime() - tw;
    if (e) {
        mm_read_t rec;
        const int have = bad >= 0 && mm_freq_read_record(h, g->ticket, bad, &rec) == 0;
        int32_t in_batch = bad;
        for (int m = 0; m < g->n && in_batch >= g->n_reads[m]; m++) in_batch -= g->n_reads[m];
        die_read_record(e, in_batch, have ? &rec : NULL, hdr);
    }
    g->ticket = -1; g->n = 0;
}

/* view: wait for a group's launch, order its rows on the device, print them (print_view_output per db_t, src/mod.c:560-626: the
 * batches' rows one after the other are the launch's rows in read order) */
static void retire_view_group(mm_freq_t *h, group_t *g, const mm_bam_hdr_t *hdr, const fopt_t *o, mm_pool_t *pool, double *wait_time, double *output_time) {
    if (g->ticket < 0) return;
    double tw = mmh_realtime();
    int32_t bad = -1;
    const mm_view_row_t *rows = NULL;
    int64_t n = mm_view_fetch(h, g->ticket, &rows, &bad);
    *wait_time += mmh_realtime() - tw;
    if (n < 0) {
        int32_t in_batch = bad;
        if (g->dev_before >= 0) in_batch = batch_index_in_file_ctx((uint64_t)g->dev_before + (uint64_t)(bad > 0 ? bad : 0), bad);
        else for (int m = 0; m < g->n && in_batch >= g->n_reads[m]; m++) in_batch -= g->n_reads[m];
        die_read_record((int)-n, in_batch, (bad >= 0 && (size_t)bad < g->j.n) ? &g->j.reads[bad] : NULL, hdr);
    }
    double to = mmh_realtime();
    const char *codes[MM_MAX_CODES];
    int n_codes = code_names(h, codes);
    mmh_print_view_rows_of(o->out, pool, rows, n, g->j.reads, g->j.name_off, g->j.names, hdr, codes, n_codes, o->insertions, o->haplotypes);
    *output_time += mmh_realtime() - to;
    g->ticket = -1; g->n = 0; journal_reset(&g->j);
}

static double replay_fetch_seconds;   /* of the replay's seconds: waiting for the second handle's launches and their rows */
/* a replay run's second handle: a launch's calls (view rows with group ordinals) go into the tie-order replay, from the group's
 * journal (the records and MM text of the batches that went into the launch) */
static void replay_group(mm_freq_t *hv, mmh_tie_t *tie, group_t *g, const mm_bam_hdr_t *hdr, mm_pool_t *pool,
                         const uint8_t *const *klass_of_code, double *seconds) {
    if (g->ticket < 0) return;
    double t0 = mmh_realtime();
    int32_t bad = -1;
    const mm_view_row_t *rows = NULL;
    int64_t n = mm_view_fetch(hv, g->ticket, &rows, &bad);
    replay_fetch_seconds += mmh_realtime() - t0;
    if (n < 0) {
        int32_t in_batch = bad;
        for (int m = 0; m < g->n && in_batch >= g->n_reads[m]; m++) in_batch -= g->n_reads[m];
        die_read_record((int)-n, in_batch, (bad >= 0 && (size_t)bad < g->j.n) ? &g->j.reads[bad] : NULL, hdr);
    }
    const char *codes[MM_MAX_CODES];
    int n_codes = code_names(hv, codes);
    mm_batch_t jb;
    memset(&jb, 0, sizeof jb);
    jb.reads = g->j.reads; jb.mm = g->j.mm; jb.n_reads = (int32_t)g->j.n; jb.n_mm_bytes = g->j.mm_len;
    (void)mmh_tie_add_batch(tie, pool, &jb, rows, n, klass_of_code, codes, n_codes);   /* a failed replay is reported once, at the end */
    *seconds += mmh_realtime() - t0;
    g->ticket = -1; g->n = 0; journal_reset(&g->j);
}


/* The same on the device (round 5): the launch's rows stay in GPU memory (mm_view_fetch_device), the batch is the one the second handle
 * launched (mm_freq_ticket_batch: the staging area's for host batches, the arena's for the device reader's), and k_tie_reads stamps
 * every key (include/minimod_tie.h).  A replay that gives up is reported once, at the end. */
static void tie_codes(mm_tie_t *dtie, mm_freq_t *hv, const uint8_t *const *klass_of_code, int *have) {
    const char *codes[MM_MAX_CODES];
    const int n_codes = code_names(hv, codes);
    if (n_codes == *have) return;
    (void)mm_tie_set_codes(dtie, n_codes, codes, klass_of_code);
    *have = n_codes;
}
/* Where the device-side replay of the row order gives up (a haplotype tag above 61, a read with 2^18 calls, a launch it could not take: mm_tie_failed), the
 * run is handed to the host's replay, which has none of those limits (tieorder.c restates src/khash.h and src/ksort.h serially): the SAME command line with
 * --host-replay, as a child process -- nothing has been printed yet, a freq run prints when its rows are ordered -- and this process leaves with the child's
 * exit code.  The canonical order is printed only when it is asked for (--canonical-order): a run never prints other bytes than the reference's by itself. */
static int g_run_argc = 0;
static char **g_run_argv = NULL;   /* run_main's arguments: "freq", options, reference, BAM */
static int g_is_worker = 0;        /* a --devices worker: the parent starts the run again, the worker only says that it gave up */
static int g_dev_replay_gave_up = 0;
extern char **environ;
static void rerun_with_host_replay(const char *why, unsigned bits) {
    MMH_WARNING("the device-side replay of minimod's row order gave up (%s, reason bits 0x%x): the run starts again with --host-replay", why, bits);
    fflush(stderr);
    char **av = (char **)calloc((size_t)g_run_argc + 4, sizeof(char *));
    if (!av || g_run_argc < 1) { MMH_ERROR("%s", "Could not start the run again"); _exit(EXIT_FAILURE); }
    int n = 0;
    av[n++] = (char *)"minimod"; av[n++] = g_run_argv[0]; av[n++] = (char *)"--host-replay";
    for (int i = 1; i < g_run_argc; i++) av[n++] = g_run_argv[i];
    av[n] = NULL;
    pid_t pid = 0;
    if (posix_spawn(&pid, "/proc/self/exe", NULL, NULL, av, environ) != 0) { MMH_ERROR("%s", "Could not start the run again (posix_spawn)"); _exit(EXIT_FAILURE); }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) { }
    _exit(WIFEXITED(status) ? WEXITSTATUS(status) : EXIT_FAILURE);   /* (_exit: what this process buffered for its own output is not written behind the child's) */
}
static void dev_replay_failed(const char *why, unsigned bits) {
    if (g_is_worker) { if (!g_dev_replay_gave_up) MMH_WARNING("the device-side replay gave up in a worker (%s, reason bits 0x%x)", why, bits); g_dev_replay_gave_up = 1; return; }
    rerun_with_host_replay(why, bits);
}
static void replay_ticket_dev(mm_freq_t *hv, mm_tie_t *dtie, int32_t ticket, const int32_t *n_reads, int n_batches, const mm_bam_hdr_t *hdr,
                              const uint8_t *const *klass_of_code, int *have_codes, double *seconds) {
    if (ticket < 0) return;
    double t0 = mmh_realtime();
    int32_t bad = -1;
    const void *rows = NULL;
    int64_t n = mm_view_fetch_device(hv, ticket, &rows, &bad);
    replay_fetch_seconds += mmh_realtime() - t0;
    if (n < 0) {
        mm_read_t rec;
        const int have = bad >= 0 && mm_freq_read_record(hv, ticket, bad, &rec) == 0;
        int32_t in_batch = bad;
        for (int m = 0; m < n_batches && in_batch >= n_reads[m]; m++) in_batch -= n_reads[m];
        die_read_record((int)-n, in_batch, have ? &rec : NULL, hdr);
    }
    mm_batch_t db;
    if (g_dev_replay_gave_up) { *seconds += mmh_realtime() - t0; return; }
    if (mm_freq_ticket_batch(hv, ticket, &db) != 0) dev_replay_failed("a launch's batch is gone", mm_tie_failed(dtie));
    else {
        tie_codes(dtie, hv, klass_of_code, have_codes);
        const int32_t rc = mm_tie_add_launch(dtie, &db, rows, n, NULL);   /* (a launch it does not take marks the replay failed: its keys' stamps would be missing) */
        if (rc != 0) dev_replay_failed(mm_strerror(-rc), mm_tie_failed(dtie));
    }
    *seconds += mmh_realtime() - t0;
}
static void replay_group_dev(mm_freq_t *hv, mm_tie_t *dtie, group_t *g, const mm_bam_hdr_t *hdr, const uint8_t *const *klass_of_code, int *have_codes, double *seconds) {
    if (g->ticket < 0) return;
    replay_ticket_dev(hv, dtie, g->ticket, g->n_reads, g->n, hdr, klass_of_code, have_codes, seconds);
    g->ticket = -1; g->n = 0; journal_reset(&g->j);
}

/* ---- workers of `--devices`: rows of other shares, sections of formatted text ---- */
static int row_key_cmp(const mm_row_t *a, const mm_row_t *b);
typedef struct { const int *rank; } rowcmp_t;
static int row_full_cmp(const rowcmp_t *c, const mm_row_t *a, const mm_row_t *b) {   /* mm_freq_finalize's order: contig by name, then the key */
    if (a->tid != b->tid) return c->rank[a->tid] < c->rank[b->tid] ? -1 : 1;
    return row_key_cmp(a, b);
}
/* rank of every contig name in strcmp order (cmp_key_fast, src/mod.c:59-76); header order among equal names */
static int *contig_ranks(const mm_bam_hdr_t *hdr) {
    const int nt = hdr->n_targets;
    int *order = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1)), *rank = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    for (int t = 0; t < nt; t++) order[t] = t;
    for (int a = 1; a < nt; a++) {
        int x = order[a], b = a - 1;
        while (b >= 0 && strcmp(hdr->target_name[order[b]], hdr->target_name[x]) > 0) { order[b + 1] = order[b]; b--; }
        order[b + 1] = x;
    }
    for (int r = 0; r < nt; r++) rank[order[r]] = r;
    free(order);
    return rank;
}
/* two ordered row arrays as one (equal keys added up); returns a malloc'd array, *n_out rows */
static mm_row_t *merge_rows(const rowcmp_t *c, const mm_row_t *a, int64_t na, const mm_row_t *b, int64_t nb, int64_t *n_out) {
    mm_row_t *out = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(na + nb > 0 ? na + nb : 1));
    int64_t i = 0, j = 0, w = 0;
    while (out && (i < na || j < nb)) {
        const int k = i >= na ? 1 : (j >= nb ? -1 : row_full_cmp(c, &a[i], &b[j]));
        if (k == 0) { mm_row_t m = a[i++]; m.n_called += b[j].n_called; m.n_mod += b[j].n_mod; j++; out[w++] = m; }
        else if (k < 0) out[w++] = a[i++];
        else out[w++] = b[j++];
    }
    *n_out = w;
    return out;
}
/* is (tid, pos) in front of the share's end?  (shares are contiguous in header order) */
static int before_hi(const wspec_t *ws, int32_t tid, int32_t pos) { return ws->last || tid < ws->hi_tid || (tid == ws->hi_tid && (int64_t)pos < ws->hi_pos); }

/* Everything behind option parsing and the reference load: the batches of one BAM (or of one share of it) through one
 * GPU.  A single run prints its rows; a worker of `--devices` sends them to the parent, which merges and prints. */
static mmh_loader_t *open_loader(const fopt_t *o, const char *bam_file, const wspec_t *ws) {
    mmh_loader_t *ld = ws->sharded
        ? mmh_loader_open_share(bam_file, o->threads, o->K, o->B, o->allow_secondary, o->skip_supplementary, ws->voffset, ws->lo_tid, ws->lo_pos,
                                ws->hi_tid, ws->hi_pos, ws->first, ws->last)
        : mmh_loader_open(bam_file, o->threads, o->K, o->B, o->allow_secondary, o->skip_supplementary);
    if (!ld) { MMH_ERROR("NULL returned: could not open or parse %s.", bam_file); exit(EXIT_FAILURE); }
    return ld;
}

/* MM_TIMELINE=1: where the process is, in seconds since its start (start-up and teardown are not in the stage timers) */
static void tl_mark(double realtime0, const char *what) {
    static int on = -1;
    if (on < 0) on = getenv("MM_TIMELINE") != NULL;
    if (on) {   /* (with the wall clock and the resident set: what lies in front of main() and behind _exit() is found from outside) */
        long pages = 0, rss = 0;
        FILE *f = fopen("/proc/self/statm", "r");
        if (f) { if (fscanf(f, "%ld %ld", &pages, &rss) != 2) rss = 0; fclose(f); }
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "[timeline] %.3f %s (epoch %.3f, resident %.2f GB)\n", mmh_realtime() - realtime0, what, (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, (double)rss * 4096.0 / 1e9);
        if (atoi(getenv("MM_TIMELINE")) >= 2 && (f = fopen("/proc/self/smaps", "r")) != NULL) {   /* the mappings that hold 64 MB or more */
            char line[512], head[512] = "";
            while (fgets(line, sizeof line, f)) {
                unsigned long kb;
                if (strchr(line, '-') && strchr(line, '-') < line + 17 && !strstr(line, "kB")) { snprintf(head, sizeof head, "%.*s", (int)strcspn(line, "\n"), line); }
                else if (sscanf(line, "Rss: %lu kB", &kb) == 1 && kb >= 65536) fprintf(stderr, "[timeline]        %7.1f MB  %s\n", (double)kb / 1024.0, head);
            }
            fclose(f);
        }
    }
}

/* The device-side reader's batches are not the reference's -K / -B batches, and the reference names a failing read by its index in its
 * batch (src/mod.c: "read %d"; load_db's limits, src/minimod.c:249): on the error path -- the run is about to exit -- the host reader walks
 * the file with the run's -K / -B up to the failing read (number `ordinal` of the accepted reads, from 0) and says where it falls. */
static int32_t batch_index_in_file(const fopt_t *o, const char *bam_file, const wspec_t *ws, uint64_t ordinal, int32_t fallback) {
    mmh_loader_t *ld = ws->sharded
        ? mmh_loader_open_share(bam_file, o->threads, o->K, o->B, o->allow_secondary, o->skip_supplementary, ws->voffset, ws->lo_tid, ws->lo_pos, ws->hi_tid, ws->hi_pos, ws->first, ws->last)
        : mmh_loader_open(bam_file, o->threads, o->K, o->B, o->allow_secondary, o->skip_supplementary);
    if (!ld) return fallback;
    uint64_t seen = 0;
    int more = 1, set = 0;
    int32_t at = fallback;
    mm_batch_t b;
    while (more) {
        const int32_t n = mmh_loader_next(ld, set, &b, &more);
        if (n < 0) break;
        if (ordinal < seen + (uint64_t)n) { at = (int32_t)(ordinal - seen); break; }
        seen += (uint64_t)n;
        set = (set + 1) % MMH_POOL_SETS;
    }
    mmh_loader_close(ld);
    return at;
}

static void *free_ref_main(void *p) { mmh_free_ref((mmh_ref_t *)p); return NULL; }
static void *close_reader_main(void *p) { mmh_devloader_close((mmh_devloader_t *)p); return NULL; }

static int32_t batch_index_in_file_ctx(uint64_t ordinal, int32_t fallback) {
    if (!err_ctx.o) return fallback;
    return batch_index_in_file((const fopt_t *)err_ctx.o, err_ctx.bam, (const wspec_t *)err_ctx.ws, ordinal, fallback);
}

/* the codes of a device batch into the handles' tables, in the order the host's walk (intern_batch_codes) meets them */
static void intern_device_codes(mmh_devloader_t *dl, const mmh_devbatch_t *db, mm_freq_t *h, mm_freq_t *hv) {
    char codes[2 * MM_MAX_CODES][MM_CODE_LEN];
    const int k = mmh_devloader_codes(dl, &db->batch, &codes[0][0], 2 * MM_MAX_CODES);
    if (k >= 0) {
        for (int i = 0; i < k; i++) { (void)mm_freq_intern_code(h, codes[i]); if (hv) (void)mm_freq_intern_code(hv, codes[i]); }
        return;
    }
    /* codes the census does not take (longer than 8 characters ...): the batch's records and MM text come over and are walked here */
    mm_batch_t b = db->batch;
    mm_read_t *rd = (mm_read_t *)malloc((size_t)(b.n_reads > 0 ? b.n_reads : 1) * sizeof(mm_read_t));
    uint8_t *mm = (uint8_t *)malloc((size_t)b.n_mm_bytes + 1);
    if (!rd || !mm || mmh_devloader_fetch(dl, rd, db->batch.reads, (size_t)b.n_reads * sizeof(mm_read_t)) != 0 || mmh_devloader_fetch(dl, mm, db->batch.mm, (size_t)b.n_mm_bytes) != 0) {
        MMH_ERROR("%s", "Could not read the batch's modification codes"); exit(EXIT_FAILURE);
    }
    b.reads = rd; b.mm = mm;
    intern_batch_codes(h, &b);
    if (hv) intern_batch_codes(hv, &b);
    free(rd); free(mm);
}

static int run_body(const fopt_t *op, const mmh_mods_t *modsp, mmh_ref_t *ref, const char *bam_file, double realtime0, const wspec_t *ws) {
    const fopt_t o = *op;
    const mmh_mods_t mods = *modsp;
    const int view = o.view;
    char err[512];
    if (ws->sharded && ws->n_iv == 0 && !ws->last && ws->fd >= 0) {   /* more workers than 64 kb pieces of genome: nothing to do, nothing allocated */
        wtotals_t tt;
        memset(&tt, 0, sizeof tt);
        if (write_all(ws->fd, &tt, sizeof tt)) { MMH_ERROR("%s", "Could not send the rows to the parent process"); exit(EXIT_FAILURE); }
        close(ws->fd);
        return 0;
    }
    /* With the device inflater the readers can only be opened once its pinned buffers exist (0.2 s): they are made on a thread of
     * their own while this one sets up the freq handle (reference to HBM, context kernels, counter planes), for which the file's
     * header is read ahead of the readers.  Without it: the readers first, they decode ahead while the handle is set up. */
    int wildcard = 0, star_ctx = 0;
    for (int i = 0; i < mods.n_mods; i++) { if (strcmp(mods.code[i], "*") == 0) wildcard = 1; if (strcmp(mods.context[i], "*") == 0) star_ctx = 1; }
    int replay = !view && !o.canonical_order && (!ws->sharded || ws->tied) && (mods.n_mods > 1 || wildcard || star_ctx || o.insertions || o.haplotypes);
    /* Round 5: the replay runs on the DEVICE (include/minimod_tie.h: a read's own table, the first-insertion stamps, the core table and the
     * sort as parallel kernels) from the second handle's rows left in GPU memory -- nothing of it crosses PCIe but the final permutation,
     * and a tied run reads its BAM with the device-side reader like any other.  --host-replay (or MINIMOD_HOST_REPLAY=1) keeps round 4's
     * path: the serial restatement in tieorder.c, which is the device's checker. */
    const int dev_replay = replay && !o.host_replay && !getenv("MINIMOD_HOST_REPLAY");
    mm_bgzf_t *bz = NULL;
    mmh_loader_t *ld = NULL;
    mmh_devloader_t *dl = NULL;   /* --gpu-ingest: the decoded BAM stays on the device (devloader.c) */
    mm_pool_t *pool = NULL;       /* the workers that format the rows (the host loader's, or one of this function's own) */
    mm_pool_t *own_pool = NULL;   /* ... the latter, to be destroyed here */
    uint64_t dev_reads_before = 0;   /* device reader: accepted reads of the batches in front of the one at hand (a failing read's place in the file) */
    dl_job_t dlj;
    memset(&dlj, 0, sizeof dlj);
    const mm_bam_hdr_t *hdr = NULL;
    mm_bam_hdr_t hdr0;
    memset(&hdr0, 0, sizeof hdr0);
    bz_job_t bzj = {o.device, NULL};
    pthread_t bz_thread;
    int bz_running = 0;
    struct stat bst;   /* (a pipe's bytes can be read once: no header read-ahead there -- and no device inflate, which wants the file mapped) */
    const int regular = stat(bam_file, &bst) == 0 && S_ISREG(bst.st_mode);
    uint64_t hdr_bytes = 0;
    /* the device loader takes every run of a regular file but the ones that replay the tie order on the host (--host-replay): view gets
     * the read names with its batches, a wildcard run the batch's codes from a census kernel (mm_ingest_batch_codes) */
    err_ctx.o = &o; err_ctx.bam = bam_file; err_ctx.ws = ws;
    int use_dev = o.gpu_ingest && (!replay || dev_replay) && regular && o.debug_break < 0 && mm_bam_peek_header2(bam_file, &hdr0, &hdr_bytes) == 0;
    if (use_dev) {
        hdr = &hdr0;
        pool = mm_pool_create(o.threads);
        if (!pool) { MMH_ERROR("%s", "Could not start the worker threads"); exit(EXIT_FAILURE); }
        own_pool = pool;
        mmh_devloader_opts_t *d = &dlj.o;
        d->device = o.device; d->n_targets = hdr0.n_targets; d->allow_secondary = o.allow_secondary; d->skip_supplementary = o.skip_supplementary;
        d->header_bytes = hdr_bytes;
        if (view) {   /* rows are printed a batch at a time: batches of one group of 512 blocks (~15 Mbases) keep the printing beside the kernels */
            d->names = 1; d->max_blocks = 512; d->target_bases = (uint64_t)16 * 1000 * 1000;
        }
        if (ws->sharded) {
            d->ranged = 1; d->first = ws->first; d->last = ws->last; d->lo_tid = ws->lo_tid; d->lo_pos = ws->lo_pos; d->hi_tid = ws->hi_tid; d->hi_pos = ws->hi_pos;
            if (ws->voffset == UINT64_MAX) d->range_done_before_start = 1; else d->voffset = ws->voffset;
        }
        dlj.path = bam_file; dlj.pool = pool;
        /* its pinned staging and device buffers are made on a thread of their own while this one sets up the freq handle */
        if (!getenv("MM_SERIAL_OPEN") && pthread_create(&bz_thread, NULL, dl_open_main, &dlj) == 0) bz_running = 2;
        else dl_open_main(&dlj);
    } else if (o.gpu_inflate && regular && mm_bam_peek_header(bam_file, &hdr0) == 0 && pthread_create(&bz_thread, NULL, bz_create_main, &bzj) == 0) {
        bz_running = 1;
        hdr = &hdr0;
    } else {
        if (o.gpu_inflate) { bz = bz_create(o.device); bz_attach(bz); }
        tl_mark(realtime0, "run_body (inflater made)");
        ld = open_loader(&o, bam_file, ws);
        hdr = mm_bam_header(ld->bam);
        tl_mark(realtime0, "loader open");
    }

    /* The HIP runtime's start (begun on a thread of its own when the process began, or here in a --devices worker) is not the
     * reference's work: waited for and reported by itself, in front of the stage the reference calls "contexts" (src/freq_main.c:470-482) */
    {
        const double tw = mmh_realtime();
        if (mm_hip_warm(o.device) != 0) { MMH_ERROR("GPU %d is not usable (no CPU fallback in this build)", o.device); exit(EXIT_FAILURE); }
        mmh_gpu_in_use = 1;
        fprintf(stderr, "[%s] GPU runtime ready %.3f sec after the process began (waited %.3f sec for it here)\n", __func__, mmh_realtime() - realtime0, mmh_realtime() - tw);
    }
    double t2 = mmh_realtime();
    fprintf(stderr, "[%s] Loading contexts in reference\n", __func__);
    mm_contig_t *ctg = (mm_contig_t *)calloc((size_t)(hdr->n_targets > 0 ? hdr->n_targets : 1), sizeof(mm_contig_t));
    for (int t = 0; t < hdr->n_targets; t++) {
        ctg[t].name = hdr->target_name[t];
        ctg[t].length = hdr->target_len[t];
        int ri = mmh_ref_find(ref, hdr->target_name[t]);
        if (ri >= 0) { ctg[t].seq = ref->seq[ri]; ctg[t].seq_length = ref->len[ri]; }
    }
    mm_freq_opts_t fo;
    mmh_fill_opts(&mods, o.insertions, o.haplotypes, o.device, &fo);
    fo.view = view;
    /* process_db is called per -K batch (src/minimod.c:344-350); the library stages consecutive batches in GPU memory and
     * launches them together.  view prints a launch's rows when the launch is retired. */
    int gather = o.gather;
    if (replay && !dev_replay && gather == MMH_MAX_GATHER) {
        /* a run that replays the reference's row order keeps a launch's batches and its calls on the host until the launch has been
         * replayed: launches of about 24 000 reads (still streamed) instead of as many as the staging takes -- the replay of one
         * launch then runs beside the loading of the next ones, and 1.3 GB less is held (3 Gbases of HiFi reads, two codes: 1.96 ->
         * 1.76 s, tools/c3_replay_sweep.sh).  (The replay on a thread of its own, beside the submits, was slower: 1.86 s -- the
         * loop is bound by the 16 cores the loader and the replay share, not by who waits for whom.) */
        int g = (24576 + (o.K > 0 ? o.K : 512) - 1) / (o.K > 0 ? o.K : 512);
        gather = g < 6 ? 6 : (g > MMH_MAX_GATHER ? MMH_MAX_GATHER : g);
    }
    fo.coalesce = gather > MMH_MAX_GATHER ? MMH_MAX_GATHER : gather;   /* (view as well since round 4: a launch's rows are printed from the group's journal) */
    mm_freq_t *h = mm_freq_create(&fo, hdr->n_targets, ctg, ws->sharded ? ws->n_iv : 0, ws->sharded ? ws->iv : NULL, err, sizeof err);
    if (!h) { MMH_ERROR("Assertion failed. %s", err); fprintf(stderr, "Exiting.\n"); exit(EXIT_FAILURE); }
    tl_mark(realtime0, "mm_freq_create done");
    if (replay && o.K >= (1 << 21)) {   /* (the rows the replay works from number a batch's reads with 21 bits) */
        MMH_ERROR("%s", "-K of 2097152 or more: the order minimod's hash table leaves rows that tie on (contig, start) in is replayed for smaller batches only. Use a smaller -K, or --canonical-order to print such rows by strand, code, ins_offset, haplotype");
        exit(EXIT_FAILURE);
    }
    mm_freq_t *hv = NULL;      /* the second handle of a replay run: the same batches in view mode (rows with group ordinals) */
    mmh_tie_t *tie = NULL;
    mm_tie_t *dtie = NULL;     /* the device-side replay (dev_replay) */
    int dtie_codes = -1;       /* code strings it has been given */
    if (replay) {
        mm_freq_opts_t fv = fo;
        fv.view = 2;
        /* (gathered like the first handle's since round 4: its launches stream, a launch's rows are replayed from the group's journal) */
        hv = mm_freq_create(&fv, hdr->n_targets, ctg, 0, NULL, err, sizeof err);
        if (dev_replay) {
            mm_tie_opts_t to;
            memset(&to, 0, sizeof to);
            to.abi_version = MM_TIE_ABI_VERSION; to.device = o.device; to.insertions = o.insertions; to.haplotypes = o.haplotypes; to.n_contigs = hdr->n_targets;
            int64_t *tl = (int64_t *)malloc(sizeof(int64_t) * (size_t)(hdr->n_targets > 0 ? hdr->n_targets : 1));
            for (int t = 0; t < hdr->n_targets; t++) tl[t] = (int64_t)hdr->target_len[t];
            char terr[256];
            terr[0] = 0;
            dtie = hv ? mm_tie_create(&to, (const char *const *)hdr->target_name, tl, terr, sizeof terr) : NULL;
            free(tl);
            if (hv && !dtie) { MMH_ERROR("Assertion failed. %s", terr); fprintf(stderr, "Exiting.\n"); exit(EXIT_FAILURE); }
        } else tie = mmh_tie_create(hdr, o.insertions, o.haplotypes);
        if (!hv || (!tie && !dtie)) { MMH_ERROR("Assertion failed. %s", hv ? "out of memory" : err); fprintf(stderr, "Exiting.\n"); exit(EXIT_FAILURE); }
    }
    free(ctg);
    {   /* the reference now lives in HBM: its host copy (hundreds of MB to unmap: 40 ms for a 400-Mb genome) is let go beside the first batches */
        pthread_t ft;
        pthread_attr_t fa;
        pthread_attr_init(&fa);
        pthread_attr_setdetachstate(&fa, PTHREAD_CREATE_DETACHED);
        if (getenv("MM_FULL_TEARDOWN") || pthread_create(&ft, &fa, free_ref_main, ref) != 0) mmh_free_ref(ref);
        pthread_attr_destroy(&fa);
    }
    fprintf(stderr, "[%s] Reference contexts loaded in %.3f sec\n", __func__, mmh_realtime() - t2);
    if (use_dev) {
        if (bz_running == 2) pthread_join(bz_thread, NULL);
        dl = dlj.dl;
        if (!dl) { MMH_ERROR("The device-side BAM reader could not be set up: %s (try --no-gpu-ingest)", dlj.err); exit(EXIT_FAILURE); }
        tl_mark(realtime0, "device loader open (beside the handle)");
    } else if (bz_running) {
        pthread_join(bz_thread, NULL);
        bz = bzj.bz;
        bz_attach(bz);
        tl_mark(realtime0, "inflater made (beside the handle)");
        ld = open_loader(&o, bam_file, ws);
        hdr = mm_bam_header(ld->bam);
        {   /* the handle was built from the header read ahead: it must be the reader's, contig by contig */
            int same = hdr->n_targets == hdr0.n_targets;
            for (int t = 0; same && t < hdr->n_targets; t++) same = hdr->target_len[t] == hdr0.target_len[t] && strcmp(hdr->target_name[t], hdr0.target_name[t]) == 0;
            if (!same) { MMH_ERROR("%s changed while it was being read", bam_file); exit(EXIT_FAILURE); }
        }
        tl_mark(realtime0, "loader open");
    }
    /* (replay, above) Rows can tie on (contig, start) when several codes are counted, both strands can be called on one
     * position (`*` contexts), insertion offsets or haplotypes are keys.  The reference prints such rows in the order its
     * hash table and its unstable sort leave them in (tieorder.c): that order is replayed from the calls of every read,
     * which a second handle in view mode delivers for the same batches.  A run whose rows cannot tie (-c m[CG]) needs
     * none of this. */
    const int header_late = !view && dev_replay;   /* (a run whose device replay gives up starts again as a child process: it must not have printed) */
    if (ws->fd < 0 && !header_late) {
        if (view) mmh_print_view_header(o.out, o.insertions, o.haplotypes);
        else mmh_print_freq_header(o.out, o.bedmethyl, o.insertions, o.haplotypes);
    }

    if (!use_dev) pool = mm_bam_pool(ld->bam);
    double load_time = 0, process_wait_time = 0, output_time = 0, replay_time = 0, submit_time = 0;
    int more = 1, counter = 0, set = 0;
    const uint8_t *klass_of_code[MM_MAX_CODES];
    for (int i = 0; i < MM_MAX_CODES; i++) {   /* a wildcard run counts every code under the one `*` entry */
        int req = i < mods.n_mods ? i : 0;
        if (wildcard) for (int m2 = 0; m2 < mods.n_mods; m2++) if (strcmp(mods.code[m2], "*") == 0) req = m2;
        klass_of_code[i] = fo.mods[req].klass;
    }
    mm_batch_t batch;
    /* freq: `cur` = the group being gathered, `prev` = the one launched last (waited for one iteration later, so that the
     * wait costs nothing); copied[set] = the ticket whose host -> device copy last read from that pool set */
    group_t *cur = (group_t *)calloc(1, sizeof(group_t)), *prev = (group_t *)calloc(1, sizeof(group_t));
    group_t *vcur = (group_t *)calloc(1, sizeof(group_t)), *vprev = (group_t *)calloc(1, sizeof(group_t));   /* the replay handle's groups */
    cur->ticket = prev->ticket = vcur->ticket = vprev->ticket = -1;
    cur->dev_before = prev->dev_before = vcur->dev_before = vprev->dev_before = -1;
    int32_t copied[MMH_POOL_SETS];
    for (int i = 0; i < MMH_POOL_SETS; i++) copied[i] = -1;
    double prog_t = mmh_realtime();
    if (use_dev) {
        /* load(N + 1) beside process(N), as in the reference's pipeline (src/freq_main.c:404-474) -- only that "load" here is the
         * device decoding the next batch while its freq kernels work on this one; the host moves compressed bytes.  Two batches'
         * tickets are kept open (their arenas stay theirs until they have been waited for), a third arena is being filled. */
        int32_t tk[2] = {-1, -1}, tv[2] = {-1, -1};   /* (tv: the same batch's ticket of the replay's second handle) */
        int32_t tvn[2] = {0, 0};
        uint64_t dev_reads_before_of[2] = {0, 0};   /* per open ticket: accepted reads of the device batches in front of it */
        int ar[2] = {-1, -1};
        group_t *vg[2] = {cur, prev};               /* view: the open tickets' journals (read records and names, for the printing) */
        void *stream = mmh_devloader_stream(dl);
        more = 1;
        while (more) {
            double tl = mmh_realtime();
            mmh_devbatch_t db;
            int32_t n = mmh_devloader_next(dl, &db, &more);
            if (n < 0) {
                const mmh_devloader_stats_t *st = mmh_devloader_stats(dl);
                if (st->err > 1 && tk[0] < 0 && st->processed_reads == 0) {
                    /* The device reader cannot take this file (a record longer than its head room, a group that does not fit an arena, a
                     * header it cannot frame ...) and nothing has been counted yet: the host reader reads such a file without trouble, so
                     * the run goes on with it -- in this process, same handle, same output (the automatic default for big
                     * files must not turn inputs that used to work into failures). */
                    MMH_WARNING("The device-side BAM reader gave up: %s; the host threads read the file", mm_ingest_strerror(st->err));
                    mmh_devloader_close(dl); dl = NULL;
                    use_dev = 0;
                    ld = open_loader(&o, bam_file, ws);
                    more = 1;
                    break;
                }
                if (st->err > 1) MMH_ERROR("The device-side BAM reader gave up: %s (try --no-gpu-ingest)", mm_ingest_strerror(st->err));
                else MMH_ERROR("%s", "Truncated or corrupt BAM file");
                exit(EXIT_FAILURE);
            }
            load_time += mmh_realtime() - tl;
            fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bases) loaded\n", __func__, mmh_realtime() - realtime0,
                    mmh_cputime() / (mmh_realtime() - realtime0), n, db.processed_bytes / (1000.0 * 1000.0));
            for (int k = 0; k < 2 && tk[0] >= 0; k++) {   /* the older ticket; both when nothing follows */
                if (view) { vg[0]->ticket = tk[0]; retire_view_group(h, vg[0], hdr, &o, pool, &process_wait_time, &output_time); }
                else {
                    double tw = mmh_realtime();
                    int32_t bad = -1;
                    int e = mm_freq_wait(h, tk[0], &bad);
                    process_wait_time += mmh_realtime() - tw;
                    if (e) {
                        mm_read_t rec;
                        const int have = bad >= 0 && mm_freq_read_record(h, tk[0], bad, &rec) == 0;
                        die_read_record(e, batch_index_in_file(&o, bam_file, ws, dev_reads_before_of[0] + (uint64_t)(bad > 0 ? bad : 0), bad), have ? &rec : NULL, hdr);
                    }
                }
                if (dev_replay) replay_ticket_dev(hv, dtie, tv[0], &tvn[0], 1, hdr, klass_of_code, &dtie_codes, &replay_time);
                mmh_devloader_release(dl, ar[0]);
                { group_t *t = vg[0]; vg[0] = vg[1]; vg[1] = t; }
                tk[0] = tk[1]; ar[0] = ar[1]; tv[0] = tv[1]; tvn[0] = tvn[1]; dev_reads_before_of[0] = dev_reads_before_of[1]; tk[1] = -1; ar[1] = -1; tv[1] = -1;
                if (more) break;
            }
            if (n > 0) {
                if (wildcard) intern_device_codes(dl, &db, h, replay ? hv : NULL);
                if (view) {
                    group_t *gj = tk[0] < 0 ? vg[0] : vg[1];
                    if (journal_from_device(&gj->j, dl, &db) != 0) { MMH_ERROR("%s", "Could not fetch the batch's read names"); exit(EXIT_FAILURE); }
                    gj->n = 1; gj->n_reads[0] = n; gj->dev_before = (int64_t)dev_reads_before;
                }
                const double t_sub = mmh_realtime();
                int32_t t = mm_freq_submit_device_now(h, &db.batch, stream, db.bases);
                submit_time += mmh_realtime() - t_sub;
                if (t < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror(t)); exit(EXIT_FAILURE); }
                int32_t t2 = -1;
                if (dev_replay) {   /* the same batch through the second handle, right behind the first on the reader's stream */
                    t2 = mm_freq_submit_device_now(hv, &db.batch, stream, db.bases);
                    if (t2 < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror(t2)); exit(EXIT_FAILURE); }
                }
                if (tk[0] < 0) { tk[0] = t; ar[0] = db.arena; tv[0] = t2; tvn[0] = n; dev_reads_before_of[0] = dev_reads_before; }
                else { tk[1] = t; ar[1] = db.arena; tv[1] = t2; tvn[1] = n; dev_reads_before_of[1] = dev_reads_before; }
                dev_reads_before += (uint64_t)n;
            }
            if (o.progress_interval <= 0 || mmh_realtime() - prog_t > o.progress_interval) {
                fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bytes) processed\t%d Entries (%.1fM bytes) skipped\n", __func__,
                        mmh_realtime() - realtime0, mmh_cputime() / (mmh_realtime() - realtime0), n, db.total_bytes / (1000.0 * 1000.0),
                        (int)(db.total_reads - (uint64_t)n), (db.total_bytes - db.processed_bytes) / (1000.0 * 1000.0));
                prog_t = mmh_realtime();
            }
            const mmh_devloader_stats_t *st = mmh_devloader_stats(dl);
            uint64_t skipped = st->total_reads - st->processed_reads;
            if (skipped > 0.9 * st->total_reads)
                MMH_WARNING("%s", "90% of the reads are skipped. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
            if (skipped == st->total_reads)
                MMH_ERROR("%s", "All reads are skipped. Quitting. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
        }
        for (int k = 0; k < 2; k++) {
            if (tk[k] < 0) continue;
            if (view) { vg[k]->ticket = tk[k]; retire_view_group(h, vg[k], hdr, &o, pool, &process_wait_time, &output_time); }
            else {
                double tw = mmh_realtime();
                int32_t bad = -1;
                int e = mm_freq_wait(h, tk[k], &bad);
                process_wait_time += mmh_realtime() - tw;
                if (e) {
                    mm_read_t rec;
                    const int have = bad >= 0 && mm_freq_read_record(h, tk[k], bad, &rec) == 0;
                    die_read_record(e, batch_index_in_file(&o, bam_file, ws, dev_reads_before_of[k] + (uint64_t)(bad > 0 ? bad : 0), bad), have ? &rec : NULL, hdr);
                }
            }
            if (dev_replay) replay_ticket_dev(hv, dtie, tv[k], &tvn[k], 1, hdr, klass_of_code, &dtie_codes, &replay_time);
            mmh_devloader_release(dl, ar[k]);
        }
    }
    while (more && !use_dev) {
        double tl = mmh_realtime();
        if (copied[set] >= 0) {   /* the batch read into this pool set MMH_POOL_SETS iterations ago must have left host memory */
            int e = mm_freq_host_done(h, copied[set]);
            if (e) { MMH_ERROR("GPU path failed: %s", mm_strerror(e)); exit(EXIT_FAILURE); }
            copied[set] = -1;
        }
        int32_t n = mmh_loader_next(ld, set, &batch, &more);
        if (n < 0) { MMH_ERROR("%s", "Truncated or corrupt BAM file"); exit(EXIT_FAILURE); }
        load_time += mmh_realtime() - tl;
        fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bases) loaded\n", __func__, mmh_realtime() - realtime0,
                mmh_cputime() / (mmh_realtime() - realtime0), n, ld->last_processed_bytes / (1000.0 * 1000.0));
        /* the previous batch's pool set is about to be reused two iterations from now: retire it first */
        if (view) retire_view_group(h, prev, hdr, &o, pool, &process_wait_time, &output_time);
        if (dev_replay) replay_group_dev(hv, dtie, vprev, hdr, klass_of_code, &dtie_codes, &replay_time);
        else if (replay) replay_group(hv, tie, vprev, hdr, pool, klass_of_code, &replay_time);
        if (!view) retire_group(h, prev, hdr, &process_wait_time);
        if (n > 0) {
            if (wildcard) { intern_batch_codes(h, &batch); if (replay) intern_batch_codes(hv, &batch); }
            const double t_sub = mmh_realtime();
            int32_t tk = mm_freq_submit(h, &batch);
            submit_time += mmh_realtime() - t_sub;
            if (tk < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror(tk)); exit(EXIT_FAILURE); }
            {
                if (tk != cur->ticket) {   /* a new group: the one before it has been launched */
                    if (view) retire_view_group(h, prev, hdr, &o, pool, &process_wait_time, &output_time);
                    else retire_group(h, prev, hdr, &process_wait_time);
                    group_t *t = prev; prev = cur; cur = t;
                    cur->ticket = tk; cur->n = 0; journal_reset(&cur->j);
                }
                if (cur->n < MMH_MAX_GATHER) cur->n_reads[cur->n++] = n;
                if (view && journal_add(&cur->j, &batch, ld, set) != 0) { MMH_ERROR("%s", "Out of memory"); exit(EXIT_FAILURE); }
                copied[set] = tk;
            }
            if (replay) {
                const int32_t vt = mm_freq_submit(hv, &batch);
                if (vt < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror(vt)); exit(EXIT_FAILURE); }
                if (vt != vcur->ticket) {
                    if (dev_replay) replay_group_dev(hv, dtie, vprev, hdr, klass_of_code, &dtie_codes, &replay_time);
                    else replay_group(hv, tie, vprev, hdr, pool, klass_of_code, &replay_time);
                    group_t *t = vprev; vprev = vcur; vcur = t;
                    vcur->ticket = vt; vcur->n = 0; journal_reset(&vcur->j);
                }
                if (vcur->n < MMH_MAX_GATHER) vcur->n_reads[vcur->n++] = n;
                if (!dev_replay && journal_add_mm(&vcur->j, &batch) != 0) { MMH_ERROR("%s", "Out of memory"); exit(EXIT_FAILURE); }
                /* (both handles copy out of the same pool set: the one checked below is the first's, whose copies were queued first ... */
                if (mm_freq_host_done(hv, vt) != 0) { MMH_ERROR("%s", "GPU path failed"); exit(EXIT_FAILURE); }   /* ... so this one is waited for here) */
            }
        }
        if (o.progress_interval <= 0 || mmh_realtime() - prog_t > o.progress_interval) {
            fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bytes) processed\t%d Entries (%.1fM bytes) skipped\n", __func__,
                    mmh_realtime() - realtime0, mmh_cputime() / (mmh_realtime() - realtime0), n, ld->last_total_bytes / (1000.0 * 1000.0),
                    ld->last_total_reads - n, (ld->last_total_bytes - ld->last_processed_bytes) / (1000.0 * 1000.0));
            prog_t = mmh_realtime();
        }
        uint64_t skipped = ld->total_reads - ld->processed_reads;
        if (skipped > 0.9 * ld->total_reads)
            MMH_WARNING("%s", "90% of the reads are skipped. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
        if (skipped == ld->total_reads)
            MMH_ERROR("%s", "All reads are skipped. Quitting. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
        set = (set + 1) % MMH_POOL_SETS;
        if (o.debug_break == counter) break;
        counter++;
    }
    if (view) { retire_view_group(h, prev, hdr, &o, pool, &process_wait_time, &output_time); retire_view_group(h, cur, hdr, &o, pool, &process_wait_time, &output_time); }
    if (dev_replay) { replay_group_dev(hv, dtie, vprev, hdr, klass_of_code, &dtie_codes, &replay_time); replay_group_dev(hv, dtie, vcur, hdr, klass_of_code, &dtie_codes, &replay_time); }
    else if (replay) { replay_group(hv, tie, vprev, hdr, pool, klass_of_code, &replay_time); replay_group(hv, tie, vcur, hdr, pool, klass_of_code, &replay_time); }
    if (!view) { retire_group(h, prev, hdr, &process_wait_time); retire_group(h, cur, hdr, &process_wait_time); }
    journal_free(&cur->j); journal_free(&prev->j); journal_free(&vcur->j); journal_free(&vprev->j);
    free(cur); free(prev); free(vcur); free(vprev);
    struct { uint64_t total_reads, total_bytes, processed_reads, processed_bytes, processed_bases; } T;
    if (use_dev) {
        const mmh_devloader_stats_t *st = mmh_devloader_stats(dl);
        T.total_reads = st->total_reads; T.total_bytes = st->total_bytes; T.processed_reads = st->processed_reads; T.processed_bytes = st->processed_bytes; T.processed_bases = st->processed_bases;
        fprintf(stderr, "[gpu-ingest] %lu groups of BGZF blocks framed and flattened on the device (%lu blocks walked again from their true entry, %lu decoded by the host), "
                        "%.3f s waiting for staged groups, %.3f s staging; device ms summed over the groups: copies %.0f, inflate %.0f, CRC32 %.0f, frame + flatten %.0f\n", (unsigned long)st->groups, (unsigned long)st->slow_blocks, (unsigned long)st->patched_blocks, st->wait_seconds, st->stage_seconds,
                st->stage_ms[0], st->stage_ms[1], st->stage_ms[2], st->stage_ms[3]);
        /* The device reader has handed over its last batch and every ticket has been waited for: its queues (a queue holds 173 MB of host memory for the
         * waves' saved state), its pinned staging and its device memory are given back on a thread of their own beside the sort and the output -- at the
         * process's death the kernel would do the same, but with the caller waiting (tools/exit_probe*.hip: 7 ms a queue, 0.14 ms a pinned MB). */
        if (!getenv("MM_FULL_TEARDOWN") && !getenv("MM_KEEP_READER")) {
            pthread_t ct; pthread_attr_t ca;
            pthread_attr_init(&ca); pthread_attr_setdetachstate(&ca, PTHREAD_CREATE_DETACHED);
            if (pthread_create(&ct, &ca, close_reader_main, dl) == 0) dl = NULL;
            pthread_attr_destroy(&ca);
        }
    } else {
        T.total_reads = ld->total_reads; T.total_bytes = ld->total_bytes; T.processed_reads = ld->processed_reads; T.processed_bytes = ld->processed_bytes; T.processed_bases = ld->processed_bases;
    }
    double sort_time = 0;
    if (!view && ws->sharded && ws->fd >= 0) {
        /* The halo behind a cut inside a contig: the counters this worker's reads left there go to the right-hand neighbour as
         * ONE slab (mm_freq_slab_export: position-dense, planes x strands x halo words), which adds them to its own
         * (mm_freq_slab_add); here they are cleared.  The left neighbour's slab is taken in first. */
        /* Transport (round 4): the slab goes from GPU to GPU -- the sender packs it into a device buffer and sends that buffer's HIP
         * IPC handle (64 bytes) over the socket between the two workers; the receiver opens it, copies device to device (xGMI between two
         * GPUs of a node) and answers with one byte, after which the sender may let go of the buffer.  Where the platform gives or takes
         * no handle (answer 0) the slab goes through host memory and the socket as before. */
        if (ws->slab_in >= 0) {
            int64_t hd[4] = {0, 0, 0, 0};   /* tid, begin, length, 1 = an IPC handle follows / 0 = the words follow */
            if (read_all(ws->slab_in, hd, sizeof hd) != 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
            if (hd[2] > 0) {
                int took = 0;
                if (hd[3] == 1) {
                    unsigned char ih[MM_IPC_HANDLE_BYTES];
                    if (read_all(ws->slab_in, ih, sizeof ih) != 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
                    took = mm_freq_slab_add_ipc(h, (int32_t)hd[0], hd[1], hd[2], ih) == 0;
                    const unsigned char ack = took ? 1 : 0;
                    if (write_all(ws->slab_in, &ack, 1) != 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
                    if (took) fprintf(stderr, "[%s] halo slab of %ld positions taken from the left neighbour's GPU through a HIP IPC handle (device to device)\n", __func__, (long)hd[2]);
                }
                if (!took) {
                    const int64_t nw = mm_freq_slab_words(h, hd[2]);
                    uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nw);
                    if (!buf || read_all(ws->slab_in, buf, sizeof(uint64_t) * (size_t)nw) != 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
                    int e = mm_freq_slab_add_host(h, (int32_t)hd[0], hd[1], hd[2], buf);
                    if (e) { MMH_ERROR("GPU path failed: %s", mm_strerror(e)); exit(EXIT_FAILURE); }
                    fprintf(stderr, "[%s] halo slab of %ld positions taken from the left neighbour through host memory\n", __func__, (long)hd[2]);
                    free(buf);
                }
            }
            close(ws->slab_in);
        }
        if (ws->slab_out >= 0) {
            int64_t hd[4] = {0, 0, 0, 0};
            const mm_interval_t *iv = ws->n_iv > 0 ? &ws->iv[ws->n_iv - 1] : NULL;
            int sent = 0;
            if (iv && iv->halo > 0) {
                hd[0] = iv->tid; hd[1] = iv->end;
                hd[2] = iv->end + iv->halo <= (int64_t)hdr->target_len[iv->tid] ? iv->halo : (int64_t)hdr->target_len[iv->tid] - iv->end;
                unsigned char ih[MM_IPC_HANDLE_BYTES];
                if (!getenv("MM_NO_IPC_SLABS") && mm_freq_slab_export_ipc(h, (int32_t)hd[0], hd[1], hd[2], ih) == 0) {
                    hd[3] = 1;
                    unsigned char ack = 0;
                    if (write_all(ws->slab_out, hd, sizeof hd) || write_all(ws->slab_out, ih, sizeof ih) || read_all(ws->slab_out, &ack, 1)) { MMH_ERROR("%s", "A worker of --devices lost its right neighbour"); exit(EXIT_FAILURE); }
                    sent = ack == 1;   /* (0: the neighbour could not open the handle and waits for the words now) */
                } else if (write_all(ws->slab_out, hd, sizeof hd)) { MMH_ERROR("%s", "A worker of --devices lost its right neighbour"); exit(EXIT_FAILURE); }
                if (!sent) {
                    const int64_t nw = mm_freq_slab_words(h, hd[2]);
                    uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nw > 0 ? nw : 1));
                    int e = buf ? mm_freq_slab_export_host(h, (int32_t)hd[0], hd[1], hd[2], buf) : -MM_E_NOMEM;
                    if (e) { MMH_ERROR("GPU path failed: %s", mm_strerror(e)); exit(EXIT_FAILURE); }
                    if (write_all(ws->slab_out, buf, sizeof(uint64_t) * (size_t)nw)) { MMH_ERROR("%s", "A worker of --devices lost its right neighbour"); exit(EXIT_FAILURE); }
                    free(buf);
                }
                int e = mm_freq_slab_clear(h, (int32_t)hd[0], hd[1], hd[2], NULL);
                if (e) { MMH_ERROR("GPU path failed: %s", mm_strerror(e)); exit(EXIT_FAILURE); }
            } else if (write_all(ws->slab_out, hd, sizeof hd)) { MMH_ERROR("%s", "A worker of --devices lost its right neighbour"); exit(EXIT_FAILURE); }
            close(ws->slab_out);
        }
    }
    tl_mark(realtime0, "last batch handed over");
    if (!view) {
        double ts = mmh_realtime();
        const mm_row_t *rows = NULL, *drows = NULL;
        /* a plain run (one process, the fixed order or no ties, no --region) whose rows the device formats: they stay in GPU memory between the
         * kernel that makes them and the kernels that make their text (mm_freq_finalize_device: NULL in *drows when side rows had to be merged in) */
        const char *fe = getenv("MINIMOD_FMT");
        const int rows_stay = !replay && !o.region && ws->fd < 0 && !(fe && strcmp(fe, "device") != 0) && !getenv("MM_ROWS_TO_HOST");
        int64_t nrows = rows_stay ? mm_freq_finalize_device(h, &rows, &drows) : mm_freq_finalize(h, &rows);
        if (nrows < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror((int32_t)nrows)); exit(EXIT_FAILURE); }
        if (drows && nrows < 65536 && !fe) {   /* (few rows: the worker pool formats them, and wants them here) */
            drows = NULL;
            nrows = mm_freq_finalize(h, &rows);
            if (nrows < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror((int32_t)nrows)); exit(EXIT_FAILURE); }
        }
        sort_time = mmh_realtime() - ts;
        if (ws->fd >= 0) {
            /* A worker of --devices.  Its rows behind its share's end -- what its reads called past the halo, and every row
             * the side lists hold there (inside insertions, haplotypes without a plane) -- belong to a share further right:
             * they go to the right-hand neighbour, which adds them to its own (and passes on what lies behind ITS end).  What
             * is left is this share's part of the output: formatted here, in parallel with the other workers, one section per
             * contig; the parent only puts the sections in order.  (Tied runs that want the reference's order send rows and
             * the first-insertion sequence of their keys to the parent instead, which orders and formats.) */
            int *rank = contig_ranks(hdr);
            rowcmp_t rc = {rank};
            mm_row_t *mine = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(nrows > 0 ? nrows : 1));
            mm_row_t *fwd = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(nrows > 0 ? nrows : 1));
            int64_t n_mine = 0, n_fwd = 0;
            for (int64_t i = 0; i < nrows; i++) { if (before_hi(ws, rows[i].tid, rows[i].pos)) mine[n_mine++] = rows[i]; else fwd[n_fwd++] = rows[i]; }
            if (ws->rows_in >= 0) {   /* what the shares to the left found in front of them */
                int64_t n_in = 0;
                if (read_all(ws->rows_in, &n_in, sizeof n_in) != 0 || n_in < 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
                mm_row_t *in = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(n_in > 0 ? n_in : 1));
                if (n_in && read_all(ws->rows_in, in, sizeof(mm_row_t) * (size_t)n_in) != 0) { MMH_ERROR("%s", "A worker of --devices lost its left neighbour"); exit(EXIT_FAILURE); }
                close(ws->rows_in);
                mm_row_t *in_mine = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(n_in > 0 ? n_in : 1)), *in_fwd = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(n_in > 0 ? n_in : 1));
                int64_t a1 = 0, a2 = 0, nm = 0, nf = 0;
                for (int64_t i = 0; i < n_in; i++) { if (before_hi(ws, in[i].tid, in[i].pos)) in_mine[a1++] = in[i]; else in_fwd[a2++] = in[i]; }
                mm_row_t *m2 = merge_rows(&rc, mine, n_mine, in_mine, a1, &nm), *f2 = merge_rows(&rc, fwd, n_fwd, in_fwd, a2, &nf);
                free(mine); free(fwd); free(in); free(in_mine); free(in_fwd);
                mine = m2; n_mine = nm; fwd = f2; n_fwd = nf;
            }
            if (ws->rows_out >= 0) {
                if (write_all(ws->rows_out, &n_fwd, sizeof n_fwd) || write_all(ws->rows_out, fwd, sizeof(mm_row_t) * (size_t)n_fwd)) { MMH_ERROR("%s", "A worker of --devices lost its right neighbour"); exit(EXIT_FAILURE); }
                close(ws->rows_out);
                n_fwd = 0;
            }
            if (n_fwd) { /* (the last worker keeps everything: before_hi is always true there) */ }
            wtotals_t tt;
            memset(&tt, 0, sizeof tt);
            tt.total_reads = T.total_reads; tt.total_bytes = T.total_bytes; tt.processed_reads = T.processed_reads;
            tt.processed_bytes = T.processed_bytes; tt.processed_bases = T.processed_bases;
            tt.load_time = load_time; tt.wait_time = process_wait_time; tt.sort_time = sort_time;
            if (ws->tied) {
                const void *tk = NULL; const uint32_t *th = NULL;
                int64_t ntk = -1;
                int pal = 0;
                void *dk = NULL; uint32_t *dh = NULL;
                if (dev_replay) {   /* the sequence from the device: every key this worker's reads entered, in the order they did, as keys the parent's table takes */
                    const int64_t nk = g_dev_replay_gave_up ? -1 : mm_tie_sequence_size(dtie);
                    mm_row_t *kr = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(nk > 0 ? nk : 1));
                    dk = malloc(16 * (size_t)(nk > 0 ? nk : 1)); dh = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(nk > 0 ? nk : 1));
                    int32_t pl = 0;
                    if (nk >= 0 && kr && dk && dh && mm_tie_sequence(dtie, kr, dh, nk, &pl) == nk) { mmh_tie_keys_from_rows(kr, NULL, nk, dk); tk = dk; th = dh; ntk = nk; pal = pl; }
                    else MMH_WARNING("the device-side replay gave up in a worker (reason bits 0x%x): the parent starts the run again with --host-replay", mm_tie_failed(dtie));
                    free(kr);
                } else if (replay) ntk = mmh_tie_export2(tie, &tk, &th, &pal);
                tt.n_rows = n_mine; tt.n_tie_keys = ntk; tt.tie_put_after = pal;
                if (write_all(ws->fd, &tt, sizeof tt) || write_all(ws->fd, mine, sizeof(mm_row_t) * (size_t)n_mine) ||
                    (ntk > 0 && (write_all(ws->fd, tk, 16 * (size_t)ntk) || write_all(ws->fd, th, 4 * (size_t)ntk)))) {
                    MMH_ERROR("%s", "Could not send the rows to the parent process"); exit(EXIT_FAILURE);
                }
                free(dk); free(dh);
            } else {
                FILE *pf = fopen(ws->part_path, "wb");
                if (!pf) { MMH_ERROR("Cannot open file %s for writing", ws->part_path); exit(EXIT_FAILURE); }
                const char *codes[MM_MAX_CODES];
                int n_codes = code_names(h, codes);
                wsection_t *sec = (wsection_t *)malloc(sizeof(wsection_t) * (size_t)(hdr->n_targets > 0 ? hdr->n_targets : 1));
                int64_t n_sec = 0;
                for (int64_t i = 0; i < n_mine;) {   /* rows come contig by contig */
                    int64_t j = i;
                    while (j < n_mine && mine[j].tid == mine[i].tid) j++;
                    if (mmh_emit_flush() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
                    const int64_t at = (int64_t)ftello(pf);
                    print_freq_rows_any(pf, pool, mine + i, NULL, j - i, hdr, codes, n_codes, o.bedmethyl, o.insertions, o.haplotypes, o.device);
                    if (mmh_emit_flush() != 0 || fflush(pf) != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
                    sec[n_sec].tid = mine[i].tid; sec[n_sec].pad = 0; sec[n_sec].off = at; sec[n_sec].len = (int64_t)ftello(pf) - at;
                    n_sec++;
                    i = j;
                }
                if (mmh_emit_finish() != 0 || fclose(pf) != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
                tt.n_sections = n_sec;
                if (write_all(ws->fd, &tt, sizeof tt) || write_all(ws->fd, sec, sizeof(wsection_t) * (size_t)n_sec)) { MMH_ERROR("%s", "Could not send the sections to the parent process"); exit(EXIT_FAILURE); }
                free(sec);
            }
            close(ws->fd);
            free(mine); free(fwd); free(rank);
            mm_freq_destroy(h);
            if (hv) mm_freq_destroy(hv);
            mmh_tie_destroy(tie); mm_tie_destroy(dtie);
            close_loaders(ld, dl, own_pool); bz_stop(bz);
            mm_bam_hdr_free(&hdr0);
            return 0;
        }
        mm_row_t *ordered = NULL;
        if (replay && nrows > 0) {   /* the same rows, in the order the reference's table and sort leave them in */
            double tr = mmh_realtime();
            ordered = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)nrows);
            int ord_rc = -1;
            if (ordered && dev_replay) {
                ord_rc = g_dev_replay_gave_up ? -1 : mm_tie_order_rows2(dtie, rows, nrows, NULL, ordered);
                if (ord_rc != 0) rerun_with_host_replay("the final ordering", mm_tie_failed(dtie));   /* (does not return) */
                else {
                    uint64_t ts[8];
                    (void)mm_tie_last_stats(ts);
                    fprintf(stderr, "[%s] tie order on the device: %ld rows, %lu kernel launches, %lu growths of the core table in %lu passes, %lu placement rounds, %lu sort levels, %.1f ms\n", __func__, (long)nrows,
                            (unsigned long)ts[0], (unsigned long)ts[1], (unsigned long)ts[2], (unsigned long)ts[3], (unsigned long)ts[4], ts[6] / 1000.0);
                }
            } else if (ordered) {
                memcpy(ordered, rows, sizeof(mm_row_t) * (size_t)nrows);
                ord_rc = mmh_tie_order_rows_mt(tie, pool, ordered, nrows);
            }
            if (!ordered || ord_rc != 0) {   /* (the host's replay has no limit but memory) */
                MMH_ERROR("%s", "The order of minimod's hash table could not be replayed for this input (out of memory?). --canonical-order prints rows that tie on (contig, start) by strand, code, ins_offset, haplotype instead");
                exit(EXIT_FAILURE);
            } else rows = ordered;
            replay_time += mmh_realtime() - tr;
            sort_time += mmh_realtime() - tr;
        }
        double to = mmh_realtime();
        mm_row_t *inside = NULL;
        if (o.region && nrows > 0) {   /* (ordered first, over every key the run made: the order of what is left is the run's) */
            inside = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)nrows);
            if (!inside) { MMH_ERROR("%s", "Out of memory"); exit(EXIT_FAILURE); }
            int64_t k = 0;
            for (int64_t i = 0; i < nrows; i++) if (rows[i].tid == o.region_tid && (int64_t)rows[i].pos >= o.region_beg && (int64_t)rows[i].pos < o.region_end) inside[k++] = rows[i];
            fprintf(stderr, "[%s] --region %s: %ld of the %ld rows the region's reads made lie inside it\n", __func__, o.region, (long)k, (long)nrows);
            rows = inside; nrows = k;
        }
        const char *codes[MM_MAX_CODES];
        int n_codes = code_names(h, codes);
        if (header_late) mmh_print_freq_header(o.out, o.bedmethyl, o.insertions, o.haplotypes);
        print_freq_rows_any(o.out, pool, rows, drows, nrows, hdr, codes, n_codes, o.bedmethyl, o.insertions, o.haplotypes, o.device);
        if (mmh_emit_flush() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
        output_time += mmh_realtime() - to;
        if (fmt_device_rows) fprintf(stderr, "[%s] %ld rows formatted on the device (k_fmt_len + scan + k_fmt_write: %.3f ms)\n", __func__, (long)fmt_device_rows, fmt_device_ms);
        free(inside);
        free(ordered);
    }
    if (replay) fprintf(stderr, "[%s] Row order replay (the reference's hash table and sort, %s): %.3f sec (%.3f of them waiting for the calls of the second handle's launches)\n", __func__, dev_replay ? "on the device" : "on the host", replay_time, replay_fetch_seconds);
    if (mmh_emit_finish() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
    if (o.out != stdout) fclose(o.out);
    else fflush(stdout);
    if (view && ws->fd >= 0) {   /* a view worker of --devices: its rows are in its part file; the parent wants the totals */
        wtotals_t tt;
        memset(&tt, 0, sizeof tt);
        tt.total_reads = T.total_reads; tt.total_bytes = T.total_bytes; tt.processed_reads = T.processed_reads;
        tt.processed_bytes = T.processed_bytes; tt.processed_bases = T.processed_bases;
        tt.load_time = load_time; tt.wait_time = process_wait_time;
        if (write_all(ws->fd, &tt, sizeof tt)) { MMH_ERROR("%s", "Could not send the totals to the parent process"); exit(EXIT_FAILURE); }
        close(ws->fd);
        mm_freq_destroy(h);
        close_loaders(ld, dl, own_pool); bz_stop(bz);
        mm_bam_hdr_free(&hdr0);
        return 0;
    }

    tl_mark(realtime0, "output written");
    fprintf(stderr, "[%s] total entries: %ld", __func__, (long)T.total_reads);
    fprintf(stderr, "\n[%s] total bytes: %.1f M", __func__, T.total_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total skipped entries: %ld", __func__, (long)(T.total_reads - T.processed_reads));
    fprintf(stderr, "\n[%s] total skipped bytes: %.1f M", __func__, (T.total_bytes - T.processed_bytes) / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed entries: %ld", __func__, (long)T.processed_reads);
    fprintf(stderr, "\n[%s] total processed bytes: %.1f M", __func__, T.processed_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed bases: %.1f M", __func__, T.processed_bases / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] Data loading time: %.3f sec", __func__, load_time);
    fprintf(stderr, "\n[%s] Batch hand-over time: %.3f sec (mm_freq_submit: host -> device copies queued, launches)", __func__, submit_time);
    fprintf(stderr, "\n[%s] Data processing time: %.3f sec (waiting for the GPU; the rest overlaps loading)", __func__, process_wait_time);
    fprintf(stderr, "\n[%s] Data merging time: %.3f sec", __func__, 0.0);
    fprintf(stderr, "\n[%s] Data sorting time: %.3f sec", __func__, sort_time);
    fprintf(stderr, "\n[%s] Data output time: %.3f sec", __func__, output_time);
    {
        uint64_t lc[4] = {0, 0, 0, 0};
        (void)mm_freq_launch_counts(h, lc);
        fprintf(stderr, "\n[%s] GPU launches: %lu for %lu batches (%lu with k_stream_reads)", __func__, (unsigned long)lc[0], (unsigned long)lc[2], (unsigned long)lc[1]);
    }
    fprintf(stderr, "\n");
    /* Everything the run had to say has been written and flushed.  What is left -- freeing device memory, unpinning buffers, joining
     * threads, and behind main() the HIP runtime's own exit handlers -- took 0.2 s of a 1.4 s run (12 Gbases), and the process's death
     * does it all anyway: main() leaves with _exit() unless MM_FULL_TEARDOWN is set (leak checks, sanitizers). */
    if (!getenv("MM_FULL_TEARDOWN")) { bz_report(bz); tl_mark(realtime0, "teardown left to the process's exit"); return 0; }   /* (a --devices worker too: it leaves with _exit()) */
    tl_mark(realtime0, "teardown starts");
    mm_freq_destroy(h);
    if (hv) mm_freq_destroy(hv);
    tl_mark(realtime0, "handles destroyed");
    mmh_tie_destroy(tie); mm_tie_destroy(dtie);
    close_loaders(ld, dl, own_pool);
    tl_mark(realtime0, "loader closed");
    bz_stop(bz);
    mm_bam_hdr_free(&hdr0);
    tl_mark(realtime0, "teardown done");
    return 0;
}


/* ---- `minimod freq --devices a,b,...`: the genome (BAM-header contigs that the FASTA has, in header order) is cut into one
 * contiguous share per GPU at 64 kb-aligned positions; a read belongs to the share its start
 * position lies in.  The parent forks one worker per GPU BEFORE anything touches HIP (it never does itself); a worker
 * opens the BAM at the virtual offset the .bai gives for its share, counts into dense planes over its intervals (a halo
 * behind a cut inside a contig) and sends its rows to the parent, which adds up the rows both neighbours have for the
 * positions behind a cut, puts the contigs in output order and prints. */
static int row_key_cmp(const mm_row_t *a, const mm_row_t *b) {   /* within one contig: mm_freq_finalize's order */
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    if (a->strand != b->strand) return a->strand < b->strand ? -1 : 1;
    if (a->code != b->code) return a->code < b->code ? -1 : 1;
    if (a->ins_offset != b->ins_offset) return a->ins_offset < b->ins_offset ? -1 : 1;
    int ha = a->hp < 0 ? 100000 : a->hp, hb = b->hp < 0 ? 100000 : b->hp;
    return ha < hb ? -1 : (ha > hb);
}

static int copy_bytes(FILE *in, int64_t off, int64_t len, FILE *out) {
    static char buf[1 << 20];
    if (fseeko(in, (off_t)off, SEEK_SET) != 0) return -1;
    while (len > 0) {
        size_t want = len > (int64_t)sizeof buf ? sizeof buf : (size_t)len;
        size_t got = fread(buf, 1, want, in);
        if (got == 0) return -1;
        if (fwrite(buf, 1, got, out) != got) return -1;
        len -= (int64_t)got;
    }
    return 0;
}

static int run_devices(const fopt_t *o, const mmh_mods_t *mods, mmh_ref_t *ref, const char *bam_file, double realtime0) {
    int star_ctx = 0;
    for (int i = 0; i < mods->n_mods; i++) {
        if (strcmp(mods->code[i], "*") == 0) { MMH_ERROR("%s", "--devices cannot be combined with the wildcard code -c '*' (code indices are per worker)"); exit(EXIT_FAILURE); }
        if (strcmp(mods->context[i], "*") == 0) star_ctx = 1;
    }
    int dev[MMH_MAX_DEVICES], nd = 0;
    for (const char *p = o->devices;;) {   /* ordinals separated by commas; one GPU may be listed more than once (a worker each) */
        char *end = NULL;
        errno = 0;
        long v = strtol(p, &end, 10);
        if (end == p || errno || v < 0 || v > 1023 || (*end != ',' && *end != 0)) { MMH_ERROR("--devices takes GPU ordinals separated by commas, e.g. 0,1,2,3. You entered %s", o->devices); exit(EXIT_FAILURE); }
        if (nd >= MMH_MAX_DEVICES) { MMH_ERROR("--devices takes at most %d GPUs", MMH_MAX_DEVICES); exit(EXIT_FAILURE); }
        dev[nd++] = (int)v;
        if (*end == 0) break;
        p = end + 1;
    }
    if (nd < 2) { MMH_ERROR("%s", "--devices needs at least two GPUs"); exit(EXIT_FAILURE); }
    char bai_path[4096];
    snprintf(bai_path, sizeof bai_path, "%s.bai", bam_file);
    mm_bai_t *bai = mm_bai_load(bai_path);
    if (!bai) { MMH_ERROR("Could not read the index %s (--devices needs it: samtools index reads.bam)", bai_path); exit(EXIT_FAILURE); }
    mm_bam_t *hb = mm_bam_open(bam_file, 1);
    if (!hb) { MMH_ERROR("NULL returned: could not open or parse %s.", bam_file); exit(EXIT_FAILURE); }
    const mm_bam_hdr_t *hdr = mm_bam_header(hb);
    const int nt = hdr->n_targets;
    /* the genome to share: contigs the FASTA has; the others cannot carry reads (src/mod.c:793) */
    int64_t *off = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
    int *has = (int *)calloc((size_t)nt + 1, sizeof(int));
    int64_t total = 0;
    for (int t = 0; t < nt; t++) {
        off[t] = total;
        has[t] = mmh_ref_find(ref, hdr->target_name[t]) >= 0;
        if (has[t]) total += hdr->target_len[t];
    }
    off[nt] = total;
    if (total <= 0) { MMH_ERROR("%s", "No contig of the BAM header is in the reference"); exit(EXIT_FAILURE); }
    /* rows that tie on (contig, start) in the reference's order: the parent replays it from the workers' key sequences */
    const int tied = !o->view && !o->canonical_order && (mods->n_mods > 1 || star_ctx || o->insertions || o->haplotypes);
    wspec_t *ws = (wspec_t *)calloc((size_t)nd, sizeof(wspec_t));
    for (int r = 0; r < nd; r++) {
        wspec_t *w = &ws[r];
        int64_t lo = r == 0 ? 0 : (total / nd * r) / MMH_SHARE_ALIGN * MMH_SHARE_ALIGN;
        int64_t hi = r == nd - 1 ? total : (total / nd * (r + 1)) / MMH_SHARE_ALIGN * MMH_SHARE_ALIGN;
        w->sharded = 1; w->first = r == 0; w->last = r == nd - 1; w->fd = -1; w->tied = tied;
        w->slab_in = w->slab_out = w->rows_in = w->rows_out = -1;
        w->lo_tid = -1; w->hi_tid = nt; w->hi_pos = 0;
        for (int t = 0; t < nt; t++) {
            if (!has[t]) continue;
            int64_t b = lo > off[t] ? lo : off[t], e = hi < off[t] + hdr->target_len[t] ? hi : off[t] + (int64_t)hdr->target_len[t];
            if (b >= e) continue;
            if (w->n_iv >= 64) { MMH_ERROR("%s", "too many contigs in one share"); exit(EXIT_FAILURE); }
            mm_interval_t *iv = &w->iv[w->n_iv++];
            iv->tid = t; iv->begin = b - off[t]; iv->end = e - off[t]; iv->halo = 0;
            if (w->lo_tid < 0) { w->lo_tid = t; w->lo_pos = iv->begin; }
            w->hi_tid = t; w->hi_pos = iv->end;
        }
        if (w->n_iv == 0) { w->lo_tid = nt; w->lo_pos = 0; w->hi_tid = nt; w->hi_pos = 0; w->voffset = UINT64_MAX; }
        else w->voffset = w->first ? 0 : mm_bai_start(bai, w->lo_tid, w->lo_pos);
        if (w->first) { w->lo_tid = -1; w->lo_pos = 0; }
        /* a share's reads begin where the share before it stops reading (not at its own first interval): an alignment on a
         * header contig the FASTA lacks, lying between two shares, is then read by the right-hand worker, which fails on it
         * like a single run does (src/mod.c:793) */
        if (r > 0 && w->n_iv > 0) {
            int q = r - 1;
            while (q > 0 && ws[q].n_iv == 0) q--;
            if (ws[q].n_iv > 0 && !(ws[q].hi_tid == w->lo_tid && ws[q].hi_pos == w->lo_pos)) {
                w->lo_tid = ws[q].hi_tid; w->lo_pos = ws[q].hi_pos;
                w->voffset = mm_bai_start(bai, w->lo_tid, w->lo_pos);
            }
        }
    }
    mm_bai_free(bai);
    /* halos: dense counters behind a cut that falls inside a contig, as far as the NEXT share's piece of that contig goes (at
     * most MMH_SHARE_HALO): the slab then lies inside the neighbour's own counters; what a read calls further right is a side row */
    int live[MMH_MAX_DEVICES], nlive = 0;
    for (int r = 0; r < nd; r++) if (ws[r].n_iv > 0 || ws[r].last) live[nlive++] = r;
    for (int k = 0; k + 1 < nlive; k++) {
        wspec_t *w = &ws[live[k]], *nx = &ws[live[k + 1]];
        if (w->n_iv == 0 || nx->n_iv == 0) continue;
        mm_interval_t *iv = &w->iv[w->n_iv - 1];
        if (iv->end < (int64_t)hdr->target_len[iv->tid] && nx->iv[0].tid == iv->tid && nx->iv[0].begin == iv->end) {
            int64_t room = nx->iv[0].end - nx->iv[0].begin;
            iv->halo = room < MMH_SHARE_HALO ? room : MMH_SHARE_HALO;
        }
    }
    /* the workers' own text goes to files next to each other in a scratch directory, put together by the parent */
    char tmpl[384];
    const char *td = getenv("TMPDIR");
    snprintf(tmpl, sizeof tmpl, "%s/minimod_devices_XXXXXX", td && *td ? td : "/tmp");
    if (!mkdtemp(tmpl)) { MMH_ERROR("Cannot create a scratch directory %s", tmpl); exit(EXIT_FAILURE); }
    for (int r = 0; r < nd; r++) snprintf(ws[r].part_path, sizeof ws[r].part_path, "%s/part_%03d", tmpl, r);
    /* neighbour pipes between the workers that have something to do */
    int slab_pipe[MMH_MAX_DEVICES][2], rows_pipe[MMH_MAX_DEVICES][2];
    for (int k = 0; k + 1 < nlive; k++) {
        /* (the slab's channel runs both ways -- the receiver answers -- so it is a socket pair: [1] the left worker's end, [0] the right one's) */
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, slab_pipe[k]) != 0 || pipe(rows_pipe[k]) != 0) { MMH_ERROR("%s", "pipe failed"); exit(EXIT_FAILURE); }
        if (!o->view) {
            ws[live[k]].slab_out = slab_pipe[k][1]; ws[live[k + 1]].slab_in = slab_pipe[k][0];
            ws[live[k]].rows_out = rows_pipe[k][1]; ws[live[k + 1]].rows_in = rows_pipe[k][0];
        }
    }
    /* the workers: forked before anything in this process has touched HIP */
    pid_t pid[MMH_MAX_DEVICES];
    int rfd[MMH_MAX_DEVICES];
    fflush(NULL);
    for (int r = 0; r < nd; r++) {
        int pp[2];
        if (pipe(pp) != 0) { MMH_ERROR("%s", "pipe failed"); exit(EXIT_FAILURE); }
        pid[r] = fork();
        if (pid[r] < 0) { MMH_ERROR("%s", "fork failed"); exit(EXIT_FAILURE); }
        if (pid[r] == 0) {
            close(pp[0]);
            for (int q = 0; q < r; q++) close(rfd[q]);
            for (int k = 0; k + 1 < nlive; k++) {   /* only this worker's ends of the neighbour pipes stay open */
                if (slab_pipe[k][1] != ws[r].slab_out) close(slab_pipe[k][1]);
                if (slab_pipe[k][0] != ws[r].slab_in) close(slab_pipe[k][0]);
                if (rows_pipe[k][1] != ws[r].rows_out) close(rows_pipe[k][1]);
                if (rows_pipe[k][0] != ws[r].rows_in) close(rows_pipe[k][0]);
            }
            fopt_t wo = *o;
            wo.device = dev[r];
            wo.threads = o->threads / nd > 0 ? o->threads / nd : 1;
            ws[r].fd = pp[1];
            g_is_worker = 1;
            if (o->view) {   /* a view worker prints its share's rows into its part file, no header */
                wo.out = fopen(ws[r].part_path, "wb");
                if (!wo.out) { MMH_ERROR("Cannot open file %s for writing", ws[r].part_path); _exit(EXIT_FAILURE); }
            }
            int rc = run_body(&wo, mods, ref, bam_file, realtime0, &ws[r]);
            fflush(NULL);
            mmh_leave_teardown_behind();   /* (the parent's wait for this worker ends with its last word, not with the kernel's clearing away of its address space) */
            _exit(rc);
        }
        close(pp[1]);
        rfd[r] = pp[0];
    }
    for (int k = 0; k + 1 < nlive; k++) { close(slab_pipe[k][0]); close(slab_pipe[k][1]); close(rows_pipe[k][0]); close(rows_pipe[k][1]); }
    mmh_free_ref(ref);
    /* what every worker reports, in rank order */
    wtotals_t tot[MMH_MAX_DEVICES];
    mm_row_t *wrows[MMH_MAX_DEVICES];
    wsection_t *wsec[MMH_MAX_DEVICES];
    void *wtk[MMH_MAX_DEVICES]; uint32_t *wth[MMH_MAX_DEVICES];
    int failed = 0;
    for (int r = 0; r < nd; r++) {
        wrows[r] = NULL; wsec[r] = NULL; wtk[r] = NULL; wth[r] = NULL;
        memset(&tot[r], 0, sizeof tot[r]);
        if (read_all(rfd[r], &tot[r], sizeof tot[r]) != 0) failed = 1;
        else {
            if (tot[r].n_rows > 0) {
                wrows[r] = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)tot[r].n_rows);
                if (!wrows[r] || read_all(rfd[r], wrows[r], sizeof(mm_row_t) * (size_t)tot[r].n_rows) != 0) failed = 1;
            }
            if (!failed && tot[r].n_tie_keys > 0) {
                wtk[r] = malloc(16 * (size_t)tot[r].n_tie_keys); wth[r] = (uint32_t *)malloc(4 * (size_t)tot[r].n_tie_keys);
                if (!wtk[r] || !wth[r] || read_all(rfd[r], wtk[r], 16 * (size_t)tot[r].n_tie_keys) != 0 || read_all(rfd[r], wth[r], 4 * (size_t)tot[r].n_tie_keys) != 0) failed = 1;
            }
            if (!failed && tot[r].n_sections > 0) {
                wsec[r] = (wsection_t *)malloc(sizeof(wsection_t) * (size_t)tot[r].n_sections);
                if (!wsec[r] || read_all(rfd[r], wsec[r], sizeof(wsection_t) * (size_t)tot[r].n_sections) != 0) failed = 1;
            }
        }
        close(rfd[r]);
    }
    for (int r = 0; r < nd; r++) {
        int st = 0;
        if (waitpid(pid[r], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (WIFSIGNALED(st)) fprintf(stderr, "[%s] worker %d (device %d) was ended by signal %d\n", __func__, r, dev[r], WTERMSIG(st));
            else if (WIFEXITED(st)) fprintf(stderr, "[%s] worker %d (device %d) exited with status %d\n", __func__, r, dev[r], WEXITSTATUS(st));
            failed = 1;
        }
    }
    if (failed) { MMH_ERROR("%s", "A worker of --devices failed"); exit(EXIT_FAILURE); }
    double ts = mmh_realtime(), sort_time = 0, output_time = 0;
    int *order = (int *)malloc(sizeof(int) * (size_t)(nt > 0 ? nt : 1));
    for (int t = 0; t < nt; t++) order[t] = t;
    for (int a = 1; a < nt; a++) {   /* contigs by name like cmp_key_fast (src/mod.c:59-76); insertion sort keeps header order among equal names */
        int x = order[a], b = a - 1;
        while (b >= 0 && strcmp(hdr->target_name[order[b]], hdr->target_name[x]) > 0) { order[b + 1] = order[b]; b--; }
        order[b + 1] = x;
    }
    if (o->view) {
        /* every read is one worker's: the parts in rank order are the rows in file order */
        mmh_print_view_header(o->out, o->insertions, o->haplotypes);
        for (int r = 0; r < nd && !failed; r++) {
            FILE *pf = fopen(ws[r].part_path, "rb");
            if (!pf) continue;   /* (a worker with nothing to do wrote nothing) */
            fseeko(pf, 0, SEEK_END);
            const int64_t len = (int64_t)ftello(pf);
            if (len > 0 && copy_bytes(pf, 0, len, o->out) != 0) failed = 1;
            fclose(pf);
        }
    } else if (!tied) {
        /* the workers' sections: a contig's rows are its pieces in rank order (= position order; no two workers hold the same key) */
        mmh_print_freq_header(o->out, o->bedmethyl, o->insertions, o->haplotypes);
        FILE *pf[MMH_MAX_DEVICES];
        for (int r = 0; r < nd; r++) pf[r] = tot[r].n_sections > 0 ? fopen(ws[r].part_path, "rb") : NULL;
        for (int k = 0; k < nt && !failed; k++)
            for (int r = 0; r < nd && !failed; r++)
                for (int64_t i = 0; i < tot[r].n_sections; i++)
                    if (wsec[r][i].tid == order[k] && wsec[r][i].len > 0 && (!pf[r] || copy_bytes(pf[r], wsec[r][i].off, wsec[r][i].len, o->out) != 0)) failed = 1;
        for (int r = 0; r < nd; r++) if (pf[r]) fclose(pf[r]);
    } else {
        /* tied rows in the reference's order: the first-insertion sequence of the whole run is the workers' sequences one after
         * the other (a worker's reads all lie in front of the next worker's in the file, src/mod.c:743-774); rows: a contig's
         * pieces in rank order */
        int64_t n_all = 0;
        for (int r = 0; r < nd; r++) n_all += tot[r].n_rows;
        mm_row_t *out_rows = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)(n_all > 0 ? n_all : 1));
        int64_t n_out = 0;
        for (int k = 0; k < nt; k++)
            for (int r = 0; r < nd; r++) {
                int64_t a = 0, n = tot[r].n_rows;
                while (a < n && wrows[r][a].tid != order[k]) a++;
                int64_t b = a;
                while (b < n && wrows[r][b].tid == order[k]) b++;
                if (b > a) { memcpy(out_rows + n_out, wrows[r] + a, sizeof(mm_row_t) * (size_t)(b - a)); n_out += b - a; }
            }
        mmh_tie_t *tie = mmh_tie_create(hdr, o->insertions, o->haplotypes);
        int ok = tie != NULL;
        for (int r = 0; r < nd && ok; r++) {
            if (tot[r].n_tie_keys < 0) ok = 0;
            else if (tot[r].n_tie_keys > 0 && mmh_tie_import2(tie, wtk[r], wth[r], tot[r].n_tie_keys, (int)tot[r].tie_put_after) != 0) ok = 0;
        }
        if (!ok && !o->host_replay && !getenv("MINIMOD_HOST_REPLAY")) rerun_with_host_replay("a worker of --devices", 0u);   /* (does not return; nothing is printed yet) */
        if (!ok || mmh_tie_order_rows(tie, out_rows, n_out) != 0) {
            MMH_ERROR("%s", "The order of minimod's hash table could not be replayed for this input (out of memory?). --canonical-order prints rows that tie on (contig, start) by strand, code, ins_offset, haplotype instead");
            exit(EXIT_FAILURE);
        }
        mmh_tie_destroy(tie);
        sort_time = mmh_realtime() - ts;
        double to = mmh_realtime();
        mm_pool_t *pool = mm_pool_create(o->threads);
        const char *codes[MM_MAX_MODS];
        for (int i = 0; i < mods->n_mods; i++) codes[i] = mods->code[i];
        mmh_print_freq_header(o->out, o->bedmethyl, o->insertions, o->haplotypes);
        mmh_print_freq_rows(o->out, pool, out_rows, n_out, hdr, codes, mods->n_mods, o->bedmethyl, o->insertions, o->haplotypes);
        if (mmh_emit_finish() != 0) failed = 1;
        mm_pool_destroy(pool);
        free(out_rows);
        output_time = mmh_realtime() - to;
    }
    if (!tied || o->view) output_time = mmh_realtime() - ts;
    for (int r = 0; r < nd; r++) { unlink(ws[r].part_path); free(wrows[r]); free(wsec[r]); free(wtk[r]); free(wth[r]); }
    rmdir(tmpl);
    if (failed) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
    if (o->out != stdout) fclose(o->out);
    else fflush(stdout);
    wtotals_t s;
    memset(&s, 0, sizeof s);
    for (int r = 0; r < nd; r++) {
        s.total_reads += tot[r].total_reads; s.total_bytes += tot[r].total_bytes; s.processed_reads += tot[r].processed_reads;
        s.processed_bytes += tot[r].processed_bytes; s.processed_bases += tot[r].processed_bases;
        if (tot[r].load_time > s.load_time) s.load_time = tot[r].load_time;
        if (tot[r].wait_time > s.wait_time) s.wait_time = tot[r].wait_time;
        if (tot[r].sort_time > s.sort_time) s.sort_time = tot[r].sort_time;
    }
    fprintf(stderr, "[%s] devices: %d (one worker process each, shares of %.1f Mb)", __func__, nd, total / (double)nd / 1e6);
    for (int r = 0; r < nd; r++)
        fprintf(stderr, "\n[%s] worker %d (device %d): %ld entries, %.1f Mbases, loading %.3f sec, waiting for the GPU %.3f sec, finalize %.3f sec", __func__, r, dev[r],
                (long)tot[r].processed_reads, tot[r].processed_bases / 1e6, tot[r].load_time, tot[r].wait_time, tot[r].sort_time);
    fprintf(stderr, "\n[%s] total entries: %ld", __func__, (long)s.total_reads);
    fprintf(stderr, "\n[%s] total bytes: %.1f M", __func__, s.total_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total skipped entries: %ld", __func__, (long)(s.total_reads - s.processed_reads));
    fprintf(stderr, "\n[%s] total skipped bytes: %.1f M", __func__, (s.total_bytes - s.processed_bytes) / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed entries: %ld", __func__, (long)s.processed_reads);
    fprintf(stderr, "\n[%s] total processed bytes: %.1f M", __func__, s.processed_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed bases: %.1f M", __func__, s.processed_bases / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] Data loading time: %.3f sec (slowest worker)", __func__, s.load_time);
    fprintf(stderr, "\n[%s] Data processing time: %.3f sec (slowest worker waiting for its GPU)", __func__, s.wait_time);
    fprintf(stderr, "\n[%s] Data merging time: %.3f sec (%s)", __func__, sort_time, tied ? "the workers' key sequences replayed, rows put in the reference's order" : "nothing to merge: halo slabs and side rows went from neighbour to neighbour");
    fprintf(stderr, "\n[%s] Data sorting time: %.3f sec (slowest worker's finalize)", __func__, s.sort_time);
    fprintf(stderr, "\n[%s] Data output time: %.3f sec (%s)", __func__, output_time, tied ? "formatted by the parent" : "the workers' sections put in order");
    fprintf(stderr, "\n");
    (void)realtime0;
    free(order); free(off); free(has); free(ws);
    mm_bam_close(hb);
    return 0;
}

static int run_main(int argc, char **argv, int view) {
    double realtime0 = mmh_realtime();
    g_run_argc = argc; g_run_argv = argv;   /* (getopt_long permutes argv in place: still the same command line) */
    const char *optstring = view ? "c:t:B:K:v:p:o:hV" : "m:c:t:B:K:v:p:o:hVb";   /* src/view_main.c:168, src/freq_main.c:185 */
    const struct option *lopts = view ? view_long_options : long_options;
    int longindex = 0, c;
    FILE *fp_help = stderr;
    fopt_t o;
    memset(&o, 0, sizeof(o));
    o.K = 512; o.B = 20 * 1000 * 1000; o.threads = 8; o.debug_break = -1; o.out = stdout;   /* init_opt, src/minimod.c:485-513 */
    o.view = view; o.gather = MMH_MAX_GATHER; o.gpu_inflate = -1; o.gpu_ingest = -1;   /* (--gather: as many batches as the staging takes -- a launch is sized by its bases, not by -K) */
    while ((c = getopt_long(argc, argv, optstring, lopts, &longindex)) >= 0) {
        const char *lname = c == 0 ? lopts[longindex].name : "";
        if (c == 'B') {
            o.B = mmh_parse_num(optarg);
            if (o.B <= 0) { MMH_ERROR("%s", "Maximum number of bases should be larger than 0."); exit(EXIT_FAILURE); }
        } else if (c == 'K') {
            o.K = atoi(optarg);
            if (o.K < 1) { MMH_ERROR("Batch size should larger than 0. You entered %d", o.K); exit(EXIT_FAILURE); }
        } else if (c == 't') {
            o.threads = atoi(optarg);
            if (o.threads < 1) { MMH_ERROR("Number of threads should larger than 0. You entered %d", o.threads); exit(EXIT_FAILURE); }
        } else if (c == 'v') {
            mmh_log_level = atoi(optarg);
        } else if (c == 'p') {
            if (atoi(optarg) < 0) { MMH_ERROR("Progress interval should be 0 or positive. You entered %d", atoi(optarg)); exit(EXIT_FAILURE); }
            o.progress_interval = atoi(optarg);
        } else if (c == 'o') {
            FILE *fp = fopen(optarg, "w");
            if (fp == NULL) { MMH_ERROR("Cannot open file %s for writing", optarg); exit(EXIT_FAILURE); }
            o.out_path = optarg; o.out = fp;
        } else if (c == 'V') {
            fprintf(stdout, "minimod %s\n", MMH_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'm') {
            o.threshes = optarg;
        } else if (c == 'c') {
            o.codes = optarg;
        } else if (c == 'b') {
            o.bedmethyl = 1;
        } else if (c == 0 && strcmp(lname, "debug-break") == 0) { o.debug_break = atoi(optarg);
        } else if (c == 0 && strcmp(lname, "insertions") == 0) { o.insertions = 1;
        } else if (c == 0 && strcmp(lname, "haplotypes") == 0) { o.haplotypes = 1;
        } else if (c == 0 && strcmp(lname, "allow-secondary") == 0) { o.allow_secondary = 1;
        } else if (c == 0 && strcmp(lname, "include-non-ref") == 0) { /* accepted and ignored like the reference */
        } else if (c == 0 && strcmp(lname, "skip-supplementary") == 0) { o.skip_supplementary = 1;
        } else if (c == 0 && strcmp(lname, "device") == 0) { o.device = atoi(optarg);
        } else if (c == 0 && strcmp(lname, "devices") == 0) { o.devices = optarg;
        } else if (c == 0 && strcmp(lname, "canonical-order") == 0) { o.canonical_order = 1;
        } else if (c == 0 && strcmp(lname, "gpu-inflate") == 0) { o.gpu_inflate = 1;
        } else if (c == 0 && strcmp(lname, "no-gpu-inflate") == 0) { o.gpu_inflate = 0;
        } else if (c == 0 && strcmp(lname, "gpu-ingest") == 0) { o.gpu_ingest = 1;
        } else if (c == 0 && strcmp(lname, "no-gpu-ingest") == 0) { o.gpu_ingest = 0;
        } else if (c == 0 && strcmp(lname, "host-replay") == 0) { o.host_replay = 1;
        } else if (c == 0 && strcmp(lname, "region") == 0) { o.region = optarg;
        } else if (c == 0 && strcmp(lname, "gather") == 0) {
            o.gather = atoi(optarg);
            if (o.gather < 1) { MMH_ERROR("--gather should be at least 1. You entered %d", o.gather); exit(EXIT_FAILURE); }
        } else {
            print_help(fp_help, &o);
            exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
        }
    }
    char err[512];
    mmh_mods_t mods;
    if (o.codes == NULL || strlen(o.codes) == 0) {
        MMH_INFO("%s", "Modification codes not provided. Using default modification code m");
        o.codes = "m";
    }
    if (mmh_parse_mod_codes(o.codes, &mods, err, sizeof err)) { MMH_ERROR("%s", err); exit(EXIT_FAILURE); }
    char defthr[MM_MAX_MODS * 4 + 1];
    if (o.threshes == NULL || strlen(o.threshes) == 0) {
        if (!view) MMH_INFO("%s", "Modification threshold not provided. Using default threshold 0.8");   /* view has no thresholds */
        defthr[0] = 0;
        for (int i = 0; i < mods.n_mods; i++) { strcat(defthr, "0.8"); if (i < mods.n_mods - 1) strcat(defthr, ","); }
        o.threshes = defthr;
    }
    if (mmh_parse_mod_threshes(o.threshes, &mods, err, sizeof err)) { MMH_ERROR("%s", err); exit(EXIT_FAILURE); }
    if (argc - optind != 2 || fp_help == stdout) {
        MMH_WARNING("%s", "Missing arguments");
        print_help(fp_help, &o);
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    const char *ref_file = argv[optind], *bam_file = argv[optind + 1];
    if (access(bam_file, F_OK) == -1) { MMH_ERROR("BAM file %s does not exist", bam_file); exit(EXIT_FAILURE); }

    tl_mark(realtime0, "options parsed");
    if (o.gpu_inflate < 0) {
        /* The device inflater costs about 0.3 s of pinned allocations and their release, and wins that back once the host threads
         * would be inflating for longer: a wash up to 3.4 GiB, 0.2 - 0.6 s faster at 5.1 GiB. */
        struct stat sb;
        int n_dev = 1;
        if (o.devices) for (const char *q = o.devices; *q; q++) if (*q == ',') n_dev++;
        o.gpu_inflate = stat(bam_file, &sb) == 0 && (int64_t)sb.st_size / n_dev >= ((int64_t)4 << 30);
    }
    if (o.gpu_ingest < 0) {
        struct stat sb;
        int n_dev = 1;
        if (o.devices) for (const char *q = o.devices; *q; q++) if (*q == ',') n_dev++;
        /* (view: from 2 GiB -- tools/view_batch_sweep.sh: on a 656-MB file the device reader's own start, pinned staging and pools, costs what its faster
         * decode wins, 0.38 - 0.66 s against the host threads' steady 0.40 s) */
        o.gpu_ingest = stat(bam_file, &sb) == 0 && (int64_t)sb.st_size / n_dev >= (view ? ((int64_t)2 << 30) : ((int64_t)512 << 20));   /* (tools/ingest_threshold.sh: the device reader's own start -- 192 MiB of pinned staging, 3 GiB of pools, their release at exit -- is 0.1 - 0.3 s by box; the host threads win below ~0.5 GiB, lose from ~1 GiB, between them it depends on the box) */
    }
    /* the HIP runtime's start (~0.2 s) beside the reference's load -- unless this process is going to fork workers (--devices a,b,...:
     * the parent must not have touched HIP) */
    pthread_t warm_thread;
    int warming = 0;
    static int warm_device;
    if (!(o.devices && strchr(o.devices, ','))) {
        warm_device = o.devices ? atoi(o.devices) : o.device;
        warming = pthread_create(&warm_thread, NULL, hip_warm_main, &warm_device) == 0;
    }
    double t1 = mmh_realtime();
    fprintf(stderr, "[%s] Loading reference genome %s\n", __func__, ref_file);
    mmh_ref_t *ref = mmh_load_ref_mt(ref_file, o.threads > 0 ? o.threads : 1);
    if (!ref) { MMH_ERROR("Could not to open file %s", ref_file); exit(EXIT_FAILURE); }
    fprintf(stderr, "[%s] Reference genome loaded in %.3f sec\n", __func__, mmh_realtime() - t1);
    if (o.devices && strchr(o.devices, ',') && o.region) { MMH_ERROR("%s", "--region and --devices a,b,... do not go together"); exit(EXIT_FAILURE); }
    if (o.devices && strchr(o.devices, ',')) return run_devices(&o, &mods, ref, bam_file, realtime0);
    if (o.devices) {   /* one ordinal: the same as --device */
        char *end = NULL;
        errno = 0;
        long v = strtol(o.devices, &end, 10);
        if (end == o.devices || errno || v < 0 || v > 1023 || *end != 0) { MMH_ERROR("--devices takes GPU ordinals separated by commas, e.g. 0,1,2,3. You entered %s", o.devices); exit(EXIT_FAILURE); }
        o.device = (int)v;
    }
    wspec_t ws;
    memset(&ws, 0, sizeof ws);
    ws.fd = -1;
    if (o.region) {
        /* a region run is a run over the reads that can reach the region -- from the virtual offset the index's
         * linear part gives for the 16 kb window of its first position (every read overlapping that window lies behind it) to the
         * first read that starts behind its end -- with the dense counters over the region alone; what those reads call outside it is
         * dropped when the rows are printed.  The counts are the whole file's for those positions (tests/test_cli_gpu.py). */
        if (view) { MMH_ERROR("%s", "--region is a freq option"); exit(EXIT_FAILURE); }
        mm_bam_hdr_t rh;
        memset(&rh, 0, sizeof rh);
        if (mm_bam_peek_header(bam_file, &rh) != 0) { MMH_ERROR("Could not read the header of %s (--region needs a regular, indexed BAM file)", bam_file); exit(EXIT_FAILURE); }
        char name[1024];
        int64_t from = 1, to = INT64_MAX;
        const char *colon = strrchr(o.region, ':');
        int ranged = 0;
        if (colon && colon[1]) {   /* digits, commas, one '-' behind the last colon: a range; anything else: part of the name */
            const char *q = colon + 1;
            int ok = 1, dash = 0;
            for (const char *z = q; *z; z++) { if (*z == '-') dash++; else if (!((*z >= '0' && *z <= '9') || *z == ',')) ok = 0; }
            if (ok && dash <= 1 && q[0] != '-') {
                ranged = 1;
                int64_t v[2] = {0, 0}; int k = 0, any[2] = {0, 0};
                for (const char *z = q; *z; z++) { if (*z == '-') k = 1; else if (*z != ',') { v[k] = v[k] * 10 + (*z - '0'); any[k] = 1; } }
                from = any[0] ? v[0] : 1;
                to = any[1] ? v[1] : INT64_MAX;
            }
        }
        const size_t nl = ranged ? (size_t)(colon - o.region) : strlen(o.region);
        if (nl == 0 || nl >= sizeof name) { MMH_ERROR("--region takes chr:from-to. You entered %s", o.region); exit(EXIT_FAILURE); }
        memcpy(name, o.region, nl); name[nl] = 0;
        int32_t tid = -1;
        for (int32_t t = 0; t < rh.n_targets; t++) if (strcmp(rh.target_name[t], name) == 0) { tid = t; break; }
        if (tid < 0) { MMH_ERROR("--region: no contig %s in the header of %s", name, bam_file); exit(EXIT_FAILURE); }
        if (from < 1) from = 1;
        if (to > (int64_t)rh.target_len[tid]) to = (int64_t)rh.target_len[tid];
        if (to < from) { MMH_ERROR("--region: an empty range (%s)", o.region); exit(EXIT_FAILURE); }
        o.region_tid = tid; o.region_beg = from - 1; o.region_end = to;
        char bai_path[4096];
        snprintf(bai_path, sizeof bai_path, "%s.bai", bam_file);
        mm_bai_t *bai = mm_bai_load(bai_path);
        if (!bai) { MMH_ERROR("Could not read the index %s (--region needs it: samtools index reads.bam)", bai_path); exit(EXIT_FAILURE); }
        ws.sharded = 1; ws.tied = 1; ws.first = 0; ws.last = 0;
        ws.n_iv = 1; ws.iv[0].tid = tid; ws.iv[0].begin = o.region_beg; ws.iv[0].end = o.region_end; ws.iv[0].halo = 0;
        ws.lo_tid = tid; ws.lo_pos = 0; ws.hi_tid = tid; ws.hi_pos = o.region_end;
        ws.voffset = mm_bai_start(bai, tid, o.region_beg);
        mm_bai_free(bai);
        mm_bam_hdr_free(&rh);
    }
    const int rc = run_body(&o, &mods, ref, bam_file, realtime0, &ws);   /* (its first HIP call waits for the runtime's start if that is still under way) */
    if (warming) pthread_join(warm_thread, NULL);
    return rc;
}

int mmh_freq_main(int argc, char **argv) { return run_main(argc, argv, 0); }
int mmh_view_main(int argc, char **argv) { return run_main(argc, argv, 1); }
