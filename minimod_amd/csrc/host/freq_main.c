/* freq_main.c -- `minimod freq` and `minimod view` with the hot path on the MI355X library.  Same options, defaults,
 * progress lines, output and exit behaviour as the reference's freq_main (src/freq_main.c:46-64 option table, :166-519
 * driver) and view_main (src/view_main.c:46-63, :166-470), with load(N+1) overlapping process(N) like their 3-stage
 * pipeline (freq_main.c:404-474): the batch is handed to mm_freq_submit (H2D + kernels, asynchronous) and the next
 * batch is decoded meanwhile.  view prints a batch's rows when the batch is retired (print_view_output per db_t,
 * src/view_main.c:142-160); freq prints once at the end. */
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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
    {0, 0, 0, 0}};

typedef struct {
    int32_t K; int64_t B; int threads, debug_break, bedmethyl, insertions, haplotypes, allow_secondary, skip_supplementary;
    int progress_interval, device, view;
    const char *codes, *threshes, *out_path;
    FILE *out;
} fopt_t;

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
}

/* the reference's message for a per-read device status (src/mod.c line in brackets), then exit(1) like it does */
static void die_read_error(int code, int32_t read, const mm_batch_t *b, const mm_bam_hdr_t *hdr) {
    (void)mmh_emit_flush();   /* rows of earlier batches reach the output as they did with stdio */
    const mm_read_t *rd = (read >= 0 && read < b->n_reads) ? &b->reads[read] : NULL;
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

/* merge_db's place in the pipeline: wait for the batch; freq has nothing to merge, view prints the batch's rows
 * (print_view_output, src/mod.c:560-626).  `pool_set` is the loader pool set the batch was read into. */
static int code_names(mm_freq_t *h, const char **codes) {
    int n = mm_freq_n_codes(h);
    if (n > MM_MAX_CODES) n = MM_MAX_CODES;
    for (int i = 0; i < n; i++) codes[i] = mm_freq_code_name(h, i);
    return n;
}

static void retire_batch(mm_freq_t *h, int32_t ticket, const mm_batch_t *b, const mmh_loader_t *ld, int pool_set, const mm_bam_hdr_t *hdr, const fopt_t *o,
                         mm_pool_t *pool, double *wait_time, double *output_time) {
    double tw = mmh_realtime();
    int32_t bad = -1;
    if (!o->view) {
        int e = mm_freq_wait(h, ticket, &bad);
        *wait_time += mmh_realtime() - tw;
        if (e) die_read_error(e, bad, b, hdr);
        return;
    }
    const mm_view_row_t *rows = NULL;
    int64_t n = mm_view_fetch(h, ticket, &rows, &bad);
    *wait_time += mmh_realtime() - tw;
    if (n < 0) die_read_error((int)-n, bad, b, hdr);
    double to = mmh_realtime();
    const char *codes[MM_MAX_CODES];
    int n_codes = code_names(h, codes);
    mmh_print_view_rows(o->out, pool, rows, n, b, ld, pool_set, hdr, codes, n_codes, o->insertions, o->haplotypes);
    *output_time += mmh_realtime() - to;
}

static int run_main(int argc, char **argv, int view) {
    double realtime0 = mmh_realtime();
    const char *optstring = view ? "c:t:B:K:v:p:o:hV" : "m:c:t:B:K:v:p:o:hVb";   /* src/view_main.c:168, src/freq_main.c:185 */
    const struct option *lopts = view ? view_long_options : long_options;
    int longindex = 0, c;
    FILE *fp_help = stderr;
    fopt_t o;
    memset(&o, 0, sizeof(o));
    o.K = 512; o.B = 20 * 1000 * 1000; o.threads = 8; o.debug_break = -1; o.out = stdout;   /* init_opt, src/minimod.c:485-513 */
    o.view = view;
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

    double t1 = mmh_realtime();
    fprintf(stderr, "[%s] Loading reference genome %s\n", __func__, ref_file);
    mmh_ref_t *ref = mmh_load_ref(ref_file);
    if (!ref) { MMH_ERROR("Could not to open file %s", ref_file); exit(EXIT_FAILURE); }
    fprintf(stderr, "[%s] Reference genome loaded in %.3f sec\n", __func__, mmh_realtime() - t1);

    mmh_loader_t *ld = mmh_loader_open(bam_file, o.threads, o.K, o.B, o.allow_secondary, o.skip_supplementary);
    if (!ld) { MMH_ERROR("NULL returned: could not open or parse %s.", bam_file); exit(EXIT_FAILURE); }
    const mm_bam_hdr_t *hdr = mm_bam_header(ld->bam);

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
    mm_freq_t *h = mm_freq_create(&fo, hdr->n_targets, ctg, 0, NULL, err, sizeof err);
    if (!h) { MMH_ERROR("Assertion failed. %s", err); fprintf(stderr, "Exiting.\n"); exit(EXIT_FAILURE); }
    free(ctg);
    mmh_free_ref(ref);   /* the reference now lives in HBM */
    fprintf(stderr, "[%s] Reference contexts loaded in %.3f sec\n", __func__, mmh_realtime() - t2);
    int wildcard = 0;
    for (int i = 0; i < mods.n_mods; i++) if (strcmp(mods.code[i], "*") == 0) wildcard = 1;

    if (view) mmh_print_view_header(o.out, o.insertions, o.haplotypes);
    else mmh_print_freq_header(o.out, o.bedmethyl, o.insertions, o.haplotypes);

    double load_time = 0, process_wait_time = 0, output_time = 0;
    int more = 1, counter = 0, set = 0;
    int32_t pending_ticket = -1;
    mm_batch_t pending_batch, batch;
    memset(&pending_batch, 0, sizeof pending_batch);
    double prog_t = mmh_realtime();
    while (more) {
        double tl = mmh_realtime();
        int32_t n = mmh_loader_next(ld, set, &batch, &more);
        if (n < 0) { MMH_ERROR("%s", "Truncated or corrupt BAM file"); exit(EXIT_FAILURE); }
        load_time += mmh_realtime() - tl;
        fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bases) loaded\n", __func__, mmh_realtime() - realtime0,
                mmh_cputime() / (mmh_realtime() - realtime0), n, ld->last_processed_bytes / (1000.0 * 1000.0));
        /* the previous batch's pool set is about to be reused two iterations from now: retire it first */
        if (pending_ticket >= 0) {
            retire_batch(h, pending_ticket, &pending_batch, ld, set ^ 1, hdr, &o, mm_bam_pool(ld->bam), &process_wait_time, &output_time);
            pending_ticket = -1;
        }
        if (n > 0) {
            if (wildcard) intern_batch_codes(h, &batch);
            int32_t tk = mm_freq_submit(h, &batch);
            if (tk < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror(tk)); exit(EXIT_FAILURE); }
            pending_ticket = tk; pending_batch = batch;
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
        set ^= 1;
        if (o.debug_break == counter) break;
        counter++;
    }
    if (pending_ticket >= 0) retire_batch(h, pending_ticket, &pending_batch, ld, set ^ 1, hdr, &o, mm_bam_pool(ld->bam), &process_wait_time, &output_time);
    double sort_time = 0;
    if (!view) {
        double ts = mmh_realtime();
        const mm_row_t *rows = NULL;
        int64_t nrows = mm_freq_finalize(h, &rows);
        if (nrows < 0) { MMH_ERROR("GPU path failed: %s", mm_strerror((int32_t)nrows)); exit(EXIT_FAILURE); }
        sort_time = mmh_realtime() - ts;
        double to = mmh_realtime();
        const char *codes[MM_MAX_CODES];
        int n_codes = code_names(h, codes);
        mmh_print_freq_rows(o.out, mm_bam_pool(ld->bam), rows, nrows, hdr, codes, n_codes, o.bedmethyl, o.insertions, o.haplotypes);
        if (mmh_emit_flush() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
        output_time += mmh_realtime() - to;
    }
    if (mmh_emit_finish() != 0) { MMH_ERROR("%s", "Could not write the output"); exit(EXIT_FAILURE); }
    if (o.out != stdout) fclose(o.out);
    else fflush(stdout);

    fprintf(stderr, "[%s] total entries: %ld", __func__, (long)ld->total_reads);
    fprintf(stderr, "\n[%s] total bytes: %.1f M", __func__, ld->total_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total skipped entries: %ld", __func__, (long)(ld->total_reads - ld->processed_reads));
    fprintf(stderr, "\n[%s] total skipped bytes: %.1f M", __func__, (ld->total_bytes - ld->processed_bytes) / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed entries: %ld", __func__, (long)ld->processed_reads);
    fprintf(stderr, "\n[%s] total processed bytes: %.1f M", __func__, ld->processed_bytes / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] total processed bases: %.1f M", __func__, ld->processed_bases / (float)(1000 * 1000));
    fprintf(stderr, "\n[%s] Data loading time: %.3f sec", __func__, load_time);
    fprintf(stderr, "\n[%s] Data processing time: %.3f sec (waiting for the GPU; the rest overlaps loading)", __func__, process_wait_time);
    fprintf(stderr, "\n[%s] Data merging time: %.3f sec", __func__, 0.0);
    fprintf(stderr, "\n[%s] Data sorting time: %.3f sec", __func__, sort_time);
    fprintf(stderr, "\n[%s] Data output time: %.3f sec", __func__, output_time);
    fprintf(stderr, "\n");
    mm_freq_destroy(h);
    mmh_loader_close(ld);
    return 0;
}

int mmh_freq_main(int argc, char **argv) { return run_main(argc, argv, 0); }
int mmh_view_main(int argc, char **argv) { return run_main(argc, argv, 1); }
