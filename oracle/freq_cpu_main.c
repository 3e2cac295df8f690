/*
 * freq_cpu_main.c -- `minimod freq` end to end on the CPU: the oracle (freq_oracle.c) behind the same BAM/FASTA readers,
 * batch limits and row formatter as the product CLI.  TEST INFRASTRUCTURE, NOT PRODUCT: only bench.py's cpu_baseline leg
 * and tests/ build or run it.  It is the "reference `-t N` path timed on the same host" of BASELINE.md section 3, with the
 * oracle standing in for the reference binary (which needs htslib 1.9, absent here): kind "port".
 *
 * Shape of the run = reference src/freq_main.c:404-474: load(N+1) on the main thread while batch N is processed by a
 * helper thread (process_db -> work_db on -t threads, then merge_db), output once at the end (output_core).  Prints the
 * reference's stage timers (src/freq_main.c:505-509) to stderr.
 *
 *   freq_cpu [-b] [-c codes] [-m thresholds] [-K n] [-B bytes] [-t threads] [-o file] [--insertions] [--haplotypes] ref.fa reads.bam
 */
#include <getopt.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

/* freq_oracle.c */
void *orc_create(int n_mods, const char **codes, const char **ctxs, const double *thresh, int insertions, int haplotypes, int n_contigs);
int orc_add_contig(void *h, int tid, const char *name, const uint8_t *raw, int64_t len);
int orc_name_contig(void *h, int tid, const char *name);
int orc_process(void *h, const void *reads, int n, const uint32_t *cigar, const uint8_t *seq, const uint8_t *mm, const uint8_t *ml, int n_threads);
int64_t orc_n_rows(void *h);
int64_t orc_error_read(void *h);
typedef struct { int32_t tid, pos, strand, code, ins_off, hp; uint32_t n_called, n_mod; } orc_row_t;
void orc_rows(void *h, orc_row_t *out);
int orc_n_codes(void *h);
const char *orc_code_name(void *h, int id);
double orc_merge_seconds(void *h);
void orc_destroy(void *h);

typedef struct { void *orc; mm_batch_t batch; int threads; int err; double seconds; } job_t;

static void *process_main(void *arg) {
    job_t *j = (job_t *)arg;
    double t0 = mmh_realtime();
    j->err = orc_process(j->orc, j->batch.reads, j->batch.n_reads, j->batch.cigar, j->batch.seq, j->batch.mm, j->batch.ml, j->threads);
    j->seconds = mmh_realtime() - t0;
    return NULL;
}

int main(int argc, char **argv) {
    static struct option lo[] = {{"insertions", no_argument, 0, 1000}, {"haplotypes", no_argument, 0, 1001},
                                 {"allow-secondary", no_argument, 0, 1002}, {"skip-supplementary", no_argument, 0, 1003}, {0, 0, 0, 0}};
    double realtime0 = mmh_realtime();
    int32_t K = 512; int64_t B = 20 * 1000 * 1000; int threads = 8, bed = 0, ins = 0, hap = 0, sec = 0, nosup = 0;
    const char *codes = "m", *thr = NULL, *out_path = NULL;
    int c;
    mmh_log_level = 1;
    while ((c = getopt_long(argc, argv, "bc:m:K:B:t:o:", lo, NULL)) >= 0) {
        if (c == 'b') bed = 1; else if (c == 'c') codes = optarg; else if (c == 'm') thr = optarg;
        else if (c == 'K') K = atoi(optarg); else if (c == 'B') B = mmh_parse_num(optarg); else if (c == 't') threads = atoi(optarg);
        else if (c == 'o') out_path = optarg; else if (c == 1000) ins = 1; else if (c == 1001) hap = 1; else if (c == 1002) sec = 1;
        else if (c == 1003) nosup = 1; else { fprintf(stderr, "usage: freq_cpu [options] ref.fa reads.bam\n"); return 1; }
    }
    if (argc - optind != 2 || K < 1 || B < 1 || threads < 1) { fprintf(stderr, "usage: freq_cpu [options] ref.fa reads.bam\n"); return 1; }
    char err[512];
    mmh_mods_t mods;
    if (mmh_parse_mod_codes(codes, &mods, err, sizeof err)) { fprintf(stderr, "%s\n", err); return 1; }
    char defthr[MM_MAX_MODS * 4 + 1] = "";
    if (!thr) { for (int i = 0; i < mods.n_mods; i++) { strcat(defthr, "0.8"); if (i < mods.n_mods - 1) strcat(defthr, ","); } thr = defthr; }
    if (mmh_parse_mod_threshes(thr, &mods, err, sizeof err)) { fprintf(stderr, "%s\n", err); return 1; }
    FILE *out = out_path ? fopen(out_path, "w") : stdout;
    if (!out) { fprintf(stderr, "cannot open %s\n", out_path); return 1; }

    double t1 = mmh_realtime();
    mmh_ref_t *ref = mmh_load_ref(argv[optind]);
    if (!ref) { fprintf(stderr, "cannot open %s\n", argv[optind]); return 1; }
    double ref_time = mmh_realtime() - t1;
    mmh_loader_t *ld = mmh_loader_open(argv[optind + 1], threads, K, B, sec, nosup);
    if (!ld) { fprintf(stderr, "cannot open %s\n", argv[optind + 1]); return 1; }
    const mm_bam_hdr_t *hdr = mm_bam_header(ld->bam);
    double t2 = mmh_realtime();
    const char *cs[MM_MAX_MODS], *xs[MM_MAX_MODS];
    for (int i = 0; i < mods.n_mods; i++) { cs[i] = mods.code[i]; xs[i] = mods.context[i]; }
    void *orc = orc_create(mods.n_mods, cs, xs, mods.thresh, ins, hap, hdr->n_targets);
    for (int t = 0; t < hdr->n_targets; t++) {
        orc_name_contig(orc, t, hdr->target_name[t]);
        int ri = mmh_ref_find(ref, hdr->target_name[t]);
        if (ri >= 0) {
            if ((int64_t)hdr->target_len[t] != ref->len[ri]) { fprintf(stderr, "ref_len:%lld target_len:%u for contig %s\n", (long long)ref->len[ri], hdr->target_len[t], hdr->target_name[t]); return 1; }
            orc_add_contig(orc, t, hdr->target_name[t], ref->seq[ri], ref->len[ri]);   /* load_ref_contexts, src/ref.c:177-229 */
        }
    }
    mmh_free_ref(ref);
    double ctx_time = mmh_realtime() - t2;
    mmh_print_freq_header(out, bed, ins, hap);

    double load_time = 0, process_time = 0;
    int more = 1, set = 0, have_job = 0;
    pthread_t th;
    job_t job;
    memset(&job, 0, sizeof job);
    while (more) {
        mm_batch_t batch;
        double tl = mmh_realtime();
        int32_t n = mmh_loader_next(ld, set, &batch, &more);
        if (n < 0) { fprintf(stderr, "Truncated or corrupt BAM file\n"); return 1; }
        load_time += mmh_realtime() - tl;
        if (have_job) {   /* the pool set about to be reused two iterations from now belongs to the job in flight */
            pthread_join(th, NULL);
            process_time += job.seconds;
            have_job = 0;
            if (job.err) { fprintf(stderr, "read %lld failed with status %d\n", (long long)orc_error_read(orc), job.err); return 1; }
        }
        if (n > 0) {
            job.orc = orc; job.batch = batch; job.threads = threads; job.err = 0;
            if (pthread_create(&th, NULL, process_main, &job) != 0) { fprintf(stderr, "pthread_create failed\n"); return 1; }
            have_job = 1;
        }
        set ^= 1;
    }
    if (have_job) {
        pthread_join(th, NULL);
        process_time += job.seconds;
        if (job.err) { fprintf(stderr, "read %lld failed with status %d\n", (long long)orc_error_read(orc), job.err); return 1; }
    }
    double ts = mmh_realtime();
    int64_t nrows = orc_n_rows(orc);
    orc_row_t *orows = (orc_row_t *)malloc(sizeof(orc_row_t) * (size_t)(nrows > 0 ? nrows : 1));
    mm_row_t *rows = (mm_row_t *)calloc((size_t)(nrows > 0 ? nrows : 1), sizeof(mm_row_t));
    if (!orows || !rows) { fprintf(stderr, "out of memory\n"); return 1; }
    orc_rows(orc, orows);   /* collect + order (print_freq_output up to ks_introsort, src/mod.c:644-664) */
    for (int64_t i = 0; i < nrows; i++) {
        rows[i].tid = orows[i].tid; rows[i].pos = orows[i].pos; rows[i].strand = (uint8_t)orows[i].strand;
        rows[i].ins_offset = (uint16_t)orows[i].ins_off; rows[i].code = (int16_t)orows[i].code; rows[i].hp = (int16_t)orows[i].hp;
        rows[i].n_called = orows[i].n_called; rows[i].n_mod = orows[i].n_mod;
    }
    double sort_time = mmh_realtime() - ts;
    double to = mmh_realtime();
    const char *cn[MM_MAX_CODES];
    int n_codes = orc_n_codes(orc);
    if (n_codes > MM_MAX_CODES) n_codes = MM_MAX_CODES;
    for (int i = 0; i < n_codes; i++) cn[i] = orc_code_name(orc, i);
    mmh_print_freq_rows(out, mm_bam_pool(ld->bam), rows, nrows, hdr, cn, n_codes, bed, ins, hap);
    if (mmh_emit_finish() != 0) { fprintf(stderr, "Could not write the output\n"); return 1; }
    if (out != stdout) fclose(out); else fflush(stdout);
    double output_time = mmh_realtime() - to;
    double merge_time = orc_merge_seconds(orc);
    fprintf(stderr, "[freq_cpu] threads: %d", threads);
    fprintf(stderr, "\n[freq_cpu] total processed entries: %ld", (long)ld->processed_reads);
    fprintf(stderr, "\n[freq_cpu] total processed bases: %.1f M", ld->processed_bases / (float)(1000 * 1000));
    fprintf(stderr, "\n[freq_cpu] Reference loading time: %.3f sec", ref_time);
    fprintf(stderr, "\n[freq_cpu] Reference contexts time: %.3f sec", ctx_time);
    fprintf(stderr, "\n[freq_cpu] Data loading time: %.3f sec", load_time);
    fprintf(stderr, "\n[freq_cpu] Data processing time: %.3f sec", process_time - merge_time);
    fprintf(stderr, "\n[freq_cpu] Data merging time: %.3f sec", merge_time);
    fprintf(stderr, "\n[freq_cpu] Data sorting time: %.3f sec", sort_time);
    fprintf(stderr, "\n[freq_cpu] Data output time: %.3f sec", output_time);
    fprintf(stderr, "\n[freq_cpu] Real time: %.3f sec; CPU time: %.3f sec; Peak RAM: %.3f GB\n", mmh_realtime() - realtime0, mmh_cputime(),
            mmh_peakrss() / 1024.0 / 1024.0 / 1024.0);
    free(orows); free(rows);
    orc_destroy(orc);
    mmh_loader_close(ld);
    return 0;
}
