/* summary_main.c -- `minimod summary reads.bam`: per read, the set of (canonical base | modification codes | status flag)
 * of its MM groups that list at least one call (reference src/summary_main.c:46-60,156-420 driver, summary_single
 * src/mod.c:1426-1555, print_summary_output src/mod.c:1373-1400).  Two ways to the same bytes: this file's host walk (the default: a
 * census of a few hundred characters a read needs no GPU, and `minimod summary` must run on a box without one), and with --gpu the
 * census kernel of include/minimod_summary.h (k_sum_reads: a thread per read, the read's table replayed as khash fills it) -- SURVEY
 * section 8(f) row 4 on the device, checked against the reference's goldens in the GPU suite.
 *
 * The reference keeps a read's keys in a khash string map and prints them in the table's slot order.  To print the same
 * bytes this file keeps a table with the same observable behaviour -- X31 string hash, power-of-two slots, probe
 * sequence i, i+1, i+3, i+6, ..., growth to twice the size once 77 % of the slots are taken, re-insertion in old slot
 * order with displaced keys re-inserted at once -- written from that description (klib's khash.h, which the reference
 * vendors as src/khash.h), not from its code. */
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "minimod_summary.h"
#include "mmhost.h"

/* ------------------------------------------------------------------ slot-order string set */
typedef struct {
    uint32_t n_slots, n_keys, grow_at;
    char **key;          /* NULL = free slot */
} slotset_t;

static uint32_t x31(const char *s) {
    uint32_t h = (uint32_t)(unsigned char)*s;
    if (h) for (++s; *s; ++s) h = h * 31u + (uint32_t)(unsigned char)*s;
    return h;
}

/* first free slot on the probe sequence of `key` in a table of n slots whose occupancy is `used` */
static uint32_t probe_free(const uint8_t *used, uint32_t n, const char *key) {
    uint32_t mask = n - 1, i = x31(key) & mask, step = 0;
    while (used[i]) i = (i + (++step)) & mask;
    return i;
}

static void slotset_grow(slotset_t *t) {
    uint32_t n_new = t->n_slots ? t->n_slots * 2 : 4;
    /* The move happens inside one array: new occupancy is tracked apart from the old keys, which are visited in slot
     * order; a key that lands on a slot still holding an unvisited old key takes its place and the displaced key is
     * placed next. */
    char **key = (char **)realloc(t->key, sizeof(char *) * n_new);
    for (uint32_t i = t->n_slots; i < n_new; i++) key[i] = NULL;
    uint8_t *placed = (uint8_t *)calloc(n_new, 1);      /* slot holds a key of the NEW layout */
    uint8_t *pending = (uint8_t *)calloc(n_new, 1);     /* slot still holds an old key waiting for its turn */
    for (uint32_t i = 0; i < t->n_slots; i++) pending[i] = key[i] != NULL;
    for (uint32_t j = 0; j < t->n_slots; j++) {
        if (!pending[j]) continue;
        char *k = key[j];
        pending[j] = 0; key[j] = NULL;
        for (;;) {
            uint32_t i = probe_free(placed, n_new, k);
            placed[i] = 1;
            if (i < t->n_slots && pending[i]) {          /* evict the old key that sat there and place it next */
                char *ev = key[i];
                key[i] = k; pending[i] = 0;
                k = ev;
            } else {
                key[i] = k;
                break;
            }
        }
    }
    free(placed); free(pending);
    t->key = key; t->n_slots = n_new;
    t->grow_at = (uint32_t)(n_new * 0.77 + 0.5);
}

/* insert unless present; the set owns the string */
static void slotset_add(slotset_t *t, char *k) {
    if (t->n_keys >= t->grow_at) slotset_grow(t);
    uint32_t mask = t->n_slots - 1, i = x31(k) & mask, step = 0;
    while (t->key[i]) {
        if (strcmp(t->key[i], k) == 0) { free(k); return; }
        i = (i + (++step)) & mask;
    }
    t->key[i] = k;
    t->n_keys++;
}

static void slotset_clear(slotset_t *t) {
    for (uint32_t i = 0; i < t->n_slots; i++) free(t->key[i]);
    free(t->key);
    memset(t, 0, sizeof(*t));
}

/* ------------------------------------------------------------------ summary_single (src/mod.c:1426-1555) */
static int valid_base(int c) { return c && strchr("ACGTUNacgtun", c) != NULL; }

static void die(const char *msg) {
    MMH_ERROR("%s", msg);
    fprintf(stderr, "Exiting.\n");
    exit(EXIT_FAILURE);
}

static void summarise_read(const char *mm, size_t n, slotset_t *set) {
    size_t i = 0;
    while (i < n) {
        int status = '.';
        int modbase = 0;
        if (i < n) {
            if (!valid_base((unsigned char)mm[i])) die("Assertion failed. Invalid base in the MM tag");
            modbase = mm[i] == 'U' ? 'T' : mm[i];
            i++;
        }
        if (i < n) {
            if (mm[i] != '+' && mm[i] != '-') die("Assertion failed. Invalid strand in the MM tag");
            i++;
        }
        size_t c0 = i;
        int has_nums = 0, has_alpha = 0;
        while (i < n && mm[i] != ',' && mm[i] != ';' && mm[i] != '?' && mm[i] != '.') {
            if (mm[i] >= '0' && mm[i] <= '9') has_nums = 1;
            else if ((mm[i] >= 'A' && mm[i] <= 'Z') || (mm[i] >= 'a' && mm[i] <= 'z')) has_alpha = 1;
            else die("Invalid base modification code. Modification codes should be either numeric or alphabetic.");
            i++;
        }
        size_t clen = i - c0;
        if (clen == 0) die("Assertion failed. Invalid modification codes. Modification codes cannot be empty.");
        if (has_nums && has_alpha) die("Assertion failed. Invalid modification codes. Modification codes should be either numeric or alphabetic, not both.");
        if (i < n && (mm[i] == '?' || mm[i] == '.')) { status = mm[i]; i++; }
        size_t n_skips = 0;
        while (i < n && mm[i] != ';') {
            if (mm[i] == ',') { i++; continue; }
            size_t l = 0;
            while (i < n && mm[i] != ',' && mm[i] != ';') {
                i++; l++;
                if (l >= 10) die("Assertion failed. Skip count longer than 9 characters");
            }
            n_skips++;
        }
        i++;
        if (n_skips == 0) continue;          /* no calls listed: the group is not reported */
        char *key = (char *)malloc(clen + 5);
        key[0] = (char)modbase; key[1] = '|';
        memcpy(key + 2, mm + c0, clen);
        key[2 + clen] = '|'; key[3 + clen] = (char)status; key[4 + clen] = 0;
        slotset_add(set, key);
    }
}

/* ------------------------------------------------------------------ driver */
static struct option long_options[] = {
    {"threads", required_argument, 0, 't'},
    {"batchsize", required_argument, 0, 'K'},
    {"max-bytes", required_argument, 0, 'B'},
    {"verbose", required_argument, 0, 'v'},
    {"help", no_argument, 0, 'h'},
    {"version", no_argument, 0, 'V'},
    {"prog-interval", required_argument, 0, 'p'},
    {"debug-break", required_argument, 0, 0},
    {"output", required_argument, 0, 'o'},
    {"allow-secondary", no_argument, 0, 0},
    {"skip-supplementary", no_argument, 0, 0},
    {"gpu", no_argument, 0, 0},
    {"device", required_argument, 0, 0},
    {0, 0, 0, 0}};

static void print_help(FILE *fp, int threads, int32_t K, int64_t B, int prog, const char *out, int sec, int sup) {
    fprintf(fp, "Usage: minimod summary reads.bam\n");
    fprintf(fp, "\nbasic options:\n");
    fprintf(fp, "   -t INT                     number of BAM decoding threads [%d]\n", threads);
    fprintf(fp, "   -K INT                     batch size (max number of reads loaded at once) [%d]\n", K);
    fprintf(fp, "   -B FLOAT[K/M/G]            max number of bases loaded at once [%.1fM]\n", B / (float)(1000 * 1000));
    fprintf(fp, "   -h                         help\n");
    fprintf(fp, "   -p INT                     print progress every INT seconds (0: per batch) [%d]\n", prog);
    fprintf(fp, "   -o FILE                    output file [%s]\n", out == NULL ? "stdout" : out);
    fprintf(fp, "   --verbose INT              verbosity level [%d]\n", mmh_log_level);
    fprintf(fp, "   --version                  print version\n");
    fprintf(fp, "   --allow-secondary          allow secondary alignments [%s]\n", sec ? "yes" : "no");
    fprintf(fp, "   --skip-supplementary       skip supplementary alignments [%s]\n", sup ? "yes" : "no");
    fprintf(fp, "\nadvanced options:\n");
    fprintf(fp, "   --debug-break INT          break after processing the specified no. of batches\n");
    fprintf(fp, "   --gpu                      the census on the GPU (a thread per read; the same bytes)\n");
    fprintf(fp, "   --device INT               GPU to use with --gpu [0]\n");
}

int mmh_summary_main(int argc, char **argv) {
    double realtime0 = mmh_realtime();
    const char *optstring = "c:t:B:K:v:p:o:hV";
    int longindex = 0, c;
    FILE *fp_help = stderr, *out = stdout;
    int32_t K = 512; int64_t B = 20 * 1000 * 1000;
    int threads = 8, debug_break = -1, prog = 0, sec = 0, sup = 0, gpu = 0, device = 0;
    const char *out_path = NULL;
    while ((c = getopt_long(argc, argv, optstring, long_options, &longindex)) >= 0) {
        const char *lname = c == 0 ? long_options[longindex].name : "";
        if (c == 'B') {
            B = mmh_parse_num(optarg);
            if (B <= 0) { MMH_ERROR("%s", "Maximum number of bases should be larger than 0."); exit(EXIT_FAILURE); }
        } else if (c == 'K') {
            K = atoi(optarg);
            if (K < 1) { MMH_ERROR("Batch size should larger than 0. You entered %d", K); exit(EXIT_FAILURE); }
        } else if (c == 't') {
            threads = atoi(optarg);
            if (threads < 1) { MMH_ERROR("Number of threads should larger than 0. You entered %d", threads); exit(EXIT_FAILURE); }
        } else if (c == 'v') {
            mmh_log_level = atoi(optarg);
        } else if (c == 'p') {
            if (atoi(optarg) < 0) { MMH_ERROR("Progress interval should be 0 or positive. You entered %d", atoi(optarg)); exit(EXIT_FAILURE); }
            prog = atoi(optarg);
        } else if (c == 'o') {
            FILE *fp = fopen(optarg, "w");
            if (fp == NULL) { MMH_ERROR("Cannot open file %s for writing", optarg); exit(EXIT_FAILURE); }
            out_path = optarg; out = fp;
        } else if (c == 'V') {
            fprintf(stdout, "minimod %s\n", MMH_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'c') { /* accepted and unused, like the reference (src/summary_main.c:212) */
        } else if (c == 0 && strcmp(lname, "debug-break") == 0) { debug_break = atoi(optarg);
        } else if (c == 0 && strcmp(lname, "allow-secondary") == 0) { sec = 1;
        } else if (c == 0 && strcmp(lname, "skip-supplementary") == 0) { sup = 1;
        } else if (c == 0 && strcmp(lname, "gpu") == 0) { gpu = 1;
        } else if (c == 0 && strcmp(lname, "device") == 0) { device = atoi(optarg);
        } else {
            print_help(fp_help, threads, K, B, prog, out_path, sec, sup);
            exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
        }
    }
    if (argc - optind != 1 || fp_help == stdout) {
        MMH_WARNING("%s", "Missing arguments");
        print_help(fp_help, threads, K, B, prog, out_path, sec, sup);
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    const char *bam_file = argv[optind];
    if (access(bam_file, F_OK) == -1) { MMH_ERROR("BAM file %s does not exist", bam_file); exit(EXIT_FAILURE); }
    mmh_loader_t *ld = mmh_loader_open(bam_file, threads, K, B, sec, sup);
    if (!ld) { MMH_ERROR("NULL returned: could not open or parse %s.", bam_file); exit(EXIT_FAILURE); }

    mm_summary_t *census = NULL;
    if (gpu) {
        char cerr[256];
        census = mm_summary_create(device, cerr, sizeof cerr);
        if (!census) { MMH_ERROR("--gpu: %s", cerr); exit(EXIT_FAILURE); }   /* (asked for by name: no quiet fallback) */
    }
    fprintf(out, "read_id\t modifications\n");     /* print_summary_header, src/mod.c:1370 */
    int more = 1, counter = 0, set = 0;
    mm_batch_t batch;
    while (more) {
        int32_t n = mmh_loader_next(ld, set, &batch, &more);
        if (n < 0) { MMH_ERROR("%s", "Truncated or corrupt BAM file"); exit(EXIT_FAILURE); }
        fprintf(stderr, "[%s::%.3f*%.2f] %d Entries (%.1fM bases) loaded\n", __func__, mmh_realtime() - realtime0,
                mmh_cputime() / (mmh_realtime() - realtime0), n, ld->last_processed_bytes / (1000.0 * 1000.0));
        if (census && n > 0) {
            const char *text = NULL; const uint64_t *off = NULL; const uint32_t *len = NULL;
            int32_t bad = -1;
            const int32_t e = mm_summary_batch(census, &batch, &text, &off, &len, &bad);
            if (e) {
                switch (e) {   /* the reference's messages, src/mod.c:1467-1530 */
                    case MM_E_MMBASE: die("Assertion failed. Invalid base in the MM tag");
                    case MM_E_MMSTRAND: die("Assertion failed. Invalid strand in the MM tag");
                    case MM_E_MMCODE: die("Invalid base modification code. Modification codes should be either numeric or alphabetic.");
                    case MM_E_MMEMPTY: die("Assertion failed. Invalid modification codes. Modification codes cannot be empty.");
                    case MM_E_MMMIXED: die("Assertion failed. Invalid modification codes. Modification codes should be either numeric or alphabetic, not both.");
                    case MM_E_SKIPLEN: die("Assertion failed. Skip count longer than 9 characters");
                    default: MMH_ERROR("GPU path failed: %s", mm_strerror(e)); exit(EXIT_FAILURE);
                }
            }
            for (int32_t i = 0; i < n; i++) {
                fputs(mmh_loader_qname(ld, set, i), out); fputc('\t', out);
                if (len[i]) fwrite(text + off[i], 1, len[i], out);
                fputc('\n', out);
            }
        } else
        for (int32_t i = 0; i < n; i++) {
            const mm_read_t *rd = &batch.reads[i];
            slotset_t keys;
            memset(&keys, 0, sizeof keys);
            summarise_read((const char *)batch.mm + rd->mm_off, rd->mm_len, &keys);
            fprintf(out, "%s\t", mmh_loader_qname(ld, set, i));
            for (uint32_t s = 0; s < keys.n_slots; s++) if (keys.key[s]) fprintf(out, "%s ", keys.key[s]);
            fputc('\n', out);
            slotset_clear(&keys);
        }
        uint64_t skipped = ld->total_reads - ld->processed_reads;
        if (skipped > 0.9 * ld->total_reads)
            MMH_WARNING("%s", "90% of the reads are skipped. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
        if (skipped == ld->total_reads)
            MMH_ERROR("%s", "All reads are skipped. Quitting. Possible causes: unmapped bam, zero sequence lengths, or missing MM, ML tags (not performed base modification aware basecalling). Refer https://github.com/warp9seq/minimod for more information.");
        set ^= 1;
        if (debug_break == counter) break;
        counter++;
    }
    if (out != stdout) fclose(out);
    fprintf(stderr, "[%s] total entries: %ld", __func__, (long)ld->total_reads);
    fprintf(stderr, "\n[%s] total skipped entries: %ld", __func__, (long)(ld->total_reads - ld->processed_reads));
    fprintf(stderr, "\n[%s] total processed entries: %ld\n", __func__, (long)ld->processed_reads);
    if (census) { fprintf(stderr, "[%s] census on the device (k_sum_reads)\n", __func__); mm_summary_destroy(census); }
    mmh_loader_close(ld);
    return 0;
}
