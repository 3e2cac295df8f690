/* emit.c -- print_freq_header / print_freq_output, reference src/mod.c:628-728: same columns, same "%f".
 *
 * Rows are formatted by the worker pool in pieces of a few thousand rows, each into its own buffer, and the buffers are
 * written in order by a writer thread while the next round is being formatted (the row formatter; at
 * 30x whole-genome depth the bedmethyl is gigabytes of text).  Nothing here touches the GPU library: code names come
 * in as an array. */
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

void mmh_print_freq_header(FILE *fp, int bedmethyl, int insertions, int haplotypes) {
    if (bedmethyl) return;
    fprintf(fp, "contig\tstart\tend\tstrand\tn_called\tn_mod\tfreq\tmod_code%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
}

/* ------------------------------------------------------------------ buffers, rounds, writer */
typedef struct { char *p; size_t len, cap; } mbuf_t;
static char *mbuf_room(mbuf_t *b, size_t need) {
    if (b->len + need > b->cap) {
        size_t cap = b->cap ? b->cap * 2 : (size_t)1 << 16;
        while (cap < b->len + need) cap *= 2;
        char *q = (char *)realloc(b->p, cap);
        if (!q) { fprintf(stderr, "out of memory formatting rows\n"); exit(EXIT_FAILURE); }
        b->p = q; b->cap = cap;
    }
    return b->p + b->len;
}

#define EMIT_KEEP 1024   /* buffers kept for reuse (a fresh 300 KB buffer per piece is an mmap, page faults and an munmap) */
static struct {
    pthread_t th;
    int started, busy, quit, err;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    FILE *fp;
    mbuf_t *bufs;
    int n;
    mbuf_t keep[EMIT_KEEP];
    int n_keep;
} W = {.mu = PTHREAD_MUTEX_INITIALIZER, .cv = PTHREAD_COND_INITIALIZER};

static void *writer_main(void *arg) {
    (void)arg;
    pthread_mutex_lock(&W.mu);
    for (;;) {
        while (!W.busy && !W.quit) pthread_cond_wait(&W.cv, &W.mu);
        if (!W.busy && W.quit) break;
        FILE *fp = W.fp; mbuf_t *bufs = W.bufs; int n = W.n;
        pthread_mutex_unlock(&W.mu);
        int err = 0;
        for (int i = 0; i < n; i++)
            if (bufs[i].len && fwrite(bufs[i].p, 1, bufs[i].len, fp) != bufs[i].len) err = 1;
        pthread_mutex_lock(&W.mu);
        for (int i = 0; i < n; i++) {
            if (bufs[i].p && W.n_keep < EMIT_KEEP) { bufs[i].len = 0; W.keep[W.n_keep++] = bufs[i]; }
            else free(bufs[i].p);
        }
        free(bufs);
        if (err) W.err = 1;
        W.busy = 0;
        pthread_cond_broadcast(&W.cv);
    }
    pthread_mutex_unlock(&W.mu);
    return NULL;
}
/* hand a round of buffers (ownership included) to the writer; waits for the round before it */
static void writer_submit(FILE *fp, mbuf_t *bufs, int n) {
    pthread_mutex_lock(&W.mu);
    if (!W.started) {
        if (pthread_create(&W.th, NULL, writer_main, NULL) != 0) { fprintf(stderr, "cannot start the writer thread\n"); exit(EXIT_FAILURE); }
        W.started = 1;
    }
    while (W.busy) pthread_cond_wait(&W.cv, &W.mu);
    W.fp = fp; W.bufs = bufs; W.n = n; W.busy = 1;
    pthread_cond_broadcast(&W.cv);
    pthread_mutex_unlock(&W.mu);
}
int mmh_emit_flush(void) {
    pthread_mutex_lock(&W.mu);
    while (W.busy) pthread_cond_wait(&W.cv, &W.mu);
    int err = W.err;
    pthread_mutex_unlock(&W.mu);
    return err ? -1 : 0;
}
int mmh_emit_finish(void) {
    int r = mmh_emit_flush();
    pthread_mutex_lock(&W.mu);
    int started = W.started;
    W.quit = 1;
    pthread_cond_broadcast(&W.cv);
    pthread_mutex_unlock(&W.mu);
    if (started) pthread_join(W.th, NULL);
    for (int i = 0; i < W.n_keep; i++) free(W.keep[i].p);
    W.n_keep = 0; W.started = 0; W.quit = 0; W.err = 0;
    return r;
}

typedef void (*piece_fn)(const void *ctx, int64_t lo, int64_t hi, mbuf_t *out);
typedef struct { piece_fn fn; const void *ctx; int64_t base, grain; mbuf_t *bufs; } round_t;
static void round_piece(void *arg, int64_t lo, int64_t hi) {
    round_t *r = (round_t *)arg;
    r->fn(r->ctx, r->base + lo, r->base + hi, &r->bufs[lo / r->grain]);
}
#define EMIT_GRAIN 4096
static void emit_rows(FILE *fp, mm_pool_t *pool, int64_t n, piece_fn fn, const void *ctx) {
    const int nt = pool ? mm_pool_threads(pool) : 1;
    const int64_t per_round = (int64_t)EMIT_GRAIN * (nt > 1 ? 2 * nt : 4);
    for (int64_t base = 0; base < n; base += per_round) {
        const int64_t m = n - base < per_round ? n - base : per_round;
        const int pieces = (int)((m + EMIT_GRAIN - 1) / EMIT_GRAIN);
        round_t r = {fn, ctx, base, EMIT_GRAIN, (mbuf_t *)calloc((size_t)pieces, sizeof(mbuf_t))};
        if (!r.bufs) { fprintf(stderr, "out of memory formatting rows\n"); exit(EXIT_FAILURE); }
        pthread_mutex_lock(&W.mu);
        for (int i = 0; i < pieces && W.n_keep > 0; i++) r.bufs[i] = W.keep[--W.n_keep];
        pthread_mutex_unlock(&W.mu);
        if (nt > 1) mm_pool_for(pool, m, EMIT_GRAIN, round_piece, &r);
        else for (int64_t lo = 0; lo < m; lo += EMIT_GRAIN) round_piece(&r, lo, lo + EMIT_GRAIN < m ? lo + EMIT_GRAIN : m);
        writer_submit(fp, r.bufs, pieces);
    }
}

/* "%f" of n_mod / n_called (x100 for bedmethyl) through snprintf once per distinct pair of small counts: rows repeat
 * the same few hundred ratios millions of times, and the text must be printf's own (src/mod.c:685,703) */
#define FREQ_MEMO 256
static char (*memo[2])[FREQ_MEMO][12];
static _Atomic uint8_t (*memo_state[2])[FREQ_MEMO];   /* 0 empty, 1..11 = length (ready), 255 = being filled */
static pthread_once_t memo_once = PTHREAD_ONCE_INIT;
static void memo_init(void) {
    for (int k = 0; k < 2; k++) {
        memo[k] = (char (*)[FREQ_MEMO][12])calloc(FREQ_MEMO, sizeof(*memo[k]));
        memo_state[k] = (_Atomic uint8_t (*)[FREQ_MEMO])calloc(FREQ_MEMO, sizeof(*memo_state[k]));
        if (!memo[k] || !memo_state[k]) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); }
    }
}
static const char *freq_str(uint32_t n_mod, uint32_t n_called, int percent, int *len, char *tmp) {
    double f = percent ? (double)n_mod * 100 / n_called : (double)n_mod / n_called;
    if (n_called >= FREQ_MEMO || n_mod >= FREQ_MEMO) { *len = snprintf(tmp, 32, "%f", f); return tmp; }
    _Atomic uint8_t *st = &memo_state[percent][n_called][n_mod];
    uint8_t v = atomic_load_explicit(st, memory_order_acquire);
    if (v == 0) {
        uint8_t expect = 0;
        if (atomic_compare_exchange_strong_explicit(st, &expect, 255, memory_order_acquire, memory_order_acquire)) {
            v = (uint8_t)snprintf(memo[percent][n_called][n_mod], 12, "%f", f);
            atomic_store_explicit(st, v, memory_order_release);
        } else {
            v = expect;
        }
    }
    if (v == 255) { *len = snprintf(tmp, 32, "%f", f); return tmp; }   /* another thread is writing the entry right now */
    *len = v;
    return memo[percent][n_called][n_mod];
}

static char *put_int(char *p, long v) {
    char tmp[24];
    int n = 0;
    unsigned long u = v < 0 ? (unsigned long)(-v) : (unsigned long)v;
    if (v < 0) *p++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
static char *put_str(char *p, const char *s, size_t n) { memcpy(p, s, n); return p + n; }

typedef struct {
    const mm_row_t *rows;
    const mm_bam_hdr_t *hdr;
    const char *const *codes;
    int n_codes, bedmethyl, insertions, haplotypes;
} freq_ctx_t;

static void freq_piece(const void *vctx, int64_t lo, int64_t hi, mbuf_t *out) {
    const freq_ctx_t *c = (const freq_ctx_t *)vctx;
    char tmp[40];
    int32_t last_tid = -2;
    const char *contig = "*";
    size_t clen = 1;
    for (int64_t i = lo; i < hi; i++) {
        const mm_row_t *r = &c->rows[i];
        if (r->tid != last_tid) {
            last_tid = r->tid;
            contig = (r->tid >= 0 && r->tid < c->hdr->n_targets) ? c->hdr->target_name[r->tid] : "*";
            clen = strlen(contig);
        }
        const char *code = (r->code >= 0 && r->code < c->n_codes) ? c->codes[r->code] : "";
        size_t codelen = strlen(code);
        char *p = mbuf_room(out, clen + codelen + 160), *p0 = p;
        char strand = r->strand ? '-' : '+';
        int flen;
        const char *fs = freq_str(r->n_mod, r->n_called, c->bedmethyl, &flen, tmp);
        if (c->bedmethyl) {   /* src/mod.c:685 */
            p = put_str(p, contig, clen); *p++ = '\t';
            p = put_int(p, r->pos); *p++ = '\t'; p = put_int(p, (long)r->pos + 1); *p++ = '\t';
            p = put_str(p, code, codelen); *p++ = '\t';
            p = put_int(p, (long)r->n_called); *p++ = '\t'; *p++ = strand; *p++ = '\t';
            p = put_int(p, r->pos); *p++ = '\t'; p = put_int(p, (long)r->pos + 1);
            p = put_str(p, "\t255,0,0\t", 9);
            p = put_int(p, (long)r->n_called); *p++ = '\t';
            p = put_str(p, fs, (size_t)flen);
        } else {           /* src/mod.c:703-715 */
            p = put_str(p, contig, clen); *p++ = '\t';
            p = put_int(p, r->pos); *p++ = '\t'; p = put_int(p, r->pos); *p++ = '\t';
            *p++ = strand; *p++ = '\t';
            p = put_int(p, (long)r->n_called); *p++ = '\t'; p = put_int(p, (long)r->n_mod); *p++ = '\t';
            p = put_str(p, fs, (size_t)flen); *p++ = '\t';
            p = put_str(p, code, codelen);
            if (c->insertions) { *p++ = '\t'; p = put_int(p, r->ins_offset); }
            if (c->haplotypes) { *p++ = '\t'; if (r->hp == -1) *p++ = '*'; else p = put_int(p, r->hp); }
        }
        *p++ = '\n';
        out->len += (size_t)(p - p0);
    }
}

void mmh_print_freq_rows(FILE *fp, mm_pool_t *pool, const mm_row_t *rows, int64_t n, const mm_bam_hdr_t *hdr,
                         const char *const *codes, int n_codes, int bedmethyl, int insertions, int haplotypes) {
    pthread_once(&memo_once, memo_init);
    freq_ctx_t c = {rows, hdr, codes, n_codes, bedmethyl, insertions, haplotypes};
    emit_rows(fp, pool, n, freq_piece, &c);
}

/* ---- view: print_view_header / print_view_output, reference src/mod.c:545-626.  One line per row; the line is
 * assembled by hand (a batch is millions of rows and fprintf("%f") dominates otherwise): the 256 possible mod_prob
 * strings are printed once with the reference's own format. */
void mmh_print_view_header(FILE *fp, int insertions, int haplotypes) {
    fprintf(fp, "ref_contig\tref_pos\tstrand\tread_id\tread_pos\tmod_code\tmod_prob%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
}


static char view_prob[256][16];
static int view_prob_len[256];
static pthread_once_t view_once = PTHREAD_ONCE_INIT;
static void view_init(void) {
    for (int x = 0; x < 256; x++) view_prob_len[x] = snprintf(view_prob[x], sizeof view_prob[x], "%f", (x + 0.5) / 256.0);   /* THRESH_UINT8_TO_DBL */
}

typedef struct {
    const mm_view_row_t *rows;
    const mm_batch_t *batch;
    const mmh_loader_t *ld;
    int pool_set;
    const mm_read_t *reads;          /* mmh_print_view_rows_of: the launch's read records and names (batch / ld not used then) */
    const uint64_t *name_off;
    const char *names;
    const mm_bam_hdr_t *hdr;
    const char *const *codes;
    int n_codes, insertions, haplotypes;
} view_ctx_t;

static void view_piece(const void *vctx, int64_t lo, int64_t hi, mbuf_t *out) {
    const view_ctx_t *c = (const view_ctx_t *)vctx;
    uint32_t last_read = 0xFFFFFFFFu;
    const char *qname = "", *contig = "*";
    size_t qlen = 0, clen = 1;
    const mm_read_t *rd = NULL;
    for (int64_t i = lo; i < hi; i++) {
        const mm_view_row_t *r = &c->rows[i];
        if (r->read != last_read) {
            last_read = r->read;
            rd = c->reads ? &c->reads[r->read] : &c->batch->reads[r->read];
            qname = c->reads ? c->names + c->name_off[r->read] : mmh_loader_qname(c->ld, c->pool_set, (int32_t)r->read);
            qlen = strlen(qname);
            contig = (rd->tid >= 0 && rd->tid < c->hdr->n_targets) ? c->hdr->target_name[rd->tid] : "*";
            clen = strlen(contig);
        }
        const char *code = r->code < c->n_codes ? c->codes[r->code] : "";
        size_t codelen = strlen(code);
        char *p = mbuf_room(out, clen + qlen + codelen + 128), *p0 = p;
        p = put_str(p, contig, clen); *p++ = '\t';
        p = put_int(p, r->pos); *p++ = '\t';
        *p++ = (rd->flag & 0x10) ? '-' : '+'; *p++ = '\t';
        p = put_str(p, qname, qlen); *p++ = '\t';
        p = put_int(p, (long)r->read_pos); *p++ = '\t';
        p = put_str(p, code, codelen); *p++ = '\t';
        p = put_str(p, view_prob[r->prob], (size_t)view_prob_len[r->prob]);
        if (c->insertions) { *p++ = '\t'; p = put_int(p, r->ins_offset); }
        if (c->haplotypes) { *p++ = '\t'; p = put_int(p, rd->hp); }
        *p++ = '\n';
        out->len += (size_t)(p - p0);
    }
}

/* The rows' text is complete when this returns (the batch and its pools may be reused); writing it may still be under way. */
void mmh_print_view_rows(FILE *fp, mm_pool_t *pool, const mm_view_row_t *rows, int64_t n, const mm_batch_t *batch, const mmh_loader_t *ld, int pool_set,
                         const mm_bam_hdr_t *hdr, const char *const *codes, int n_codes, int insertions, int haplotypes) {
    pthread_once(&view_once, view_init);
    view_ctx_t c = {rows, batch, ld, pool_set, NULL, NULL, NULL, hdr, codes, n_codes, insertions, haplotypes};
    emit_rows(fp, pool, n, view_piece, &c);
}
/* the same for the rows of a gathered launch: `reads` are the records of every batch that went into it, one behind the other (a row
 * names its read by its index in the launch), names[name_off[i]] read i's name */
void mmh_print_view_rows_of(FILE *fp, mm_pool_t *pool, const mm_view_row_t *rows, int64_t n, const mm_read_t *reads, const uint64_t *name_off, const char *names,
                            const mm_bam_hdr_t *hdr, const char *const *codes, int n_codes, int insertions, int haplotypes) {
    pthread_once(&view_once, view_init);
    view_ctx_t c = {rows, NULL, NULL, 0, reads, name_off, names, hdr, codes, n_codes, insertions, haplotypes};
    emit_rows(fp, pool, n, view_piece, &c);
}
