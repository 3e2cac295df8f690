/* fasta.c -- FASTA(.gz) loader.  Mirrors load_ref, reference src/ref.c:46-89 with kseq semantics (src/kseq.h:195): the
 * name ends at the first whitespace, sequence lines are concatenated, '>' at the start of a line starts a record.  Letters
 * are kept raw: upper-casing and U->T happen on the device (kernel K0).
 *
 * Two parsers with the same result: a byte-at-a-time one over zlib's gzread (compressed files, pipes), and, for a plain file,
 * one over the mapped bytes in three parallel passes -- find the '>' that start lines, count the letters of every piece of
 * every record, copy them to their places (a human genome is 3 GB of text: 4 s for the first parser on one core). */
#include <fcntl.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "bamio.h"
#include "mmhost.h"

static mmh_ref_t *load_ref_stream(const char *path) {
    gzFile fp = gzopen(path, "r");
    if (!fp) return NULL;
    gzbuffer(fp, 1 << 20);
    mmh_ref_t *r = (mmh_ref_t *)calloc(1, sizeof(*r));
    int cap = 0;
    size_t scap = 0, slen = 0;
    uint8_t *seq = NULL;
    char *buf = (char *)malloc(1 << 20);
    int in_name = 0, at_line_start = 1, cur = -1;
    char name[1024]; size_t nl = 0; int name_done = 0;
    int got;
    while ((got = gzread(fp, buf, 1 << 20)) > 0) {
        for (int i = 0; i < got; i++) {
            char c = buf[i];
            if (in_name) {
                if (c == '\n') {
                    name[nl] = 0; in_name = 0; at_line_start = 1;
                    if (r->n == cap) {
                        cap = cap ? cap * 2 : 64;
                        r->name = (char **)realloc(r->name, sizeof(char *) * cap);
                        r->seq = (uint8_t **)realloc(r->seq, sizeof(uint8_t *) * cap);
                        r->len = (int64_t *)realloc(r->len, sizeof(int64_t) * cap);
                    }
                    cur = r->n++;
                    r->name[cur] = strdup(name);
                    seq = NULL; scap = 0; slen = 0;
                } else if (!name_done) {
                    if (c == ' ' || c == '\t' || c == '\r') name_done = 1;
                    else if (nl < sizeof(name) - 1) name[nl++] = c;
                }
                continue;
            }
            if (at_line_start && c == '>') {
                if (cur >= 0) { r->seq[cur] = seq; r->len[cur] = (int64_t)slen; }
                in_name = 1; nl = 0; name_done = 0;
                continue;
            }
            at_line_start = (c == '\n');
            if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;   /* kseq keeps only isgraph() characters */
            if (cur < 0) continue;
            if (slen + 1 > scap) { scap = scap ? scap * 2 : (1 << 20); seq = (uint8_t *)realloc(seq, scap); }
            seq[slen++] = (uint8_t)c;
        }
    }
    if (cur >= 0) { r->seq[cur] = seq; r->len[cur] = (int64_t)slen; }
    free(buf);
    gzclose(fp);
    return r;
}

/* ------------------------------------------------------------------ the mapped parser */
#define FA_PIECE ((size_t)4 << 20)
static uint8_t fa_keep[256];   /* 1: a character of the sequence (everything but newline, carriage return, blank, tab) */

typedef struct { const uint8_t *d; size_t n; size_t *out; size_t *n_starts, *first; size_t chunk; } fa_find_t;
/* '>' at the start of a line: counted per chunk (out == NULL), then written in text order from the chunk's place on */
static void fa_find_range(void *arg, int64_t lo, int64_t hi) {
    fa_find_t *f = (fa_find_t *)arg;
    for (int64_t c = lo; c < hi; c++) {
        const size_t a = (size_t)c * f->chunk, b = a + f->chunk < f->n ? a + f->chunk : f->n;
        size_t k = 0;
        const uint8_t *p = f->d + a, *e = f->d + b;
        while (p < e && (p = (const uint8_t *)memchr(p, '>', (size_t)(e - p))) != NULL) {
            const size_t i = (size_t)(p - f->d);
            if (i == 0 || f->d[i - 1] == '\n') { if (f->out) f->out[f->first[c] + k] = i; k++; }
            p++;
        }
        f->n_starts[c] = k;
    }
}

typedef struct { const uint8_t *src; size_t len; uint8_t *dst; size_t kept; } fa_piece_t;
static void fa_count_range(void *arg, int64_t lo, int64_t hi) {
    fa_piece_t *ps = (fa_piece_t *)arg;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *s = ps[i].src;
        size_t k = 0;
        for (size_t j = 0; j < ps[i].len; j++) k += fa_keep[s[j]];
        ps[i].kept = k;
    }
}
static void fa_copy_range(void *arg, int64_t lo, int64_t hi) {
    fa_piece_t *ps = (fa_piece_t *)arg;
    for (int64_t i = lo; i < hi; i++) {
        const uint8_t *s = ps[i].src, *e = s + ps[i].len;
        uint8_t *o = ps[i].dst;
        /* line by line where the line is clean (the usual case: one memchr, one memcpy), character by character where not */
        while (s < e) {
            const uint8_t *nl = (const uint8_t *)memchr(s, '\n', (size_t)(e - s));
            const uint8_t *le = nl ? nl : e;
            const size_t L = (size_t)(le - s);
            if (!memchr(s, '\r', L) && !memchr(s, ' ', L) && !memchr(s, '\t', L)) { memcpy(o, s, L); o += L; }
            else for (const uint8_t *q = s; q < le; q++) if (fa_keep[*q]) *o++ = *q;
            s = nl ? nl + 1 : e;
        }
    }
}

static double fa_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static mmh_ref_t *load_ref_mapped(const uint8_t *d, size_t n, int threads) {
    const double t0 = fa_now();
    double t1 = t0, t2 = t0, t3 = t0;
    for (int c = 0; c < 256; c++) fa_keep[c] = !(c == '\n' || c == '\r' || c == ' ' || c == '\t');
    mm_pool_t *pool = threads > 1 ? mm_pool_create(threads) : NULL;
    mmh_ref_t *r = (mmh_ref_t *)calloc(1, sizeof(*r));
    /* pass 1: the record starts */
    const size_t chunk = (size_t)8 << 20, n_chunks = (n + chunk - 1) / chunk;
    size_t *n_starts = (size_t *)calloc(2 * (n_chunks ? n_chunks : 1), sizeof(size_t)), *first = n_starts ? n_starts + (n_chunks ? n_chunks : 1) : NULL;
    size_t *rs = NULL, *body = NULL, n_rec = 0, n_pieces = 0;
    fa_piece_t *ps = NULL;
    if (!r || !n_starts) goto fail;
    {
        fa_find_t f = {d, n, NULL, n_starts, first, chunk};
        mm_pool_for(pool, (int64_t)n_chunks, 1, fa_find_range, &f);
        for (size_t c = 0; c < n_chunks; c++) { first[c] = n_rec; n_rec += n_starts[c]; }
        rs = (size_t *)malloc(sizeof(size_t) * (n_rec + 1));
        if (!rs) goto fail;
        f.out = rs;
        mm_pool_for(pool, (int64_t)n_chunks, 1, fa_find_range, &f);
        rs[n_rec] = n;
    }
    t1 = fa_now();
    /* the names; a header line the file ends in has no record (the first parser makes the record at the line's end) */
    r->name = (char **)calloc(n_rec ? n_rec : 1, sizeof(char *));
    r->seq = (uint8_t **)calloc(n_rec ? n_rec : 1, sizeof(uint8_t *));
    r->len = (int64_t *)calloc(n_rec ? n_rec : 1, sizeof(int64_t));
    body = (size_t *)malloc(sizeof(size_t) * (n_rec + 1));
    if (!r->name || !r->seq || !r->len || !body) goto fail;
    for (size_t k = 0; k < n_rec; k++) {
        const uint8_t *h = d + rs[k] + 1, *he = (const uint8_t *)memchr(h, '\n', rs[k + 1] - rs[k] - 1);
        if (!he) { body[k] = (size_t)-1; continue; }   /* (only the last record can be like that) */
        size_t nl = 0;
        while (h + nl < he && h[nl] != ' ' && h[nl] != '\t' && h[nl] != '\r') nl++;
        if (nl > 1023) nl = 1023;
        char *nm = (char *)malloc(nl + 1);
        memcpy(nm, h, nl); nm[nl] = 0;
        r->name[r->n] = nm;
        body[r->n] = (size_t)(he + 1 - d);
        rs[r->n] = rs[k];            /* (compacted in step with the names) */
        const size_t blen = rs[k + 1] - body[r->n];
        n_pieces += (blen + FA_PIECE - 1) / FA_PIECE;
        r->len[r->n] = (int64_t)blen;   /* raw length for now */
        r->n++;
    }
    /* pass 2: letters per piece; pass 3: the copies */
    ps = (fa_piece_t *)malloc(sizeof(fa_piece_t) * (n_pieces ? n_pieces : 1));
    if (!ps) goto fail;
    {
        size_t pi = 0;
        for (int k = 0; k < r->n; k++)
            for (size_t o = 0; o < (size_t)r->len[k]; o += FA_PIECE, pi++) {
                ps[pi].src = d + body[k] + o;
                ps[pi].len = (size_t)r->len[k] - o < FA_PIECE ? (size_t)r->len[k] - o : FA_PIECE;
                ps[pi].dst = NULL; ps[pi].kept = 0;
            }
        mm_pool_for(pool, (int64_t)n_pieces, 1, fa_count_range, ps);
        t2 = fa_now();
        pi = 0;
        for (int k = 0; k < r->n; k++) {
            const size_t raw = (size_t)r->len[k], first = pi;
            size_t tot = 0;
            for (size_t o = 0; o < raw; o += FA_PIECE, pi++) tot += ps[pi].kept;
            /* (a record without letters has a NULL sequence of length 0, like the first parser's) */
            r->seq[k] = tot ? (uint8_t *)malloc(tot) : NULL;
            if (tot && !r->seq[k]) goto fail;
            r->len[k] = (int64_t)tot;
            size_t at = 0;
            for (size_t q = first; q < pi; q++) { ps[q].dst = r->seq[k] + at; at += ps[q].kept; }
        }
        mm_pool_for(pool, (int64_t)n_pieces, 1, fa_copy_range, ps);
        t3 = fa_now();
    }
    if (getenv("MM_LOADER_TIMING"))
        fprintf(stderr, "[fasta] %d records, %zu bytes, %d threads: record starts %.3f s, letter counts %.3f s, copies %.3f s\n",
                r->n, n, threads, t1 - t0, t2 - t1, t3 - t2);
    free(ps); free(rs); free(body); free(n_starts);
    if (pool) mm_pool_destroy(pool);
    return r;
fail:
    free(n_starts); free(ps); free(rs); free(body);
    if (pool) mm_pool_destroy(pool);
    mmh_free_ref(r);
    return NULL;
}

/* threads <= 0: the stream parser whatever the file is */
mmh_ref_t *mmh_load_ref_mt(const char *path, int threads) {
    if (threads > 0) {
        int fd = open(path, O_RDONLY);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 2) {
            void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                const uint8_t *d = (const uint8_t *)m;
                mmh_ref_t *r = NULL;
                const int gz = d[0] == 0x1f && d[1] == 0x8b;
                if (!gz) r = load_ref_mapped(d, (size_t)st.st_size, threads);
                munmap(m, (size_t)st.st_size);
                if (!gz) { close(fd); return r; }
            }
        }
        if (fd >= 0) close(fd);
    }
    return load_ref_stream(path);
}
mmh_ref_t *mmh_load_ref(const char *path) { return mmh_load_ref_mt(path, 1); }

int mmh_ref_find(const mmh_ref_t *r, const char *name) {
    for (int i = r->n - 1; i >= 0; i--) if (strcmp(r->name[i], name) == 0) return i;
    return -1;
}

void mmh_free_ref(mmh_ref_t *r) {
    if (!r) return;
    for (int i = 0; i < r->n; i++) { free(r->name[i]); free(r->seq[i]); }
    free(r->name); free(r->seq); free(r->len); free(r);
}
