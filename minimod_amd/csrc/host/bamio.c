/* bamio.c -- see bamio.h.  BGZF = a series of gzip members, each with a 'BC' extra subfield giving the block
 * size, at most 64 KiB of payload per block (SAM/BAM specification section 4.1).  The reader pulls a group of
 * blocks from the file, inflates them in parallel (raw deflate, zlib) into one contiguous window and parses BAM
 * records out of that window, carrying a partial record over to the next group. */
#include "bamio.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define GROUP_BLOCKS 256
#define RAW_CAP (GROUP_BLOCKS * 65536 + 65536)

typedef struct {
    const uint8_t *cdata;
    uint32_t clen, isize;
    uint8_t *out;
    int err;
} blk_t;

struct mm_bam {
    FILE *fp;
    int n_threads;
    mm_bam_hdr_t hdr;
    /* compressed side */
    uint8_t *cbuf;
    size_t ccap, clen, cpos;
    int eof;
    /* decompressed window */
    uint8_t *win;
    size_t wcap, wlen, wpos;
    blk_t blk[GROUP_BLOCKS];
    /* pool */
    int job_n, job_next;
    pthread_mutex_t mu;
};

static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static int inflate_block(blk_t *b) {
    if (b->isize == 0) return 0;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    zs.next_in = (Bytef *)b->cdata; zs.avail_in = b->clen;
    zs.next_out = b->out; zs.avail_out = b->isize;
    int r = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return (r == Z_STREAM_END && zs.avail_out == 0) ? 0 : -1;
}

static void *worker(void *arg) {
    mm_bam_t *b = (mm_bam_t *)arg;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        int i = b->job_next < b->job_n ? b->job_next++ : -1;
        pthread_mutex_unlock(&b->mu);
        if (i < 0) break;
        b->blk[i].err = inflate_block(&b->blk[i]);
    }
    return NULL;
}

/* refill the compressed buffer so that at least `need` bytes are available at cpos (or EOF) */
static void cfill(mm_bam_t *b, size_t need) {
    if (b->clen - b->cpos >= need || b->eof) return;
    memmove(b->cbuf, b->cbuf + b->cpos, b->clen - b->cpos);
    b->clen -= b->cpos; b->cpos = 0;
    while (b->clen < b->ccap && !b->eof) {
        size_t n = fread(b->cbuf + b->clen, 1, b->ccap - b->clen, b->fp);
        if (n == 0) { b->eof = 1; break; }
        b->clen += n;
    }
}

/* inflate the next group of blocks behind the unread tail of the window; returns bytes added, 0 at EOF, <0 on error */
static long refill(mm_bam_t *b) {
    size_t tail = b->wlen - b->wpos;
    memmove(b->win, b->win + b->wpos, tail);
    b->wlen = tail; b->wpos = 0;
    cfill(b, (size_t)GROUP_BLOCKS * 65536);
    int n = 0;
    size_t added = 0;
    while (n < GROUP_BLOCKS) {
        /* never move the compressed buffer while earlier blocks of this group still point into it */
        if (b->clen - b->cpos < 28) {
            if (n > 0) break;
            cfill(b, 65536 + 28);
            if (b->clen - b->cpos < 28) break;
        }
        const uint8_t *h = b->cbuf + b->cpos;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
        uint32_t xlen = rd_u16(h + 10);
        if (b->clen - b->cpos < 12 + (size_t)xlen) { if (n > 0) break; return -1; }
        const uint8_t *x = h + 12, *xe = x + xlen;
        int bsize = -1;
        while (x + 4 <= xe) {
            uint32_t sl = rd_u16(x + 2);
            if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = rd_u16(x + 4);
            x += 4 + sl;
        }
        if (bsize < 0) return -1;
        size_t total = (size_t)bsize + 1;
        if (b->clen - b->cpos < total) {
            if (n > 0) break;
            cfill(b, total);
            h = b->cbuf + b->cpos;
            if (b->clen - b->cpos < total) return -1;
        }
        blk_t *k = &b->blk[n];
        k->cdata = h + 12 + xlen;
        k->clen = (uint32_t)(total - xlen - 12 - 8);
        k->isize = rd_u32(h + total - 4);
        if (k->isize > 65536) return -1;
        if (b->wlen + added + k->isize > b->wcap) {
            /* cannot happen: the window holds a whole group plus the largest record tail; grow to be safe */
            size_t ncap = b->wcap * 2;
            uint8_t *nw = (uint8_t *)realloc(b->win, ncap);
            if (!nw) return -1;
            for (int j = 0; j < n; j++) b->blk[j].out = nw + (b->blk[j].out - b->win);
            b->win = nw; b->wcap = ncap;
        }
        k->out = b->win + b->wlen + added;
        k->err = 0;
        added += k->isize;
        b->cpos += total;
        n++;
        /* cdata pointers stay valid: cfill only moves data when it needs more, and we asked for a whole group */
    }
    if (n == 0) return 0;
    b->job_n = n; b->job_next = 0;
    int nt = b->n_threads;
    if (nt > n) nt = n;
    if (nt <= 1) worker(b);
    else {
        pthread_t th[64];
        if (nt > 64) nt = 64;
        for (int t = 0; t < nt; t++) pthread_create(&th[t], NULL, worker, b);
        for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    }
    for (int i = 0; i < n; i++) if (b->blk[i].err) return -1;
    b->wlen += added;
    return (long)added + 1; /* +1: a group of empty blocks (EOF marker) is not EOF by itself */
}

/* make `need` bytes available at wpos; 1 ok, 0 clean EOF, -1 error/truncated */
static int want(mm_bam_t *b, size_t need) {
    while (b->wlen - b->wpos < need) {
        if (need + 65536 > b->wcap - (size_t)GROUP_BLOCKS * 65536) {
            size_t ncap = need + (size_t)GROUP_BLOCKS * 65536 + 65536;
            size_t tail = b->wlen - b->wpos;
            uint8_t *nw = (uint8_t *)malloc(ncap);
            if (!nw) return -1;
            memcpy(nw, b->win + b->wpos, tail);
            free(b->win);
            b->win = nw; b->wcap = ncap; b->wlen = tail; b->wpos = 0;
        }
        long r = refill(b);
        if (r < 0) return -1;
        if (r == 0) return (b->wlen - b->wpos) == 0 ? 0 : -1;
    }
    return 1;
}

mm_bam_t *mm_bam_open(const char *path, int n_threads) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    mm_bam_t *b = (mm_bam_t *)calloc(1, sizeof(*b));
    b->fp = fp;
    b->n_threads = n_threads < 1 ? 1 : n_threads;
    b->ccap = (size_t)GROUP_BLOCKS * 65536 * 2 + 65536;
    b->cbuf = (uint8_t *)malloc(b->ccap);
    b->wcap = (size_t)RAW_CAP * 2;
    b->win = (uint8_t *)malloc(b->wcap);
    pthread_mutex_init(&b->mu, NULL);
    if (!b->cbuf || !b->win) { mm_bam_close(b); return NULL; }
    /* header */
    if (want(b, 12) != 1 || memcmp(b->win + b->wpos, "BAM\1", 4) != 0) { mm_bam_close(b); return NULL; }
    uint32_t l_text = rd_u32(b->win + b->wpos + 4);
    if (want(b, 12 + (size_t)l_text) != 1) { mm_bam_close(b); return NULL; }
    b->wpos += 8 + l_text;
    int32_t n_ref = (int32_t)rd_u32(b->win + b->wpos);
    b->wpos += 4;
    b->hdr.n_targets = n_ref;
    b->hdr.target_name = (char **)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(char *));
    b->hdr.target_len = (uint32_t *)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(uint32_t));
    for (int32_t i = 0; i < n_ref; i++) {
        if (want(b, 4) != 1) { mm_bam_close(b); return NULL; }
        uint32_t l_name = rd_u32(b->win + b->wpos);
        if (want(b, 8 + (size_t)l_name) != 1) { mm_bam_close(b); return NULL; }
        b->hdr.target_name[i] = (char *)malloc(l_name + 1);
        memcpy(b->hdr.target_name[i], b->win + b->wpos + 4, l_name);
        b->hdr.target_name[i][l_name] = 0;
        b->hdr.target_len[i] = rd_u32(b->win + b->wpos + 4 + l_name);
        b->wpos += 8 + l_name;
    }
    return b;
}

const mm_bam_hdr_t *mm_bam_header(const mm_bam_t *b) { return &b->hdr; }

int mm_bam_next(mm_bam_t *b, mm_bam_rec_t *r) {
    int w = want(b, 4);
    if (w <= 0) return w;
    uint32_t bs = rd_u32(b->win + b->wpos);
    if (bs < 32) return -1;
    w = want(b, 4 + (size_t)bs);
    if (w <= 0) return -1;
    const uint8_t *p = b->win + b->wpos + 4;
    r->tid = (int32_t)rd_u32(p); r->pos = (int32_t)rd_u32(p + 4);
    r->l_read_name = p[8]; r->mapq = p[9];
    r->n_cigar = rd_u16(p + 12); r->flag = rd_u16(p + 14);
    r->l_qseq = (int32_t)rd_u32(p + 16);
    size_t o = 32;
    r->qname = (const char *)(p + o); o += r->l_read_name;
    r->cigar = (const uint32_t *)(p + o); o += 4 * (size_t)r->n_cigar;
    r->seq = p + o; o += ((size_t)r->l_qseq + 1) / 2;
    o += (size_t)r->l_qseq;
    if (o > bs) return -1;
    r->aux = p + o; r->l_aux = (int32_t)(bs - o);
    r->l_data = (int32_t)(bs - 32 + ((4 - (r->l_read_name & 3)) & 3)); /* htslib pads qname to a multiple of 4 */
    b->wpos += 4 + (size_t)bs;
    return 1;
}

void mm_bam_close(mm_bam_t *b) {
    if (!b) return;
    if (b->fp) fclose(b->fp);
    for (int32_t i = 0; i < b->hdr.n_targets; i++) free(b->hdr.target_name[i]);
    free(b->hdr.target_name); free(b->hdr.target_len);
    free(b->cbuf); free(b->win);
    pthread_mutex_destroy(&b->mu);
    free(b);
}

const uint8_t *mm_aux_get(const uint8_t *aux, int32_t l_aux, const char tag[2]) {
    const uint8_t *p = aux, *e = aux + l_aux;
    while (p + 3 <= e) {
        const uint8_t *t = p + 2;
        int match = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
        size_t sz;
        switch (*t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': {
                const uint8_t *z = t + 1;
                while (z < e && *z) z++;
                sz = (size_t)(z - (t + 1)) + 1;
                break;
            }
            case 'B': {
                if (t + 6 > e) return NULL;
                uint32_t n = rd_u32(t + 2);
                size_t es = (t[1] == 'c' || t[1] == 'C') ? 1 : (t[1] == 's' || t[1] == 'S') ? 2 : 4;
                sz = 5 + (size_t)n * es;
                break;
            }
            default: return NULL;
        }
        if (match) return t;
        p = t + 1 + sz;
    }
    return NULL;
}
