/*
 * synth.c -- synthetic reference + ONT / HiFi shaped reads, generated straight into the flattened batch layout of
 * include/minimod_hip.h (SURVEY.md section 8d "Synthetic inputs").  Deterministic: every read is a pure function of
 * (seed, read index), so any batch can be regenerated independently and on any number of threads.
 *
 * The reads are what load_db (reference src/minimod.c:235-333) would hand to the hot path: mapped, primary or
 * supplementary, with MM:Z / ML:B:C in the style of dorado/remora 5mC+5hmC CpG models: "C+h?,...;C+m?,...;" listing
 * every CpG of the ORIGINAL read (so for reverse-strand reads the skip counts run over the complement strand from
 * the end of SEQ, reference src/mod.c:1092-1114).
 */
#include "synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- splitmix64 / xoshiro-free small RNG ---- */
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double rng_u(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint32_t rng_below(rng_t *r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }
static inline double rng_normal(rng_t *r) {
    double u1 = rng_u(r), u2 = rng_u(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}
/* geometric with mean m (>= 1), support 1.. */
static inline uint32_t rng_geom(rng_t *r, double m) {
    if (m <= 1.0) return 1;
    double u = rng_u(r);
    if (u < 1e-300) u = 1e-300;
    double p = 1.0 / m;
    return 1u + (uint32_t)(log(u) / log(1.0 - p));
}

/* ---- reference: i.i.d. ACGT with CpG depletion, soft-masked stretches, a few N runs ---- */
/* Chunks of 1 MiB are pure functions of (seed, chunk index): any MiB-aligned slice can be generated alone. */
void mm_synth_reference_slice(uint64_t seed, int64_t begin, int64_t len, uint8_t *out_slice) {
    static const char B[4] = {'A', 'C', 'G', 'T'};
    const int64_t CH = 1 << 20;
    int64_t c0 = begin / CH;
    uint8_t *out = out_slice - c0 * CH; /* index by absolute position */
    len += c0 * CH;
    int64_t nchunks = (len + CH - 1) / CH;
    for (int64_t c = c0; c < nchunks; c++) {
        rng_t r = {seed * 0x9E3779B97F4A7C15ull + (uint64_t)c * 0xD1B54A32D192ED03ull + 12345};
        int64_t lo = c * CH, hi = lo + CH < len ? lo + CH : len;
        int prev = 'A'; /* chunks are independent */
        int64_t i = lo;
        while (i < hi) {
            uint64_t x = rng_next(&r);
            for (int k = 0; k < 12 && i < hi; k++, x >>= 5) {
                int b = B[x & 3];
                /* human-like CpG depletion: most G after C are redrawn (CpG ~ 1% of dinucleotides) */
                if ((prev == 'C') && b == 'G' && ((x >> 2) & 7) != 0) b = B[(x >> 2) & 1 ? 0 : 3];
                out[i++] = (uint8_t)b;
                prev = b;
            }
        }
        /* one soft-masked stretch and (rarely) one N run per chunk */
        rng_t q = {seed ^ (0xA5A5A5A5ull + (uint64_t)c * 77)};
        int64_t span = hi - lo;
        if (span > 4096) {
            int64_t s = lo + rng_below(&q, (uint32_t)(span - 2048)), l = 200 + rng_below(&q, 1800);
            for (int64_t j = s; j < s + l && j < hi; j++) out[j] = (uint8_t)(out[j] | 0x20);
            if ((c & 15) == 7) {
                int64_t s2 = lo + rng_below(&q, (uint32_t)(span - 1024)), l2 = 50 + rng_below(&q, 500);
                for (int64_t j = s2; j < s2 + l2 && j < hi; j++) out[j] = 'N';
            }
        }
    }
}

/* ---- growable byte pools ---- */
typedef struct { uint8_t *p; size_t n, cap; } buf_t;
static void buf_need(buf_t *b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t nc = b->cap ? b->cap * 2 : (1 << 20);
    while (nc < b->n + extra) nc *= 2;
    b->p = (uint8_t *)realloc(b->p, nc);
    b->cap = nc;
}
static void buf_pad(buf_t *b, size_t align) {
    size_t r = (align - (b->n % align)) % align;
    buf_need(b, r);
    memset(b->p + b->n, 0, r);
    b->n += r;
}

static inline int nt16(int c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}
static inline int upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

static size_t put_uint(uint8_t *dst, uint32_t v) {
    char tmp[12];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; i++) dst[i] = (uint8_t)tmp[n - 1 - i];
    return (size_t)n;
}

/* One read -> appended to the pools.  Returns the number of MM-listed calls (per code). */
static uint32_t gen_read(const mm_synth_opts_t *o, const uint8_t *ref, int64_t idx, mm_read_t *rd, buf_t *cig, buf_t *seq,
                         buf_t *mm, buf_t *ml, uint8_t **scratch, size_t *scratch_cap) {
    rng_t r = {o->seed * 0x2545F4914F6CDD1Dull + (uint64_t)idx * 0x9E3779B97F4A7C15ull + 99};
    (void)rng_next(&r);
    /* length */
    double L;
    if (o->shape == MM_SYNTH_HIFI) {
        L = 15000.0 + 3000.0 * rng_normal(&r);
        if (L < 1000) L = 1000;
    } else {
        L = exp(log(o->median_len > 0 ? o->median_len : 12000.0) + 0.668 * rng_normal(&r));
        if (L < 200) L = 200;
        if (L > (o->max_len > 0 ? o->max_len : 200000.0)) L = o->max_len > 0 ? o->max_len : 200000.0;
    }
    uint32_t want = (uint32_t)L;
    /* stratified, hence sorted, start positions */
    int64_t region = o->region_len > 0 ? o->region_len : o->contig_len;
    double slot = (double)region / (double)(o->n_reads_total > 0 ? o->n_reads_total : 1);
    int64_t pos = o->region_begin + (int64_t)(((double)idx + rng_u(&r)) * slot);
    if (pos >= o->contig_len - 64) pos = o->contig_len - 64;
    if (pos < 0) pos = 0;
    int rev = (rng_next(&r) >> 40) & 1;
    double op_gap = o->shape == MM_SYNTH_HIFI ? 650.0 : 33.0; /* mean M-run length: ~0.057 (ONT) / ~0.003 (HiFi) ops per base */
    double mism = o->shape == MM_SYNTH_HIFI ? 0.001 : 0.02;

    if (*scratch_cap < (size_t)want + 4096) {
        *scratch_cap = (size_t)want * 2 + 8192;
        *scratch = (uint8_t *)realloc(*scratch, *scratch_cap);
    }
    uint8_t *s = *scratch; /* read bases as letters, BAM orientation */
    uint32_t q = 0;
    size_t cig0 = cig->n;
    buf_need(cig, ((size_t)want / 8 + 64) * 4 + 64);
#define PUSH_OP(len, op)                                                    \
    do {                                                                    \
        buf_need(cig, 8);                                                   \
        uint32_t w_ = ((uint32_t)(len) << 4) | (uint32_t)(op);              \
        memcpy(cig->p + cig->n, &w_, 4);                                    \
        cig->n += 4;                                                        \
    } while (0)
    static const char B[4] = {'A', 'C', 'G', 'T'};
    /* leading soft clip (about 4% of the read mass at each end for 60% of ONT reads) */
    uint32_t clip5 = 0, clip3 = 0;
    if (o->shape != MM_SYNTH_HIFI) {
        if (rng_u(&r) < 0.6) clip5 = (uint32_t)(rng_u(&r) * 0.13 * want);
        if (rng_u(&r) < 0.6) clip3 = (uint32_t)(rng_u(&r) * 0.13 * want);
    }
    if (clip5) {
        for (uint32_t i = 0; i < clip5; i++) s[q++] = (uint8_t)B[rng_next(&r) >> 62];
        PUSH_OP(clip5, 4);
    }
    int64_t rp = pos;
    uint32_t body = want - clip5 - clip3;
    uint32_t made = 0;
    while (made < body && rp < o->contig_len - 1) {
        uint32_t m = rng_geom(&r, op_gap);
        if (m > body - made) m = body - made;
        if ((int64_t)m > o->contig_len - rp) m = (uint32_t)(o->contig_len - rp);
        if (m == 0) break;
        for (uint32_t i = 0; i < m; i++) {
            int c = upper(ref[rp + i]);
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') c = B[rng_next(&r) >> 62];
            if (rng_u(&r) < mism) c = B[rng_next(&r) >> 62];
            s[q++] = (uint8_t)c;
        }
        PUSH_OP(m, 0);
        rp += m; made += m;
        if (made >= body || rp >= o->contig_len - 1) break;
        if (rng_u(&r) < 0.4) { /* insertion; C5-style longer insertions carry CpGs */
            uint32_t l = rng_geom(&r, o->long_insertions ? 6.0 : 1.3);
            if (l > body - made) l = body - made;
            for (uint32_t i = 0; i < l; i++) s[q++] = (uint8_t)((o->long_insertions && (i & 1)) ? 'G' : (o->long_insertions ? 'C' : B[rng_next(&r) >> 62]));
            PUSH_OP(l, 1);
            made += l;
        } else {
            uint32_t l = rng_geom(&r, 1.75);
            if ((int64_t)l > o->contig_len - 1 - rp) l = (uint32_t)(o->contig_len - 1 - rp);
            if (l) { PUSH_OP(l, 2); rp += l; }
        }
    }
    /* a CIGAR must not end on I/D before the clip: close with a match if needed */
    {
        uint32_t lastw;
        if (cig->n > cig0) {
            memcpy(&lastw, cig->p + cig->n - 4, 4);
            if (((lastw & 15) == 1 || (lastw & 15) == 2) && rp < o->contig_len) {
                int c = upper(ref[rp]);
                if (c != 'A' && c != 'C' && c != 'G' && c != 'T') c = 'A';
                s[q++] = (uint8_t)c;
                PUSH_OP(1, 0);
                rp++;
            }
        }
    }
    if (clip3) {
        for (uint32_t i = 0; i < clip3; i++) s[q++] = (uint8_t)B[rng_next(&r) >> 62];
        PUSH_OP(clip3, 4);
    }
#undef PUSH_OP
    uint32_t lq = q;
    uint32_t ncig = (uint32_t)((cig->n - cig0) / 4);
    buf_pad(cig, 16);

    /* packed sequence */
    size_t seq0 = seq->n;
    buf_need(seq, (size_t)lq / 2 + 32);
    for (uint32_t i = 0; i + 1 < lq; i += 2) seq->p[seq->n++] = (uint8_t)((nt16(s[i]) << 4) | nt16(s[i + 1]));
    if (lq & 1) seq->p[seq->n++] = (uint8_t)(nt16(s[lq - 1]) << 4);
    buf_pad(seq, 16);

    /* MM / ML */
    size_t mm0 = mm->n, ml0 = ml->n;
    int n_groups = o->single_code ? 1 : 2;
    int dot = o->dot_fraction > 0 && rng_u(&r) < o->dot_fraction;
    /* per-read methylation state so that sites are bimodal */
    uint32_t ncalls = 0;
    for (int g = 0; g < n_groups; g++) {
        int is_m = o->single_code ? 1 : (g == 1);
        buf_need(mm, 8);
        mm->p[mm->n++] = 'C'; mm->p[mm->n++] = '+'; mm->p[mm->n++] = (uint8_t)(is_m ? 'm' : 'h');
        mm->p[mm->n++] = (uint8_t)(dot ? '.' : '?');
        rng_t rq = {o->seed ^ ((uint64_t)idx * 0xC2B2AE3D27D4EB4Full + (uint64_t)g * 977 + 5)};
        uint32_t skip = 0, calls = 0;
        if (!rev) {
            for (uint32_t i = 0; i < lq; i++) {
                if (s[i] != 'C') continue;
                int listed = i + 1 < lq && s[i + 1] == 'G';
                if (!listed) { skip++; continue; }
                buf_need(mm, 12); buf_need(ml, 1);
                mm->p[mm->n++] = ',';
                mm->n += put_uint(mm->p + mm->n, skip);
                skip = 0; calls++;
                double u = rng_u(&rq);
                uint32_t v = is_m ? (u < 0.70 ? rng_below(&rq, 21) : (u < 0.95 ? 235 + rng_below(&rq, 21) : rng_below(&rq, 256)))
                                  : (u < 0.93 ? rng_below(&rq, 16) : rng_below(&rq, 256));
                ml->p[ml->n++] = (uint8_t)v;
            }
        } else {
            for (uint32_t k = lq; k-- > 0;) {
                if (s[k] != 'G') continue;
                int listed = k > 0 && s[k - 1] == 'C';
                if (!listed) { skip++; continue; }
                buf_need(mm, 12); buf_need(ml, 1);
                mm->p[mm->n++] = ',';
                mm->n += put_uint(mm->p + mm->n, skip);
                skip = 0; calls++;
                double u = rng_u(&rq);
                uint32_t v = is_m ? (u < 0.70 ? rng_below(&rq, 21) : (u < 0.95 ? 235 + rng_below(&rq, 21) : rng_below(&rq, 256)))
                                  : (u < 0.93 ? rng_below(&rq, 16) : rng_below(&rq, 256));
                ml->p[ml->n++] = (uint8_t)v;
            }
        }
        buf_need(mm, 2);
        mm->p[mm->n++] = ';';
        ncalls += calls;
    }
    uint32_t mm_len = (uint32_t)(mm->n - mm0), ml_len = (uint32_t)(ml->n - ml0);
    buf_need(mm, 1);
    mm->p[mm->n++] = 0;
    buf_pad(mm, 16);
    buf_pad(ml, 4);

    memset(rd, 0, sizeof(*rd));
    rd->cigar_off = cig0 / 4; rd->seq_off = seq0; rd->mm_off = mm0; rd->ml_off = ml0;
    rd->tid = o->tid; rd->pos = (int32_t)pos; rd->l_qseq = lq; rd->n_cigar = ncig;
    rd->mm_len = mm_len; rd->ml_len = ml_len;
    rd->flag = (uint16_t)((rev ? 0x10 : 0) | ((rng_u(&r) < 0.03) ? 0x800 : 0));
    rd->hp = 0;
    if (o->haplotypes) { uint32_t h = rng_below(&r, 3); rd->hp = (uint8_t)h; }
    return ncalls;
}

int mm_synth_batch(const mm_synth_opts_t *o, const uint8_t *ref, int64_t first_read, int32_t n_reads, mm_host_batch_t *out) {
    if (!o || !ref || !out || n_reads < 0) return -1;
    memset(out, 0, sizeof(*out));
    buf_t cig = {0}, seq = {0}, mm = {0}, ml = {0};
    mm_read_t *reads = (mm_read_t *)calloc((size_t)(n_reads > 0 ? n_reads : 1), sizeof(mm_read_t));
    uint8_t *scratch = NULL;
    size_t scap = 0;
    uint64_t bases = 0, calls = 0;
    uint32_t max_cig = 0, max_l = 0;
    for (int32_t i = 0; i < n_reads; i++) {
        calls += gen_read(o, ref, first_read + i, &reads[i], &cig, &seq, &mm, &ml, &scratch, &scap);
        bases += reads[i].l_qseq;
        if (reads[i].n_cigar > max_cig) max_cig = reads[i].n_cigar;
        if (reads[i].l_qseq > max_l) max_l = reads[i].l_qseq;
    }
    free(scratch);
    /* >= 64 bytes of zero slack on every pool */
    buf_need(&cig, 64); memset(cig.p + cig.n, 0, 64); cig.n += 64;
    buf_need(&seq, 64); memset(seq.p + seq.n, 0, 64); seq.n += 64;
    buf_need(&mm, 64); memset(mm.p + mm.n, 0, 64); mm.n += 64;
    buf_need(&ml, 64); memset(ml.p + ml.n, 0, 64); ml.n += 64;
    out->b.reads = reads;
    out->b.cigar = (const uint32_t *)cig.p; out->b.seq = seq.p; out->b.mm = mm.p; out->b.ml = ml.p;
    out->b.order = NULL;
    out->b.n_reads = n_reads;
    out->b.n_cigar_words = cig.n / 4; out->b.n_seq_bytes = seq.n; out->b.n_mm_bytes = mm.n; out->b.n_ml_bytes = ml.n;
    out->b.max_n_cigar = max_cig; out->b.max_l_qseq = max_l;
    out->n_bases = bases; out->n_listed_calls = calls;
    return 0;
}

void mm_synth_batch_free(mm_host_batch_t *b) {
    if (!b) return;
    free((void *)b->b.reads); free((void *)b->b.cigar); free((void *)b->b.seq); free((void *)b->b.mm); free((void *)b->b.ml);
    free((void *)b->b.order);
    memset(b, 0, sizeof(*b));
}

/* longest-first processing order (counting sort on l_qseq/256) */
int mm_batch_make_order(mm_host_batch_t *hb) {
    int32_t n = hb->b.n_reads;
    if (n <= 0) return 0;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!order) return -1;
    enum { NB = 4096 };
    uint32_t *cnt = (uint32_t *)calloc(NB + 1, sizeof(uint32_t));
    for (int32_t i = 0; i < n; i++) { uint32_t k = hb->b.reads[i].l_qseq >> 8; if (k >= NB) k = NB - 1; cnt[NB - 1 - k + 1]++; }
    for (int k = 0; k < NB; k++) cnt[k + 1] += cnt[k];
    for (int32_t i = 0; i < n; i++) { uint32_t k = hb->b.reads[i].l_qseq >> 8; if (k >= NB) k = NB - 1; order[cnt[NB - 1 - k]++] = i; }
    free(cnt);
    free((void *)hb->b.order);
    hb->b.order = order;
    hb->b.n_order = n;
    return 0;
}

void mm_synth_reference(uint64_t seed, int64_t len, uint8_t *out) { mm_synth_reference_slice(seed, 0, len, out); }

/* ------------------------------------------------------------------------------------------------------------
 * BAM / FASTA writers: any flattened host batch as a real BGZF-compressed BAM (64 KB blocks + EOF marker), so the
 * same synthetic reads can be fed to the CLI (and to real htslib tools).  Filter fodder is interleaved: every 97th
 * record is followed by an unmapped copy, every 89th by a secondary copy, every 83rd by a copy without MM/ML --
 * load_db (reference src/minimod.c:260-284) must drop all three.
 * ------------------------------------------------------------------------------------------------------------ */
#include <stdio.h>
#include <zlib.h>

typedef struct { FILE *fp; uint8_t *buf; size_t n; uint64_t coff; } bgzf_w;   /* coff: file offset of the block being filled */
static int bgzf_flush_block(bgzf_w *w, const uint8_t *data, size_t len) {
    uint8_t out[65536 + 1024];
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return -1;
    zs.next_in = (Bytef *)data; zs.avail_in = (uInt)len;
    zs.next_out = out + 18; zs.avail_out = sizeof(out) - 18 - 8;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return -1; }
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    size_t total = 18 + clen + 8;
    out[16] = (uint8_t)((total - 1) & 0xFF); out[17] = (uint8_t)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len), isz = (uint32_t)len;
    memcpy(out + 18 + clen, &crc, 4); memcpy(out + 18 + clen + 4, &isz, 4);
    w->coff += total;
    return fwrite(out, 1, total, w->fp) == total ? 0 : -1;
}
static int bgzf_put(bgzf_w *w, const void *data, size_t len) {
    const uint8_t *p = (const uint8_t *)data;
    while (len) {
        size_t room = 0xFF00 - w->n, k = len < room ? len : room;
        memcpy(w->buf + w->n, p, k);
        w->n += k; p += k; len -= k;
        if (w->n == 0xFF00) { if (bgzf_flush_block(w, w->buf, w->n)) return -1; w->n = 0; }
    }
    return 0;
}

static int put_record(bgzf_w *w, const mm_batch_t *b, int32_t i, int variant, uint64_t serial) {
    const mm_read_t *rd = &b->reads[i];
    char qname[40];
    int lq = snprintf(qname, sizeof qname, "synth_%010llu", (unsigned long long)serial) + 1;
    uint32_t n_cigar = variant == 1 ? 0 : rd->n_cigar;   /* variant 1 = unmapped */
    uint16_t flag = rd->flag;
    if (variant == 1) flag |= 0x4;
    if (variant == 2) flag |= 0x100;
    int with_mm = variant != 3;
    uint32_t l_seq = rd->l_qseq;
    size_t aux = 0;
    if (with_mm) aux += 3 + rd->mm_len + 1 + 4 + 4 + rd->ml_len;
    if (rd->hp) aux += 4;
    uint32_t block = 32 + (uint32_t)lq + 4 * n_cigar + (l_seq + 1) / 2 + l_seq + (uint32_t)aux;
    uint8_t fixed[36];
    int32_t refid = variant == 1 ? -1 : rd->tid, pos = variant == 1 ? -1 : rd->pos, m1 = -1, zero = 0;
    uint32_t bin_mq_nl = ((uint32_t)4680 << 16) | (60u << 8) | (uint32_t)lq;
    uint32_t flag_nc = ((uint32_t)flag << 16) | n_cigar;
    memcpy(fixed, &block, 4); memcpy(fixed + 4, &refid, 4); memcpy(fixed + 8, &pos, 4); memcpy(fixed + 12, &bin_mq_nl, 4);
    memcpy(fixed + 16, &flag_nc, 4); memcpy(fixed + 20, &l_seq, 4); memcpy(fixed + 24, &m1, 4); memcpy(fixed + 28, &m1, 4);
    memcpy(fixed + 32, &zero, 4);
    if (bgzf_put(w, fixed, 36) || bgzf_put(w, qname, (size_t)lq)) return -1;
    if (n_cigar && bgzf_put(w, b->cigar + rd->cigar_off, 4 * (size_t)n_cigar)) return -1;
    if (bgzf_put(w, b->seq + rd->seq_off, (l_seq + 1) / 2)) return -1;
    {
        uint8_t q[4096];
        memset(q, 0xFF, sizeof q);
        for (uint32_t left = l_seq; left;) { uint32_t k = left < sizeof q ? left : (uint32_t)sizeof q; if (bgzf_put(w, q, k)) return -1; left -= k; }
    }
    if (with_mm) {
        uint8_t t[8] = {'M', 'M', 'Z'};
        if (bgzf_put(w, t, 3) || bgzf_put(w, b->mm + rd->mm_off, rd->mm_len) || bgzf_put(w, "", 1)) return -1;
        uint8_t t2[8] = {'M', 'L', 'B', 'C'};
        memcpy(t2 + 4, &rd->ml_len, 4);
        if (bgzf_put(w, t2, 8) || (rd->ml_len && bgzf_put(w, b->ml + rd->ml_off, rd->ml_len))) return -1;
    }
    if (rd->hp) { uint8_t t3[4] = {'H', 'P', 'C', rd->hp}; if (bgzf_put(w, t3, 4)) return -1; }
    return 0;
}

/* BAI index under construction (SAM specification section 5.2): per reference the bins with their chunks and the linear
 * index (smallest virtual offset of an alignment overlapping each 16 kb window) */
typedef struct { uint32_t bin; uint64_t beg, end; } bai_chunk_t;
typedef struct { bai_chunk_t *chunks; size_t n, cap; uint64_t *lin; size_t n_lin, cap_lin; } bai_ref_t;
typedef struct mm_bam_writer { bgzf_w w; uint64_t serial; int flags; bai_ref_t *idx; int32_t n_ref; uint64_t n_no_coor; char *path; } mm_bam_writer_t;

static int reg2bin(int64_t beg, int64_t end) {   /* SAM specification section 5.3 */
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
static uint64_t voffset(const bgzf_w *w) { return (w->coff << 16) | (uint64_t)w->n; }
static void bai_add(mm_bam_writer_t *bw, int32_t tid, int64_t beg, int64_t end, uint64_t v0, uint64_t v1) {
    if (!bw->idx) return;
    if (tid < 0 || tid >= bw->n_ref) { bw->n_no_coor++; return; }
    bai_ref_t *r = &bw->idx[tid];
    if (end <= beg) end = beg + 1;
    uint32_t bin = (uint32_t)reg2bin(beg, end);
    if (r->n && r->chunks[r->n - 1].bin == bin && r->chunks[r->n - 1].end == v0) r->chunks[r->n - 1].end = v1;   /* the record continues the chunk */
    else {
        if (r->n == r->cap) { r->cap = r->cap ? r->cap * 2 : 1024; r->chunks = (bai_chunk_t *)realloc(r->chunks, r->cap * sizeof(bai_chunk_t)); }
        r->chunks[r->n].bin = bin; r->chunks[r->n].beg = v0; r->chunks[r->n].end = v1; r->n++;
    }
    size_t w1 = (size_t)((end - 1) >> 14) + 1;
    if (w1 > r->cap_lin) {
        size_t nc = r->cap_lin ? r->cap_lin : 1024;
        while (nc < w1) nc *= 2;
        r->lin = (uint64_t *)realloc(r->lin, nc * sizeof(uint64_t));
        memset(r->lin + r->cap_lin, 0, (nc - r->cap_lin) * sizeof(uint64_t));
        r->cap_lin = nc;
    }
    for (size_t w = (size_t)(beg >> 14); w < w1; w++) if (r->lin[w] == 0) r->lin[w] = v0 + 1;   /* kept + 1: 0 is "not set", and a piece without a header has a record AT 0 */
    if (w1 > r->n_lin) r->n_lin = w1;
}
static int chunk_cmp(const void *a, const void *b) {
    const bai_chunk_t *x = (const bai_chunk_t *)a, *y = (const bai_chunk_t *)b;
    if (x->bin != y->bin) return x->bin < y->bin ? -1 : 1;
    return x->beg < y->beg ? -1 : (x->beg > y->beg);
}
static int bai_write(mm_bam_writer_t *bw, const char *path) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fwrite("BAI\1", 1, 4, fp);
    fwrite(&bw->n_ref, 4, 1, fp);
    for (int32_t t = 0; t < bw->n_ref; t++) {
        bai_ref_t *r = &bw->idx[t];
        qsort(r->chunks, r->n, sizeof(bai_chunk_t), chunk_cmp);
        int32_t n_bin = 0;
        for (size_t i = 0; i < r->n; i++) if (i == 0 || r->chunks[i].bin != r->chunks[i - 1].bin) n_bin++;
        fwrite(&n_bin, 4, 1, fp);
        for (size_t i = 0; i < r->n;) {
            size_t j = i;
            while (j < r->n && r->chunks[j].bin == r->chunks[i].bin) j++;
            int32_t n_chunk = (int32_t)(j - i);
            fwrite(&r->chunks[i].bin, 4, 1, fp); fwrite(&n_chunk, 4, 1, fp);
            for (size_t k = i; k < j; k++) { fwrite(&r->chunks[k].beg, 8, 1, fp); fwrite(&r->chunks[k].end, 8, 1, fp); }
            i = j;
        }
        int32_t n_intv = (int32_t)r->n_lin;
        fwrite(&n_intv, 4, 1, fp);
        for (size_t w = 1; w < r->n_lin; w++) if (r->lin[w] == 0) r->lin[w] = r->lin[w - 1];   /* windows nothing overlaps, as samtools fills them */
        for (size_t w = 0; w < r->n_lin; w++) if (r->lin[w]) r->lin[w]--;
        if (n_intv) fwrite(r->lin, 8, (size_t)n_intv, fp);
    }
    fwrite(&bw->n_no_coor, 8, 1, fp);
    return fclose(fp);
}

/* flags: MM_BAMW_NO_HEADER = a later piece of a file written in pieces (BGZF members concatenate), MM_BAMW_NO_EOF = a piece
 * that is not the last; first_serial numbers the records (read names, filter fodder) as one writer would have */
mm_bam_writer_t *mm_bam_writer_open_piece(const char *path, int32_t n_contigs, const char *const *names, const int64_t *lens,
                                          int flags, uint64_t first_serial) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return NULL;
    mm_bam_writer_t *bw = (mm_bam_writer_t *)calloc(1, sizeof(*bw));
    if (!bw) { fclose(fp); return NULL; }
    bw->w.fp = fp; bw->w.buf = (uint8_t *)malloc(0x10000);
    bw->flags = flags; bw->serial = first_serial;
    if (flags & MM_BAMW_INDEX) {
        bw->n_ref = n_contigs; bw->idx = (bai_ref_t *)calloc((size_t)(n_contigs > 0 ? n_contigs : 1), sizeof(bai_ref_t));
        bw->path = (char *)malloc(strlen(path) + 5);
        if (bw->path) { strcpy(bw->path, path); strcat(bw->path, ".bai"); }
    }
    if (!bw->w.buf) { fclose(fp); free(bw); return NULL; }
    if (flags & MM_BAMW_NO_HEADER) return bw;
    char text[256];
    int lt = snprintf(text, sizeof text, "@HD\tVN:1.6\tSO:coordinate\n");
    bgzf_put(&bw->w, "BAM\1", 4);
    int32_t l_text = lt;
    bgzf_put(&bw->w, &l_text, 4); bgzf_put(&bw->w, text, (size_t)lt);
    bgzf_put(&bw->w, &n_contigs, 4);
    for (int32_t i = 0; i < n_contigs; i++) {
        int32_t ln = (int32_t)strlen(names[i]) + 1, ll = (int32_t)lens[i];
        bgzf_put(&bw->w, &ln, 4); bgzf_put(&bw->w, names[i], (size_t)ln); bgzf_put(&bw->w, &ll, 4);
    }
    return bw;
}
mm_bam_writer_t *mm_bam_writer_open(const char *path, int32_t n_contigs, const char *const *names, const int64_t *lens) {
    return mm_bam_writer_open_piece(path, n_contigs, names, lens, 0, 0);
}
static int64_t ref_end(const mm_batch_t *b, const mm_read_t *rd) {
    int64_t e = rd->pos;
    for (uint32_t k = 0; k < rd->n_cigar; k++) {
        uint32_t w = b->cigar[rd->cigar_off + k], op = w & 15u;
        if ((0x18Du >> op) & 1u) e += w >> 4;   /* M D N = X consume the reference */
    }
    return e;
}
static int put_indexed(mm_bam_writer_t *bw, const mm_batch_t *b, int32_t i, int variant) {
    const uint64_t v0 = voffset(&bw->w);
    if (put_record(&bw->w, b, i, variant, bw->serial)) return -1;
    if (bw->idx) {
        const mm_read_t *rd = &b->reads[i];
        if (variant == 1) bai_add(bw, -1, 0, 0, v0, voffset(&bw->w));
        else bai_add(bw, rd->tid, rd->pos, ref_end(b, rd), v0, voffset(&bw->w));
    }
    return 0;
}
int mm_bam_writer_put_batch(mm_bam_writer_t *bw, const mm_batch_t *b, int with_filter_fodder) {
    for (int32_t i = 0; i < b->n_reads; i++) {
        if (put_indexed(bw, b, i, 0)) return -1;
        if (with_filter_fodder) {
            if (bw->serial % 97 == 5 && put_indexed(bw, b, i, 1)) return -1;
            if (bw->serial % 89 == 7 && put_indexed(bw, b, i, 2)) return -1;
            if (bw->serial % 83 == 11 && put_indexed(bw, b, i, 3)) return -1;
        }
        bw->serial++;
    }
    return 0;
}
int mm_bam_writer_close(mm_bam_writer_t *bw) {
    int r = 0;
    if (bw->w.n) r |= bgzf_flush_block(&bw->w, bw->w.buf, bw->w.n);
    if (!(bw->flags & MM_BAMW_NO_EOF)) r |= bgzf_flush_block(&bw->w, bw->w.buf, 0);   /* the 28-byte EOF marker: an empty block */
    r |= fclose(bw->w.fp);
    if (bw->idx) {
        if (!bw->path || bai_write(bw, bw->path)) r |= -1;
        for (int32_t t = 0; t < bw->n_ref; t++) { free(bw->idx[t].chunks); free(bw->idx[t].lin); }
        free(bw->idx); free(bw->path);
    }
    free(bw->w.buf); free(bw);
    return r;
}
int mm_write_fasta(const char *path, const char *name, const uint8_t *seq, int64_t len) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fprintf(fp, ">%s synthetic\n", name);
    for (int64_t i = 0; i < len; i += 80) {
        int64_t k = len - i < 80 ? len - i : 80;
        fwrite(seq + i, 1, (size_t)k, fp); fputc('\n', fp);
    }
    return fclose(fp);
}
