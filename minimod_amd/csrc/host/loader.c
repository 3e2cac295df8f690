/* loader.c -- load_db (reference src/minimod.c:235-333) writing straight into the flattened batch of
 * include/minimod_hip.h instead of per-read mallocs: read filters, MM/ML/HP extraction (src/mod.c:123-202),
 * -K / -B batch limits.  A batch is built in two steps: the records are framed and filtered in file order (that part is
 * sequential by definition: the -K / -B limits count ACCEPTED reads), then the worker pool copies them into the pools
 * in parallel, every record to offsets fixed beforehand.  Two pool sets let the caller fill batch N+1 while batch N is
 * still being uploaded. */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

#include "mmhost.h"

enum { P_READS = 0, P_CIGAR, P_SEQ, P_MM, P_ML, P_QOFF, P_QNAME };   /* the last two stay on the host (view prints read names) */
#define NPOOL 7

typedef struct { uint8_t *p; size_t n, cap; int foreign; } pool_t;   /* foreign: from the caller's allocator */


/* one accepted record: where its parts lie in the reader's buffers and where they go in the pools */
typedef struct {
    const uint8_t *cigar, *seq, *mm, *ml;
    const char *qname;
    uint32_t n_cigar, l_qseq, mm_len, ml_len, qlen;
    int32_t tid, pos;
    uint16_t flag;
    uint8_t hp;
    size_t o_cigar, o_seq, o_mm, o_ml, o_qname;   /* byte offsets */
} item_t;

/* per-loader state: two pool sets and the item array (two loaders in one process do not share anything) */
typedef struct loader_priv {
    pool_t sets[MMH_POOL_SETS][NPOOL];
    item_t *items;
    size_t items_cap;
    /* a worker of a sharded run reads only the alignments that START inside its share [lo, hi) of the genome */
    int ranged, first, last, seen, done;
    double t_frame, t_copy;   /* seconds in step 1 (framing + filters, waits for decoded data included) and step 2 (the copies) */
    int32_t lo_tid, hi_tid;
    int64_t lo_pos, hi_pos;
} loader_priv_t;
#define PRIV(ld) ((loader_priv_t *)(ld)->priv)

/* The device-bound pools (records, CIGARs, sequences, MM, ML) may come from an allocator of the caller's: pinned memory makes
 * mm_freq_submit's host -> device copies DMA transfers instead of the runtime's staged copies (mmh_loader_set_allocator). */
static void *(*pool_alloc_fn)(size_t) = NULL;
static void (*pool_free_fn)(void *) = NULL;
void mmh_loader_set_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *)) { pool_alloc_fn = alloc_fn; pool_free_fn = free_fn; }

static void pool_release(pool_t *b) {
    if (b->p) { if (b->foreign) pool_free_fn(b->p); else free(b->p); }
    b->p = NULL; b->cap = 0; b->foreign = 0;
}
static void pool_reserve_kind(pool_t *b, size_t bytes, int device_bound) {
    if (bytes > b->cap) {
        size_t nc = b->cap ? b->cap : ((size_t)1 << 22);
        while (nc < bytes) nc *= 2;
        pool_release(b);                  /* contents are rebuilt for every batch */
        if (device_bound && pool_alloc_fn && pool_free_fn) {
            /* pinning costs by the byte and the pools of a run settle at the size of its biggest batch: no doubling, a quarter more */
            nc = bytes + bytes / 4 + ((size_t)1 << 20);
            b->p = (uint8_t *)pool_alloc_fn(nc);
            b->foreign = b->p != NULL;
        }
        if (!b->p) { b->p = (uint8_t *)malloc(nc); b->foreign = 0; }
        b->cap = nc;
    }
    b->n = bytes;
}
static void pool_reserve(pool_t *b, size_t bytes) { pool_reserve_kind(b, bytes, 0); }
static double loader_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

mmh_loader_t *mmh_loader_open(const char *bam_path, int threads, int32_t K, int64_t B, int allow_secondary, int skip_supplementary) {
    mm_bam_t *bam = mm_bam_open(bam_path, threads);
    if (!bam) return NULL;
    mmh_loader_t *ld = (mmh_loader_t *)calloc(1, sizeof(*ld));
    loader_priv_t *pv = (loader_priv_t *)calloc(1, sizeof(*pv));
    if (!ld || !pv) { free(ld); free(pv); mm_bam_close(bam); return NULL; }
    ld->priv = pv;
    ld->bam = bam; ld->K = K; ld->B = B;
    ld->allow_secondary = allow_secondary; ld->skip_supplementary = skip_supplementary;
    return ld;
}

mmh_loader_t *mmh_loader_open_share(const char *bam_path, int threads, int32_t K, int64_t B, int allow_secondary, int skip_supplementary,
                                    uint64_t voffset, int32_t lo_tid, int64_t lo_pos, int32_t hi_tid, int64_t hi_pos, int first, int last) {
    mm_bam_t *bam = voffset == UINT64_MAX ? mm_bam_open(bam_path, threads) : mm_bam_open_at(bam_path, threads, voffset);
    if (!bam) return NULL;
    mmh_loader_t *ld = (mmh_loader_t *)calloc(1, sizeof(*ld));
    loader_priv_t *pv = (loader_priv_t *)calloc(1, sizeof(*pv));
    if (!ld || !pv) { free(ld); free(pv); mm_bam_close(bam); return NULL; }
    ld->priv = pv;
    ld->bam = bam; ld->K = K; ld->B = B;
    ld->allow_secondary = allow_secondary; ld->skip_supplementary = skip_supplementary;
    pv->ranged = 1; pv->first = first; pv->last = last;
    pv->lo_tid = lo_tid; pv->lo_pos = lo_pos; pv->hi_tid = hi_tid; pv->hi_pos = hi_pos;
    pv->done = voffset == UINT64_MAX;   /* the index has nothing at or behind the share's start */
    return ld;
}

static int aux2i(const uint8_t *t) {   /* bam_aux2i on the type byte */
    switch (*t) {
        case 'c': return (int8_t)t[1];
        case 'C': return t[1];
        case 's': return (int16_t)(t[1] | (t[2] << 8));
        case 'S': return (uint16_t)(t[1] | (t[2] << 8));
        case 'i': case 'I': return (int)((uint32_t)t[1] | ((uint32_t)t[2] << 8) | ((uint32_t)t[3] << 16) | ((uint32_t)t[4] << 24));
        default: return 0;
    }
}

typedef struct { pool_t *P; const item_t *it; } copy_ctx_t;

/* copy records [lo, hi): every byte of the pools between the items (alignment padding) is zeroed by the item behind it */
static unsigned long long copy_busy_ns;   /* diagnostics: nanoseconds inside copy_range, all workers */
static void copy_range(void *arg, int64_t lo, int64_t hi) {
    const double t_c = loader_now();
    const copy_ctx_t *c = (const copy_ctx_t *)arg;
    pool_t *P = c->P;
    for (int64_t i = lo; i < hi; i++) {
        const item_t *it = &c->it[i];
        const item_t *pv = i > 0 ? &c->it[i - 1] : NULL;
        size_t e;
        mm_read_t *rd = (mm_read_t *)P[P_READS].p + i;
        memset(rd, 0, sizeof(*rd));
        e = pv ? pv->o_cigar + 4 * (size_t)pv->n_cigar : 0;
        memset(P[P_CIGAR].p + e, 0, it->o_cigar - e);
        memcpy(P[P_CIGAR].p + it->o_cigar, it->cigar, 4 * (size_t)it->n_cigar);
        size_t sb = ((size_t)it->l_qseq + 1) / 2;
        e = pv ? pv->o_seq + ((size_t)pv->l_qseq + 1) / 2 : 0;
        memset(P[P_SEQ].p + e, 0, it->o_seq - e);
        memcpy(P[P_SEQ].p + it->o_seq, it->seq, sb);
        if (it->l_qseq & 1) P[P_SEQ].p[it->o_seq + sb - 1] &= 0xF0;   /* the unused low nibble must be zero for the device's base counts */
        e = pv ? pv->o_mm + pv->mm_len + 1 : 0;
        memset(P[P_MM].p + e, 0, it->o_mm - e);
        memcpy(P[P_MM].p + it->o_mm, it->mm, it->mm_len);
        P[P_MM].p[it->o_mm + it->mm_len] = 0;
        e = pv ? pv->o_ml + pv->ml_len : 0;
        memset(P[P_ML].p + e, 0, it->o_ml - e);
        if (it->ml_len) memcpy(P[P_ML].p + it->o_ml, it->ml, it->ml_len);
        memcpy(P[P_QNAME].p + it->o_qname, it->qname, it->qlen + 1);   /* bam_get_qname, printed by view (src/mod.c:571) */
        uint64_t qo = it->o_qname;
        memcpy(P[P_QOFF].p + 8 * (size_t)i, &qo, sizeof qo);
        rd->cigar_off = it->o_cigar / 4; rd->seq_off = it->o_seq; rd->mm_off = it->o_mm; rd->ml_off = it->o_ml;
        rd->tid = it->tid; rd->pos = it->pos; rd->l_qseq = it->l_qseq; rd->n_cigar = it->n_cigar;
        rd->mm_len = it->mm_len; rd->ml_len = it->ml_len; rd->flag = it->flag; rd->hp = it->hp;
    }
    __atomic_fetch_add(&copy_busy_ns, (unsigned long long)((loader_now() - t_c) * 1e9), __ATOMIC_RELAXED);
}

int32_t mmh_loader_next(mmh_loader_t *ld, int set, mm_batch_t *out, int *more) {
    loader_priv_t *lp = PRIV(ld);
    pool_t *P = lp->sets[(unsigned)set % MMH_POOL_SETS];
    int32_t n = 0, total = 0;
    int64_t total_bytes = 0, proc_bytes = 0;
    uint32_t max_cig = 0, max_l = 0;
    size_t o_cigar = 0, o_seq = 0, o_mm = 0, o_ml = 0, o_qname = 0;
    mm_bam_rec_t rec;
    int rc = 1;
    const double t_1 = loader_now();
    /* ---- step 1: frame + filter, in file order (minimod.c:249-333) */
    while (n < ld->K && proc_bytes < ld->B) {           /* minimod.c:249 */
        if (lp->done) { rc = 0; break; }
        rc = mm_bam_next(ld->bam, &rec);
        if (rc <= 0) break;
        if (lp->ranged) {
            if (!(rec.flag & 0x4) && rec.tid >= 0) {
                if (rec.tid < lp->lo_tid || (rec.tid == lp->lo_tid && rec.pos < lp->lo_pos)) continue;   /* the left neighbour's read */
                if (!lp->last && (rec.tid > lp->hi_tid || (rec.tid == lp->hi_tid && rec.pos >= lp->hi_pos))) { lp->done = 1; rc = 0; break; }
                lp->seen = 1;
            } else if (!lp->seen && !lp->first) continue;   /* unplaced records in front of the share are the left neighbour's to count */
        }
        total++; total_bytes += rec.l_data;
        if (rec.flag & 0x4) continue;                                        /* unmapped, :260 */
        if (!ld->allow_secondary && (rec.flag & 0x100)) continue;           /* secondary, :265 */
        if (ld->skip_supplementary && (rec.flag & 0x800)) continue;         /* supplementary, :270 */
        if (rec.l_qseq == 0) continue;                                      /* :275 */
        /* one walk over the tags: first MM (get_mm_tag_ptr), first ML (get_ml_tag: B:C with len > 0), first HP (get_hp_tag) */
        const uint8_t *mmt = mm_aux_get(rec.aux, rec.l_aux, "MM");
        if (!mmt || (*mmt != 'Z' && *mmt != 'H')) continue;                 /* :280-284 */
        const char *mm = (const char *)(mmt + 1);
        size_t mm_len = strlen(mm);
        const uint8_t *mlt = mm_aux_get(rec.aux, rec.l_aux, "ML");
        const uint8_t *ml = NULL; uint32_t ml_len = 0;
        if (mlt && mlt[0] == 'B' && mlt[1] == 'C') {
            ml_len = (uint32_t)mlt[2] | ((uint32_t)mlt[3] << 8) | ((uint32_t)mlt[4] << 16) | ((uint32_t)mlt[5] << 24);
            ml = mlt + 6;
            if ((size_t)ml_len > (size_t)(rec.aux + rec.l_aux - ml)) continue;   /* cannot happen after mm_aux_get's bounds check */
        }
        const uint8_t *hpt = mm_aux_get(rec.aux, rec.l_aux, "HP");
        if ((size_t)n == lp->items_cap) {
            size_t ncap = lp->items_cap ? lp->items_cap * 2 : 1024;
            item_t *grown = (item_t *)realloc(lp->items, ncap * sizeof(item_t));
            if (!grown) { rc = -1; break; }
            lp->items = grown; lp->items_cap = ncap;
        }
        item_t *it = &lp->items[n];
        it->cigar = (const uint8_t *)rec.cigar; it->seq = rec.seq; it->mm = (const uint8_t *)mm; it->ml = ml;
        it->qname = rec.qname; it->qlen = (uint32_t)strlen(rec.qname);
        it->n_cigar = rec.n_cigar; it->l_qseq = (uint32_t)rec.l_qseq; it->mm_len = (uint32_t)mm_len; it->ml_len = ml_len;
        it->tid = rec.tid; it->pos = rec.pos; it->flag = rec.flag;
        it->hp = hpt ? (uint8_t)aux2i(hpt) : 0;
        /* pool offsets: cigar/seq/mm items start on 16-byte boundaries, ml on 4 */
        o_cigar = align_up(o_cigar, 16); it->o_cigar = o_cigar; o_cigar += 4 * (size_t)rec.n_cigar;
        o_seq = align_up(o_seq, 16); it->o_seq = o_seq; o_seq += ((size_t)rec.l_qseq + 1) / 2;
        o_mm = align_up(o_mm, 16); it->o_mm = o_mm; o_mm += mm_len + 1;
        o_ml = align_up(o_ml, 4); it->o_ml = o_ml; o_ml += ml_len;
        it->o_qname = o_qname; o_qname += it->qlen + 1;
        if (rec.n_cigar > max_cig) max_cig = rec.n_cigar;
        if ((uint32_t)rec.l_qseq > max_l) max_l = (uint32_t)rec.l_qseq;
        n++;
        proc_bytes += rec.l_data;
        ld->processed_bases += (uint64_t)rec.l_qseq;
    }
    const double t_2 = loader_now();
    lp->t_frame += t_2 - t_1;
    /* ---- step 2: sizes are known: reserve, copy in parallel, zero the tails (every pool ends in >= 64 zero bytes) */
    size_t e_cigar = o_cigar, e_seq = o_seq, e_mm = o_mm, e_ml = o_ml;
    o_cigar = align_up(o_cigar, 16) + 64; o_seq = align_up(o_seq, 16) + 64; o_mm = align_up(o_mm, 16) + 64; o_ml = align_up(o_ml, 4) + 64;
    pool_reserve_kind(&P[P_READS], sizeof(mm_read_t) * (size_t)(n > 0 ? n : 1), 1);
    pool_reserve_kind(&P[P_CIGAR], o_cigar, 1); pool_reserve_kind(&P[P_SEQ], o_seq, 1); pool_reserve_kind(&P[P_MM], o_mm, 1); pool_reserve_kind(&P[P_ML], o_ml, 1);
    pool_reserve(&P[P_QOFF], 8 * (size_t)(n > 0 ? n : 1)); pool_reserve(&P[P_QNAME], o_qname + 1);
    copy_ctx_t cc = {P, lp->items};
    mm_pool_t *pool = mm_bam_pool(ld->bam);
    int nt = mm_pool_threads(pool);
    int64_t grain = n / (4 * (nt > 0 ? nt : 1)) + 1;
    mm_pool_for(pool, n, grain, copy_range, &cc);
    memset(P[P_CIGAR].p + e_cigar, 0, o_cigar - e_cigar); memset(P[P_SEQ].p + e_seq, 0, o_seq - e_seq);
    memset(P[P_MM].p + e_mm, 0, o_mm - e_mm); memset(P[P_ML].p + e_ml, 0, o_ml - e_ml);
    mm_bam_release(ld->bam);
    lp->t_copy += loader_now() - t_2;
    memset(out, 0, sizeof(*out));
    out->reads = (const mm_read_t *)P[P_READS].p;
    out->cigar = (const uint32_t *)P[P_CIGAR].p; out->seq = P[P_SEQ].p; out->mm = P[P_MM].p; out->ml = P[P_ML].p;
    out->n_reads = n;
    out->n_cigar_words = P[P_CIGAR].n / 4; out->n_seq_bytes = P[P_SEQ].n; out->n_mm_bytes = P[P_MM].n; out->n_ml_bytes = P[P_ML].n;
    out->max_n_cigar = max_cig; out->max_l_qseq = max_l;
    ld->last_total_reads = total; ld->last_total_bytes = total_bytes; ld->last_processed_bytes = proc_bytes;
    ld->total_reads += (uint64_t)total; ld->total_bytes += (uint64_t)total_bytes;
    ld->processed_reads += (uint64_t)n; ld->processed_bytes += (uint64_t)proc_bytes;
    *more = (n >= ld->K || proc_bytes >= ld->B);         /* freq_main.c:410 */
    if (rc < 0) return -1;
    return n;
}

const char *mmh_loader_qname(const mmh_loader_t *ld, int set, int32_t read) {
    const pool_t *P = PRIV(ld)->sets[(unsigned)set % MMH_POOL_SETS];
    uint64_t qo;
    memcpy(&qo, P[P_QOFF].p + 8 * (size_t)read, sizeof qo);
    return (const char *)P[P_QNAME].p + qo;
}

void mmh_loader_close(mmh_loader_t *ld) {
    if (!ld) return;
    if (getenv("MM_LOADER_TIMING")) {
        double busy = 0, sub = 0; unsigned long long jobs = 0;
        mm_pool_stats(mm_bam_pool(ld->bam), &busy, &jobs, &sub);
        fprintf(stderr, "[loader] framing + filters %.3f s (of which waiting for decoded data %.3f s), copies %.3f s, %d threads; "
                        "pool: %llu jobs, %.3f s inside jobs (%.3f s of them copies), %.3f s queueing\n",
                PRIV(ld)->t_frame, mm_bam_wait_seconds(ld->bam), PRIV(ld)->t_copy, mm_pool_threads(mm_bam_pool(ld->bam)),
                jobs, busy, 1e-9 * (double)copy_busy_ns, sub);
    }
    mm_bam_close(ld->bam);
    loader_priv_t *lp = PRIV(ld);
    for (int s = 0; s < MMH_POOL_SETS; s++) for (int i = 0; i < NPOOL; i++) pool_release(&lp->sets[s][i]);
    free(lp->items);
    free(lp);
    free(ld);
}
