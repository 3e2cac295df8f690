/* loader.c -- load_db (reference src/minimod.c:235-333) writing straight into the flattened batch of
 * include/minimod_hip.h instead of per-read mallocs: read filters, MM/ML/HP extraction (src/mod.c:123-202),
 * -K / -B batch limits.  Two pool sets let the caller fill batch N+1 while batch N is still being uploaded. */
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

enum { P_READS = 0, P_CIGAR, P_SEQ, P_MM, P_ML, P_QOFF, P_QNAME };   /* the last two stay on the host (view prints read names) */
#define NPOOL 7

typedef struct { uint8_t *p; size_t n, cap; } pool_t;

static pool_t g_sets[2][NPOOL];

static uint8_t *pool_take(pool_t *b, size_t bytes, size_t align) {
    size_t pad = (align - (b->n % align)) % align;
    size_t need = b->n + pad + bytes + 128;
    if (need > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : (1 << 22);
        while (nc < need) nc *= 2;
        b->p = (uint8_t *)realloc(b->p, nc);
        b->cap = nc;
    }
    memset(b->p + b->n, 0, pad);
    b->n += pad;
    uint8_t *r = b->p + b->n;
    b->n += bytes;
    return r;
}

mmh_loader_t *mmh_loader_open(const char *bam_path, int threads, int32_t K, int64_t B, int allow_secondary, int skip_supplementary) {
    mm_bam_t *bam = mm_bam_open(bam_path, threads);
    if (!bam) return NULL;
    mmh_loader_t *ld = (mmh_loader_t *)calloc(1, sizeof(*ld));
    ld->bam = bam; ld->K = K; ld->B = B;
    ld->allow_secondary = allow_secondary; ld->skip_supplementary = skip_supplementary;
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

int32_t mmh_loader_next(mmh_loader_t *ld, int set, mm_batch_t *out, int *more) {
    pool_t *P = g_sets[set & 1];
    for (int i = 0; i < NPOOL; i++) P[i].n = 0;
    int32_t n = 0, total = 0;
    int64_t total_bytes = 0, proc_bytes = 0;
    uint32_t max_cig = 0, max_l = 0;
    mm_bam_rec_t rec;
    int rc = 1;
    while (n < ld->K && proc_bytes < ld->B) {           /* minimod.c:249 */
        rc = mm_bam_next(ld->bam, &rec);
        if (rc <= 0) break;
        total++; total_bytes += rec.l_data;
        if (rec.flag & 0x4) continue;                                        /* unmapped, :260 */
        if (!ld->allow_secondary && (rec.flag & 0x100)) continue;           /* secondary, :265 */
        if (ld->skip_supplementary && (rec.flag & 0x800)) continue;         /* supplementary, :270 */
        if (rec.l_qseq == 0) continue;                                      /* :275 */
        const uint8_t *mmt = mm_aux_get(rec.aux, rec.l_aux, "MM");          /* get_mm_tag_ptr */
        if (!mmt || (*mmt != 'Z' && *mmt != 'H')) continue;                 /* :280-284 */
        const char *mm = (const char *)(mmt + 1);
        size_t mm_len = strlen(mm);
        const uint8_t *mlt = mm_aux_get(rec.aux, rec.l_aux, "ML");          /* get_ml_tag: B:C with len > 0, else none */
        const uint8_t *ml = NULL; uint32_t ml_len = 0;
        if (mlt && mlt[0] == 'B' && mlt[1] == 'C') {
            ml_len = (uint32_t)mlt[2] | ((uint32_t)mlt[3] << 8) | ((uint32_t)mlt[4] << 16) | ((uint32_t)mlt[5] << 24);
            ml = mlt + 6;
        }
        const uint8_t *hpt = mm_aux_get(rec.aux, rec.l_aux, "HP");          /* get_hp_tag */
        mm_read_t *rd = (mm_read_t *)pool_take(&P[P_READS], sizeof(mm_read_t), 64);
        memset(rd, 0, sizeof(*rd));
        uint8_t *c = pool_take(&P[P_CIGAR], 4 * (size_t)rec.n_cigar, 16);
        memcpy(c, rec.cigar, 4 * (size_t)rec.n_cigar);
        rd->cigar_off = (uint64_t)(c - P[P_CIGAR].p) / 4;
        size_t sb = ((size_t)rec.l_qseq + 1) / 2;
        uint8_t *s = pool_take(&P[P_SEQ], sb, 16);
        memcpy(s, rec.seq, sb);
        if (rec.l_qseq & 1) s[sb - 1] &= 0xF0;   /* the unused low nibble must be zero for the device's base counts */
        rd->seq_off = (uint64_t)(s - P[P_SEQ].p);
        uint8_t *m = pool_take(&P[P_MM], mm_len + 1, 16);
        memcpy(m, mm, mm_len + 1);
        rd->mm_off = (uint64_t)(m - P[P_MM].p);
        uint8_t *l = pool_take(&P[P_ML], ml_len, 4);
        if (ml_len) memcpy(l, ml, ml_len);
        rd->ml_off = (uint64_t)(l - P[P_ML].p);
        rd->tid = rec.tid; rd->pos = rec.pos; rd->l_qseq = (uint32_t)rec.l_qseq; rd->n_cigar = rec.n_cigar;
        rd->mm_len = (uint32_t)mm_len; rd->ml_len = ml_len; rd->flag = rec.flag;
        rd->hp = hpt ? (uint8_t)aux2i(hpt) : 0;
        {   /* bam_get_qname, printed by view (src/mod.c:571) */
            size_t ql = strlen(rec.qname);
            uint8_t *q = pool_take(&P[P_QNAME], ql + 1, 1);
            memcpy(q, rec.qname, ql + 1);
            uint64_t qo = (uint64_t)(q - P[P_QNAME].p);
            memcpy(pool_take(&P[P_QOFF], sizeof qo, 8), &qo, sizeof qo);
        }
        if (rec.n_cigar > max_cig) max_cig = rec.n_cigar;
        if ((uint32_t)rec.l_qseq > max_l) max_l = (uint32_t)rec.l_qseq;
        n++;
        proc_bytes += rec.l_data;
        ld->processed_bases += (uint64_t)rec.l_qseq;
    }
    /* padding between aligned items must not count as slack: add the zero tail every pool needs */
    (void)pool_take(&P[P_CIGAR], 64, 16); (void)pool_take(&P[P_SEQ], 64, 16);
    (void)pool_take(&P[P_MM], 64, 16); (void)pool_take(&P[P_ML], 64, 4);
    memset(P[P_CIGAR].p + P[P_CIGAR].n - 64, 0, 64); memset(P[P_SEQ].p + P[P_SEQ].n - 64, 0, 64);
    memset(P[P_MM].p + P[P_MM].n - 64, 0, 64); memset(P[P_ML].p + P[P_ML].n - 64, 0, 64);
    /* reads were taken with 64-byte alignment from an empty pool: contiguous */
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

const char *mmh_loader_qname(int set, int32_t read) {
    const pool_t *P = g_sets[set & 1];
    uint64_t qo;
    memcpy(&qo, P[P_QOFF].p + 8 * (size_t)read, sizeof qo);
    return (const char *)P[P_QNAME].p + qo;
}

void mmh_loader_close(mmh_loader_t *ld) {
    if (!ld) return;
    mm_bam_close(ld->bam);
    for (int s = 0; s < 2; s++) for (int i = 0; i < NPOOL; i++) { free(g_sets[s][i].p); memset(&g_sets[s][i], 0, sizeof(pool_t)); }
    free(ld);
}
