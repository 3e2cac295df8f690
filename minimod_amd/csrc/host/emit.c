/* emit.c -- print_freq_header / print_freq_output, reference src/mod.c:628-728: same columns, same "%f". */
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

void mmh_print_freq_header(FILE *fp, int bedmethyl, int insertions, int haplotypes) {
    if (bedmethyl) return;
    fprintf(fp, "contig\tstart\tend\tstrand\tn_called\tn_mod\tfreq\tmod_code%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
}

/* "%f" of n_mod / n_called (x100 for bedmethyl) through snprintf once per distinct pair of small counts: rows repeat
 * the same few hundred ratios millions of times, and the text must be printf's own (src/mod.c:685,703) */
#define FREQ_MEMO 256
static const char *freq_str(uint32_t n_mod, uint32_t n_called, int percent, int *len, char *tmp) {
    static char (*memo[2])[FREQ_MEMO][12];
    static uint8_t (*memo_len[2])[FREQ_MEMO];
    double f = percent ? (double)n_mod * 100 / n_called : (double)n_mod / n_called;
    if (n_called >= FREQ_MEMO || n_mod >= FREQ_MEMO) { *len = snprintf(tmp, 32, "%f", f); return tmp; }
    if (!memo[percent]) {
        memo[percent] = (char (*)[FREQ_MEMO][12])calloc(FREQ_MEMO, sizeof(*memo[percent]));
        memo_len[percent] = (uint8_t (*)[FREQ_MEMO])calloc(FREQ_MEMO, sizeof(*memo_len[percent]));
    }
    if (!memo_len[percent][n_called][n_mod])
        memo_len[percent][n_called][n_mod] = (uint8_t)snprintf(memo[percent][n_called][n_mod], 12, "%f", f);
    *len = memo_len[percent][n_called][n_mod];
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

void mmh_print_freq_rows(FILE *fp, const mm_row_t *rows, int64_t n, const mm_bam_hdr_t *hdr, mm_freq_t *h, int bedmethyl,
                         int insertions, int haplotypes) {
    const size_t cap = 1 << 20;
    char *buf = (char *)malloc(cap + 4096), *p = buf, tmp[40];
    int32_t last_tid = -2;
    const char *contig = "*";
    size_t clen = 1;
    for (int64_t i = 0; i < n; i++) {
        const mm_row_t *r = &rows[i];
        if (r->tid != last_tid) {
            last_tid = r->tid;
            contig = (r->tid >= 0 && r->tid < hdr->n_targets) ? hdr->target_name[r->tid] : "*";
            clen = strlen(contig);
        }
        const char *code = mm_freq_code_name(h, r->code);
        size_t codelen = strlen(code);
        if ((size_t)(p - buf) + clen + codelen + 160 > cap) { fwrite(buf, 1, (size_t)(p - buf), fp); p = buf; }
        if (clen + codelen + 160 > cap) continue;   /* cannot happen: names are far shorter than the buffer */
        char strand = r->strand ? '-' : '+';
        int flen;
        const char *fs = freq_str(r->n_mod, r->n_called, bedmethyl, &flen, tmp);
        if (bedmethyl) {   /* src/mod.c:685 */
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
            if (insertions) { *p++ = '\t'; p = put_int(p, r->ins_offset); }
            if (haplotypes) { *p++ = '\t'; if (r->hp == -1) *p++ = '*'; else p = put_int(p, r->hp); }
        }
        *p++ = '\n';
    }
    if (p > buf) fwrite(buf, 1, (size_t)(p - buf), fp);
    free(buf);
}

/* ---- view: print_view_header / print_view_output, reference src/mod.c:545-626.  One line per row; the line is
 * assembled by hand (a batch is millions of rows and fprintf("%f") dominates otherwise): the 256 possible mod_prob
 * strings are printed once with the reference's own format. */
void mmh_print_view_header(FILE *fp, int insertions, int haplotypes) {
    fprintf(fp, "ref_contig\tref_pos\tstrand\tread_id\tread_pos\tmod_code\tmod_prob%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
}


void mmh_print_view_rows(FILE *fp, const mm_view_row_t *rows, int64_t n, const mm_batch_t *batch, int pool_set,
                         const mm_bam_hdr_t *hdr, mm_freq_t *h, int insertions, int haplotypes) {
    static char prob[256][16];
    static int prob_len[256];
    if (!prob_len[0])
        for (int x = 0; x < 256; x++) prob_len[x] = snprintf(prob[x], sizeof prob[x], "%f", (x + 0.5) / 256.0);   /* THRESH_UINT8_TO_DBL */
    const size_t cap = 1 << 20;
    char *buf = (char *)malloc(cap + 4096), *p = buf;
    uint32_t last_read = 0xFFFFFFFFu;
    const char *qname = "", *contig = "*";
    size_t qlen = 0, clen = 1;
    const mm_read_t *rd = NULL;
    for (int64_t i = 0; i < n; i++) {
        const mm_view_row_t *r = &rows[i];
        if (r->read != last_read) {
            last_read = r->read;
            rd = &batch->reads[r->read];
            qname = mmh_loader_qname(pool_set, (int32_t)r->read);
            qlen = strlen(qname);
            contig = (rd->tid >= 0 && rd->tid < hdr->n_targets) ? hdr->target_name[rd->tid] : "*";
            clen = strlen(contig);
        }
        const char *code = mm_freq_code_name(h, r->code);
        if ((size_t)(p - buf) + clen + qlen + 128 > cap) { fwrite(buf, 1, (size_t)(p - buf), fp); p = buf; }
        if (clen + qlen + 128 > cap) {   /* absurdly long names: let stdio do it */
            fprintf(fp, "%s\t%d\t%c\t%s\t%d\t%s\t%s", contig, r->pos, (rd->flag & 0x10) ? '-' : '+', qname, (int)r->read_pos, code, prob[r->prob]);
            if (insertions) fprintf(fp, "\t%d", (int)r->ins_offset);
            if (haplotypes) fprintf(fp, "\t%d", (int)rd->hp);
            fputc('\n', fp);
            continue;
        }
        p = put_str(p, contig, clen); *p++ = '\t';
        p = put_int(p, r->pos); *p++ = '\t';
        *p++ = (rd->flag & 0x10) ? '-' : '+'; *p++ = '\t';
        p = put_str(p, qname, qlen); *p++ = '\t';
        p = put_int(p, (long)r->read_pos); *p++ = '\t';
        p = put_str(p, code, strlen(code)); *p++ = '\t';
        p = put_str(p, prob[r->prob], (size_t)prob_len[r->prob]);
        if (insertions) { *p++ = '\t'; p = put_int(p, r->ins_offset); }
        if (haplotypes) { *p++ = '\t'; p = put_int(p, rd->hp); }
        *p++ = '\n';
    }
    if (p > buf) fwrite(buf, 1, (size_t)(p - buf), fp);
    free(buf);
}
