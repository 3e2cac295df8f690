/* emit.c -- print_freq_header / print_freq_output, reference src/mod.c:628-728: same columns, same "%f". */
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

void mmh_print_freq_header(FILE *fp, int bedmethyl, int insertions, int haplotypes) {
    if (bedmethyl) return;
    fprintf(fp, "contig\tstart\tend\tstrand\tn_called\tn_mod\tfreq\tmod_code%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
}

void mmh_print_freq_rows(FILE *fp, const mm_row_t *rows, int64_t n, const mm_bam_hdr_t *hdr, mm_freq_t *h, int bedmethyl,
                         int insertions, int haplotypes) {
    for (int64_t i = 0; i < n; i++) {
        const mm_row_t *r = &rows[i];
        const char *contig = (r->tid >= 0 && r->tid < hdr->n_targets) ? hdr->target_name[r->tid] : "*";
        const char *code = mm_freq_code_name(h, r->code);
        char strand = r->strand ? '-' : '+';
        if (bedmethyl) {
            double f = (double)r->n_mod * 100 / r->n_called;
            int end = r->pos + 1;
            fprintf(fp, "%s\t%d\t%d\t%s\t%d\t%c\t%d\t%d\t255,0,0\t%d\t%f\n", contig, r->pos, end, code, (int)r->n_called, strand,
                    r->pos, end, (int)r->n_called, f);
        } else {
            double f = (double)r->n_mod / r->n_called;
            fprintf(fp, "%s\t%d\t%d\t%c\t%d\t%d\t%f\t%s", contig, r->pos, r->pos, strand, (int)r->n_called, (int)r->n_mod, f, code);
            if (insertions) fprintf(fp, "\t%d", (int)r->ins_offset);
            if (haplotypes) { if (r->hp == -1) fputs("\t*", fp); else fprintf(fp, "\t%d", (int)r->hp); }
            fputc('\n', fp);
        }
    }
}

/* ---- view: print_view_header / print_view_output, reference src/mod.c:545-626.  One line per row; the line is
 * assembled by hand (a batch is millions of rows and fprintf("%f") dominates otherwise): the 256 possible mod_prob
 * strings are printed once with the reference's own format. */
void mmh_print_view_header(FILE *fp, int insertions, int haplotypes) {
    fprintf(fp, "ref_contig\tref_pos\tstrand\tread_id\tread_pos\tmod_code\tmod_prob%s%s\n", insertions ? "\tins_offset" : "",
            haplotypes ? "\thaplotype" : "");
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
