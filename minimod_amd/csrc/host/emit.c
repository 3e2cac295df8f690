/* emit.c -- print_freq_header / print_freq_output, reference src/mod.c:628-728: same columns, same "%f". */
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
