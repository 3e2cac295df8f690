/* fasta.c -- FASTA(.gz) loader (own parser over zlib's gzread).  Mirrors load_ref, reference src/ref.c:46-89 with
 * kseq semantics (src/kseq.h:195): the name ends at the first whitespace, sequence lines are concatenated, '>' starts
 * a record.  Letters are kept raw: upper-casing and U->T happen on the device (kernel K0). */
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "mmhost.h"

mmh_ref_t *mmh_load_ref(const char *path) {
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

int mmh_ref_find(const mmh_ref_t *r, const char *name) {
    for (int i = r->n - 1; i >= 0; i--) if (strcmp(r->name[i], name) == 0) return i;
    return -1;
}

void mmh_free_ref(mmh_ref_t *r) {
    if (!r) return;
    for (int i = 0; i < r->n; i++) { free(r->name[i]); free(r->seq[i]); }
    free(r->name); free(r->seq); free(r->len); free(r);
}
