/* fasta_check.c -- the mapped FASTA parser (fasta.c, parallel passes) against the stream parser on the files given; built with
 * sanitizers by tools/sanitize_host.sh.  Usage: fasta_check file.fa [more files] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mmhost.h"
int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        mmh_ref_t *want = mmh_load_ref_mt(argv[a], 0);
        if (!want) { fprintf(stderr, "cannot read %s\n", argv[a]); return 1; }
        for (int t = 1; t <= 5; t += 2) {
            mmh_ref_t *got = mmh_load_ref_mt(argv[a], t);
            int bad = !got || got->n != want->n;
            for (int i = 0; !bad && i < want->n; i++)
                bad = strcmp(got->name[i], want->name[i]) != 0 || got->len[i] != want->len[i] ||
                      (want->len[i] && memcmp(got->seq[i], want->seq[i], (size_t)want->len[i]) != 0);
            if (bad) { fprintf(stderr, "%s: the parsers disagree at %d threads\n", argv[a], t); return 1; }
            mmh_free_ref(got);
        }
        printf("%s: %d records, parsers agree\n", argv[a], want->n);
        mmh_free_ref(want);
    }
    return 0;
}
