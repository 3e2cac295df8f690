/* emit_check.c -- the row formatter (emit.c) alone: formats N synthetic freq rows with T pool threads into a file and,
 * with T > 1, checks the bytes against the serial run.  Built with sanitizers by tools/sanitize_host.sh; also times it.
 * Usage: emit_check n_rows threads out_path */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mmhost.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static char *slurp(const char *path, long *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); *len = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)*len + 1);
    if (fread(b, 1, (size_t)*len, f) != (size_t)*len) { fclose(f); free(b); return NULL; }
    fclose(f);
    return b;
}
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: emit_check n_rows threads out_path\n"); return 2; }
    long n = atol(argv[1]);
    int th = atoi(argv[2]);
    mm_row_t *rows = (mm_row_t *)calloc((size_t)n, sizeof(mm_row_t));
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        rows[i].tid = (int32_t)(i * 3 / n); rows[i].pos = (int32_t)(s % 250000000u); rows[i].strand = (uint8_t)((s >> 40) & 1);
        rows[i].code = (int16_t)((s >> 41) % 3); rows[i].hp = -1;
        rows[i].n_called = 1 + (uint32_t)((s >> 44) % ((s >> 60) == 0 ? 5000 : 60));
        rows[i].n_mod = (uint32_t)((s >> 20) % (rows[i].n_called + 1));
    }
    char *names[3] = {"chr1", "chr2", "chrM"};
    uint32_t lens[3] = {0, 0, 0};
    mm_bam_hdr_t hdr = {3, names, lens};
    const char *codes[3] = {"m", "h", "21839"};
    char path1[4096];
    snprintf(path1, sizeof path1, "%s.serial", argv[3]);
    double t[2];
    for (int pass = 0; pass < 2; pass++) {
        mm_pool_t *pool = (pass == 1 && th > 1) ? mm_pool_create(th) : NULL;
        FILE *fp = fopen(pass ? argv[3] : path1, "wb");
        if (!fp) { perror("fopen"); return 1; }
        double t0 = now();
        mmh_print_freq_rows(fp, pool, rows, n, &hdr, codes, 3, 1, 0, 0);
        if (mmh_emit_flush() != 0) { fprintf(stderr, "write failed\n"); return 1; }
        t[pass] = now() - t0;
        fclose(fp);
        if (pool) mm_pool_destroy(pool);
    }
    if (mmh_emit_finish() != 0) return 1;
    long la = 0, lb = 0;
    char *a = slurp(path1, &la), *b = slurp(argv[3], &lb);
    int same = a && b && la == lb && memcmp(a, b, (size_t)la) == 0;
    printf("%ld rows, %ld bytes: serial %.3f s (%.2f GB/s), %d threads %.3f s (%.2f GB/s), %s\n", n, la, t[0], la / t[0] / 1e9, th, t[1],
           lb / t[1] / 1e9, same ? "identical" : "DIFFERENT");
    free(a); free(b); free(rows);
    remove(path1);
    return same ? 0 : 1;
}
