/* inflate_check.c -- the BGZF reader's DEFLATE decoder (inflate_fast.c) against zlib: synthetic streams of every block
 * type, then damaged streams (random bytes, bit flips, truncation), which must be rejected or decoded to something, never
 * read or write out of bounds (run under ASan/UBSan by tools/sanitize_host.sh), then a speed line.
 * Usage: inflate_check [n_fuzz] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include "inflate_fast.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static unsigned long long rs = 88172645463325252ull;
static unsigned rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 32); }
static void fill(unsigned char *src, size_t n, int kind) {
    for (size_t i = 0; i < n; i++) {
        switch (kind) {
            case 0: src[i] = (unsigned char)rnd(); break;
            case 1: src[i] = "ACGT"[rnd() & 3]; break;
            case 2: src[i] = (unsigned char)(i % 7 == 0 ? rnd() : src[i ? i - 1 : 0]); break;
            case 3: src[i] = (unsigned char)(i >= 300 ? src[i - 300 + (rnd() % 3 == 0)] : rnd()); break;
            case 4: src[i] = (unsigned char)((rnd() % 100 < 95) ? '!' + (rnd() % 40) : rnd()); break;
            default: src[i] = (unsigned char)(i & 255); break;
        }
    }
}
static size_t deflate_raw(const unsigned char *src, size_t n, unsigned char *cmp, size_t cap, int level, int strat) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strat);
    zs.next_in = (Bytef *)src; zs.avail_in = (uInt)n; zs.next_out = cmp; zs.avail_out = (uInt)cap;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate failed\n"); exit(2); }
    size_t clen = zs.total_out;
    deflateEnd(&zs);
    return clen;
}
int main(int argc, char **argv) {
    int n_fuzz = argc > 1 ? atoi(argv[1]) : 20000;
    int fails = 0, cases = 0;
    static const int strats[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
    for (int kind = 0; kind < 6; kind++) for (int level = 0; level <= 9; level++) for (int st = 0; st < 4; st++) for (int rep = 0; rep < 5; rep++) {
        size_t n = rep == 0 ? 0 : rep == 1 ? 1 : (size_t)(rnd() % 65536u);
        unsigned char *src = malloc(n + 8), *dst = malloc(n + 8), *cmp = malloc(n * 2 + 1024);
        fill(src, n, kind);
        size_t clen = deflate_raw(src, n, cmp, n * 2 + 1024, level, strats[st]);
        memset(dst, 0xEE, n + 8);
        int r = mm_inflate_raw(cmp, clen, dst, n);
        cases++;
        if (r != 0 || memcmp(src, dst, n) != 0 || dst[n] != 0xEE) { fails++; if (fails < 10) printf("FAIL kind %d level %d strat %d n %zu r %d\n", kind, level, st, n, r); }
        if (n > 10) {
            if (mm_inflate_raw(cmp, clen / 2, dst, n) == 0 && memcmp(src, dst, n) != 0) { fails++; printf("truncated stream accepted\n"); }
            if (mm_inflate_raw(cmp, clen, dst, n - 1) == 0) { fails++; printf("short output accepted\n"); }
        }
        free(src); free(dst); free(cmp);
    }
    /* damage: the decoder may accept or reject, it must stay inside its buffers (exact-size heap blocks: ASan sees any slip) */
    int accepted = 0;
    for (int f = 0; f < n_fuzz; f++) {
        size_t n = 1 + rnd() % 20000u;
        unsigned char *src = malloc(n), *cmp = malloc(n * 2 + 1024);
        fill(src, n, (int)(rnd() % 6));
        size_t clen = deflate_raw(src, n, cmp, n * 2 + 1024, 1 + (int)(rnd() % 9), strats[rnd() % 4]);
        int mode = (int)(rnd() % 3);
        size_t in_len = clen;
        if (mode == 0) { for (int k = 0; k < 1 + (int)(rnd() % 4); k++) cmp[rnd() % clen] ^= (unsigned char)(1u << (rnd() % 8)); }
        else if (mode == 1) { in_len = rnd() % (clen + 1); }
        else { for (size_t i = 0; i < clen; i++) cmp[i] = (unsigned char)rnd(); }
        unsigned char *in_exact = malloc(in_len ? in_len : 1);
        memcpy(in_exact, cmp, in_len);
        size_t out_len = (rnd() % 4 == 0) ? rnd() % (2 * n + 1) : n;
        unsigned char *dst = malloc(out_len ? out_len : 1);
        if (mm_inflate_raw(in_exact, in_len, dst, out_len) == 0) {
            accepted++;
            /* whatever the own decoder accepts, zlib must accept too and decode to the same bytes */
            unsigned char *ref = malloc(out_len ? out_len : 1);
            z_stream z;
            memset(&z, 0, sizeof z);
            inflateInit2(&z, -15);
            z.next_in = in_exact; z.avail_in = (uInt)in_len; z.next_out = ref; z.avail_out = (uInt)out_len;
            int zr = inflate(&z, Z_FINISH);
            if (zr != Z_STREAM_END || z.total_out != out_len || memcmp(ref, dst, out_len) != 0) { fails++; if (fails < 10) printf("accepted a stream zlib does not decode the same way (zlib %d, %lu of %zu bytes)\n", zr, z.total_out, out_len); }
            inflateEnd(&z);
            free(ref);
        }
        free(src); free(cmp); free(in_exact); free(dst);
    }
    printf("%d round trips, %d failures; %d damaged streams, %d of them still decoded to the promised size\n", cases, fails, n_fuzz, accepted);
    size_t n = 60000, reps = 2000;
    unsigned char *src = malloc(n), *dst = malloc(n + 8), *cmp = malloc(n * 2);
    for (size_t i = 0; i < n; i++) src[i] = (unsigned char)((i % 3 == 0) ? "ACGT"[rnd() & 3] : '!' + (rnd() % 30));
    size_t clen = deflate_raw(src, n, cmp, n * 2, 6, Z_DEFAULT_STRATEGY);
    double t0 = now();
    for (size_t r = 0; r < reps; r++) mm_inflate_raw(cmp, clen, dst, n);
    double t1 = now();
    for (size_t r = 0; r < reps; r++) { z_stream z; memset(&z, 0, sizeof z); inflateInit2(&z, -15); z.next_in = cmp; z.avail_in = (uInt)clen; z.next_out = dst; z.avail_out = (uInt)n; inflate(&z, Z_FINISH); inflateEnd(&z); }
    double t2 = now();
    printf("ratio %.2f: own decoder %.0f MB/s, zlib %.0f MB/s\n", (double)n / clen, n * reps / (t1 - t0) / 1e6, n * reps / (t2 - t1) / 1e6);
    free(src); free(dst); free(cmp);
    return fails ? 1 : 0;
}
