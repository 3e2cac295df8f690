// fmt_check.cpp -- csrc/fmt_core.h against snprintf (built and run by tests/test_fmt_cpu.py): every count pair up to a bound, random
// pairs over the whole 32-bit range, and the doubles that sit exactly on a rounding tie.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "fmt_core.h"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static long check_f(double q) {
    char a[64], b[64];
    uint64_t ip; uint32_t fr;
    mm_f6_parts(q, &ip, &fr);
    char* e = mm_put_f6(a, ip, fr); *e = 0;
    snprintf(b, sizeof b, "%f", q);
    if (strcmp(a, b) != 0 || (int)(e - a) != mm_f6_len(ip)) { printf("MISMATCH %a: %s != %s\n", q, a, b); return 1; }
    return 0;
}
static long check_row(mm_fmt_row_in_t r) {
    char a[256], b[256];
    char* e = mm_row_write(a, &r); *e = 0;
    const double f = r.bedmethyl ? (double)r.n_mod * 100 / r.n_called : (double)r.n_mod / r.n_called;
    int n;
    if (r.bedmethyl) n = snprintf(b, sizeof b, "%s\t%d\t%d\t%s\t%ld\t%c\t%d\t%d\t255,0,0\t%ld\t%f\n", r.contig, r.pos, r.pos + 1, r.code, (long)r.n_called, r.strand ? '-' : '+', r.pos, r.pos + 1, (long)r.n_called, f);
    else {
        n = snprintf(b, sizeof b, "%s\t%d\t%d\t%c\t%ld\t%ld\t%f\t%s", r.contig, r.pos, r.pos, r.strand ? '-' : '+', (long)r.n_called, (long)r.n_mod, f, r.code);
        if (r.insertions) n += snprintf(b + n, sizeof b - n, "\t%d", r.ins_offset);
        if (r.haplotypes) { if (r.hp == -1) n += snprintf(b + n, sizeof b - n, "\t*"); else n += snprintf(b + n, sizeof b - n, "\t%d", r.hp); }
        n += snprintf(b + n, sizeof b - n, "\n");
    }
    if (strcmp(a, b) != 0 || mm_row_len(&r) != n || (int)(e - a) != n) { printf("ROW MISMATCH:\n%s%s(len %d / %d / %d)\n", a, b, mm_row_len(&r), (int)(e - a), n); return 1; }
    return 0;
}

int main(int argc, char** argv) {
    const long bound = argc > 1 ? atol(argv[1]) : 1500, nrand = argc > 2 ? atol(argv[2]) : 3000000;
    long bad = 0, n = 0;
    for (long c = 1; c <= bound; c++)
        for (long m = 0; m <= c; m++) { bad += check_f((double)m / c); bad += check_f((double)m * 100 / c); n += 2; }
    for (long i = 0; i < nrand; i++) {
        uint32_t c = (uint32_t)(rnd() >> (rnd() % 33 + 31)); if (!c) c = 1;
        uint32_t m = (uint32_t)(rnd() % ((uint64_t)c + 1));
        bad += check_f((double)m / c); bad += check_f((double)m * 100 / c); n += 2;
    }
    // exact ties: (2j + 1) * 15625 / 2^k  (the seventh decimal is a 5 and nothing follows), and neighbours one ulp away
    for (int k = 7; k <= 30; k++)
        for (long j = 0; j < 4000; j++) {
            const double q = (double)((2 * j + 1) * 15625ull) / (double)(1ull << k);
            if (q >= 100.5) break;
            bad += check_f(q); n++;
            uint64_t b; memcpy(&b, &q, 8);
            uint64_t b1 = b + 1, b0 = b - 1; double q1, q0; memcpy(&q1, &b1, 8); memcpy(&q0, &b0, 8);
            bad += check_f(q1); bad += check_f(q0); n += 2;
        }
    const double specials[] = {0.0, 1.0, 100.0, 99.9999995, 99.99999949999999, 0.9999995, 0.99999949999, 0.0000005, 0.00000049999999, 1e-300, 4.9e-324, 0.5, 0.0078125, 33.333333333333336, 66.66666666666667};
    for (double q : specials) { bad += check_f(q); n++; }
    // whole rows
    const char* contigs[] = {"chr1", "chrX", "c", "a_rather_long_contig_name.1"};
    const char* codes[] = {"m", "h", "76792", "hm"};
    for (long i = 0; i < 400000; i++) {
        mm_fmt_row_in_t r;
        r.contig = contigs[rnd() % 4]; r.contig_len = (int)strlen(r.contig);
        r.code = codes[rnd() % 4]; r.code_len = (int)strlen(r.code);
        r.pos = (int32_t)(rnd() >> (rnd() % 30 + 33));
        r.n_called = (uint32_t)(rnd() >> (rnd() % 31 + 32)); if (!r.n_called) r.n_called = 1;
        r.n_mod = (uint32_t)(rnd() % ((uint64_t)r.n_called + 1));
        r.strand = (int)(rnd() & 1); r.ins_offset = (int)(rnd() % 65536); r.hp = (int)(rnd() % 6) - 1;
        r.bedmethyl = (int)(rnd() % 3 == 0); r.insertions = (int)(rnd() & 1); r.haplotypes = (int)(rnd() & 1);
        bad += check_row(r); n++;
    }
    printf("%ld checks, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
