// fmt_core.h -- the text of one output row of `minimod freq` (print_freq_output, reference src/mod.c:666-719), character for character,
// without printf: plain integer arithmetic that compiles for the device (csrc/fmt_api.hip: k_fmt_len / k_fmt_write) and for the host
// (tests/test_fmt_cpu.py builds it with g++ and compares it with snprintf on millions of count pairs).
//
//   TSV        "%s\t%d\t%d\t%c\t%d\t%d\t%f\t%s" [ "\t%d" ins_offset ] [ "\t*" | "\t%d" haplotype ] "\n"     freq = (double)n_mod / n_called
//   bedmethyl  "%s\t%d\t%d\t%s\t%d\t%c\t%d\t%d\t255,0,0\t%d\t%f\n"                                         freq = (double)n_mod * 100 / n_called
//
// "%f" is six decimals of the EXACT value of the double, rounded to nearest, ties to even (what glibc does): the integer part is taken
// off (exact), the fraction's 53-bit mantissa times 10^6 is a 73-bit integer, shifted right by the fraction's binary exponent with the
// remainder compared against one half.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define MM_HD __host__ __device__ inline
#else
#define MM_HD static inline
#endif

MM_HD int mm_u64_digits(uint64_t v) {
    int n = 1;
    while (v >= 10ull) { v /= 10ull; n++; }
    return n;
}
MM_HD int mm_i64_len(int64_t v) { return v < 0 ? 1 + mm_u64_digits((uint64_t)(-v)) : mm_u64_digits((uint64_t)v); }
MM_HD char* mm_put_u64(char* p, uint64_t v) {
    const int n = mm_u64_digits(v);
    for (int i = n - 1; i >= 0; i--) { p[i] = (char)('0' + (int)(v % 10ull)); v /= 10ull; }
    return p + n;
}
MM_HD char* mm_put_i64(char* p, int64_t v) {
    if (v < 0) { *p++ = '-'; return mm_put_u64(p, (uint64_t)(-v)); }
    return mm_put_u64(p, (uint64_t)v);
}
MM_HD void mm_mul_64x64(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
    const uint64_t a0 = a & 0xFFFFFFFFull, a1 = a >> 32, b0 = b & 0xFFFFFFFFull, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (p01 & 0xFFFFFFFFull) + (p10 & 0xFFFFFFFFull);
    *lo = (p00 & 0xFFFFFFFFull) | (mid << 32);
    *hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}
// q >= 0 finite, below 2^32: *ip = digits in front of the point, *frac = the six behind it (0 .. 999999), as "%f" prints them
MM_HD void mm_f6_parts(double q, uint64_t* ip, uint32_t* frac) {
    uint64_t whole = (uint64_t)q;          // truncation: exact
    const double fr = q - (double)whole;   // exact (both are multiples of q's last bit, the difference is smaller than q)
    uint32_t r = 0;
    if (fr > 0.0) {
        uint64_t bits;
        memcpy(&bits, &fr, 8);
        const int e = (int)((bits >> 52) & 0x7FFull);
        uint64_t mant = bits & 0xFFFFFFFFFFFFFull;
        int shift;                          // fr = mant * 2^-shift
        if (e == 0) shift = 1074; else { mant |= 1ull << 52; shift = 1075 - e; }
        if (shift < 75) {                   // (from 75 on the product, below 2^73, is less than a quarter: rounds to nothing)
            uint64_t hi, lo;
            mm_mul_64x64(mant, 1000000ull, &hi, &lo);
            uint64_t res, rem_hi, rem_lo, half_hi, half_lo;
            if (shift >= 64) {
                const int s = shift - 64;
                res = s ? (hi >> s) : hi;
                rem_hi = s ? (hi & ((1ull << s) - 1ull)) : 0ull; rem_lo = lo;
                half_hi = s ? (1ull << (s - 1)) : 0ull; half_lo = s ? 0ull : (1ull << 63);
            } else {                        // 53 <= shift < 64 (fr < 1)
                res = (hi << (64 - shift)) | (lo >> shift);
                rem_hi = 0ull; rem_lo = lo & ((1ull << shift) - 1ull);
                half_hi = 0ull; half_lo = 1ull << (shift - 1);
            }
            const int gt = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
            const int eq = rem_hi == half_hi && rem_lo == half_lo;
            if (gt || (eq && (res & 1ull))) res++;
            r = (uint32_t)res;
            if (r >= 1000000u) { r -= 1000000u; whole++; }
        }
    }
    *ip = whole; *frac = r;
}
MM_HD int mm_f6_len(uint64_t ip) { return mm_u64_digits(ip) + 7; }
MM_HD char* mm_put_f6(char* p, uint64_t ip, uint32_t frac) {
    p = mm_put_u64(p, ip);
    *p++ = '.';
    for (int i = 5; i >= 0; i--) { p[i] = (char)('0' + (int)(frac % 10u)); frac /= 10u; }
    return p + 6;
}

typedef struct mm_fmt_row_in {
    const char* contig; int contig_len;
    const char* code; int code_len;
    int32_t pos; uint32_t n_called, n_mod; int strand, ins_offset, hp;
    int bedmethyl, insertions, haplotypes;
} mm_fmt_row_in_t;

MM_HD double mm_row_freq(const mm_fmt_row_in_t* r) {
    return r->bedmethyl ? (double)r->n_mod * 100 / r->n_called : (double)r->n_mod / r->n_called;
}
MM_HD int mm_row_len(const mm_fmt_row_in_t* r) {
    uint64_t ip; uint32_t fr;
    mm_f6_parts(mm_row_freq(r), &ip, &fr);
    int n = r->contig_len + 1;
    if (r->bedmethyl) {
        const int lp = mm_i64_len(r->pos), le = mm_i64_len((int64_t)r->pos + 1), lc = mm_u64_digits(r->n_called);
        n += lp + 1 + le + 1 + r->code_len + 1 + lc + 1 + 1 + 1 + lp + 1 + le + 9 + lc + 1 + mm_f6_len(ip);
    } else {
        const int lp = mm_i64_len(r->pos);
        n += lp + 1 + lp + 1 + 1 + 1 + mm_u64_digits(r->n_called) + 1 + mm_u64_digits(r->n_mod) + 1 + mm_f6_len(ip) + 1 + r->code_len;
        if (r->insertions) n += 1 + mm_u64_digits((uint64_t)r->ins_offset);
        if (r->haplotypes) n += 1 + (r->hp == -1 ? 1 : mm_i64_len(r->hp));
    }
    return n + 1;
}
MM_HD char* mm_put_bytes(char* p, const char* s, int n) { for (int i = 0; i < n; i++) p[i] = s[i]; return p + n; }
MM_HD char* mm_row_write(char* p, const mm_fmt_row_in_t* r) {
    uint64_t ip; uint32_t fr;
    mm_f6_parts(mm_row_freq(r), &ip, &fr);
    const char strand = r->strand ? '-' : '+';
    p = mm_put_bytes(p, r->contig, r->contig_len); *p++ = '\t';
    if (r->bedmethyl) {   /* src/mod.c:685 */
        p = mm_put_i64(p, r->pos); *p++ = '\t'; p = mm_put_i64(p, (int64_t)r->pos + 1); *p++ = '\t';
        p = mm_put_bytes(p, r->code, r->code_len); *p++ = '\t';
        p = mm_put_u64(p, r->n_called); *p++ = '\t'; *p++ = strand; *p++ = '\t';
        p = mm_put_i64(p, r->pos); *p++ = '\t'; p = mm_put_i64(p, (int64_t)r->pos + 1);
        p = mm_put_bytes(p, "\t255,0,0\t", 9);
        p = mm_put_u64(p, r->n_called); *p++ = '\t';
        p = mm_put_f6(p, ip, fr);
    } else {              /* src/mod.c:703-715 */
        p = mm_put_i64(p, r->pos); *p++ = '\t'; p = mm_put_i64(p, r->pos); *p++ = '\t';
        *p++ = strand; *p++ = '\t';
        p = mm_put_u64(p, r->n_called); *p++ = '\t'; p = mm_put_u64(p, r->n_mod); *p++ = '\t';
        p = mm_put_f6(p, ip, fr); *p++ = '\t';
        p = mm_put_bytes(p, r->code, r->code_len);
        if (r->insertions) { *p++ = '\t'; p = mm_put_u64(p, (uint64_t)r->ins_offset); }
        if (r->haplotypes) { *p++ = '\t'; if (r->hp == -1) *p++ = '*'; else p = mm_put_i64(p, r->hp); }
    }
    *p++ = '\n';
    return p;
}
