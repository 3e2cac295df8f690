/* crc32_fast.c -- CRC-32 (gzip polynomial, reflected 0xEDB88320) of a BGZF block's payload.
 *
 * Every BGZF block ends in CRC32 + ISIZE of its decoded bytes (RFC 1952 section 2.3.1); htslib -- the reference's
 * reader -- fails a file whose block does not match (bgzf.c check_header/inflate_block), so this reader checks it too
 * (bamio.c).  zlib's table-driven crc32() runs at ~1 GB/s per core, a tenth of what a worker inflates at, so on x86-64
 * hosts with carry-less multiply the CRC is folded 64 bytes at a time (the method of Gopal et al., "Fast CRC Computation
 * for Generic Polynomials Using PCLMULQDQ", Intel 2009; constants for this polynomial as published there and used by
 * zlib-ng / Chromium): x^(n) mod P for the fold distances, then a Barrett reduction.  Anything else -- short buffers,
 * other CPUs -- goes to zlib. */
#include "crc32_fast.h"

#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>

__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_pclmul(uint32_t crc, const uint8_t *buf, size_t len) {
    /* len >= 64 and a multiple of 16 on entry */
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};   /* x^(4*128+32), x^(4*128-32) mod P */
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};   /* x^(128+32), x^(128-32) mod P */
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};   /* x^64 mod P */
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ull, 0x01f7011641ull};   /* P', mu */
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {   /* four lanes, 64 bytes per step */
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    /* four lanes into one */
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {   /* single lane, 16 bytes per step */
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    /* 128 -> 64 bits */
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    /* Barrett reduction 64 -> 32 bits */
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

static int have_pclmul(void) {   /* (every thread that finds -1 stores the same answer: relaxed atomics keep it race-free) */
    static int known = -1;
    int k = __atomic_load_n(&known, __ATOMIC_RELAXED);
    if (k < 0) { k = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); __atomic_store_n(&known, k, __ATOMIC_RELAXED); }
    return k;
}
#endif

uint32_t mm_crc32(const uint8_t *buf, size_t len) {
    uint32_t crc = 0;   /* crc32(0, NULL, 0) */
#if defined(__x86_64__)
    if (len >= 64 && have_pclmul()) {
        size_t body = len & ~(size_t)15;
        crc = ~crc32_pclmul(~crc, buf, body);
        buf += body; len -= body;
    }
#endif
    while (len) {   /* zlib takes uInt lengths */
        uInt k = len > 0x40000000u ? 0x40000000u : (uInt)len;
        crc = (uint32_t)crc32(crc, buf, k);
        buf += k; len -= k;
    }
    return crc;
}
