/* inflate_fast.c -- raw DEFLATE decoder for the BGZF reader (block inflate is the end-to-end limit).
 *
 * Own code, written from RFC 1951.  What makes it quicker than a general streaming inflate for this job: the whole
 * input and the whole output of a block are in memory (no state machine that can stop anywhere), a 64-bit bit buffer
 * refilled eight bytes at a time, two-level decode tables (11 bits for literals/lengths, 8 for distances) whose entries
 * carry the extra-bit count and base value so that a symbol costs one lookup, a second literal decoded from the same
 * refill, and matches copied eight bytes at a time. */
#include "inflate_fast.h"

#include <string.h>

#define LL_BITS 11
#define D_BITS 8
#define LL_SUB_MAX 1334   /* generous bounds on second-level entries (at most 15 - LL_BITS / 15 - D_BITS extra bits) */
#define D_SUB_MAX 402

/* table entry: bits 0-3 code length consumed at this level (0 = invalid), bits 4-7 kind, bits 8-12 extra bits,
 * bits 16-31 value (literal byte, base length / distance, or sub-table offset) */
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_DIST = 4, K_BAD = 15 };
#define E_INVALID ((uint32_t)K_BAD << 4)
#define ENT(len, kind, extra, val) ((uint32_t)(len) | ((uint32_t)(kind) << 4) | ((uint32_t)(extra) << 8) | ((uint32_t)(val) << 16))
#define E_LEN(e) ((e) & 15u)
#define E_KIND(e) (((e) >> 4) & 15u)
#define E_EXTRA(e) (((e) >> 8) & 31u)
#define E_VAL(e) ((e) >> 16)

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t rev_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

static uint32_t sym_entry(int is_dist, int sym, int len_here) {
    if (is_dist) return sym < 30 ? ENT(len_here, K_DIST, dist_extra[sym], dist_base[sym]) : E_INVALID;
    if (sym < 256) return ENT(len_here, K_LIT, 0, sym);
    if (sym == 256) return ENT(len_here, K_EOB, 0, 0);
    if (sym <= 285) return ENT(len_here, K_LEN, len_extra[sym - 257], len_base[sym - 257]);
    return E_INVALID;   /* 286, 287: never valid in a stream */
}

/* Canonical Huffman code (lens[0..n)) -> two-level table.  Returns 0, or -1 for a code zlib would refuse too: an
 * over-subscribed one, or an incomplete one unless it has no symbols at all or a single one-bit code (unused slots stay
 * invalid). */
static int build_table(const uint8_t *lens, int n, int is_dist, int root, uint32_t *tab, int sub_max) {
    int count[16] = {0}, next[16];
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1, max_len = 0;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return -1; if (count[l]) max_len = l; }
    if (left > 0 && max_len > 1) return -1;
    int code = 0;
    for (int l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
    const int root_size = 1 << root;
    for (int i = 0; i < root_size; i++) tab[i] = E_INVALID;
    /* sub-tables: one per distinct root prefix of the long codes, sized for the longest code under that prefix */
    int sub_bits[1 << LL_BITS];
    int any_long = 0;
    for (int l = root + 1; l <= 15; l++) if (count[l]) any_long = 1;
    if (any_long) {
        for (int i = 0; i < root_size; i++) sub_bits[i] = 0;
        int nx[16];
        for (int l = 1; l <= 15; l++) nx[l] = next[l];
        for (int s = 0; s < n; s++) {
            int l = lens[s];
            if (l <= root) { if (l) nx[l]++; continue; }
            uint32_t c = rev_bits((uint32_t)nx[l]++, l);
            int prefix = (int)(c & (uint32_t)(root_size - 1));
            if (l - root > sub_bits[prefix]) sub_bits[prefix] = l - root;
        }
        int off = root_size;
        for (int i = 0; i < root_size; i++) {
            if (!sub_bits[i]) continue;
            if (off + (1 << sub_bits[i]) > root_size + sub_max) return -1;
            tab[i] = ENT(root, K_SUB, sub_bits[i], off);
            for (int j = 0; j < (1 << sub_bits[i]); j++) tab[off + j] = E_INVALID;
            off += 1 << sub_bits[i];
        }
    }
    for (int s = 0; s < n; s++) {
        int l = lens[s];
        if (!l) continue;
        uint32_t c = rev_bits((uint32_t)next[l]++, l);
        if (l <= root) {
            uint32_t e = sym_entry(is_dist, s, l);
            for (uint32_t i = c; i < (uint32_t)root_size; i += 1u << l) tab[i] = e;
        } else {
            uint32_t prefix = c & (uint32_t)(root_size - 1);
            uint32_t sb = E_EXTRA(tab[prefix]), off = E_VAL(tab[prefix]);
            uint32_t e = sym_entry(is_dist, s, l - root);
            for (uint32_t i = c >> root; i < (1u << sb); i += 1u << (l - root)) tab[off + i] = e;
        }
    }
    return 0;
}

static uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }   /* little-endian hosts */

typedef struct {
    const uint8_t *in, *in_end;
    uint64_t bits;
    int nbits;
} bitrd_t;

/* at least 56 valid bits after a refill while input lasts; past the end zero bits are shifted in (and detected by
 * the callers through position checks) */
static inline void refill(bitrd_t *b) {
    if (b->in + 8 <= b->in_end) {
        b->bits |= load64(b->in) << b->nbits;
        int take = (63 - b->nbits) >> 3;
        b->in += take;
        b->nbits += take << 3;
    } else {
        while (b->nbits <= 56 && b->in < b->in_end) { b->bits |= (uint64_t)*b->in++ << b->nbits; b->nbits += 8; }
    }
}
static inline uint32_t peek(const bitrd_t *b, int n) { return (uint32_t)(b->bits & ((1ull << n) - 1ull)); }
static inline void drop(bitrd_t *b, int n) { b->bits >>= n; b->nbits -= n; }

int mm_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    static const uint8_t clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t ll_tab[(1 << LL_BITS) + LL_SUB_MAX], d_tab[(1 << D_BITS) + D_SUB_MAX];
    bitrd_t b = {in, in + in_len, 0, 0};
    uint8_t *o = out, *const o_end = out + out_len;
    int final = 0;
    while (!final) {
        refill(&b);
        if (b.nbits < 3) return -1;
        final = (int)peek(&b, 1); drop(&b, 1);
        uint32_t type = peek(&b, 2); drop(&b, 2);
        if (type == 0) {   /* stored */
            drop(&b, b.nbits & 7);
            refill(&b);
            if (b.nbits < 32) return -1;
            uint32_t len = peek(&b, 16); drop(&b, 16);
            uint32_t nlen = peek(&b, 16); drop(&b, 16);
            if ((len ^ nlen) != 0xFFFFu) return -1;
            /* bytes still in the bit buffer belong to the stored data */
            while (len && b.nbits >= 8) { if (o >= o_end) return -1; *o++ = (uint8_t)peek(&b, 8); drop(&b, 8); len--; }
            if (len) {   /* the bit buffer is empty now (whole bytes only); what refill() left above its valid bits is a
                          * preview of the bytes at b.in and must go before b.in jumps ahead */
                if ((size_t)(b.in_end - b.in) < len || (size_t)(o_end - o) < len) return -1;
                memcpy(o, b.in, len); o += len; b.in += len;
                b.bits = 0; b.nbits = 0;
            }
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {   /* fixed codes */
            uint8_t lens[288 + 32];
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            for (i = 0; i < 32; i++) lens[288 + i] = 5;
            if (build_table(lens, 288, 0, LL_BITS, ll_tab, LL_SUB_MAX) || build_table(lens + 288, 32, 1, D_BITS, d_tab, D_SUB_MAX)) return -1;
        } else {           /* dynamic codes */
            refill(&b);
            if (b.nbits < 14) return -1;
            int hlit = (int)peek(&b, 5) + 257; drop(&b, 5);
            int hdist = (int)peek(&b, 5) + 1; drop(&b, 5);
            int hclen = (int)peek(&b, 4) + 4; drop(&b, 4);
            if (hlit > 286 || hdist > 30) return -1;
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; i++) {
                refill(&b);
                if (b.nbits < 3) return -1;
                cl[clen_order[i]] = (uint8_t)peek(&b, 3); drop(&b, 3);
            }
            uint32_t cl_tab[1 << 7];
            {   /* code-length code: 7 bits at most, one level */
                int count[8] = {0}, next[8], left = 1, code = 0;
                for (int i = 0; i < 19; i++) count[cl[i]]++;
                count[0] = 0;
                for (int l = 1; l <= 7; l++) { left = (left << 1) - count[l]; if (left < 0) return -1; }
                if (left > 0) return -1;   /* the code-length code must be complete */
                for (int l = 1; l <= 7; l++) { code = (code + count[l - 1]) << 1; next[l] = code; }
                for (int i = 0; i < 128; i++) cl_tab[i] = 0;
                for (int s = 0; s < 19; s++) {
                    int l = cl[s];
                    if (!l) continue;
                    uint32_t c = rev_bits((uint32_t)next[l]++, l);
                    for (uint32_t i = c; i < 128u; i += 1u << l) cl_tab[i] = ENT(l, 0, 0, s);
                }
            }
            uint8_t lens[286 + 30 + 138];
            int n = 0, total = hlit + hdist;
            while (n < total) {
                refill(&b);
                uint32_t e = cl_tab[peek(&b, 7)];
                if (!E_LEN(e) || (int)E_LEN(e) > b.nbits) return -1;
                drop(&b, (int)E_LEN(e));
                uint32_t s = E_VAL(e);
                if (s < 16) { lens[n++] = (uint8_t)s; continue; }
                int rep, val = 0;
                if (s == 16) { if (n == 0) return -1; val = lens[n - 1]; rep = 3 + (int)peek(&b, 2); drop(&b, 2); }
                else if (s == 17) { rep = 3 + (int)peek(&b, 3); drop(&b, 3); }
                else { rep = 11 + (int)peek(&b, 7); drop(&b, 7); }
                if (b.nbits < 0 || n + rep > total) return -1;
                while (rep--) lens[n++] = (uint8_t)val;
            }
            if (lens[256] == 0) return -1;   /* no end-of-block code */
            if (build_table(lens, hlit, 0, LL_BITS, ll_tab, LL_SUB_MAX) || build_table(lens + hlit, hdist, 1, D_BITS, d_tab, D_SUB_MAX)) return -1;
        }
        /* ---- symbols of the block */
        for (;;) {
            /* fast path: far enough from both ends that nothing needs a bounds check (a symbol pair consumes at most 48
             * bits = 6 bytes of input and produces at most 258 bytes; copies may run 8 bytes over) */
            while ((size_t)(b.in_end - b.in) >= 16 && (size_t)(o_end - o) >= 3 + 258 + 16) {
                b.bits |= load64(b.in) << b.nbits;
                { int take = (63 - b.nbits) >> 3; b.in += take; b.nbits += take << 3; }
                uint32_t e = ll_tab[b.bits & ((1u << LL_BITS) - 1u)];
                if ((e & 0xF0u) == 0) {   /* literal; up to two more from the same refill */
                    *o++ = (uint8_t)(e >> 16); b.bits >>= (e & 15u); b.nbits -= (int)(e & 15u);
                    e = ll_tab[b.bits & ((1u << LL_BITS) - 1u)];
                    if ((e & 0xF0u) == 0) {
                        *o++ = (uint8_t)(e >> 16); b.bits >>= (e & 15u); b.nbits -= (int)(e & 15u);
                        e = ll_tab[b.bits & ((1u << LL_BITS) - 1u)];
                        if ((e & 0xF0u) == 0) { *o++ = (uint8_t)(e >> 16); b.bits >>= (e & 15u); b.nbits -= (int)(e & 15u); }
                    }
                    continue;
                }
                if (E_KIND(e) == K_SUB) {
                    uint32_t sb = E_EXTRA(e), off = E_VAL(e);
                    b.bits >>= LL_BITS; b.nbits -= LL_BITS;
                    e = ll_tab[off + (uint32_t)(b.bits & ((1u << sb) - 1u))];
                    if ((e & 0xF0u) == 0) { *o++ = (uint8_t)(e >> 16); b.bits >>= (e & 15u); b.nbits -= (int)(e & 15u); continue; }
                }
                b.bits >>= (e & 15u); b.nbits -= (int)(e & 15u);
                uint32_t kind = E_KIND(e);
                if (kind != K_LEN) {
                    if (kind == K_EOB) goto block_done;
                    return -1;
                }
                uint32_t len = E_VAL(e) + (uint32_t)(b.bits & ((1u << E_EXTRA(e)) - 1u));
                b.bits >>= E_EXTRA(e); b.nbits -= (int)E_EXTRA(e);
                uint32_t d = d_tab[b.bits & ((1u << D_BITS) - 1u)];
                if (E_KIND(d) == K_SUB) {
                    uint32_t sb = E_EXTRA(d), off = E_VAL(d);
                    b.bits >>= D_BITS; b.nbits -= D_BITS;
                    d = d_tab[off + (uint32_t)(b.bits & ((1u << sb) - 1u))];
                }
                if (E_KIND(d) != K_DIST) return -1;
                b.bits >>= (d & 15u); b.nbits -= (int)(d & 15u);
                uint32_t dist = E_VAL(d) + (uint32_t)(b.bits & ((1u << E_EXTRA(d)) - 1u));
                b.bits >>= E_EXTRA(d); b.nbits -= (int)E_EXTRA(d);
                if (dist > (size_t)(o - out)) return -1;
                const uint8_t *src = o - dist;
                uint8_t *dst = o;
                o += len;
                if (dist >= 8) {
                    do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < o);
                } else if (dist == 1) {
                    memset(dst, *src, len);
                } else {
                    do { *dst++ = *src++; } while (dst < o);
                }
            }
            /* careful path: one symbol, every access checked */
            refill(&b);
            uint32_t e = ll_tab[peek(&b, LL_BITS)];
            if (E_KIND(e) == K_SUB) {
                uint32_t sb = E_EXTRA(e), off = E_VAL(e);
                drop(&b, LL_BITS);
                e = ll_tab[off + peek(&b, (int)sb)];
            }
            if (E_KIND(e) == K_BAD) return -1;
            drop(&b, (int)E_LEN(e));
            if (b.nbits < 0) return -1;
            uint32_t kind = E_KIND(e);
            if (kind == K_LIT) {
                if (o >= o_end) return -1;
                *o++ = (uint8_t)E_VAL(e);
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_LEN) return -1;
            uint32_t len = E_VAL(e) + peek(&b, (int)E_EXTRA(e));
            drop(&b, (int)E_EXTRA(e));
            refill(&b);
            uint32_t d = d_tab[peek(&b, D_BITS)];
            if (E_KIND(d) == K_SUB) {
                uint32_t sb = E_EXTRA(d), off = E_VAL(d);
                drop(&b, D_BITS);
                d = d_tab[off + peek(&b, (int)sb)];
            }
            if (E_KIND(d) != K_DIST) return -1;
            drop(&b, (int)E_LEN(d));
            uint32_t dist = E_VAL(d) + peek(&b, (int)E_EXTRA(d));
            drop(&b, (int)E_EXTRA(d));
            if (b.nbits < 0) return -1;
            if (dist > (size_t)(o - out) || len > (size_t)(o_end - o)) return -1;
            const uint8_t *src = o - dist;
            while (len--) *o++ = *src++;
        }
block_done:;
    }
    return o == o_end ? 0 : -1;
}
