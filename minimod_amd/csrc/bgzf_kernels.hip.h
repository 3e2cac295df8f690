// bgzf_kernels.hip.h -- BGZF blocks inflated on the device (parallel ingestion: the reference leaves
// this to htslib's thread pool, src/minimod.c:249 sam_read1; on a node with 16-24 host cores per MI355X the inflate is the
// end-to-end limit).
//
// A BGZF block (RFC 1952 member with a BC subfield, <= 64 KB decoded) is an independent unit and a wavefront's work:
//   * the bit reader is SCALAR: the block's compressed bytes stand in two vector registers as a 512-byte window (lane l holds
//     dword l), and the next 64 bits at any bit position are two v_readlane away -- no memory trip per symbol;
//   * the Huffman tables (RFC 1951) are built by the wavefront in its slice of LDS: codes numbered with ballots (a symbol's code
//     is its length's first code plus the number of symbols of that length in front of it), a first-level table of 10 bits for
//     literals / lengths and 8 for distances filled by the lanes, longer codes decoded canonically (first code and count per
//     length: the rare path);
//   * tokens are decoded in BURSTS: every lane decodes the token (literal, or length + distance with their extra bits) that would
//     start at its own bit offset behind the reader's position, and the scalar unit only walks the chain from token to token;
//     what a burst cannot take (end of block, long codes) goes through the plain path, one LDS lookup a symbol;
//   * the block's latest 2 KB of output stand in an LDS ring: a literal is a byte store there, a match is copied by all lanes at
//     once, out[o + i] = out[o - dist + i mod dist], from the ring when its source is that near (else from global memory, behind
//     a release fence if those bytes were flushed since the last one); the ring's older half goes to global memory in whole
//     dwords every 1 KB;
//   * the CRC32 of the decoded bytes (the gzip trailer covers them) is a kernel of its own: 64 slices a block, a table-driven
//     CRC per lane, the slices' registers combined by multiplying with x^(8 * bytes behind the slice) modulo the polynomial.
// Measured: 24 GB/s of decoded bytes per MI355X on a C2-shape BAM in a launch of 7 000 blocks (the scalar
// symbol loop alone: 9.6; literals in bursts 11.0; whole tokens in bursts 16.7; smaller tables and ring for 16 wavefronts per CU
// 24.2).  Issue-bound by then: about a hundred instructions a token.
// Anything the decoder does not like (a malformed stream, a size or CRC mismatch) is a status word per block: the host's own
// decoder (csrc/host/inflate_fast.c, then zlib) has the last word on such a block, so error behaviour stays what it was.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmbgzf {

constexpr int kLL = 10, kD = 8;                 // first-level table bits.  The wavefront's LDS is 8.1 KB: FOUR workgroups a CU (32.3 KB each), on purpose --
                                                // five (7.6 KB a wavefront with kD = 7) decoded 5 % faster alone, but a workgroup holds its CU's LDS and registers for
                                                // ~9 ms, and with five of them nothing else fits: the record framing behind the inflate and k_stream_reads (22.5 KB of
                                                // LDS, 72 registers) waited for workgroups to retire
constexpr int kWaves = 4;                       // wavefronts per workgroup
constexpr size_t kPad = 1024;                   // readable bytes behind a launch's compressed bytes
enum { S_OK = 0, S_BAD_BLOCK_TYPE = 1, S_BAD_STORED = 2, S_BAD_CODE_LENGTHS = 3, S_BAD_SYMBOL = 4, S_BAD_DISTANCE = 5, S_OVERRUN_OUT = 6,
       S_OVERRUN_IN = 7, S_SIZE = 8, S_CRC = 9 };

// table entry: bits 0-3 code length (0: not a code of this level: the canonical path decides), 4-7 kind, 8-12 extra bits, 16-31 value
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_DIST = 4 };
__host__ __device__ constexpr uint32_t ent(uint32_t len, uint32_t kind, uint32_t extra, uint32_t val) { return len | (kind << 4) | (extra << 8) | (val << 16); }
// Round 5: the tables in LDS hold COMPACT entries, 16 bits -- code length 0-3, kind 4-5 (0 literal, 1 length, 2 end of block, 3 distance),
// the literal's byte or the length / distance SYMBOL from bit 6 -- and a symbol's base value and extra-bit count are computed where the
// entry is read (RFC 1951 3.2.5's tables are arithmetic: length symbol s >= 8 has (s - 4) >> 2 extra bits and base 3 + ((4 + (s & 3)) << extra),
// distance symbol d >= 4 has (d - 2) >> 1 and 1 + ((2 + (d & 1)) << extra)).  A wavefront's tables are 2.5 KB instead of 5: its LDS 5.6 KB
// instead of 8.1, and what bounds the kernel is how many wavefronts' chains a SIMD has to interleave (a block lasts ~8 ms whatever shares its
// SIMD): five or six workgroups a CU instead of four.

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#ifdef MM_INFLATE_U32   // (A/B: round 4's 32-bit entries)
typedef uint32_t tab_t;
constexpr int kSymShift = 16;
__host__ __device__ inline uint32_t cent(uint32_t len, uint32_t kind2, uint32_t val) {
    return kind2 == 0u ? ent(len, K_LIT, 0, val) : (kind2 == 2u ? ent(len, K_EOB, 0, 0) : (kind2 == 1u ? ent(len, K_LEN, c_len_extra[val], c_len_base[val]) : ent(len, K_DIST, c_dist_extra[val], c_dist_base[val])));
}
__device__ __forceinline__ uint32_t expand_ll(uint32_t c) { return c; }
__device__ __forceinline__ uint32_t expand_d(uint32_t c) { return c; }
#else
typedef uint16_t tab_t;
constexpr int kSymShift = 6;
__host__ __device__ constexpr uint32_t cent(uint32_t len, uint32_t kind2, uint32_t val) { return len | (kind2 << 4) | (val << 6); }
__device__ __forceinline__ uint32_t expand_ll(uint32_t c) {   // compact literal / length entry -> the 32-bit form the decoder works with
    const uint32_t len = c & 15u, k = (c >> 4) & 3u, v = c >> 6;
    const uint32_t xl = v < 8u ? 0u : (v >= 28u ? 0u : (v - 4u) >> 2);
    const uint32_t base = v < 8u ? 3u + v : (v >= 28u ? 258u : 3u + ((4u + (v & 3u)) << xl));
    const uint32_t lit = len | ((uint32_t)K_LIT << 4) | (v << 16);
    const uint32_t mat = len | ((uint32_t)K_LEN << 4) | (xl << 8) | (base << 16);
    const uint32_t eob = len | ((uint32_t)K_EOB << 4);
    return k == 0u ? lit : (k == 1u ? mat : eob);
}
__device__ __forceinline__ uint32_t expand_d(uint32_t c) {
    const uint32_t len = c & 15u, v = c >> 6;
    const uint32_t xd = v < 4u ? 0u : (v - 2u) >> 1;
    const uint32_t base = v < 4u ? 1u + v : 1u + ((2u + (v & 1u)) << xd);
    return ((c >> 4) & 3u) == 3u ? (len | ((uint32_t)K_DIST << 4) | (xd << 8) | (base << 16)) : 0u;
}
#endif

// Round 5's loop looks a length / distance symbol's base and extra-bit count up in a table of the workgroup's (64 words: 0-28 the length
// symbols, 32-61 the distance symbols, in the 32-bit entries' form without the code length) instead of computing them: the kernel is
// bound by the instructions it issues, and the arithmetic above is twenty-five of them a window
__device__ __forceinline__ void fill_xtab(uint32_t* xtab) {
    const uint32_t t = threadIdx.x;
    if (t < 64u) {
        uint32_t e = 0;
        if (t < 29u) e = ent(0, K_LEN, c_len_extra[t], c_len_base[t]);
        else if (t == 31u) e = ent(0, K_EOB, 0, 0);            // (expand_ll_t: kind 2 looks here)
        else if (t >= 32u && t < 62u) e = ent(0, K_DIST, c_dist_extra[t - 32u], c_dist_base[t - 32u]);
        xtab[t] = e;
    }
}
__device__ __forceinline__ uint32_t expand_ll_t(uint32_t c, const uint32_t* xtab) {
    const uint32_t len = c & 15u, k = (c >> 4) & 3u, v = c >> 6;
    const uint32_t x = xtab[(v | (31u & (0u - (k >> 1)))) & 31u];   // (a length symbol's entry; the end of the block's is entry 31; read whatever the entry is)
    return len | (k == 0u ? v << 16 : x);
}
__device__ __forceinline__ uint32_t expand_d_t(uint32_t c, const uint32_t* xtab) {
    const uint32_t x = xtab[32u + ((c >> 6) & 31u)];
    return ((c >> 4) & 3u) == 3u ? ((c & 15u) | x) : 0u;
}

struct Block {            // one BGZF block of a launch
    uint32_t c_off;       // its deflate payload in the launch's compressed bytes
    uint32_t c_len;
    uint32_t o_off;       // where its decoded bytes go
    uint32_t isize;       // how many they are (ISIZE of the trailer)
    uint32_t crc;         // CRC32 of the trailer
};

struct CodeLds {          // one Huffman code: the canonical description for codes longer than its first-level table
    uint16_t cnt[16];     // codes per length
    uint16_t sorted[320]; // symbols by (length, symbol)
};
struct CodeLdsD {         // the same for the distance code (30 symbols)
    uint16_t cnt[16];
    uint16_t sorted[32];
};
#ifndef MM_RING
#define MM_RING 2048   // bytes of a block's latest output kept in LDS (a power of two; build-time experiments: 4096)
#endif
struct WaveLds {
    tab_t ll[1 << kLL];   // compact entries (its first 128 are the code-length code's table while a block's code lengths are read)
    tab_t dt[1 << kD];
    CodeLds cl_ll;
    CodeLdsD cl_d;
    uint8_t lens[320];    // litlen lengths, then distance lengths
    uint8_t ring[MM_RING];   // the block's latest output (kRing)
    uint32_t win[128 + 4];   // round 5: the 512 compressed bytes around the reader's position (what Bits::va / vb hold), for the lanes' unaligned reads; the ring's over-read pad in front of it

};

typedef uint32_t u32_unal __attribute__((aligned(1)));
typedef uint16_t u16_unal __attribute__((aligned(1)));
__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ void lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }
// inclusive prefix sum over the 64 lanes: DPP row shifts inside each row of 16, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 (no LDS trip)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    // (v_add_u32 with the DPP modifier on its first source: one instruction a step -- the builtin makes a v_mov_dpp and an add of it; a lane
    // whose source does not exist adds 0, bound_ctrl:0; a VGPR written by the VALU can be read through DPP two wait states later)
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}

// ---- the bit reader: wave-uniform state, the bytes in two registers
struct Bits {
    const uint8_t* in;    // the block's payload
    uint32_t c_len;
    uint32_t wpos;        // byte offset of the window's first byte (a multiple of 256)
    uint32_t va, vb;      // the window: bytes [wpos, wpos + 256) and [wpos + 256, wpos + 512), dword `lane` of each
    uint64_t bitpos;      // next bit of the stream

    // dword `lane` of the 256 bytes at `at` (zeros behind the payload).  No branches: a lane-dependent branch in here is enough
    // for the compiler to call the reader's whole state divergent (see settle()).  Reads up to 516 bytes behind the payload:
    // the launch's compressed bytes have kPad bytes behind them.
    __device__ __forceinline__ uint32_t load_win(uint32_t at) const {
        const uint32_t o = at + 4u * (uint32_t)lane();
        uint32_t w;
        __builtin_memcpy(&w, in + o, 4);
        const int rem = (int)c_len - (int)o;
        const uint32_t mask = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        return w & mask;
    }
    __device__ __forceinline__ void start(const uint8_t* p, uint32_t n) {
        in = p; c_len = n; wpos = 0; bitpos = 0;
        va = load_win(0); vb = load_win(256);
    }
    __device__ __forceinline__ uint32_t dword_at(uint32_t wi) const {   // wi < 128, uniform
        return wi < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)va, (int)wi) : (uint32_t)__builtin_amdgcn_readlane((int)vb, (int)(wi - 64u));
    }
    __device__ __forceinline__ uint32_t dword_sel(uint32_t wi) const {  // the same without a branch: both halves read, one kept
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)va, (int)(wi & 63u)), y = (uint32_t)__builtin_amdgcn_readlane((int)vb, (int)(wi & 63u));
        return wi < 64u ? x : y;
    }
    // the window follows the position: its byte lies in the window's first 256
    __device__ __forceinline__ void place() {
        const uint32_t byte = (uint32_t)(bitpos >> 3);
        if (byte - wpos >= 512u) { wpos = byte & ~255u; va = load_win(wpos); vb = load_win(wpos + 256u); }   // (behind a stored block)
        else if (byte - wpos >= 256u) { va = vb; wpos += 256u; vb = load_win(wpos + 256u); }
    }
    // the next 32 bits (the stream's later bits in the higher positions)
    __device__ __forceinline__ uint32_t peek32() {
        place();
        const uint32_t byte = (uint32_t)(bitpos >> 3);
        const uint32_t rel = byte - wpos, wi = rel >> 2;
        const uint64_t w = (uint64_t)dword_at(wi) | ((uint64_t)dword_at(wi + 1u) << 32);
        return (uint32_t)(w >> ((rel & 3u) * 8u + (uint32_t)(bitpos & 7u)));
    }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = peek32() & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1u)); bitpos += (uint64_t)n; return v; }
    __device__ __forceinline__ bool overrun() const { return bitpos > 8ull * (uint64_t)c_len; }
    // The reader's position IS wave-uniform, but the compiler stops believing it at the first loop that lanes leave one by one
    // (the window's byte-wise tail, the table fills): from there on it keeps the decoder's state in vector registers and turns
    // every `if` into exec masking.  Passing the state through readfirstlane at the top of every loop says what it is.
    __device__ __forceinline__ void settle() {
        wpos = uni(wpos);
        bitpos = (uint64_t)uni((uint32_t)bitpos) | ((uint64_t)uni((uint32_t)(bitpos >> 32)) << 32);
    }
};

// ---- one Huffman code from its lengths lens[0..n) (in LDS): table `tab` of `root` bits, canonical description `cd`.
// Returns false for a code zlib refuses too (over-subscribed, or incomplete with more than one code).
template <typename CD>
__device__ __forceinline__ bool build_code(const uint8_t* lens, int n, bool is_dist, int root, tab_t* tab, CD& cd) {
    const int l = lane();
    // codes per length, the first code of every length, each symbol's place among the symbols of its length
    uint32_t cnt[16];
#pragma unroll
    for (int k = 0; k < 16; k++) cnt[k] = 0;
    uint32_t my_rank[5], my_len[5];
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int s = 64 * r + l;
        const uint32_t ln = s < n ? lens[s] : 0u;
        my_len[r] = ln; my_rank[r] = 0;
#pragma unroll
        for (int k = 1; k < 16; k++) {
            const uint64_t m = __ballot(ln == (uint32_t)k);
            if (ln == (uint32_t)k) my_rank[r] = cnt[k] + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
            cnt[k] += (uint32_t)__popcll(m);
        }
    }
    int left = 1, max_len = 0;
    uint32_t first[16], offs[16];
    uint32_t code = 0, at = 0;
    first[0] = 0; offs[0] = 0;
#pragma unroll
    for (int k = 1; k < 16; k++) {
        left = (left << 1) - (int)cnt[k];
        if (cnt[k]) max_len = k;
        code = (code + cnt[k - 1]) << 1; first[k] = code;
        offs[k] = at; at += cnt[k];
    }
    if (left < 0) return false;
    if (left > 0 && max_len > 1) return false;
    {
        uint32_t mine = 0;
#pragma unroll
        for (int k = 1; k < 16; k++) if (l == k) mine = cnt[k];
        if (l < 16) cd.cnt[l] = (uint16_t)mine;
    }
    const uint32_t root_size = 1u << root;
    for (uint32_t i = (uint32_t)l; i < root_size; i += 64u) tab[i] = (tab_t)0;
    lds_sync();
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int s = 64 * r + l;
        const uint32_t ln = my_len[r];
        if (ln == 0u) continue;
        uint32_t fk = 0, ok = 0;
#pragma unroll
        for (int k = 1; k < 16; k++) if (ln == (uint32_t)k) { fk = first[k]; ok = offs[k]; }
        cd.sorted[ok + my_rank[r]] = (uint16_t)s;
        if (ln <= (uint32_t)root) {
            uint32_t e;
            if (is_dist) e = s < 30 ? cent(ln, 3u, (uint32_t)s) : 0u;
            else if (s < 256) e = cent(ln, 0u, (uint32_t)s);
            else if (s == 256) e = cent(ln, 2u, 0u);
            else e = s <= 285 ? cent(ln, 1u, (uint32_t)(s - 257)) : 0u;
            const uint32_t c = rev_bits(fk + my_rank[r], (int)ln);
            for (uint32_t i = c; i < root_size; i += 1u << ln) tab[i] = (tab_t)e;
        }
    }
    lds_sync();
    return true;
}

// a symbol of a code whose first-level entry says "longer than the table": canonical decoding, one bit at a time (the stream's
// bits are a code's most significant first).  Returns the entry (length 0: no such code).
template <typename CD>
__device__ __forceinline__ uint32_t decode_long(Bits& b, const CD& cd, bool is_dist) {
    const uint32_t bits = b.peek32();
    uint32_t code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (bits >> (len - 1)) & 1u;
        const uint32_t count = uni(cd.cnt[len]);
        if (code < first + count) {
            // (uni: the tables' loads are vector loads to the compiler; one divergent value here and the whole decoder's state
            // moves to vector registers and exec masks)
            const uint32_t s = uni(cd.sorted[index + (code - first)]);
            if (is_dist) return s < 30u ? uni(ent((uint32_t)len, K_DIST, c_dist_extra[s], c_dist_base[s])) : 0u;
            if (s < 256u) return ent((uint32_t)len, K_LIT, 0, s);
            if (s == 256u) return ent((uint32_t)len, K_EOB, 0, 0);
            return s <= 285u ? uni(ent((uint32_t)len, K_LEN, c_len_extra[s - 257u], c_len_base[s - 257u])) : 0u;
        }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return 0u;
}

// ---- the tables of one fixed / dynamic DEFLATE block.  A function of its own, not inlined: the code numbering keeps a hundred
// scalar values alive (counts, first codes, lane masks), and inlined they are spilled and reloaded in every round of the symbol
// loop.  Its arguments and results travel in vector registers (the reader's state comes back through readfirstlane).
struct TablesRet { uint32_t wpos, va, vb, bit_lo, bit_hi; int st; };
__device__ __noinline__ TablesRet read_tables(const uint8_t* in_, uint32_t c_len_, uint32_t wpos_, uint32_t va_, uint32_t vb_, uint32_t bit_lo, uint32_t bit_hi,
                                              uint32_t type_, WaveLds* Sp) {
    WaveLds& S = *Sp;
    const int l = lane();
    Bits b;
    b.in = (const uint8_t*)(((uint64_t)uni((uint32_t)((uint64_t)in_ >> 32)) << 32) | (uint64_t)uni((uint32_t)(uint64_t)in_));
    b.c_len = uni(c_len_); b.wpos = uni(wpos_); b.va = va_; b.vb = vb_;
    b.bitpos = (uint64_t)uni(bit_lo) | ((uint64_t)uni(bit_hi) << 32);
    const uint32_t type = uni(type_);
    TablesRet r;
    r.st = S_OK;
    int n_ll = 288, n_d = 32;
    if (type == 1u) {   // fixed codes
        for (int i = l; i < 288; i += 64) S.lens[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
        if (l < 32) S.lens[288 + l] = 5;
        lds_sync();
    } else {            // dynamic codes
        const int hlit = (int)b.take(5) + 257, hdist = (int)b.take(5) + 1, hclen = (int)b.take(4) + 4;
        if (b.overrun() || hlit > 286 || hdist > 30) r.st = S_BAD_CODE_LENGTHS;
        if (r.st == S_OK) {
            // the code-length code: 19 lengths of 3 bits, one level of 7 bits
            uint32_t cl_len = 0;   // lane s < 19 holds the length of code-length symbol s
            for (int i = 0; i < hclen; i++) { b.settle(); const uint32_t v = b.take(3); if (l == (int)c_clen_order[i]) cl_len = v; }
            if (b.overrun()) r.st = S_OVERRUN_IN;
            uint32_t cnt[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cnt[k] = 0;
            uint32_t rank = 0;
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const uint64_t m = __ballot(l < 19 && cl_len == (uint32_t)k);
                if (l < 19 && cl_len == (uint32_t)k) rank = cnt[k] + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
                cnt[k] = (uint32_t)__popcll(m);
            }
            int left = 1;
            uint32_t first[8], code = 0;
            first[0] = 0;
#pragma unroll
            for (int k = 1; k < 8; k++) { left = (left << 1) - (int)cnt[k]; code = (code + cnt[k - 1]) << 1; first[k] = code; }
            if (left != 0 && r.st == S_OK) r.st = S_BAD_CODE_LENGTHS;   // the code-length code must be complete
            S.ll[l] = (tab_t)0; S.ll[64 + l] = (tab_t)0;
            lds_sync();
            if (l < 19 && cl_len) {
                uint32_t fk = 0;
#pragma unroll
                for (int k = 1; k < 8; k++) if (cl_len == (uint32_t)k) fk = first[k];
                const uint32_t c = rev_bits(fk + rank, (int)cl_len);
                for (uint32_t i = c; i < 128u; i += 1u << cl_len) S.ll[i] = (tab_t)(kSymShift == 16 ? ent(cl_len, 0, 0, (uint32_t)l) : cent(cl_len, 0u, (uint32_t)l));
            }
            lds_sync();
        }
        // the lengths themselves (serial: every symbol depends on the bits in front of it)
        const int total = hlit + hdist;
        int n = 0;
        uint32_t prev = 0;
        uint32_t guard = 0;
        while (r.st == S_OK && n < total) {
            b.settle(); n = (int)uni((uint32_t)n); prev = uni(prev); guard = uni(guard);
            if (++guard > 400u) { r.st = S_BAD_CODE_LENGTHS; break; }
            if (b.bitpos > 8ull * (uint64_t)b.c_len + 64ull) { r.st = S_OVERRUN_IN; break; }   // (a cut-off stream reads as zeros behind its end: never further than this)
            const uint32_t bits = b.peek32();
            const uint32_t e = uni((uint32_t)S.ll[bits & 127u]);
            const uint32_t el = e & 15u;
            if (!el) { r.st = S_BAD_CODE_LENGTHS; break; }
            b.bitpos += el;
            const uint32_t s = e >> kSymShift;
            if (s < 16u) { if (l == 0) S.lens[n] = (uint8_t)s; prev = s; n++; continue; }
            uint32_t rep, val = 0;
            if (s == 16u) { if (n == 0) { r.st = S_BAD_CODE_LENGTHS; break; } val = prev; rep = 3u + b.take(2); }
            else if (s == 17u) { rep = 3u + b.take(3); }
            else { rep = 11u + b.take(7); }
            if (n + (int)rep > total) { r.st = S_BAD_CODE_LENGTHS; break; }
            for (uint32_t i = (uint32_t)l; i < rep; i += 64u) S.lens[n + (int)i] = (uint8_t)val;
            n = (int)uni((uint32_t)n + rep); prev = uni(val);
            b.settle();
        }
        if (r.st == S_OK && b.overrun()) r.st = S_OVERRUN_IN;
        lds_sync();
        if (r.st == S_OK && uni(S.lens[256]) == 0u) r.st = S_BAD_CODE_LENGTHS;   // no end-of-block code
        n_ll = hlit; n_d = hdist;
    }
    if (r.st == S_OK && !build_code(S.lens, n_ll, false, kLL, S.ll, S.cl_ll)) r.st = S_BAD_CODE_LENGTHS;
    if (r.st == S_OK && !build_code(S.lens + n_ll, n_d, true, kD, S.dt, S.cl_d)) r.st = S_BAD_CODE_LENGTHS;
    r.wpos = b.wpos; r.va = b.va; r.vb = b.vb; r.bit_lo = (uint32_t)b.bitpos; r.bit_hi = (uint32_t)(b.bitpos >> 32);
    return r;
}

// ---- the output: the last kRing bytes of a block's output stand in LDS.  Literals and matches are written there; a match whose
// source lies that near (nearly all of them) is an LDS copy; every kFlush bytes the ring's older half goes to global memory in
// whole dwords.  Only a match that reaches further back reads global memory -- bytes flushed long before -- after a release
// fence if they were flushed since the last one.
constexpr uint32_t kRing = MM_RING, kFlush = 1024, kNear = kRing - 320;   // a match is at most 258 bytes: positions >= o - kNear are in the ring
__device__ __forceinline__ void ring_flush(uint8_t* out, const uint8_t* ring, uint32_t from, uint32_t upto) {   // from: a multiple of 4
    for (uint32_t i = from + 4u * (uint32_t)lane(); i < upto; i += 256u) {
        uint32_t w;
        __builtin_memcpy(&w, ring + (i & (kRing - 1u)), 4);
        if (i + 4u <= upto) __builtin_memcpy(out + i, &w, 4);
        else for (uint32_t k = 0; i + k < upto; k++) out[i + k] = (uint8_t)(w >> (8u * k));
    }
}

// four bytes at any address of the block's flushed output, from L2 (sc1: a line of the vector cache may be older than the flush's stores)
__device__ __forceinline__ uint32_t load_far(const uint8_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ---- one block
__device__ __forceinline__ int inflate_block(const uint8_t* in, uint32_t c_len, uint8_t* out, uint32_t isize, WaveLds& S, const uint32_t* xtab) {
    const int l = lane();
    Bits b;
    b.start(in, c_len);
    uint32_t o = 0;               // bytes produced
    uint32_t f = 0;               // bytes flushed to global memory (a multiple of kFlush until the end)
    uint32_t fenced = 0;          // flushed bytes known to have reached L2
    uint32_t guard = 0;           // rounds + symbols so far: a stream of c_len bytes has fewer than 8 c_len + 3 symbols (a bound, in case)
    const uint32_t guard_max = 16u * c_len + 4096u;   // (rounds and burst literals are both counted)
    int st = S_OK;
    for (;;) {
        b.settle(); o = uni(o); f = uni(f); fenced = uni(fenced); guard = uni(guard);
        if (++guard > guard_max) return S_OVERRUN_IN;
        const uint32_t final_block = b.take(1);
        const uint32_t type = b.take(2);
        if (b.overrun()) return S_OVERRUN_IN;
        if (type == 3u) return S_BAD_BLOCK_TYPE;
        if (type == 0u) {
            // stored: skip to the byte boundary, LEN, NLEN, the bytes (straight to global memory; the last of them into the ring too)
            b.bitpos = (b.bitpos + 7ull) & ~7ull;
            const uint32_t len = b.take(16), nlen = b.take(16);
            if (b.overrun() || (len ^ nlen) != 0xFFFFu) return S_BAD_STORED;
            const uint32_t src = (uint32_t)(b.bitpos >> 3);
            if (src + len > c_len) return S_OVERRUN_IN;
            if (o + len > isize) return S_OVERRUN_OUT;
            ring_flush(out, S.ring, f, o);
            for (uint32_t i = (uint32_t)l; i < len; i += 64u) {
                const uint8_t v = in[src + i];
                out[o + i] = v;
                if (i + kRing >= len) S.ring[(o + i) & (kRing - 1u)] = v;
            }
            o += len;
            f = o & ~3u;   // (a flush starts at a whole dword of the block's output: the up to three bytes behind f are in the ring too)
            b.bitpos += 8ull * (uint64_t)len;
            lds_sync();
        } else {
            const TablesRet t = read_tables(in, c_len, b.wpos, b.va, b.vb, (uint32_t)b.bitpos, (uint32_t)(b.bitpos >> 32), type, &S);
            b.wpos = uni(t.wpos); b.va = t.va; b.vb = t.vb;
            b.bitpos = (uint64_t)uni(t.bit_lo) | ((uint64_t)uni(t.bit_hi) << 32);
            st = (int)uni((uint32_t)t.st);
            if (st != S_OK) return st;
            // a match's bytes: out[at + i] = out[at - dist + i mod dist], all lanes at once, from the ring when the source is that near
            auto copy_match = [&](uint32_t at, uint32_t len, uint32_t dist) {
                const uint32_t s0 = at - dist;
                if (s0 + kNear < at) {
                    // a far source: flushed bytes, in global memory -- behind a fence if they were flushed since the last one
                    // (agent-scope loads: from L2, where the stores are -- a line of the vector cache may be older than they)
                    const uint32_t src_end = s0 + (len < dist ? len : dist);
                    if (src_end > fenced) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); fenced = f; }
                }
                lds_sync();   // (the bytes in front of the match are other lanes' stores)
                for (uint32_t i = (uint32_t)l; i < len; i += 64u) {
                    const uint32_t p = s0 + (dist >= len ? i : i % dist);
                    uint8_t v;
                    if (p + kNear >= at) v = S.ring[p & (kRing - 1u)];
                    else v = __hip_atomic_load(out + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S.ring[(at + i) & (kRing - 1u)] = v;
                }
            };
            // ---- the block's symbols, a WINDOW of 64 bit offsets at a time (round 4; round 3 walked the chain of tokens one by one on the
            // scalar unit and copied every match, three bytes or three hundred, with all 64 lanes behind a wave barrier: ~90
            // instructions a token, and a BAM's packed sequence is literals in runs of two or three between matches of three or four
            // bytes -- 13 600 tokens a block).
            //   1. every lane decodes the TOKEN that would start at ITS bit offset behind the reader's position: a literal, or a length
            //      with its extra bits, its distance code and that code's extra bits (the window's five dwords are scalars; a lane's 64
            //      bits are two v_alignbit away; two table lookups for 64 candidate tokens);
            //   2. pointer doubling over "the token behind this one" (four rounds, three ds_bpermute each: sixteen tokens) gives every offset the set of
            //      offsets on ITS chain, where that chain leaves the window (or meets something the plain path must take) and how many
            //      bytes it produces -- so the chain from the reader's position is a readlane away, and every token on it knows where its
            //      output goes;
            //   3. the tokens of the chain write their bytes TOGETHER: a literal its byte, a match its (up to sixteen) bytes from the ring
            //      or, further back, from global memory -- as long as no match reads what this step writes (distance >= bytes in front of
            //      it in the step + its length).  A match that does (a run: distance < length), or a long one, ends the step and is
            //      copied the old way, all lanes on one match; the same tables serve the rest of the window.
#ifdef MM_INFLATE_V1
            bool eob = false;
            for (;;) {
                b.settle(); o = uni(o); f = uni(f); fenced = uni(fenced); guard = uni(guard);
                if (++guard > guard_max) return S_OVERRUN_IN;
                // a stream that ends in the middle of a symbol reads as zeros from there on, and a code of zeros may well be a literal
                // or a match: without this the loop would run on to ISIZE, up to 120 KB behind the payload (the window's loads are
                // real loads).  With it the reader stays within a window's bits (64 + a token) of the payload's end.
                if (b.bitpos > 8ull * (uint64_t)c_len + 64ull) return S_OVERRUN_IN;
                (void)b.peek32();                                  // places the window: the position's byte lies in its first 256
                const uint64_t base = b.bitpos;
                const uint32_t rel = (uint32_t)(base - 8ull * (uint64_t)b.wpos);
                const uint32_t i0 = rel >> 5, sh = rel & 31u;
                const uint32_t w0 = b.dword_sel(i0), w1 = b.dword_sel(i0 + 1u), w2 = b.dword_sel(i0 + 2u), w3 = b.dword_sel(i0 + 3u), w4 = b.dword_sel(i0 + 4u);
                const uint32_t sb = sh + (uint32_t)l, kq = sb >> 5, rq = sb & 31u;
                const uint32_t a0 = kq == 0u ? w0 : (kq == 1u ? w1 : w2), a1 = kq == 0u ? w1 : (kq == 1u ? w2 : w3), a2 = kq == 0u ? w2 : (kq == 1u ? w3 : w4);
                const uint64_t bits64 = (uint64_t)__builtin_amdgcn_alignbit(a1, a0, rq) | ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, rq) << 32);
                const uint32_t e1 = expand_ll((uint32_t)S.ll[(uint32_t)bits64 & ((1u << kLL) - 1u)]);
                const uint32_t l1 = e1 & 15u, k1 = (e1 >> 4) & 15u;
                uint32_t t_type = 2u, t_bits = 0u, t_val = 0u, t_dist = 0u, t_out = 0u;   // 0 literal (val = byte), 1 match (val = length), 2 the plain path's
                if (l1 != 0u && k1 == (uint32_t)K_LIT) { t_type = 0u; t_bits = l1; t_val = e1 >> 16; t_out = 1u; }
                else if (l1 != 0u && k1 == (uint32_t)K_LEN) {
                    const uint32_t xl = (e1 >> 8) & 31u;
                    const uint32_t mlen = (e1 >> 16) + ((uint32_t)(bits64 >> l1) & ((1u << xl) - 1u));
                    const uint32_t used = l1 + xl;
                    const uint32_t dbits = (uint32_t)(bits64 >> used);
                    const uint32_t d = expand_d((uint32_t)S.dt[dbits & ((1u << kD) - 1u)]);
                    const uint32_t dl = d & 15u;
                    if (dl != 0u && ((d >> 4) & 15u) == (uint32_t)K_DIST) {
                        const uint32_t xd = (d >> 8) & 31u;
                        t_dist = (d >> 16) + ((dbits >> dl) & ((1u << xd) - 1u));
                        t_type = 1u; t_bits = used + dl + xd; t_val = mlen; t_out = mlen;
                    }
                }
                const bool tok = t_type != 2u;
                // the chain behind every offset: next offset (>= 64: out of the window; itself: the plain path's) | bytes produced << 8; the offsets on it
                uint32_t nxs = (tok ? (uint32_t)l + t_bits : (uint32_t)l) | (t_out << 8);
                uint32_t r_lo = tok && l < 32 ? 1u << l : 0u, r_hi = tok && l >= 32 ? 1u << (l - 32) : 0u;
#pragma unroll
                for (int step = 0; step < 4; step++) {
                    const int tgt = (int)(nxs & 63u);
                    const uint32_t an = (uint32_t)__shfl((int)nxs, tgt), a_lo = (uint32_t)__shfl((int)r_lo, tgt), a_hi = (uint32_t)__shfl((int)r_hi, tgt);
                    if ((nxs & 255u) < 64u) { r_lo |= a_lo; r_hi |= a_hi; nxs = (an & 255u) | (((nxs >> 8) + (an >> 8)) << 8); }
                }
                const uint64_t tok_mask = __ballot(tok);
                uint32_t pos = 0;
                for (;;) {
                    o = uni(o); f = uni(f); fenced = uni(fenced); guard = uni(guard); pos = uni(pos);
                    if (++guard > guard_max) return S_OVERRUN_IN;
                    if (o - f >= kFlush + 256u) { lds_sync(); ring_flush(out, S.ring, f, f + kFlush); f += kFlush; }
                    if ((tok_mask >> pos) & 1ull) {
                        const uint32_t m_lo = (uint32_t)__builtin_amdgcn_readlane((int)r_lo, (int)pos), m_hi = (uint32_t)__builtin_amdgcn_readlane((int)r_hi, (int)pos);
                        const uint32_t pp = (uint32_t)__builtin_amdgcn_readlane((int)nxs, (int)pos);
                        const uint32_t stop = pp & 255u, s_tot = pp >> 8;
                        const bool on_chain = (l < 32 ? (m_lo >> l) : (m_hi >> (l - 32))) & 1u;
                        if (stop >= 64u || !((tok_mask >> stop) & 1ull)) {
                            // (the chain has been followed to its end: every token on it knows the bytes in front of it)
                            const uint32_t off = s_tot - (nxs >> 8);                       // bytes the chain produces in front of this token
                            const bool dep = on_chain && t_type == 1u && (t_dist < off + t_val || t_val > 16u);
                            const bool cut = on_chain && off + t_out > 256u;   // (a step writes at most 256 + 16 bytes: the ring's near sources stay whole)
                            const uint64_t db = __ballot(dep), sbm = db | __ballot(cut);
                            const uint32_t first = sbm ? (uint32_t)__builtin_ctzll(sbm) : 64u;
                            const bool first_dep = sbm && ((db >> (first & 63u)) & 1ull);
                            const bool in_step = on_chain && (uint32_t)l < first;
                            const uint32_t n_step = sbm ? (uint32_t)__builtin_amdgcn_readlane((int)off, (int)(first & 63u)) : s_tot;
                            if (o + n_step > isize) return S_OVERRUN_OUT;
                            const bool is_m = in_step && t_type == 1u;
                            const uint32_t at = o + off;
                            if (__ballot(is_m && t_dist > at)) return S_BAD_DISTANCE;
                            const bool far = is_m && t_dist > kNear;
                            // a far source: flushed bytes, in global memory -- behind a fence if they were flushed since the last one
                            if (__ballot(far && at - t_dist + t_val > fenced)) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); fenced = f; }
                            lds_sync();   // (the bytes in front of the step are other lanes' stores)
                            const uint32_t my_n = in_step ? t_out : 0u;
                            // four bytes a trip (a BAM's matches are mostly three or four bytes): one unaligned LDS read, one or two writes
                            for (uint32_t i = 0; __ballot(i < my_n); i += 4u) {
                                if (i < my_n) {
                                    const uint32_t n4 = min(4u, my_n - i);
                                    uint32_t v = t_val;
                                    if (t_type == 1u) {
                                        const uint32_t sp = at - t_dist + i;
                                        if (far) {
                                            uint32_t q0 = __hip_atomic_load(out + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q1 = 0, q2 = 0, q3 = 0;
                                            if (n4 > 1u) q1 = __hip_atomic_load(out + sp + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            if (n4 > 2u) q2 = __hip_atomic_load(out + sp + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            if (n4 > 3u) q3 = __hip_atomic_load(out + sp + 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            v = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
                                        } else {
                                            const uint32_t rs = sp & (kRing - 1u);
                                            if (rs <= kRing - 4u) v = *reinterpret_cast<const u32_unal*>(S.ring + rs);
                                            else v = (uint32_t)S.ring[rs] | ((uint32_t)S.ring[(rs + 1u) & (kRing - 1u)] << 8) | ((uint32_t)S.ring[(rs + 2u) & (kRing - 1u)] << 16) | ((uint32_t)S.ring[(rs + 3u) & (kRing - 1u)] << 24);
                                        }
                                    }
                                    const uint32_t rd = (at + i) & (kRing - 1u);
                                    if (rd <= kRing - 4u) {
                                        if (n4 == 4u) *reinterpret_cast<u32_unal*>(S.ring + rd) = v;
                                        else {
                                            if (n4 & 2u) { *reinterpret_cast<u16_unal*>(S.ring + rd) = (uint16_t)v; }
                                            if (n4 & 1u) S.ring[rd + (n4 & 2u)] = (uint8_t)(v >> (8u * (n4 & 2u)));
                                        }
                                    } else {
                                        for (uint32_t k = 0; k < n4; k++) S.ring[(rd + k) & (kRing - 1u)] = (uint8_t)(v >> (8u * k));
                                    }
                                }
                            }
                            o += n_step;
                            guard += (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                            if (!sbm) { pos = stop; }
                            else if (!first_dep) { pos = first; }   // (the step was cut short: the next one begins here)
                            else {
                                // the match that reads what the step wrote (or a long one): all lanes on it
                                const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)t_val, (int)first), md = (uint32_t)__builtin_amdgcn_readlane((int)t_dist, (int)first);
                                if (md > o) return S_BAD_DISTANCE;
                                if (o + mlen > isize) return S_OVERRUN_OUT;
                                if (o - f >= kFlush + 256u) { lds_sync(); ring_flush(out, S.ring, f, f + kFlush); f += kFlush; }
                                copy_match(o, mlen, md);
                                o = uni(o + mlen);
                                pos = first + (uint32_t)__builtin_amdgcn_readlane((int)t_bits, (int)first);
                            }
                        } else {
                            // more than sixteen tokens in what is left of the window (codes of three bits and less): one token, the old way
                            const uint32_t ty = (uint32_t)__builtin_amdgcn_readlane((int)t_type, (int)pos);
                            if (ty == 0u) {
                                if (o >= isize) return S_OVERRUN_OUT;
                                if ((uint32_t)l == pos) S.ring[o & (kRing - 1u)] = (uint8_t)t_val;
                                o++;
                            } else {
                                const uint32_t mlen = (uint32_t)__builtin_amdgcn_readlane((int)t_val, (int)pos), md = (uint32_t)__builtin_amdgcn_readlane((int)t_dist, (int)pos);
                                if (md > o) return S_BAD_DISTANCE;
                                if (o + mlen > isize) return S_OVERRUN_OUT;
                                copy_match(o, mlen, md);
                                o = uni(o + mlen);
                            }
                            pos += (uint32_t)__builtin_amdgcn_readlane((int)t_bits, (int)pos);
                        }
                        if (pos >= 64u) { b.bitpos = base + (uint64_t)pos; break; }
                        continue;
                    }
                    // one token of the plain path at base + pos: the end of the block, a code longer than the first-level tables
                    b.bitpos = base + (uint64_t)pos;
                    uint32_t bits = b.peek32();
                    uint32_t e = expand_ll(uni((uint32_t)S.ll[bits & ((1u << kLL) - 1u)]));
                    if ((e & 15u) == 0u) { e = decode_long(b, S.cl_ll, false); if ((e & 15u) == 0u) return S_BAD_SYMBOL; }
                    const uint32_t kind = (e >> 4) & 15u;
                    if (kind == K_LIT) {
                        b.bitpos += e & 15u;
                        if (o >= isize) return S_OVERRUN_OUT;
                        if (l == 0) S.ring[o & (kRing - 1u)] = (uint8_t)(e >> 16);
                        o++;
                    } else if (kind == K_EOB) { b.bitpos += e & 15u; eob = true; break; }
                    else {
                        // a match: length (extra bits behind the code), distance code, its extra bits
                        bits >>= e & 15u;
                        const uint32_t xl = (e >> 8) & 31u;
                        const uint32_t len = (e >> 16) + (bits & ((1u << xl) - 1u));
                        b.bitpos += (e & 15u) + xl;
                        bits = b.peek32();
                        uint32_t d = expand_d(uni((uint32_t)S.dt[bits & ((1u << kD) - 1u)]));
                        if ((d & 15u) == 0u) { d = decode_long(b, S.cl_d, true); if ((d & 15u) == 0u) return S_BAD_DISTANCE; }
                        bits >>= d & 15u;
                        const uint32_t xd = (d >> 8) & 31u;
                        const uint32_t dist = (d >> 16) + (bits & ((1u << xd) - 1u));
                        b.bitpos += (d & 15u) + xd;
                        if (dist > o) return S_BAD_DISTANCE;
                        if (o + len > isize) return S_OVERRUN_OUT;
                        copy_match(o, len, dist);
                        o = uni(o + len);
                    }
                    b.settle();
                    if (b.bitpos - base >= 64ull) break;          // the window is used up
                    pos = (uint32_t)(b.bitpos - base);
                }
                if (eob) break;
            }
#else
            // ---- the block's symbols, a WINDOW of 64 bit offsets at a time: round 5's loop (MM_INFLATE_V1 builds round 4's, above).
            // The kernel is bound by the NUMBER of instructions a wavefront issues (a SIMD issues one every four cycles whatever their kind:
            // ~600 a window in round 4's loop, measured), so this loop is written to be short:
            //   1. the compressed bytes around the reader's position stand in LDS as well (S.win, refilled with the registers' window every 256
            //      bytes): a lane's 64 bits at ITS bit offset are one unaligned ds_read_b64 and a shift (round 4: ten lane reads, five scalar
            //      and nine vector selects, two alignbits);
            //   2. every lane decodes the TOKEN that would start at its offset (two table lookups for 64 candidates), without branches;
            //   3. the chain of real tokens is WALKED by the scalar unit -- position -> readlane(next position), five scalar instructions a
            //      token in a hand-written loop, no LDS trip (round 4 doubled pointers: four dependent rounds of three ds_bpermute);
            //   4. where a token's bytes go is a prefix sum over the chain's lanes (DPP, no LDS);
            //   5. phase A: every token whose source lies in front of the window's output -- literals, matches of up to 16 bytes that reach back
            //      past everything the window writes -- writes at once: whole dwords in a loop, the last one to three bytes (a literal is that case)
            //      behind it; a source further back than the ring holds is ONE unaligned dword load from global memory;
            //   6. phase B: the others (a match that reads what the window writes, a long one, one that straddles the ring's end) in chain order,
            //      all lanes on one match.  Round 4 ended a "step" at the first such match and ran the step's bookkeeping again for what was left.
            // One exit for errors (`st`): early returns from the nest made the compiler build a state machine of scalar moves around it.
            bool eob = false;
            uint32_t win_at = 0xFFFFFFFFu;                 // the window position S.win holds
            while (!eob && st == S_OK) {
                b.settle(); o = uni(o); f = uni(f); fenced = uni(fenced); win_at = uni(win_at);
                // a stream that ends in the middle of a symbol reads as zeros from there on: the reader stays within a window's bits of the payload's end
                if (b.bitpos > 8ull * (uint64_t)c_len + 64ull) { st = S_OVERRUN_IN; break; }
                if (o - f >= kFlush + 256u) { lds_sync(); ring_flush(out, S.ring, f, f + kFlush); f += kFlush; }
                b.place();                                         // the window: the position's byte lies in its first 256
                if (b.wpos != win_at) { S.win[l] = b.va; S.win[64 + l] = b.vb; win_at = b.wpos; lds_sync(); }
                const uint64_t base = b.bitpos;
                const uint32_t rel = (uint32_t)(base - 8ull * (uint64_t)b.wpos) + (uint32_t)l;   // this lane's bit offset in the window (< 2048 + 64)
                uint64_t raw;
                __builtin_memcpy(&raw, reinterpret_cast<const uint8_t*>(S.win) + (rel >> 3), 8);
                const uint64_t bits64 = raw >> (rel & 7u);         // 57 bits and more: a token is at most 10 + 5 + 8 + 13
#ifdef MM_INFLATE_U32
                const uint32_t e1 = (uint32_t)S.ll[(uint32_t)bits64 & ((1u << kLL) - 1u)];
#else
                const uint32_t e1 = expand_ll_t((uint32_t)S.ll[(uint32_t)bits64 & ((1u << kLL) - 1u)], xtab);
#endif
                const uint32_t l1 = e1 & 15u, k1 = (e1 >> 4) & 15u;
                const uint32_t xl = (e1 >> 8) & 31u;
                const uint32_t mlen = (e1 >> 16) + ((uint32_t)(bits64 >> l1) & ((1u << xl) - 1u));
                const uint32_t used = l1 + xl;
                const uint32_t dbits = (uint32_t)(bits64 >> used);
#ifdef MM_INFLATE_U32
                const uint32_t d = (uint32_t)S.dt[dbits & ((1u << kD) - 1u)];
#else
                const uint32_t d = expand_d_t((uint32_t)S.dt[dbits & ((1u << kD) - 1u)], xtab);
#endif
                const uint32_t dl = d & 15u, xd = (d >> 8) & 31u;
                const bool is_lit = l1 != 0u && k1 == (uint32_t)K_LIT;
                const bool is_mat = l1 != 0u && k1 == (uint32_t)K_LEN && dl != 0u && ((d >> 4) & 15u) == (uint32_t)K_DIST;
                const uint32_t t_dist = (d >> 16) + ((dbits >> dl) & ((1u << xd) - 1u));
                const uint32_t t_val = is_lit ? e1 >> 16 : mlen;      // a literal's byte / a match's length
                const uint32_t t_out = is_lit ? 1u : (is_mat ? mlen : 0u);
                // the chain from the reader's position: lane p holds where the token at offset p ends (bit 7: no token there)
                const uint32_t tb = is_lit ? l1 : used + dl + xd;
                const uint32_t nx = (is_lit | is_mat) ? (uint32_t)l + tb : 0x80u;
                // The window is worked off in SEGMENTS: the chain of tokens from `pos` on, then -- where the chain ends on an offset that holds no
                // token -- that one symbol by the plain path (a long code, the end of the block), and on with the chain behind it.
                uint32_t pos = 0;
                bool next_window = false;
                while (!next_window) {
                pos = uni(pos); o = uni(o); f = uni(f); fenced = uni(fenced);
                uint32_t nxt, tmp;
                uint64_t chain;
                // the walk: A -> B = readlane(nx, A) -> A = readlane(nx, B) ...; five instructions a token (a lane select that a lane read wrote
                // wants four wait states).  Out: `chain` = the offsets visited that hold tokens, pos = the first offset not worked off
                // (>= 64: the window is used up; below: no token there)
                asm volatile(
                    "s_mov_b64 %[chain], 0\n"
                    "1:\n\t"
                    "v_readlane_b32 %[nxt], %[nx], %[pos]\n\t"
                    "s_bitset1_b64 %[chain], %[pos]\n\t"
                    "s_cmp_lt_u32 %[nxt], 64\n\t"
                    "s_cbranch_scc0 2f\n\t"
                    "s_nop 0\n\t"
                    "v_readlane_b32 %[pos], %[nx], %[nxt]\n\t"
                    "s_bitset1_b64 %[chain], %[nxt]\n\t"
                    "s_nop 0\n\t"
                    "s_cmp_lt_u32 %[pos], 64\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "s_mov_b32 %[tmp], %[pos]\n\t"
                    "s_mov_b32 %[pos], %[nxt]\n\t"
                    "s_mov_b32 %[nxt], %[tmp]\n"
                    "2:\n\t"                                    // pos = the last offset visited, nxt = what it holds
                    "s_bitcmp1_b32 %[nxt], 7\n\t"
                    "s_cbranch_scc0 3f\n\t"
                    "s_bitset0_b64 %[chain], %[pos]\n\t"       // no token there: not of the chain, and where the plain path starts
                    "s_mov_b32 %[nxt], %[pos]\n"
                    "3:\n\t"
                    "s_mov_b32 %[pos], %[nxt]"
                    : [pos] "+s"(pos), [chain] "=&s"(chain), [nxt] "=&s"(nxt), [tmp] "=&s"(tmp)
                    : [nx] "v"(nx)
                    : "scc");
                if (chain) {
                    const bool on_chain = (((l < 32 ? (uint32_t)chain : (uint32_t)(chain >> 32)) >> (l & 31)) & 1u) != 0u;
                    const uint32_t my_out = on_chain ? t_out : 0u;
                    const uint32_t incl = wave_incl_scan(my_out);
                    const uint32_t off = incl - my_out;                // bytes the chain produces in front of this token
                    // a window's parallel writes stay within 256 + 16 bytes of its start (the ring's near sources stay whole): what would go
                    // further begins the next window
                    uint32_t n_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    uint64_t cutm = 0;
                    bool mine = on_chain;
                    if (n_total > 256u) {                                 // (a window in twelve: it holds a long match)
                        cutm = __ballot(on_chain & (off != 0u) & (off + t_out > 256u));
                        if (cutm) {
                            const uint32_t first = (uint32_t)__builtin_ctzll(cutm);
                            mine = on_chain & ((uint32_t)l < first);
                            n_total = (uint32_t)__builtin_amdgcn_readlane((int)off, (int)first);
                            pos = first;
                        }
                    }
                    const bool is_m = mine & is_mat;
                    const uint32_t at = o + off;
                    if (o + n_total > isize) { st = S_OVERRUN_OUT; break; }
                    if (__ballot(is_m && t_dist > at)) { st = S_BAD_DISTANCE; break; }
                    const uint32_t src = at - t_dist;
                    // phase B's: reads what the window writes / long / source or destination across the ring's end (phase A moves whole dwords)
                    const bool far = t_dist > kNear;                   // a source further back than the ring holds: flushed bytes, in global memory
                    const bool defer = is_m & ((t_dist < off + mlen) | (mlen > 16u) | (far & (mlen > 4u)) | ((at & (kRing - 1u)) + mlen > kRing) | (!far & ((src & (kRing - 1u)) + mlen > kRing)));   // (| and &: no branches)
                    const bool use_far = is_m & !defer & far;
                    uint32_t fv = 0;
                    if (__ballot(use_far)) {
                        // (behind a fence if the bytes were flushed since the last one; ONE unaligned dword: nine matches in ten are of three or four bytes)
                        if (__ballot(use_far && src + mlen > fenced)) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); fenced = f; }
#ifndef MM_ABL_NOFAR
                        if (use_far) fv = load_far(out + src);
#endif
                    }
                    lds_sync();   // (the bytes in front of the window are other lanes' stores)
#ifdef MM_ABL_NOA   // (diagnostic builds: wrong bytes, right amount of everything else)
                    const uint32_t my_n = 0u;
#else
                    const uint32_t my_n = (mine && !defer) ? t_out : 0u;
#endif
                    uint8_t* const rdst = S.ring + (at & (kRing - 1u));
                    const uint8_t* const rsrc = S.ring + (src & (kRing - 1u));
                    // phase A: whole dwords ...
                    for (uint32_t i = 0; __ballot(i + 4u <= my_n); i += 4u) {
                        if (i + 4u <= my_n) {
                            const uint32_t rv = *reinterpret_cast<const u32_unal*>(rsrc + i);
                            *reinterpret_cast<u32_unal*>(rdst + i) = use_far ? fv : rv;
                        }
                    }
                    // ... then the last one to three bytes (a literal: its byte).  Every lane reads (its address is a ring address whatever it holds)
                    {
                        const uint32_t r = my_n & 3u, i = my_n & ~3u;
                        const uint32_t rv = *reinterpret_cast<const u32_unal*>(rsrc + i);
                        const uint32_t v = is_lit ? t_val : (use_far ? fv : rv);
                        if (r & 2u) *reinterpret_cast<u16_unal*>(rdst + i) = (uint16_t)v;
                        if (r & 1u) rdst[i + (r & 2u)] = (uint8_t)(v >> (8u * (r & 2u)));
                    }
                    // phase B: in chain order, all lanes on one match
                    uint64_t dm = __ballot(defer);
                    while (dm) {
                        const int kk = (int)__builtin_ctzll(dm);
                        dm &= dm - 1ull;
                        const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)mlen, kk), bd = (uint32_t)__builtin_amdgcn_readlane((int)t_dist, kk);
                        const uint32_t a = o + (uint32_t)__builtin_amdgcn_readlane((int)off, kk);
                        if (a - f >= kFlush + 256u) { lds_sync(); ring_flush(out, S.ring, f, f + kFlush); f += kFlush; }
#ifndef MM_ABL_NOB
                        copy_match(a, bl, bd);
#endif
                    }
                    o += n_total;
                    if (pos >= 64u || cutm) { next_window = true; break; }   // the window is used up (or cut short: the next one begins at the cut)
                }
                // one token of the plain path at base + pos: the end of the block, a code longer than the first-level tables
                b.bitpos = base + (uint64_t)pos;
                uint32_t bits = b.peek32();
                uint32_t e = expand_ll(uni((uint32_t)S.ll[bits & ((1u << kLL) - 1u)]));
                if ((e & 15u) == 0u) { e = decode_long(b, S.cl_ll, false); if ((e & 15u) == 0u) { st = S_BAD_SYMBOL; break; } }
                const uint32_t kind = (e >> 4) & 15u;
                if (kind == K_LIT) {
                    b.bitpos += e & 15u;
                    if (o >= isize) { st = S_OVERRUN_OUT; break; }
                    if (l == 0) S.ring[o & (kRing - 1u)] = (uint8_t)(e >> 16);
                    o++;
                } else if (kind == K_EOB) { b.bitpos += e & 15u; eob = true; break; }
                else {
                    // a match: length (extra bits behind the code), distance code, its extra bits
                    bits >>= e & 15u;
                    const uint32_t xl2 = (e >> 8) & 31u;
                    const uint32_t len = (e >> 16) + (bits & ((1u << xl2) - 1u));
                    b.bitpos += (e & 15u) + xl2;
                    bits = b.peek32();
                    uint32_t d2 = expand_d(uni((uint32_t)S.dt[bits & ((1u << kD) - 1u)]));
                    if ((d2 & 15u) == 0u) { d2 = decode_long(b, S.cl_d, true); if ((d2 & 15u) == 0u) { st = S_BAD_DISTANCE; break; } }
                    bits >>= d2 & 15u;
                    const uint32_t xd2 = (d2 >> 8) & 31u;
                    const uint32_t dist = (d2 >> 16) + (bits & ((1u << xd2) - 1u));
                    b.bitpos += (d2 & 15u) + xd2;
                    if (dist > o) { st = S_BAD_DISTANCE; break; }
                    if (o + len > isize) { st = S_OVERRUN_OUT; break; }
                    if (o - f >= kFlush + 256u) { lds_sync(); ring_flush(out, S.ring, f, f + kFlush); f += kFlush; }
                    copy_match(o, len, dist);
                    o = uni(o + len);
                }
                b.settle();
                pos = (uint32_t)(b.bitpos - base);
                if (pos >= 64u) { next_window = true; break; }
                }   // (segments)
                if (st != S_OK) break;
                if (!eob) b.bitpos = base + (uint64_t)pos;
            }
            if (st != S_OK) return st;
#endif
            if (b.overrun()) return S_OVERRUN_IN;
        }
        if (final_block) break;
    }
    lds_sync();
    ring_flush(out, S.ring, f, o);
    return o == isize ? S_OK : S_SIZE;
}

// a wavefront takes every n-th block of the launch
__global__ __launch_bounds__(64 * kWaves, 6) void k_bgzf_inflate(const uint8_t* __restrict__ cdata, const Block* __restrict__ blocks, int n_blocks,
                                                                 uint8_t* __restrict__ out, int32_t* __restrict__ status) {
    __shared__ WaveLds lds[kWaves];
    __shared__ uint32_t xtab[64];
    fill_xtab(xtab);
    __syncthreads();
    extern __shared__ uint8_t occupancy_pad[];   // (dynamic LDS nobody touches: the launch asks for as much as keeps the workgroups per CU at what the host wants, inflate_wgs_per_cu)
    (void)occupancy_pad;
    WaveLds& S = lds[threadIdx.x >> 6];
    const int n_waves = (int)gridDim.x * kWaves;
    const int first = (int)uni((uint32_t)((int)blockIdx.x * kWaves + (int)(threadIdx.x >> 6)));   // (uniform to the compiler as well)
    for (int i = first; i < n_blocks; i += n_waves) {
        const uint32_t c_off = uni(blocks[i].c_off), c_len = uni(blocks[i].c_len), o_off = uni(blocks[i].o_off), isize = uni(blocks[i].isize);
        int st = S_OK;
        if (isize) st = inflate_block(cdata + c_off, c_len, out + o_off, isize, S, xtab);
        if (lane() == 0) status[i] = st;
    }
}

// ---- CRC32 (IEEE 802.3, reflected, as gzip's): a * b modulo the polynomial, x^(8 n)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b) {   // zlib crc32.c multmodp
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t gf_x8n(uint32_t n, const uint32_t* x2n) {   // x^(8 n): x2n[k] = x^(2^k)
    uint32_t p = 1u << 31;
    uint32_t k = 3;
    while (n) { if (n & 1u) p = gf_mul(x2n[k & 31u], p); n >>= 1; k++; }
    return p;
}

__global__ __launch_bounds__(256) void k_bgzf_crc(const uint8_t* __restrict__ out, const Block* __restrict__ blocks, int n_blocks, int32_t* __restrict__ status) {
    __shared__ uint32_t tab[256];
    __shared__ uint32_t x2n[32];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        tab[threadIdx.x] = c;
        if (threadIdx.x == 0) {
            uint32_t p = 1u << 30;   // x^1
            x2n[0] = p;
            for (int k = 1; k < 32; k++) { p = gf_mul(p, p); x2n[k] = p; }
        }
    }
    __syncthreads();
    const int l = lane();
    const int wave = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = (int)gridDim.x * 4;
    for (int i = wave; i < n_blocks; i += n_waves) {
        const Block bk = blocks[i];
        if (status[i] != S_OK || bk.isize == 0u) { if (bk.isize == 0u && bk.crc != 0u && l == 0 && status[i] == S_OK) status[i] = S_CRC; continue; }
        // 64 slices, a multiple of 4 bytes each but the last
        const uint32_t per = ((bk.isize + 63u) / 64u + 3u) & ~3u;
        const uint32_t a = min((uint32_t)l * per, bk.isize), e = min(a + per, bk.isize);
        const uint8_t* p = out + bk.o_off;
        uint32_t c = l == 0 ? 0xFFFFFFFFu : 0u;
        // (a dword a load: a lane walks its own kilobyte, and with a byte a load a launch of 6 144 blocks kept 50 MB of half-read lines in
        // flight -- 57 GB/s against 215 in launches of 2 048)
        uint32_t j = a;
        for (; j + 4u <= e; j += 4u) {
            uint32_t w;
            __builtin_memcpy(&w, p + j, 4);
            c = tab[(c ^ w) & 255u] ^ (c >> 8);
            c = tab[(c ^ (w >> 8)) & 255u] ^ (c >> 8);
            c = tab[(c ^ (w >> 16)) & 255u] ^ (c >> 8);
            c = tab[(c ^ (w >> 24)) & 255u] ^ (c >> 8);
        }
        for (; j < e; j++) c = tab[(c ^ p[j]) & 255u] ^ (c >> 8);
        // the register of slice l has bk.isize - e bytes behind it
        uint32_t v = gf_mul(gf_x8n(bk.isize - e, x2n), c);
        if (bk.isize - e == 0u) v = c;
        for (int d = 1; d < 64; d <<= 1) v ^= (uint32_t)__shfl_xor((int)v, d);
        if (l == 0 && (v ^ 0xFFFFFFFFu) != bk.crc) status[i] = S_CRC;
    }
}

}  // namespace mmbgzf
