// freq_stream.hip.h -- k_stream_reads: the freq hot path of a whole read in ONE wavefront, with nothing written to HBM but
// the counters (reference src/mod.c:776-1370).
//
// The tile pipeline (freq_tiles.hip.h) cuts a read into independent tiles and pays for the independence: per-op CIGAR
// prefix arrays and a rank directory are written to a scratch and read back, every tile re-derives its carries, stages
// its own slices, and the three kernels read the read record three times.  In a launch big enough to hide a read inside it
// none of that is needed: ONE wavefront takes a read through get_aln, the base lists, the MM parse and the calls, with two
// small tables in LDS that are built ONCE per read (round 3; rounds 1-2 slid two windows along the read instead):
//
//   CIGAR table     one pass over the CIGAR (get_aln walks it before anything else, mod.c:776-881) gives the totals, the checks
//                   AND, for a segment of 512 ops, a checkpoint every FOUR ops: query / reference positions consumed in
//                   front of it, in stored order.  A call finds its checkpoint with a 7-step search in LDS, loads the
//                   checkpoint's four ops (16 bytes the pass has just read) and walks them.  A reverse read needs no mirrored
//                   walk: get_aln's back-to-front walk with mirrored coordinates (mod.c:813-815, :855-858) is the ordinary
//                   projection of BAM position q shifted by L - q_total (SURVEY.md section 8a').  Reads of up to 512 ops -- every
//                   HiFi read, 9 kb of ONT read -- hold their whole CIGAR in the table; longer ones move it a segment at a time
//                   (the pass holds a segment's ops in registers: 1024 ops were sixteen registers and forty more spills).
//   directory       the read's base class counted per 32-base block (popcount + one wave scan per 64 blocks), a segment of 512
//                   blocks (16 kb of read) at a time, walked from the read's END for a reverse read (whose MM counts bases of
//                   the original orientation); every lane finds its block with a 9-step search and selects the base inside it.
//   tokens          the skip list in text order, 256 characters per trip (the per-character sum of k_sum_tiles), into a ring of
//                   ranks; a round takes 64 of them: rank -> block -> read position -> checkpoint -> op -> reference position
//                   -> reference word + ML byte -> threshold -> ONE 64-bit atomic add, the same update the tile pipeline makes.
// Ranks rise inside a group, so a table only ever moves forward (backward for a reverse read's CIGAR) while a group is
// walked; all groups of a read share the tables.
//
// '.' groups: the unlisted bases of the class are calls too; tokens and the bases of the gaps in front of them are one rising
// sequence of ranks that goes through the same rounds (run_group).  That generality costs the '?' path 5 %, so the kernel has
// two instantiations (kDot) and the handle moves to the '.'-capable one when a read with a '.' group has shown up.
// kIns: runs with --insertions and / or --haplotypes ('?' groups only): a base inside an insertion is a call on the anchor
// left of it with its offset (mod.c:864-874), counters live in per-haplotype planes, what has no dense counter goes to the
// side table.
//
// What this kernel does not do, it hands on BEFORE touching a counter: reads with groups on 'N' or on different bases, more
// than four codes or eight groups, a CIGAR the checks do not pass outright -> appended to the tile pipeline's item list
// (k_scan_reads runs after this kernel, or at wait time when nothing else needs it).  Anything that goes wrong once calls have
// been counted is an input error (malformed token, rank past the last base, ML too short): the read goes on the fallback list
// and the fused kernel names the error in the reference's order, exactly as for the tile pipeline's irregular reads.
// DESIGN.md section 4 ("Streaming kernel") has the measurements.
#pragma once
#include <type_traits>
#include "freq_tiles.hip.h"

namespace mmhip {

// diagnostic builds (-DMM_STREAM_TIMING): wave time per phase of k_stream_reads, summed into stats[7..15]
#ifdef MM_STREAM_TIMING
#define KFT_LAP(slot) do { const unsigned long long _n = __builtin_amdgcn_s_memrealtime(); ftacc[slot] += _n - ft0; ft0 = _n; } while (0)
#else
#define KFT_LAP(slot) do {} while (0)
#endif

constexpr uint32_t kStreamChunk = 256;      // skip-list characters parsed per trip (+16 of look-ahead)
constexpr uint32_t kStreamRing = 256;       // token ring, a power of two: at most 63 left over + 128 of a chunk (+1)
#ifndef MM_STREAM_SEG_BLOCKS
#define MM_STREAM_SEG_BLOCKS 512
#endif
#ifndef MM_STREAM_WAVES
#define MM_STREAM_WAVES 7       // wavefronts per SIMD the kernel is built for: 72 registers (measured with 512-op CIGAR segments: C2 33.3 us
                                // per batch against 34.8 at 6 / 80 registers and 37.8 at 5 / 96; view 0.140 against 0.134)
#endif
#ifndef MM_STREAM_WAVES_INS
#define MM_STREAM_WAVES_INS 7   // ... the --insertions / --haplotypes one
#endif
#ifndef MM_STREAM_WAVES_DOT
#define MM_STREAM_WAVES_DOT 6   // ... the '.'-capable instantiation: at 72 registers it spills 55 of them (C2 with '.' flags 179 against 172 us)
#endif
constexpr uint32_t kSegBlocks = MM_STREAM_SEG_BLOCKS;   // directory segment: 32-base blocks (a multiple of 64)
#ifndef MM_SEG_OPS
#define MM_SEG_OPS 512   // (1024: the pass over a segment holds sixteen registers of ops and the kernel spills 51 registers instead of
                         // 10 -- C2 36.6 against 35.1 us per batch, C3 30.5 against 29.2; 256: an ONT read's 1 000 ops are four
                         // dependent trips instead of two, C2 39.3)
#endif
#ifndef MM_NO_CK16
#define MM_CK16 1
#endif
#ifdef MM_CK16
// Round 5: a segment is 1024 ops in the LDS 512 took -- its checkpoints are 16-bit, RELATIVE to the segment's first op.  An ONT read's
// thousand ops are ONE table: no second pass over the CIGAR when the tokens reach op 512, no round cut in two at that edge (365 vector
// instructions a read fewer with 1024-op segments of 32-bit checkpoints, which cost the seventh wavefront its LDS: DESIGN section 5).
// The pass over a segment still holds 512 ops at a time in registers (two halves, one behind the other).  A read with a segment whose
// ops consume 65 535 positions or more of the read or of the reference (a 64 kb deletion, an intron) is WIDE: its tables are round 3's,
// 512 ops with 32-bit checkpoints in the same bytes (kWideOps; seg_pass_wide).
constexpr uint32_t kSegOps = 1024;
constexpr int kSegVec = 2;                   // 16-byte words per lane of HALF a segment
constexpr uint32_t kSegCk = kSegOps / 4;
typedef uint16_t ck_t;
constexpr uint32_t kCkInf = 0xFFFFu;
constexpr uint32_t kWideOps = 512, kWideCk = kWideOps / 4;
#else
constexpr uint32_t kSegOps = MM_SEG_OPS;    // CIGAR segment: ops
constexpr int kSegVec = (int)(kSegOps / 256u);   // ... as 16-byte words per lane
constexpr uint32_t kSegCk = kSegOps / 4;    // ... and its checkpoints, one per four ops
typedef uint32_t ck_t;
constexpr uint32_t kCkInf = 0xFFFFFFFFu;
#endif
constexpr uint32_t kStreamGroups = 8;       // MM groups per read (more: tile pipeline)
constexpr uint32_t kStreamMemo = 4;         // group ordinals whose last header is remembered
constexpr uint32_t kStreamInf = 0xFFFFFFFFu; // the bound behind a table's last entry
constexpr uint32_t kStreamMaxLen = 1u << 20; // an op this long (a 1 Mb intron) is the tile pipeline's: shorter ones cannot wrap a segment's sums
constexpr uint32_t kNoPend = 0xFFFFFFFFu;

// kDotLds: the instantiation does '.' groups (their two arrays are the 516 bytes the member bits below take in the others, which
// have to fit seven workgroups into a CU's 160 KB)
template <bool kDotLds>
struct StreamLdsT {
    uint32_t mmw[kStreamChunk / 4 + 4];     // the chunk's characters
    uint32_t tok[kStreamRing];              // ranks of parsed tokens not yet called
    // the directory segment (traversal order): which of a block's 32 bases are class members (stream_block_mask's word), and the members in front of every FOURTH block, then the running total.  A call finds its four blocks with a 7-step
    // search, takes their bits with one 16-byte read and selects its base there -- round 3 kept a count per block and fetched the
    // block's 16 bytes of sequence from memory a second time (the whole read, once more, long after the L2 had let go of it)
    alignas(16) uint32_t dm[kSegBlocks];
    uint32_t cw[kSegBlocks / 4 + 1];
#ifdef MM_CK16
    union {
        struct { uint16_t cq[kSegCk + 2]; uint16_t cr[kSegCk]; };       // query / reference positions consumed in front of every fourth op of the segment, relative to its first op; then the bound
        struct { uint32_t cq32[kWideCk + 1]; uint32_t cr32[kWideCk]; }; // a WIDE read's: 512 ops, absolute
    };
#else
    uint32_t cq[kSegCk + 1];                // query positions consumed in front of every fourth op of the segment (stored order), then 0xFFFFFFFF
    uint32_t cr[kSegCk];                    // ... reference positions
#endif
    uint32_t gap_p[kDotLds ? 65 : 1];                     // '.' groups: first element (gap bases, then the token) of each token of the batch; [n] = all
    uint32_t gap_r[kDotLds ? 64 : 1];                     //             first rank of the gap in front of each token
    char hdr[16];
    int16_t g_code[16];
    // what the header pass leaves for the groups: where the group starts and its list begins, flags (bit 6 no requested code,
    // 12-14 codes per token), the codes and their packed table entries
    uint32_t g_mpos[kStreamGroups], g_lstart[kStreamGroups], g_flags[kStreamGroups], g_c01[kStreamGroups], g_c23[kStreamGroups];
    uint32_t g_end[kStreamGroups];   // the group's ';' (or the string's end)
    // side-list appends (round 4): the wavefront reserves kSideChunk records of its region at a time and hands them out itself --
    // one atomic with a return value per chunk instead of one per round and code (the lanes of the round waited for it: a trip to the
    // memory side in the middle of every round of a --insertions run).  [at, end) is what is left of the chunk, kf a key of the chunk:
    // what is left when the next chunk is taken, or when the kernel ends, is filled with (kf, increment 0) -- the lists' readers
    // take every record below the cursor.  In LDS because the appends sit in divergent code (only the lanes with an update run them).
    uint32_t sres_at, sres_end, sres_klo, sres_khi;
    uint32_t g_ci[kStreamGroups][4];
    // the headers this wave resolved last, by group ordinal: the reads of a file nearly all carry the same ones, and a header
    // whose characters are those of the memo needs neither the checks nor the code table again
    uint32_t memo_len[kStreamMemo];          // header characters incl. the flag (0 = empty)
    uint8_t memo_hdr[kStreamMemo][16];
    uint32_t memo_flags[kStreamMemo], memo_c01[kStreamMemo], memo_c23[kStreamMemo], memo_ci[kStreamMemo][4];
};

// The read's base class as a nibble pattern (C 2, G 4, T 8, N 15 in every nibble; 0 for the fifth class: whatever is none of
// those), made once per read: the match bits of a word are then five instructions and no branch on the class.
__device__ __forceinline__ uint32_t class_pattern(int cls) {
    return cls == 1 ? 0x22222222u : (cls == 2 ? 0x44444444u : (cls == 3 ? 0x88888888u : (cls == 4 ? 0xFFFFFFFFu : 0u)));
}
// bit 4n+3 set iff nibble n of x equals the pattern's nibble
__device__ __forceinline__ uint32_t pattern_eq(uint32_t x, uint32_t pat) {
    const uint32_t y = x ^ pat;
    return ~(((y & 0x77777777u) + 0x77777777u) | y) & 0x88888888u;
}
// ... or, for the fifth class, is none of 2, 4, 8, 15
__device__ __forceinline__ uint32_t pattern_other(uint32_t x) {
    return 0x88888888u & ~(nib_eq(x, 2) | nib_eq(x, 4) | nib_eq(x, 8) | nib_eq(x, 15));
}
// the match bits of a block's four words (one branch on the class for the block, none per word)
__device__ __forceinline__ void block_bits(uint4 v, uint32_t pat, uint32_t& m0, uint32_t& m1, uint32_t& m2, uint32_t& m3) {
    if (pat != 0u) { m0 = pattern_eq(v.x, pat); m1 = pattern_eq(v.y, pat); m2 = pattern_eq(v.z, pat); m3 = pattern_eq(v.w, pat); }
    else { m0 = pattern_other(v.x); m1 = pattern_other(v.y); m2 = pattern_other(v.z); m3 = pattern_other(v.w); }
}
// class members among the 32 bases of block b (KA::block_count)
__device__ __forceinline__ uint32_t stream_block_count(uint4 v, uint32_t pat, uint32_t b, uint32_t L) {
    uint32_t m0, m1, m2, m3;
    block_bits(v, pat, m0, m1, m2, m3);
    uint32_t c = __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
    if (pat == 0u) c -= 32u - min(32u, L - b * 32u);   // the padding behind the read's last base is none of C, G, T, N either
    return c;
}
// Which of block b's 32 bases are members of the class, as ONE word: bit 4n + i = nibble n of the block's word i -- the four words'
// match bits, which sit at 4n + 3, shifted into each other's gaps.  No bit is moved to its base's place here: eight blocks a lane are made
// per read and only a few of them are ever asked (select_bit does the sorting out, for one block).  (With the freq units still built with
// machine LICM -- scratch spills in the rounds -- the dearer selection made this form 2.7 % SLOWER than bits compacted to base order with
// 31 instructions a block; without the spills it is 220 vector instructions a read fewer and 2.1 % faster: the unit is the bound.)
__device__ __forceinline__ uint32_t stream_block_mask(uint4 v, uint32_t pat, uint32_t b, uint32_t L) {
    uint32_t m0, m1, m2, m3;
    block_bits(v, pat, m0, m1, m2, m3);
    if (pat == 0u) {   // the padding behind the read's last base is none of C, G, T, N either
        const int valid = (int)min(32u, L - b * 32u);
        m0 &= base_order(valid_bits(valid)); m1 &= base_order(valid_bits(valid - 8));
        m2 &= base_order(valid_bits(valid - 16)); m3 &= base_order(valid_bits(valid - 24));
    }
    return (m0 >> 3) | (m1 >> 2) | (m2 >> 1) | m3;
}
// the k-th member (k counts from 0 and is below the popcount) of such a word, in the order of the bases -> base 0 .. 31 of the block
__device__ __forceinline__ uint32_t select_bit(uint32_t c, uint32_t k) {
    const uint32_t s1 = __popc(c & 0x11111111u), s2 = s1 + __popc(c & 0x22222222u), s3 = s2 + __popc(c & 0x44444444u);
    const uint32_t word = (k >= s1 ? 1u : 0u) + (k >= s2 ? 1u : 0u) + (k >= s3 ? 1u : 0u);
    k -= word == 0u ? 0u : (word == 1u ? s1 : (word == 2u ? s2 : s3));
    uint32_t mk = base_order((c >> word) & 0x11111111u);   // bit 4n <-> base n of the word (a byte holds base 2i in its HIGH nibble)
    uint32_t n = 0, cn = __popc(mk & 0xFFFFu);
    bool ge = k >= cn;
    k -= ge ? cn : 0u; n += ge ? 4u : 0u; mk = ge ? mk >> 16 : mk;
    cn = __popc(mk & 0xFFu);
    ge = k >= cn;
    k -= ge ? cn : 0u; n += ge ? 2u : 0u; mk = ge ? mk >> 8 : mk;
    n += k >= (mk & 1u) ? 1u : 0u;
    return word * 8u + n;
}
// all ones when bit `op` of `mask` is set, else 0 (one signed bit-field extract: which ops consume the read / the reference)
__device__ __forceinline__ uint32_t op_mask(uint32_t mask, uint32_t op) { return (uint32_t)__builtin_amdgcn_sbfe((int)mask, op, 1u); }
// number of set bits below the lowest clear one (64 when all are set)
__device__ __forceinline__ uint32_t leading_ones(uint64_t m) { return ~m ? (uint32_t)__ffsll((unsigned long long)~m) - 1u : 64u; }

// largest i in [0, n) with arr[i] <= key: arr[0 .. n) rises, arr[0] <= key < arr[n] (the entry behind the last one in use is a
// bound no key of the round reaches).  The steps are those of a search over N entries, so the loop unrolls into ceil(log2 N)
// steps of a read, a compare and a select with no loop control; a probe beyond n reads the bound instead.
template <uint32_t N>
__device__ __forceinline__ uint32_t search_le(const uint32_t* arr, uint32_t key, uint32_t n) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t m = N; m > 1u; m -= m >> 1) {
        const uint32_t half = m >> 1;
        const uint32_t at = min(lo + half, n);
        lo = arr[at] <= key ? at : lo;
    }
    return lo;
}
// ... over a table whose entries behind the last one in use, up to N - 1, all hold the bound (the table's maker fills them): no probe
// has to be kept inside the part in use, which is one vector instruction a step less, fourteen a call
template <uint32_t N>
__device__ __forceinline__ uint32_t search_le_padded(const uint32_t* arr, uint32_t key) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t m = N; m > 1u; m -= m >> 1) {
        const uint32_t at = lo + (m >> 1);
        lo = arr[at] <= key ? at : lo;
    }
    return lo;
}

template <uint32_t N>
__device__ __forceinline__ uint32_t search_le_padded(const uint16_t* arr, uint32_t key) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t m = N; m > 1u; m -= m >> 1) {
        const uint32_t at = lo + (m >> 1);
        lo = (uint32_t)arr[at] <= key ? at : lo;
    }
    return lo;
}

// kDot: '.' groups (implicit calls) are this kernel's too; without it (the leaner instantiation) reads that have one go to the
// tile pipeline and a flag tells the host to launch the other instantiation from then on (a file's reads carry one flag or the other)
// kView: `minimod view` -- a call that passes the context test becomes a record (view_append) instead of a counter update
// kIns: --insertions and / or --haplotypes (freq, '?' groups)
template <typename RefWord, bool kStats, bool kDot, bool kView, bool kIns>
struct KF {
    static_assert(!(kIns && kView), "the --insertions / --haplotypes instantiations are freq");
    const TileParams& P;
    const DevParams& p;
    using StreamLds = StreamLdsT<kDot>;
    StreamLds& S;
    const uint32_t* ptab;
    uint32_t st_look, st_ml, st_dense, st_side;
    int err;
    // the read (wave-uniform)
    const uint8_t* mm;
    const uint8_t* ml;
    const uint4* sq;
    const uint32_t* cg;
    typename RefLoad<RefWord>::Base rwb;
    uint32_t L, ncig, nblk, mlen, ml_len, q_total, r_total, q_shift, seg_lo32, seg_len32, cpat;
    int32_t pos, rev, ridx_cur;
    int32_t hp, hpi;            // kIns: the read's haplotype tag (-1: haplotypes off), its dense plane (-1: none)
    uint32_t v_region, v_gord;   // view: append region of the wavefront, ordinal of the group at hand
    uint32_t v_seq;              // view: records of the read so far
    uint32_t v_sorted;           // view: 0x80000000 when the read's records are made in the order of its rows (one requested code in one
                                 // '?' group: positions rise with the ranks of a forward read and fall with those of a reverse one)
    // the group
    int32_t ncg;
    bool dot_group;             // a '.' group: unlisted bases are calls too
    bool saw_dot;               // (!kDot) a read was handed on because of a '.' group
    uint32_t gc01, gc23;        // the group's codes, two 16-bit indices a word (0xFFFF: not requested)
    uint32_t ci0, ci1, ci2, ci3;
    // the group's counters (freq_kernels.hip.h, DevClass): all requested codes of a group belong to one context class (else the
    // read is the tile pipeline's).  gcb[(site * gnp) + slot] is the counter of the code in `slot` at the read's strand and
    // haplotype plane, `site` = the rank of the position among the class's sites (gsite: 32 positions a word) -- or, for a dense
    // class (context `*`, --insertions), the position in the contig itself (gsite null).  Null gcb: no dense counters here.
    unsigned long long* gcb;
    const uint2* gsite;
    uint32_t gnp;
    int64_t ref_base_g;         // the contig's first position in the reference-word space
    bool has_dense;             // the contig has dense counters in this handle
    int32_t tid_cur;
    // kIns: the part of a side update's key that is the READ's -- contig base << 28 | strand << 27 | haplotype -- when every key of the
    // read fits the 64-bit form (side_key's range checks made once per read, not per update); side_fast says so
    unsigned long long side_kbase;
    bool side_fast;
    uint32_t ml_start;
    // TWIN groups (run()): two one-code `?` lists over the same tokens, done in one pass.  tw = 0: an ordinary group, the ML byte
    // of token k's code m is ml[ml_start + k * ncg + m]; tw = the list's tokens: ml[ml_start + k + m * tw] (the second list's
    // bytes follow the first's).  Never with one requested mod (RefNib): two requested codes are what makes a pair.
    static constexpr bool kTwinOK = !std::is_same<RefWord, RefNib>::value;
    uint32_t tw;
    // text cursor and token ring
    uint32_t cpos, nx_w0, nx_w1, qhead, qn, kdone, Rcarry, ntok_parsed;
    // a round's counter updates are ISSUED at the top of the next round, behind that round's loads (one per lane: the update of
    // the group's first code; further codes go out at once): on gfx9 an atomic counts in vmcnt like a load, so the wait for the
    // next loads would otherwise sit out the atomics' trip to the memory side as well
    uint32_t pend;   // reference position << 1 | modified; kNoPend: none (a position is below 2^31 - 1: contig lengths are int32)
    uint32_t staged_at, skip0;   // text offset of the chunk in LDS; header characters in front of the list in the group's first chunk
    bool prev_delim, closed, bad_text;
    // directory segment: blocks [d_t0, d_t0 + d_n) in traversal order (d_n = 0: none), S_lo / S_hi class members in front of it / through it
    uint32_t d_t0, d_n, S_lo, S_hi;
    // CIGAR segment: stored ops [c_o0, c_o0 + 1024) as c_n checkpoints (0: none); Q_lo / Q_hi (R_lo / R_hi) query (reference)
    // positions consumed in front of its first op / through its last op
    uint32_t c_o0, c_n, Q_lo, Q_hi, R_lo, R_hi;
#ifdef MM_CK16
    bool wide;          // the read's CIGAR tables are 512 ops of 32-bit checkpoints (a segment of 1024 ops spans 65 535 positions or more)
    __device__ __forceinline__ uint32_t seg_ops() const { return wide ? kWideOps : kSegOps; }
#else
    __device__ __forceinline__ uint32_t seg_ops() const { return kSegOps; }
#endif
    uint32_t rank_ok;   // 1 + the largest rank that has been found in the read (0: none yet): what an unrequested group's last rank is checked against
#ifdef MM_STREAM_TIMING
    unsigned long long ftacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ft0 = 0;   // 0 record + CIGAR pass, 1 headers, 2 parse, 3 directory,
                                                                           // 4 locate, 5 CIGAR segment, 6 finish, 7 --, 8 group set-up
#endif

    __device__ KF(const TileParams& tp, StreamLds& s, const uint32_t* tab)
        : P(tp), p(tp.d), S(s), ptab(tab), st_look(0), st_ml(0), st_dense(0), st_side(0), err(0), saw_dot(false) {}

    // --insertions with '.' groups: an implicit call on a base that is not aligned takes its anchor from the MIRRORED read position
    // (the reference indexes ins[] with the BAM position there and everything else with the original one, mod.c:1234, :1314):
    // a reverse read's implicit calls look a second op up, anywhere in the CIGAR.  With the whole CIGAR in the table (up to kSegOps
    // ops) that is one more search; a longer reverse read stays with the tile pipeline, which keeps every op's sums in memory.
    __device__ __forceinline__ bool dot_ins_far() const { return p.insertions && rev && ncig > seg_ops(); }
    __device__ __forceinline__ int gcode_at(int m) const { return (int)(int16_t)(((m < 2 ? gc01 : gc23) >> (16 * (m & 1))) & 0xFFFFu); }
    __device__ __forceinline__ uint32_t cinfo_at(int m) const { return m == 0 ? ci0 : (m == 1 ? ci1 : (m == 2 ? ci2 : ci3)); }

    // ------------------------------------------------------------------ text -> ranks
    __device__ __forceinline__ void fetch_chunk(uint32_t c) {
        const uint32_t lane = (uint32_t)lane_id();
        if (c + kStreamChunk + 16u <= mlen) {   // (every chunk but the text's last: nothing to replace behind the end, mm_dword's seven instructions a load)
            __builtin_memcpy(&nx_w0, mm + c + 4u * lane, 4);
            nx_w1 = 0u;
            if (lane < 4u) __builtin_memcpy(&nx_w1, mm + c + kStreamChunk + 4u * lane, 4);
            return;
        }
        nx_w0 = mm_dword(mm, mlen, c + 4u * lane);
        nx_w1 = lane < 4u ? mm_dword(mm, mlen, c + kStreamChunk + 4u * lane) : 0u;
    }
    // One chunk of the group's skip list: its tokens appended to the ring as ranks (keep), counted and summed either way.
    // The parse is k_sum_tiles' (a token belongs to the chunk it starts in; the look-ahead completes the last one).
    __device__ __forceinline__ void stage_chunk(uint32_t at) {
        const int lane = lane_id();
        wave_sync();
        S.mmw[lane] = nx_w0;
        if (lane < 4) S.mmw[64 + lane] = nx_w1;
        wave_sync();
        staged_at = at;
    }
    __device__ __forceinline__ void parse_chunk(bool keep) {
        const int lane = lane_id();
        if (staged_at != cpos) stage_chunk(cpos);
        fetch_chunk(cpos + kStreamChunk);   // the next chunk is requested before this one is parsed
        const uint32_t skip = skip0;         // the group's first chunk starts at its header: the list begins `skip` characters in
        skip0 = 0;
        const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(S.mmw);
        // Round 4: FOUR characters a lane (its dword of the chunk) and ONE scan a chunk -- rounds 2 and 3 took a character a lane, a
        // quarter of the chunk at a time, four scans one behind the other: 29 % of the kernel's time on C2 (profiles/r4q_phases.txt).
        const uint32_t w = S.mmw[lane];
        const uint32_t x4 = mb8[kStreamChunk + (lane & 15)];
        const uint32_t yc = w ^ 0x2C2C2C2Cu, ys = w ^ 0x3B3B3B3Bu, yd = w ^ 0x30303030u;
        const uint32_t cm = ~(((yc & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yc) & 0x80808080u;   // commas (bit 8j+7: character j of the dword)
        const uint32_t sm = ~(((ys & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | ys) & 0x80808080u;   // semicolons
        const uint32_t ndm = (((yd & 0x7F7F7F7Fu) + 0x76767676u) | yd) & 0x80808080u;    // not a digit
        const uint32_t dlm = cm | sm;
        const uint32_t xd = dlm >> 7;
        uint32_t d4 = (xd | (xd >> 7) | (xd >> 14) | (xd >> 21)) & 0xFu;                 // the dword's delimiters: bit j
        const uint32_t L16 = (uint32_t)__ballot(lane < 16 && (x4 == ',' || x4 == ';')) & 0xFFFFu;   // ... those of the 16 characters behind the chunk
        // where the group ends (its ';') and where this chunk's own tokens begin
        const uint64_t semis = __ballot(sm != 0u);
        uint32_t hi = kStreamChunk;
        bool cl = false;
        if (semis) {
            const int ls = __ffsll((unsigned long long)semis) - 1;
            hi = 4u * (uint32_t)ls + ((uint32_t)__ffs((int)lane_valu(sm, ls)) - 1u) / 8u;
            cl = true;
        }
        if (skip && (uint32_t)lane == (skip - 1u) >> 2) d4 |= 1u << ((skip - 1u) & 3u);   // the list's first character follows the header as if it followed a delimiter
        uint32_t lo = 0;
        if (skip) lo = skip;
        else if (!prev_delim) {   // the token that began in the chunk before is that chunk's
            const uint64_t dels = __ballot(d4 != 0u);
            lo = kStreamChunk;
            if (dels) { const int ld = __ffsll((unsigned long long)dels) - 1; lo = 4u * (uint32_t)ld + (uint32_t)__ffs((int)lane_valu(d4, ld)) - 1u; }
        }
        // the lane's window: bit 0 = the character in front of the dword, bits 1..4 the dword's, bits 5..16 the twelve behind it
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d4, 0x130, 0xF, 0xF, true);   // wave_shl:1 -- lane i takes lane i + 1's
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, 0x130, 0xF, 0xF, true);
        const uint32_t a3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a2, 0x130, 0xF, 0xF, true);
        uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d4, 0x138, 0xF, 0xF, true) >> 3;       // wave_shr:1 -- the lane in front's last character
        if (lane == 0) pv = prev_delim ? 1u : 0u;
        uint32_t W = d4 | (a1 << 4) | (a2 << 8) | (a3 << 12);
        if (lane >= 61) W |= (L16 << (4u * (64u - (uint32_t)lane))) & 0xFFFFu;
        W = (W << 1) | pv;
        const uint32_t base = 4u * (uint32_t)lane;
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, ends = 0, badv = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t wj = W >> j;   // bit 0 the character in front, bit 1 this one, bit 2 the next
            const bool own = base + (uint32_t)j - lo < hi - lo && lo < hi;
            const bool start = own && (wj & 3u) == 1u;
            const bool end = own && (wj & 6u) == 4u;
            uint32_t e = (uint32_t)__ffs((int)(wj >> 1)) - 1u;
            e = e < 10u ? e : 10u;
            const uint32_t dv = ((w >> (8 * j)) & 0xFFu) - (uint32_t)'0';
            uint32_t c = ptab[(e << 4) | (dv & 15u)];
            c = own ? c : 0u;
            c += start ? 1u : 0u;
            if (j == 0) c0 = c; else if (j == 1) c1 = c; else if (j == 2) c2 = c; else c3 = c;
            ends |= end ? 1u << j : 0u;
            badv |= (own && ((ndm & ~dlm) >> (8 * j + 7) & 1u)) || (start && e >= 10u) ? 1u : 0u;
        }
        const uint32_t p0 = c0, p1 = p0 + c1, p2 = p1 + c2, p3 = p2 + c3;
        const uint32_t run = wave_incl_scan(p3);
        const uint32_t rsum0 = lane_valu(run, 63);
        const uint64_t E1 = __ballot(ends != 0u), E2 = __ballot((ends & (ends - 1u)) != 0u);   // lanes with a token's end, with two (four characters hold no more)
        const uint32_t nends = (uint32_t)__popcll(E1) + (uint32_t)__popcll(E2);
        const uint32_t qtail = qhead + qn;
        if (keep && ends) {
            const uint32_t at = qtail + (uint32_t)__popcll(E1 & lanemask_lt()) + (uint32_t)__popcll(E2 & lanemask_lt());
            const uint32_t front = Rcarry + run - p3 - 1u;
            const uint32_t j1 = (uint32_t)__ffs((int)ends) - 1u;
            S.tok[at & (kStreamRing - 1u)] = front + (j1 == 0u ? p0 : (j1 == 1u ? p1 : (j1 == 2u ? p2 : p3)));
            const uint32_t rest = ends & (ends - 1u);
            if (rest) {
                const uint32_t j2 = (uint32_t)__ffs((int)rest) - 1u;   // 2 or 3
                S.tok[(at + 1u) & (kStreamRing - 1u)] = front + (j2 == 2u ? p2 : p3);
            }
        }
        uint64_t bad = __ballot(badv != 0u);
        uint32_t rsum_v = rsum0;
        // the chunk's last character inside a token that goes on behind it
        const bool open_tail = !cl && ((lane_valu(W, 63) >> 4) & 3u) == 0u;   // (bit 4: the chunk's last character, bit 5: the first one behind it)
        const uint64_t D_look = (uint64_t)L16 | ~0xFFFFull;
        uint32_t ntok = nends;
        if (open_tail) {
            const int k = __ffsll((unsigned long long)D_look) - 1;   // 1..16
            const uint32_t dv = x4 - (uint32_t)'0';
            uint32_t e = (uint32_t)(k - lane);
            e = e < 10u ? e : 10u;
            uint32_t c = lane < k ? ptab[(e << 4) | (dv & 15u)] : 0u;
            bad |= __ballot(lane < k && dv > 9u);
            rsum_v += lane_valu(wave_incl_scan(c), 15);
            if (keep && lane == 0) S.tok[(qtail + nends) & (kStreamRing - 1u)] = Rcarry + rsum_v - 1u;
            ntok = nends + 1u;
        }
        if (bad) bad_text = true;
        prev_delim = (lane_valu(w, 63) >> 24) == (uint32_t)',';
        closed = cl;
        // (the cursors are wave-uniform by construction; saying so keeps them and the arithmetic on them in scalar registers)
        if (keep) qn = uniu(qn + ntok);
        Rcarry = uniu(Rcarry + rsum_v);
        ntok_parsed = uniu(ntok_parsed + ntok);
        cpos += kStreamChunk;
        wave_sync();
    }

    // ------------------------------------------------------------------ sequence -> directory segment
    // blocks [t0, t0 + kSegBlocks) of the read in traversal order (from the END for a reverse read), s0 class members in front
    // of them: every block's count from one wave scan per 64 blocks, four 16-byte loads a lane in flight
    __device__ __forceinline__ void build_dir(uint32_t t0, uint32_t s0) {
        const uint32_t lane = (uint32_t)lane_id();
        const uint32_t nb = min(kSegBlocks, nblk - t0);
        uint32_t run = s0;
        wave_sync();
        for (uint32_t h = 0; h < nb; h += 256u) {
            // a lane takes FOUR CONSECUTIVE blocks (64 bytes of sequence): their bits go out as one 16-byte LDS write, the count in front of
            // them is the lane's entry of `cw`, and one scan over the lanes' totals serves 256 blocks (a block a lane, 64 blocks a row, was a
            // scan a row: four of them)
            uint4 vv[4];
            const uint32_t at = h + 4u * lane, tb = t0 + at;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = tb + (uint32_t)r;
                vv[r] = at + (uint32_t)r < nb ? sq[rev ? nblk - 1u - t : t] : make_uint4(0, 0, 0, 0);
            }
            uint32_t mk[4], total = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = tb + (uint32_t)r;
                mk[r] = at + (uint32_t)r < nb ? stream_block_mask(vv[r], cpat, rev ? nblk - 1u - t : t, L) : 0u;   // (zero behind the read's last block: a group of four is whole)
                total += __popc(mk[r]);
            }
            const uint32_t incl = wave_incl_scan(total);
            *reinterpret_cast<uint4*>(&S.dm[at]) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
            if (at < nb) S.cw[at >> 2] = run + incl - total;
            run = uniu(run + lane_valu(incl, 63));
        }
        if (lane == 0u) S.cw[(nb + 3u) >> 2] = run;
        for (uint32_t c = ((nb + 3u) >> 2) + 1u + lane; c <= kSegBlocks / 4; c += 64u) S.cw[c] = kStreamInf;   // (search_le_padded)
        d_t0 = t0; d_n = nb; S_lo = s0; S_hi = run;
        wave_sync();
    }
    // all class members of the read (only needed to check the last rank of a group nobody asked for)
    __device__ __forceinline__ uint32_t count_all() {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t sum = 0;
        for (uint32_t b0 = 0; b0 < nblk; b0 += 512u) {
            uint4 vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t b = b0 + 64u * (uint32_t)u + lane; vv[u] = b < nblk ? sq[b] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t b = b0 + 64u * (uint32_t)u + lane; if (b < nblk) sum += stream_block_count(vv[u], cpat, b, L); }
        }
        return lane_valu(wave_incl_scan(sum), 63);
    }

    // ------------------------------------------------------------------ CIGAR -> checkpoint table
    // the ops of segment [o0, o0 + 1024) as the lanes hold them: lane's word u = stored ops o0 + 256u + 4 lane .. + 3
    __device__ __forceinline__ void load_seg(uint4 (&cv)[kSegVec], uint32_t o0) const {
        const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
        for (int u = 0; u < kSegVec; u++) {
            const uint32_t i = o0 + 256u * (uint32_t)u + 4u * lane;
            cv[u] = i < ncig ? *reinterpret_cast<const uint4*>(cg + i) : make_uint4(0, 0, 0, 0);
        }
    }
    // query / reference positions the four ops of one word consume (the word holds stored ops i .. i + 3); what the ops are
    // (okops: all of M I D N S = X) and how long (lenor) is collected for the checks
    __device__ __forceinline__ void word_sums(uint4 v, uint32_t i, uint32_t& a, uint32_t& b, uint32_t& okops, uint32_t& lenor) const {
        // a word behind the read's last op counts as 0M: nothing to either sum, nothing wrong with it
        const uint32_t w4[4] = {v.x, i + 1u < ncig ? v.y : 0u, i + 2u < ncig ? v.z : 0u, i + 3u < ncig ? v.w : 0u};
        a = 0; b = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t op = w4[k] & 15u, len = w4[k] >> 4;
            a += len & op_mask(0x193u, op);
            b += len & op_mask(0x18Du, op);
            okops &= op_mask(0x19Fu, op);
            lenor |= len;
        }
    }
#ifdef MM_CK16
    // Segment o0 (1024 ops) of the CIGAR: its first half from the words the lanes hold, its second half loaded here; its totals (tq / tr)
    // and, with `build`, its checkpoint table -- 16-bit entries relative to the segment's first op, whichever end (qa, ra) counts from
    // (from_end: through the segment's last op -- a reverse read's tokens walk the CIGAR from its end): Q_lo / R_lo carry the rest.
    __device__ __forceinline__ void seg_pass(uint4 (&cv)[kSegVec], uint32_t o0, uint32_t qa, uint32_t ra, bool from_end, bool build,
                                             uint32_t& okops, uint32_t& lenor, uint32_t& tq_out, uint32_t& tr_out) {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t tq = 0, tr = 0;
        if (build) wave_sync();
        for (uint32_t hf = 0; hf < 2u; hf++) {
            const uint32_t oh = o0 + 512u * hf;
            if (oh >= ncig) break;
            if (hf) load_seg(cv, oh);
            if (!build) {
                uint32_t a = 0, b = 0;
#pragma unroll
                for (int u = 0; u < kSegVec; u++) {
                    uint32_t au, bu;
                    word_sums(cv[u], oh + 256u * (uint32_t)u + 4u * lane, au, bu, okops, lenor);
                    a += au; b += bu;
                }
                tq = uniu(tq + lane_valu(wave_incl_scan(a), 63)); tr = uniu(tr + lane_valu(wave_incl_scan(b), 63));
            } else {
#pragma unroll
                for (int u = 0; u < kSegVec; u++) {
                    if (oh + 256u * (uint32_t)u < ncig) {
                        const uint32_t i = oh + 256u * (uint32_t)u + 4u * lane;
                        uint32_t a, b;
                        word_sums(cv[u], i, a, b, okops, lenor);
                        const uint32_t iq = wave_incl_scan(a), ir = wave_incl_scan(b);
                        // (a span of 65 535 or more wraps here: such a read never gets as far as a round, run() hands it on)
                        if (i < ncig) { S.cq[128u * hf + 64u * (uint32_t)u + lane] = (ck_t)(tq + iq - a); S.cr[128u * hf + 64u * (uint32_t)u + lane] = (ck_t)(tr + ir - b); }
                        tq = uniu(tq + lane_valu(iq, 63)); tr = uniu(tr + lane_valu(ir, 63));
                    }
                }
            }
        }
        if (build) {
            const uint32_t nck = (min(kSegOps, ncig - o0) + 3u) >> 2;
            for (uint32_t c = nck + lane; c < kSegCk + 2u; c += 64u) S.cq[c] = (ck_t)kCkInf;   // (the bound, up to the table's end: search_le_padded)
            const uint32_t qb = from_end ? qa - tq : qa, rb = from_end ? ra - tr : ra;
            c_o0 = o0; c_n = nck; Q_lo = qb; Q_hi = qb + tq; R_lo = rb; R_hi = rb + tr;
            wave_sync();
        }
        tq_out = tq; tr_out = tr;
    }
    // a WIDE read's segment: 512 ops from the words the lanes hold, 32-bit checkpoints, absolute (round 3's table)
    __device__ __forceinline__ void seg_pass_wide(const uint4 (&cv)[kSegVec], uint32_t o0, uint32_t qa, uint32_t ra, bool from_end) {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t tq = 0, tr = 0, okops = 0xFFFFFFFFu, lenor = 0;
        const uint32_t q0 = from_end ? 0u : qa, r0 = from_end ? 0u : ra;
        wave_sync();
#pragma unroll
        for (int u = 0; u < kSegVec; u++) {
            if (o0 + 256u * (uint32_t)u < ncig) {
                const uint32_t i = o0 + 256u * (uint32_t)u + 4u * lane;
                uint32_t a, b;
                word_sums(cv[u], i, a, b, okops, lenor);
                const uint32_t iq = wave_incl_scan(a), ir = wave_incl_scan(b);
                if (i < ncig) { S.cq32[64u * (uint32_t)u + lane] = q0 + tq + iq - a; S.cr32[64u * (uint32_t)u + lane] = r0 + tr + ir - b; }
                tq = uniu(tq + lane_valu(iq, 63)); tr = uniu(tr + lane_valu(ir, 63));
            }
        }
        const uint32_t nck = (min(kWideOps, ncig - o0) + 3u) >> 2;
        uint32_t qb = q0, rb = r0;
        if (from_end) {
            qb = qa - tq; rb = ra - tr;
            wave_sync();
#pragma unroll
            for (int u = 0; u < kSegVec; u++) {
                const uint32_t c = 64u * (uint32_t)u + lane;
                if (c < nck) { S.cq32[c] += qb; S.cr32[c] += rb; }
            }
        }
        for (uint32_t c = nck + lane; c <= kWideCk; c += 64u) S.cq32[c] = kStreamInf;
        c_o0 = o0; c_n = nck; Q_lo = qb; Q_hi = qb + tq; R_lo = rb; R_hi = rb + tr;
        wave_sync();
    }
#else
    // Segment o0 of the CIGAR from the words the lanes hold: its totals (tq / tr) and, with `build`, its checkpoint table.
    // from_end = false: (qa, ra) are the positions consumed in front of op o0; true: through the segment's last op (a reverse
    // read's tokens walk the CIGAR from its end): the table is then written relative to the segment's start and moved.
    __device__ __forceinline__ void seg_pass(const uint4 (&cv)[kSegVec], uint32_t o0, uint32_t qa, uint32_t ra, bool from_end, bool build,
                                             uint32_t& okops, uint32_t& lenor, uint32_t& tq_out, uint32_t& tr_out) {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t tq = 0, tr = 0;
        if (!build) {
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int u = 0; u < kSegVec; u++) {
                uint32_t au, bu;
                word_sums(cv[u], o0 + 256u * (uint32_t)u + 4u * lane, au, bu, okops, lenor);
                a += au; b += bu;
            }
            tq = lane_valu(wave_incl_scan(a), 63); tr = lane_valu(wave_incl_scan(b), 63);
        } else {
            const uint32_t q0 = from_end ? 0u : qa, r0 = from_end ? 0u : ra;
            wave_sync();
#pragma unroll
            for (int u = 0; u < kSegVec; u++) {
                if (o0 + 256u * (uint32_t)u < ncig) {
                    const uint32_t i = o0 + 256u * (uint32_t)u + 4u * lane;
                    uint32_t a, b;
                    word_sums(cv[u], i, a, b, okops, lenor);
                    const uint32_t iq = wave_incl_scan(a), ir = wave_incl_scan(b);
                    if (i < ncig) { S.cq[64u * (uint32_t)u + lane] = q0 + tq + iq - a; S.cr[64u * (uint32_t)u + lane] = r0 + tr + ir - b; }
                    tq = uniu(tq + lane_valu(iq, 63)); tr = uniu(tr + lane_valu(ir, 63));
                }
            }
            const uint32_t nck = (min(kSegOps, ncig - o0) + 3u) >> 2;
            uint32_t qb = q0, rb = r0;
            if (from_end) {
                qb = qa - tq; rb = ra - tr;
                wave_sync();
#pragma unroll
                for (int u = 0; u < kSegVec; u++) {
                    const uint32_t c = 64u * (uint32_t)u + lane;
                    if (c < nck) { S.cq[c] += qb; S.cr[c] += rb; }
                }
            }
            for (uint32_t c = nck + lane; c < kSegCk; c += 64u) S.cq[c] = kStreamInf;   // (the bound, up to the table's end: search_le_padded)
            if (lane == 0u) S.cq[kSegCk] = kStreamInf;
            c_o0 = o0; c_n = nck; Q_lo = qb; Q_hi = qb + tq; R_lo = rb; R_hi = rb + tr;
            wave_sync();
        }
        tq_out = tq; tr_out = tr;
    }
#endif
    // the segment that holds query position qf (< q_total): the neighbour in the direction the tokens walk, segment by segment;
    // with no table yet, from the end of the CIGAR the walk starts at
    __device__ __forceinline__ void ensure_cig(uint32_t qf) {
        while (c_n == 0u || qf - Q_lo >= Q_hi - Q_lo) {
            uint32_t o0, qa, ra;
            bool from_end;
            if (c_n == 0u) {
                from_end = rev != 0;
                o0 = rev ? ((ncig - 1u) / seg_ops()) * seg_ops() : 0u;
                qa = rev ? q_total : 0u; ra = rev ? r_total : 0u;
            } else if (qf >= Q_hi) {
                o0 = c_o0 + seg_ops(); from_end = false; qa = Q_hi; ra = R_hi;
            } else {
                o0 = c_o0 - seg_ops(); from_end = true; qa = Q_lo; ra = R_lo;
            }
            if (o0 >= ncig) { err = MM_E_QOVER; break; }   // (cannot happen: the totals are this table's own sums)
            uint4 cv[kSegVec];
            uint32_t okops = 0xFFFFFFFFu, lenor = 0, tq, tr;
            load_seg(cv, o0);
#ifdef MM_CK16
            if (wide) { seg_pass_wide(cv, o0, qa, ra, from_end); continue; }
#endif
            seg_pass(cv, o0, qa, ra, from_end, true, okops, lenor, tq, tr);
        }
    }

    __device__ __forceinline__ void side_append(int32_t spos, uint32_t ins_off, int is_mod, int code) {
        const int tid = tid_cur;
        const int64_t ref_base = ref_base_g;   // (round 3 fetched both again here, two dependent loads a round: rare then, every round of a --insertions run)
        const int hpk = kIns ? hp : -1;
        unsigned long long key;
        const bool keyed = (kIns && side_fast) ? (key = side_kbase + ((unsigned long long)(uint32_t)spos << 28) + (((uint32_t)code << 21) | ((ins_off & 0xFFFFu) << 5)), true)
                                               : side_key(ref_base + spos, rev, code, ins_off, hpk, key);
        if (keyed) {
            if (!kIns) { if (side_insert(p.stab, p.smask, p.scur, key, is_mod ? 0x100000001ull : 1ull)) err = MM_E_SIDEFULL; return; }
            // (the lanes that are here together: those of the round with a side update that has a 64-bit key)
            const uint64_t m = __ballot(1);
            const uint32_t cnt = (uint32_t)__popcll(m), mine = (uint32_t)__popcll(m & lanemask_lt());
            const int leader = __ffsll((unsigned long long)m) - 1;
            const uint32_t region = uniu(((uint32_t)blockIdx.x * 4u + ((uint32_t)threadIdx.x >> 6)) & (kSideRegions - 1u));
            unsigned long long* const list = p.stab + 2ull * (unsigned long long)region * p.smask;
            wave_sync();   // (what the leader of the lanes that were here last wrote: a fence, no instruction -- the lanes here are not all of the wave's)
            uint32_t at = S.sres_at;
            const uint32_t end = S.sres_end;
            if (end - at < cnt) {
                const unsigned long long kf = ((unsigned long long)S.sres_khi << 32) | S.sres_klo;
                for (uint32_t i = at + mine; i < end; i += cnt)
                    if ((unsigned long long)i < p.smask) { ulonglong2 z; z.x = kf; z.y = 0ull; *reinterpret_cast<ulonglong2*>(list + 2ull * i) = z; }
                unsigned int base = 0;
                if (lane_id() == leader) base = atomicAdd(p.scur + region * kSideCurStride, kSideChunk);
                at = (uint32_t)__shfl((int)base, leader, 64);
                if (lane_id() == leader) { S.sres_end = at + kSideChunk; S.sres_klo = (uint32_t)key; S.sres_khi = (uint32_t)(key >> 32); }
            }
            const unsigned long long idx = (unsigned long long)at + mine;
            if ((uint32_t)idx >= (uint32_t)p.smask) err = MM_E_SIDEFULL;   // (a region holds fewer than 2^32 records)
            else { ulonglong2 rec; rec.x = key; rec.y = is_mod ? 0x100000001ull : 1ull; *reinterpret_cast<ulonglong2*>(list + 2ull * idx) = rec; }
            if (lane_id() == leader) S.sres_at = at + cnt;
            wave_sync();
            return;
        }
        uint64_t m = __ballot(1);
        int leader = __ffsll((unsigned long long)m) - 1;
        unsigned long long base = 0;
        if (lane_id() == leader) base = atomicAdd(p.side_count, (unsigned long long)__popcll(m));
        base = __shfl(base, leader, 64);
        unsigned long long idx = base + __popcll(m & lanemask_lt());
        if (idx < p.side_cap) {
            SideRec r;
            r.tid = tid; r.pos = spos; r.ins_off = (uint16_t)ins_off; r.strand = (uint8_t)rev;
            r.is_mod = (uint8_t)is_mod; r.code = (int16_t)code; r.hp = (int16_t)hpk;
            p.side[idx] = r;
        } else {
            err = MM_E_SIDEFULL;
        }
    }

#ifndef MM_SIDE_CHUNK
#define MM_SIDE_CHUNK 256
#endif
    static constexpr uint32_t kSideChunk = MM_SIDE_CHUNK;
    // the unused records of the wavefront's last chunk (the kernel's end; wave-uniform)
    __device__ __forceinline__ void side_fill_rest() {
        wave_sync();
        const uint32_t at = S.sres_at, end = S.sres_end;
        if (at == end) return;
        const unsigned long long kf = ((unsigned long long)S.sres_khi << 32) | S.sres_klo;
        const uint32_t region = uniu(((uint32_t)blockIdx.x * 4u + ((uint32_t)threadIdx.x >> 6)) & (kSideRegions - 1u));
        unsigned long long* const list = p.stab + 2ull * (unsigned long long)region * p.smask;
        for (uint32_t i = at + (uint32_t)lane_id(); i < end; i += 64u)
            if ((unsigned long long)i < p.smask) { ulonglong2 z; z.x = kf; z.y = 0ull; *reinterpret_cast<ulonglong2*>(list + 2ull * i) = z; }
    }

    __device__ __forceinline__ void flush_pending() {
        if (pend != kNoPend) {
            // (the first code's slot is wave-uniform: the lane keeps one word, not an address and an increment)
            const uint32_t slot = ((ci0 >> 24) & 127u) - 1u;
            // (one code a class -- the usual run -- needs no 64-bit multiply for the counter's place: that instruction issues at a quarter of the rate)
            const uint64_t at = gnp == 1u ? (uint64_t)(pend >> 1) + slot : (uint64_t)(pend >> 1) * gnp + slot;
            atomicAdd(gcb + at, (pend & 1u) ? 0x100000001ull : 1ull);
        }
        pend = kNoPend;
    }

    // ------------------------------------------------------------------ one round: n rising ranks, one a lane (n <= 64)
    // expl: the lane's rank is a listed token (its ML bytes are those of token kidx of the group), else an unlisted base of a
    // '.' group (mod.c:1206-1287, :1289-1365: called, not modified, no ML byte).  Returns the number of ranks done: those whose
    // block lies in the directory segment and whose op lies in the CIGAR segment (0: a rank beyond the read's last base of the
    // class, or something else wrong with the read); the caller comes again with the rest.
    __device__ __forceinline__ uint32_t round_core(uint32_t rho_in, uint32_t n, bool expl, uint32_t kidx) {
        const uint32_t lane = (uint32_t)lane_id();
        const bool lv = lane < n;
        const uint32_t rho = lv ? rho_in : 0xFFFFFFFFu;
        const uint32_t rho_0 = lane_valu(rho, 0);
        KFT_LAP(2);
        // the round's ML bytes are requested before anything else is waited for
        // (32 bits: a BAM record is shorter than 2^29 bytes, so a read's groups have fewer than 2^28 tokens in all; times four codes: below 2^30)
        const uint32_t mi0 = ml_start + kidx * ((kTwinOK && tw) ? 1u : (uint32_t)ncg);
        const uint32_t ml0 = (lv && expl && mi0 < ml_len) ? ml[mi0] : 0u;
        flush_pending();   // the round before's updates, behind this round's loads
        // the directory segment that holds the round's first rank (ranks rise inside a group; a new group starts over)
#ifdef MM_ABL_NODIR   // (diagnostic: a directory that claims every block holds 8 members, never built)
        if (d_n == 0u) { d_t0 = 0; d_n = nblk < kSegBlocks ? nblk : kSegBlocks; S_lo = 0; S_hi = 0x7FFFFFFFu; }
#else
        if (d_n == 0u || rho_0 < S_lo) build_dir(0u, 0u);
        while (rho_0 >= S_hi && d_t0 + d_n < nblk) build_dir(d_t0 + d_n, S_hi);
#endif
        const uint32_t n1 = leading_ones(__ballot(lv && rho < S_hi));   // ranks whose block is in the segment
        KFT_LAP(3);
        uint32_t n_done = 0;
        if (n1 > 0u) {
            // rank -> four blocks: largest g with cw[g] <= rho (cw[0] = S_lo <= rho_0) -> the block among them by its members' counts
            const bool act = lane < n1;
            const uint32_t dn4 = (d_n + 3u) >> 2;   // (entries of `cw` in use, the running total behind them: a read of 2 kb needs four steps, not seven)
            const uint32_t g4 = !act ? 0u : (dn4 < 16u ? search_le_padded<16>(S.cw, rho) : (dn4 < 64u ? search_le_padded<64>(S.cw, rho) : search_le_padded<kSegBlocks / 4>(S.cw, rho)));
            const uint4 mb = *reinterpret_cast<const uint4*>(S.dm + 4u * g4);
            const uint32_t rg = rho - S.cw[g4];
            const uint32_t e1 = __popc(mb.x), e2 = e1 + __popc(mb.y), e3 = e2 + __popc(mb.z);
            const uint32_t wb = (rg >= e1 ? 1u : 0u) + (rg >= e2 ? 1u : 0u) + (rg >= e3 ? 1u : 0u);
            const uint32_t mk = wb == 0u ? mb.x : (wb == 1u ? mb.y : (wb == 2u ? mb.z : mb.w));
            const uint32_t j = 4u * g4 + wb;
            const uint32_t r_in = rg - (wb == 0u ? 0u : (wb == 1u ? e1 : (wb == 2u ? e2 : e3))), c_b = __popc(mk);
            const uint32_t t = d_t0 + j, blk = rev ? nblk - 1u - t : t;
            const uint32_t kk = rev ? c_b - 1u - r_in : r_in;
            // the base inside the block from the segment's member bits; its code is the class's own (C 2, G 4, T 8, N 15) -- only the
            // fifth class (A and the ambiguity codes) has to look at the read's bases again
            const uint32_t q = act ? blk * 32u + select_bit(mk, kk) : 0u;
            uint32_t code = cpat & 15u;
            if (cpat == 0u) {
                const uint32_t raw = act ? reinterpret_cast<const uint32_t*>(sq)[q >> 3] : 0u;
                code = (base_order(raw) >> (4u * (q & 7u))) & 15u;
            }
            // BAM position -> query position of the CIGAR: get_aln walks a reverse read's ops back to front from position 0 of
            // the original orientation (mod.c:813-860), so its aligned part lies at BAM positions [L - q_total, L); positions
            // outside the CIGAR's query length have no call
            const uint32_t qi = q - q_shift;
            const bool live = act && qi < q_total;
            const uint64_t lm = __ballot(live);
            KFT_LAP(4);
            if (lm) ensure_cig(lane_valu(qi, __ffsll((unsigned long long)lm) - 1));
            const bool in_seg = qi - Q_lo < Q_hi - Q_lo;
            n_done = leading_ones(__ballot(act && (!live || in_seg)));
            if (__ballot(err != 0)) n_done = 0;
            KFT_LAP(5);
            const bool fin = lane < n_done && live;
#ifdef MM_ABL_NOFINISH
            if (false) {
#else
            if (__ballot(fin)) {
#endif
                // query position -> checkpoint (largest c with cq[c] <= qi) -> op: the checkpoint's four ops are walked
#ifdef MM_CK16
                uint32_t qk, ck, a0, b0;
                if (!wide) {
                    qk = qi - Q_lo;   // (the table's positions count from the segment's first op)
                    // (as many steps as the table in use needs: a HiFi read's 40 ops are ten checkpoints -- four dependent LDS reads, not eight; the entries
                    // behind the ones in use hold the bound up to the table's end, so any power of two that covers c_n will do)
                    ck = !fin ? 0u : (c_n <= 16u ? search_le_padded<16>(S.cq, qk) : (c_n <= 64u ? search_le_padded<64>(S.cq, qk) : search_le_padded<kSegCk>(S.cq, qk)));
                    a0 = S.cq[ck]; b0 = R_lo + S.cr[ck];
                } else {
                    qk = qi;
                    ck = fin ? search_le_padded<kWideCk>(S.cq32, qi) : 0u;
                    a0 = S.cq32[ck]; b0 = S.cr32[ck];
                }
#else
                const uint32_t qk = qi;
                const uint32_t ck = fin ? search_le_padded<kSegCk>(S.cq, qi) : 0u;
                const uint32_t a0 = S.cq[ck], b0 = S.cr[ck];
#endif
                const uint4 ov = fin ? *reinterpret_cast<const uint4*>(cg + c_o0 + 4u * ck) : make_uint4(0, 0, 0, 0);
                const uint32_t o0 = ov.x & 15u, o1 = ov.y & 15u, o2 = ov.z & 15u, o3 = ov.w & 15u;
                const uint32_t l0 = ov.x >> 4, l1 = ov.y >> 4, l2 = ov.z >> 4;
                const uint32_t a1 = a0 + (l0 & op_mask(0x193u, o0)), a2 = a1 + (l1 & op_mask(0x193u, o1)), a3 = a2 + (l2 & op_mask(0x193u, o2));
                const uint32_t b1 = b0 + (l0 & op_mask(0x18Du, o0)), b2 = b1 + (l1 & op_mask(0x18Du, o1)), b3 = b2 + (l2 & op_mask(0x18Du, o2));
                // the op that holds qi: the first whose end lies behind it (ops that consume no query position are passed over)
                const uint32_t kq = (qk >= a1 ? 1u : 0u) + (qk >= a2 ? 1u : 0u) + (qk >= a3 ? 1u : 0u);
                const uint32_t op = kq == 0u ? o0 : (kq == 1u ? o1 : (kq == 2u ? o2 : o3));
                const uint32_t a_s = kq == 0u ? a0 : (kq == 1u ? a1 : (kq == 2u ? a2 : a3));
                const uint32_t b_s = kq == 0u ? b0 : (kq == 1u ? b1 : (kq == 2u ? b2 : b3));
                const uint32_t e = qk - a_s;
                bool call = fin && ((0x181u >> op) & 1u);
                int32_t ref_pos = pos + (int32_t)(b_s + e);
                uint32_t ins_off = 0;
                if (kIns && p.insertions && fin && op == 1u) {
                    // a base inside an insertion: a call on the reference base left of it, with its 1-based offset truncated like
                    // make_key's uint16 (mod.c:428, :864-874)
                    ins_off = (e + 1u) & 0xFFFFu;
                    ref_pos = pos + (int32_t)b_s - 1;
                    call = ref_pos >= 0;
                }
                if (kIns && kDot && p.insertions && rev && !expl) {
                    // ... but an IMPLICIT call of a reverse read that is not on an aligned base (inside an insertion, or in a soft
                    // clip) takes the anchor of the base at the mirrored position, if that one lies inside an insertion, and no call
                    // else; its offset stays its own (0 in a clip).  The whole CIGAR is in the table here (dot_ins_far).
                    const bool odd = fin && !((0x181u >> op) & 1u);
                    if (__ballot(odd)) {
                        const uint32_t q2 = L - 1u - q, qi2 = q2 - q_shift;
                        const bool in2 = odd && qi2 < q_total;
#ifdef MM_CK16
                        uint32_t qk2, ck2, a20, b20;
                        if (!wide) {
                            qk2 = qi2 - Q_lo;
                            ck2 = in2 ? search_le_padded<kSegCk>(S.cq, qk2) : 0u;
                            a20 = S.cq[ck2]; b20 = R_lo + S.cr[ck2];
                        } else {
                            qk2 = qi2;
                            ck2 = in2 ? search_le_padded<kWideCk>(S.cq32, qi2) : 0u;
                            a20 = S.cq32[ck2]; b20 = S.cr32[ck2];
                        }
#else
                        const uint32_t qk2 = qi2;
                        const uint32_t ck2 = in2 ? search_le_padded<kSegCk>(S.cq, qi2) : 0u;
                        const uint32_t a20 = S.cq[ck2], b20 = S.cr[ck2];
#endif
                        const uint4 ov2 = in2 ? *reinterpret_cast<const uint4*>(cg + c_o0 + 4u * ck2) : make_uint4(0, 0, 0, 0);
                        const uint32_t p0 = ov2.x & 15u, p1 = ov2.y & 15u, p2 = ov2.z & 15u, p3 = ov2.w & 15u;
                        const uint32_t m0 = ov2.x >> 4, m1 = ov2.y >> 4, m2 = ov2.z >> 4;
                        const uint32_t a21 = a20 + (m0 & op_mask(0x193u, p0)), a22 = a21 + (m1 & op_mask(0x193u, p1)), a23 = a22 + (m2 & op_mask(0x193u, p2));
                        const uint32_t b21 = b20 + (m0 & op_mask(0x18Du, p0)), b22 = b21 + (m1 & op_mask(0x18Du, p1)), b23 = b22 + (m2 & op_mask(0x18Du, p2));
                        const uint32_t k2 = (qk2 >= a21 ? 1u : 0u) + (qk2 >= a22 ? 1u : 0u) + (qk2 >= a23 ? 1u : 0u);
                        const uint32_t op2 = k2 == 0u ? p0 : (k2 == 1u ? p1 : (k2 == 2u ? p2 : p3));
                        const uint32_t bs2 = k2 == 0u ? b20 : (k2 == 1u ? b21 : (k2 == 2u ? b22 : b23));
                        if (odd) {
                            ref_pos = pos + (int32_t)bs2 - 1;
                            call = in2 && op2 == 1u && ref_pos >= 0;
                            ins_off = op == 1u ? ins_off : 0u;
                        }
                    }
                }
                uint32_t w = 0;
                // Is the position a site of the group's context class, and which one: the class's site word (32 positions: which are
                // sites, how many sites lie in front) answers both.  A dense class (context `*`; --insertions, where neither context
                // nor base are looked at for ANY call, mod.c:1167-1172) counts every position: nothing to look up.
                bool in_ctx = true;
                uint32_t site = (uint32_t)ref_pos;
                if (call && gsite != nullptr) {
                    const int64_t gp = ref_base_g + (int64_t)(uint32_t)ref_pos;
                    uint2 sw;
#ifndef MM_ABL_NOREF
                    if (std::is_same<RefWord, RefNib>::value) {
                        // four-bit reference words (one requested mod): the site word carries its 32 positions' bases as well -- ONE
                        // 16-byte gather answers context, rank and base
                        const uint4 s4 = *reinterpret_cast<const uint4*>(gsite + 2 * (gp >> 5));
                        sw = make_uint2(s4.x, s4.y);
                        const uint32_t bsel = ((uint32_t)gp & 31u) * 2u;
                        w = 1u << (((bsel < 32u ? s4.z >> bsel : s4.w >> (bsel - 32u))) & 3u);
                    } else {
                        sw = gsite[gp >> 5];
                        w = RefLoad<RefWord>::at(rwb, (int64_t)(uint32_t)ref_pos);
                    }
#else
                    sw = make_uint2(0xFFFFFFFFu, (uint32_t)(gp >> 5) * 32u);
                    w = 0xFFFFFFE0u | (code & 31u) | ((uint32_t)ref_pos & 0u);
#endif
                    in_ctx = (sw.x >> ((uint32_t)gp & 31u)) & 1u;
                    site = site_rank(sw, (uint32_t)gp & 31u);
                    if (kStats) st_look++;
                }
                call = call && in_ctx;
                if (kView) {
                    // `minimod view`: a call that passes the tests is a record (add_view_entry, mod.c:931-946: no threshold, the ML byte
                    // itself; implicit calls carry probability 0, mod.c:1281-1283, :1361-1363).  Every lane takes part in the append:
                    // the wavefront numbers the read's records in the order it makes them.
                    const uint32_t refcode = w & 31u;
                    for (int m = 0; m < ncg; m++) {
                        const int ci = gcode_at(m);
                        if (ci < 0) continue;
                        const uint32_t cinfo = cinfo_at(m);
                        bool emit = call && (gsite == nullptr || ((cinfo >> 18) & 1u) || refcode == code);
                        uint32_t prob = 0;
                        if (expl) {
                            const uint32_t ml_idx = (kTwinOK && tw) ? ml_start + kidx + (uint32_t)m * tw : ml_start + kidx * (uint32_t)ncg + (uint32_t)m;
                            if (emit && ml_idx >= ml_len) { err = MM_E_MLIDX; emit = false; }
                            if (emit) prob = m == 0 ? ml0 : (uint32_t)ml[ml_idx];
                            if (kStats && emit) st_ml++;
                        }
                        v_seq = uniu(v_seq + view_append_seq(p, v_region, (uint32_t)ridx_cur, emit, (uint32_t)(ref_pos - pos + 1), rev ? L - 1u - q : q, 0u, (uint32_t)ci,
                                                             v_gord + ((kTwinOK && tw) ? (uint32_t)m : 0u), expl ? 0u : 1u, prob, v_seq | v_sorted));
                    }
                } else if (call) {
                    const uint32_t refcode = w & 31u;
                    for (int m = 0; m < ncg; m++) {
                        const int ci = gcode_at(m);
                        if (ci < 0) continue;
                        const uint32_t cinfo = cinfo_at(m);
                        const int t_hi = (int)(cinfo & 511u), t_lo = (int)((cinfo >> 9) & 511u) - 1;
                        if (gsite != nullptr && !(((cinfo >> 18) & 1u) || refcode == code)) continue;   // the base test (mod.c:1163-1164); a dense class has none
                        int is_mod = 0;
                        if (expl) {
                            const uint32_t ml_idx = (kTwinOK && tw) ? ml_start + kidx + (uint32_t)m * tw : ml_start + kidx * (uint32_t)ncg + (uint32_t)m;
                            if (ml_idx >= ml_len) { err = MM_E_MLIDX; continue; }   // (not `break`: a divergent exit would make m, and all that
                                                                                     // hangs on it -- the code's table word, its slot, the 64-bit
                                                                                     // counter offset -- vector values; the read fails either way)
                            const int mv = m == 0 ? (int)ml0 : (int)ml[ml_idx];
                            if (kStats) st_ml++;
                            if (mv >= t_hi) is_mod = 1;
                            else if (mv <= t_lo) is_mod = 0;
                            else continue;
                        }
                        const int slot = (int)((cinfo >> 24) & 127u) - 1;
                        const bool dense = gcb != nullptr && slot >= 0 && (uint32_t)ref_pos - seg_lo32 < seg_len32 && (!kIns || (ins_off == 0u && hpi >= 0));
                        if (dense) {
#ifndef MM_ABL_NOATOMIC
                            if (m == 0) pend = (site << 1) | (uint32_t)is_mod;
                            else atomicAdd(gcb + ((uint64_t)site * gnp + (uint32_t)slot), is_mod ? 0x100000001ull : 1ull);
#else
                            if (ref_pos == -12345 && is_mod) atomicAdd(gcb, 1ull);
#endif
                            if (kStats) st_dense++;
                        } else {
                            side_append(ref_pos, ins_off, is_mod, ci);
                            if (kStats) st_side++;
                        }
                    }
                }
            }
            KFT_LAP(6);
            if (n_done > 0u) rank_ok = max(rank_ok, lane_valu(rho, (int)(n_done - 1u)) + 1u);
        }
        return n_done;
    }

    // ------------------------------------------------------------------ one group's skip list [lstart, ...;)
    // returns 0, or 2 when the read has to go to the fused kernel (an input error); *ntok = tokens of the group
    __device__ __forceinline__ int run_group(uint32_t mpos, uint32_t lstart, bool wanted, uint32_t& ntok) {
        cpos = mpos; skip0 = lstart - mpos; prev_delim = true; closed = false; bad_text = false;
        pend = kNoPend;
        qhead = 0; qn = 0; kdone = 0; Rcarry = 0; ntok_parsed = 0;
        // a read of several segments starts the group at the table's first segment again (a short read's tables hold all of it)
        if (nblk > kSegBlocks) d_n = 0;
        if (ncig > seg_ops() && wanted) c_n = 0;
        if (staged_at != cpos) fetch_chunk(cpos);
        int st = 0;
        // One loop, one call of round_core: the ranks of a round come from the ring (a batch of up to 64 tokens), or -- in a
        // '.' group, where every base of the class that the list skips is a call too (called, not modified) -- from the batch's
        // tokens AND the bases of the gaps in front of them, ONE rising sequence gap, token, gap, token, ... taken 64 elements at
        // a time, and behind the last token from the rest of the read's bases (mod.c:1206-1287, :1289-1365).
        const bool dot = kDot && wanted && dot_group;
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t prev_last = 0xFFFFFFFFu;   // rank of the last token so far (-1: none)
        uint32_t n = 0, E = 0, e_done = 0;   // the batch: tokens, elements, elements done
        uint32_t r0 = 0, nb_all = 0;         // the tail: next rank, the read's bases of the class
        bool tail = false;
        for (;;) {
            if (!tail && e_done == E) {
                // the batch is done: its tokens leave the ring, the next ones come in
                if (n != 0u) {
                    if (dot) prev_last = uniu(S.tok[(qhead + n - 1u) & (kStreamRing - 1u)]);
                    qhead = uniu((qhead + n) & (kStreamRing - 1u)); qn = uniu(qn - n); kdone = uniu(kdone + n);
                    n = 0;
                }
#ifdef MM_ABL_NOPARSE   // (diagnostic: no list is ever parsed -- and so no round made; what is left is the record, the CIGAR pass, the headers, the twin comparison)
                closed = true;
#endif
                while (qn < 64u && !closed && !bad_text) parse_chunk(wanted);
                if (bad_text) { st = 2; break; }
                e_done = 0; E = 0;
                if (qn == 0u) {
                    if (!dot) break;
                    tail = true; nb_all = count_all(); r0 = prev_last + 1u;
                } else {
                    n = qn < 64u ? qn : 64u;
                    E = n;
                    if (dot) {
                        const uint32_t rho = lane < n ? S.tok[(qhead + lane) & (kStreamRing - 1u)] : 0u;
                        uint32_t prev = (uint32_t)__shfl_up((int)rho, 1, 64);
                        if (lane == 0u) prev = prev_last;
                        const uint32_t gap = lane < n ? rho - prev - 1u : 0u;   // (ranks rise strictly; -1 wraps to the right thing)
                        const uint32_t pin = wave_incl_scan(lane < n ? gap + 1u : 0u);
                        E = lane_valu(pin, 63);
                        wave_sync();
                        S.gap_p[lane] = lane < n ? pin - gap - 1u : E;   // the first element of token `lane`: its gap's first base
                        S.gap_r[lane] = prev + 1u;                       // ... whose rank is this
                        if (lane == 0u) S.gap_p[64] = E;
                        wave_sync();
                    }
                }
            }
            uint32_t rank = 0, cnt = 0, kidx = 0;
            bool is_tok = true;
            if (tail) {
                if (r0 >= nb_all) break;
                cnt = nb_all - r0 < 64u ? nb_all - r0 : 64u;
                rank = r0 + lane; is_tok = false;
            } else if (dot) {
                cnt = E - e_done < 64u ? E - e_done : 64u;
                const uint32_t e = e_done + lane;
                const uint32_t i = lane < cnt ? search_le<64>(S.gap_p, e, n) : 0u;   // the token whose gap, or self, this element is
                rank = S.gap_r[i] + (e - S.gap_p[i]);
                is_tok = rank == S.tok[(qhead + i) & (kStreamRing - 1u)];
                kidx = kdone + i;
            } else {
                cnt = E - e_done;
                rank = S.tok[(qhead + e_done + lane) & (kStreamRing - 1u)];
                kidx = kdone + e_done + lane;
            }
#ifdef MM_ABL_NOROUNDS   // (diagnostic builds, with tools/valu.sh: what the rounds cost -- their results are of course wrong)
            const uint32_t nd = cnt;
#else
            const uint32_t nd = round_core(rank, cnt, is_tok, kidx);
#endif
            const uint64_t eb = __ballot(err != 0);
            if (nd == 0u || eb) { st = 2; break; }
            if (tail) r0 = uniu(r0 + nd);
            else e_done = uniu(e_done + nd);
        }
        flush_pending();
        ntok = ntok_parsed;
        return st;
    }

    // Are the skip lists [a, a + len) and [b, b + len) of the MM string the same text, and is that text nothing but
    // ",digits,digits,...,digits"?  Then its tokens are its commas (returned; 0: not the same, or not that plain).
    __device__ __forceinline__ uint32_t twin_lists(uint32_t a, uint32_t b, uint32_t len) const {
#ifdef MM_ABL_NOTWIN   // (diagnostic: the lists are "the same" unseen)
        return len / 3u + 1u;
#endif
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t bad = 0, commas = 0;
        // 1024 characters a trip: the twelve loads of a trip are requested together (a list of a 15 kb read is one trip)
        for (uint32_t off0 = 0; off0 < len; off0 += 1024u) {
            uint32_t va[4], vb[4], vp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t o = off0 + 256u * (uint32_t)u + 4u * lane;
                // (plain loads: what lies behind a list's end -- the text goes on, or the pool's 64 bytes of slack -- is masked out by `vm` below;
                // mm_dword's own replacing of those bytes was seven instructions a load, twelve loads a trip)
                va[u] = 0u; vb[u] = 0u; vp[u] = 0u;
                if (o < len) {
                    __builtin_memcpy(&va[u], mm + a + o, 4); __builtin_memcpy(&vb[u], mm + b + o, 4);
                    __builtin_memcpy(&vp[u], mm + a + o - 1u, 4);                       // the same four characters' left neighbours (a >= 1: a header is in front)
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (off0 + 256u * (uint32_t)u >= len) continue;   // (wave-uniform: a 350-character list is two of the trip's four quarters, not four)
                const uint32_t o = off0 + 256u * (uint32_t)u + 4u * lane;
                const uint32_t wa = va[u], wb = vb[u], wp = vp[u];
                const uint32_t vm = o >= len ? 0u : (len - o >= 4u ? 0x80808080u : (0x80808080u & ((1u << (8u * (len - o))) - 1u)));   // bytes inside the list
                const uint32_t ya = wa ^ 0x2C2C2C2Cu, yp = wp ^ 0x2C2C2C2Cu, yd = wa ^ 0x30303030u;
                const uint32_t ca = ~(((ya & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | ya) & 0x80808080u;   // commas
                const uint32_t cp = ~(((yp & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | yp) & 0x80808080u;   // commas among the left neighbours
                const uint32_t nd = (((yd & 0x7F7F7F7Fu) + 0x76767676u) | yd) & 0x80808080u;     // not a digit
                bad |= ((wa ^ wb) & (vm | (vm - (vm >> 7)))) | ((nd & ~ca) & vm) | (ca & cp & vm);   // (0x80 -> 0xFF a byte without a multiply)
                if (o <= len - 1u && len - 1u < o + 4u) bad |= ca & (0x80u << (8u * (len - 1u - o)));   // the last character is a digit
                if (o == 0u) bad |= (~ca) & 0x80u;                                                    // the first one a comma
                commas += (uint32_t)__popc(ca & vm);
            }
        }
        if (__ballot(bad != 0u)) return 0u;
        return lane_valu(wave_incl_scan(commas), 63);
    }

    // ------------------------------------------------------------------ one read: 0 done, 1 -> tile pipeline, 2 -> fused kernel
    __device__ __forceinline__ int run(int ridx) {
        const uint32_t lane = (uint32_t)lane_id();
#ifdef MM_STREAM_TIMING
        ft0 = __builtin_amdgcn_s_memrealtime();
#endif
        const mm_read_t rd = scalar_load(p.reads + ridx);
        err = 0;
        const int tid = uni(rd.tid);
        pos = uni(rd.pos); ridx_cur = ridx;
        L = uniu(rd.l_qseq); ncig = uniu(rd.n_cigar); mlen = uniu(rd.mm_len); ml_len = uniu(rd.ml_len);
        rev = (uni(rd.flag) & 0x10) ? 1 : 0;
        if (kIns) {
            const int hpt = (int)(uint32_t)uni((int)rd.hp);
            hp = p.haplotypes ? hpt : -1;
            hpi = p.haplotypes ? (hpt < p.n_hp ? hpt : -1) : 0;
        } else { hp = -1; hpi = 0; }
        mm = p.mm + rd.mm_off; ml = p.ml + rd.ml_off;
        sq = reinterpret_cast<const uint4*>(p.seq + rd.seq_off);
        cg = p.cigar + rd.cigar_off;
        nblk = (L + 31u) >> 5;
        // Everything that only needs the record is requested before anything is waited for: the CIGAR (its first segment),
        // the first characters of the MM string (the first group's header and the start of its list), the contig's entries
        uint4 cv[kSegVec];
        load_seg(cv, 0u);
        staged_at = 0xFFFFFFFFu; skip0 = 0;
        fetch_chunk(0u);
        const bool tid_ok = tid >= 0 && tid < p.n_contigs;
        const int tid_c = tid_ok ? tid : 0;
        const int64_t ref_base = scalar_load(p.ref_base + tid_c);
        const int64_t ctg_len = scalar_load(p.ctg_len + tid_c);
        const int64_t seg_begin = scalar_load(p.seg_begin + tid_c), seg_len = scalar_load(p.seg_len + tid_c);
        int st = (tid_ok && ref_base >= 0 && L > 0u && ncig > 0u) ? 0 : 1;
        d_n = 0; c_n = 0; rank_ok = 0; v_seq = 0; v_sorted = 0;
#ifdef MM_ABL_NORUN   // (diagnostic, with tools/valu.sh: the hand-out and the record alone)
        return (int)(lane_valu(cv[0].x, 0) == 0xDEADBEEFu && nx_w0 == 0x12345u) + (int)(ctg_len + seg_begin + seg_len == -77) + st * 0;
#endif
        if (st == 0) {
            // the whole CIGAR once (get_aln walks it before anything else, mod.c:776-881): totals, the checks reduced to what a
            // clean record passes outright (anything else is the tile pipeline's to judge op by op) -- and, on the way, the
            // checkpoint table of the segment the tokens start in: the first one, or the last one for a reverse read
            uint32_t okops = 0xFFFFFFFFu, lenor = 0;
            uint32_t run_q = 0, run_r = 0;
            bool over = false;
            const uint32_t want_o0 = rev ? ((ncig - 1u) / kSegOps) * kSegOps : 0u;
#ifdef MM_CK16
            wide = false;
#endif
#ifdef MM_ABL_NOVALIDATE
            for (uint32_t i0 = 0; i0 < 0u; i0 += kSegOps) {
#else
            for (uint32_t i0 = 0; i0 < ncig; i0 += kSegOps) {
#endif
                if (i0 > 0u) load_seg(cv, i0);
                uint32_t tq, tr;
                seg_pass(cv, i0, run_q, run_r, false, i0 == want_o0, okops, lenor, tq, tr);
                // (no op is as long as 2^20 in a read that stays here, so a segment's sums stay below 2^30; the running totals are
                // looked at after every segment)
                run_q += tq; run_r += tr;
                over = over || run_q >= (1u << 28) || run_r >= (1u << 28);
#ifdef MM_CK16
                wide = wide || tq >= kCkInf || tr >= kCkInf;   // a segment's checkpoints do not fit 16 bits: the read's tables are the wide ones
#endif
            }
            const bool badop = __ballot(okops == 0u || lenor >= kStreamMaxLen) != 0ull;
            if (badop || over || run_q > L || pos < 0 || (int64_t)pos + (int64_t)run_r > ctg_len) { st = 1; c_n = 0; }
            q_total = run_q; r_total = run_r;
            q_shift = (rev && run_q < L) ? L - run_q : 0u;
#ifdef MM_CK16
            if (wide) c_n = 0;   // (the table the pass has just made wrapped: ensure_cig makes the wide one when the first call asks)
#endif
        }
        KFT_LAP(0);
        uint32_t ngrp = 0;
        if (st == 0) {
            // The group headers (mod.c:1003-1062): is this a read for this kernel, where are its lists, which codes do they carry.
            // A group's first kStreamChunk characters are staged in LDS: the header is read from there, and so is the group's
            // end when it lies that near.
            KA<RefWord, StreamLds> hp_(P, S);
            const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(S.mmw);
            uint32_t mpos = 0;
            int first_cls = -1;
            while (mpos < mlen && st == 0) {
                if (mpos != 0u) fetch_chunk(mpos);
                stage_chunk(mpos);
                const uint32_t ci = mpos + lane;
                const int ch = ci < mlen ? (int)mb8[lane] : 0;
                // the header ends at the first of , ; ? . (or the string's end); a flag belongs to it
                const uint64_t sb = __ballot(lane >= 1u && (ci >= mlen || ch == ',' || ch == ';' || ch == '?' || ch == '.'));
                const uint32_t e = sb ? (uint32_t)__ffsll((unsigned long long)sb) - 1u : 64u;
                const int ce = e < 64u ? lane_val(ch, (int)e) : 0;
                const uint32_t hlen = e + ((ce == '?' || ce == '.') ? 1u : 0u);
                const uint32_t slot = ngrp < kStreamMemo ? ngrp : 0u;
                const bool memo_hit = ngrp < kStreamMemo && hlen <= 16u && uniu(S.memo_len[slot]) == hlen &&
                                      !__ballot(lane < hlen && (uint32_t)ch != (uint32_t)S.memo_hdr[slot][lane & 15u]);
                uint32_t gflags, c01, c23, lstart, ci_w = 0;
                const int c0 = lane_val(ch, 0);
                if (memo_hit) {
                    gflags = uniu(S.memo_flags[slot]); c01 = uniu(S.memo_c01[slot]); c23 = uniu(S.memo_c23[slot]);
                    if (lane < 4u) ci_w = S.memo_ci[slot][lane];
                    lstart = mpos + hlen;
                    if (kIns && (gflags & 4u)) {   // a '.' group under --insertions / --haplotypes (round 4)
                        if (!kDot) { st = 1; saw_dot = true; }
                        else if (dot_ins_far()) st = 1;
                    }
                } else {
                    GroupHdr g = hp_.parse_header_ch(ch, mlen, mpos);
                    lstart = g.lstart;
                    gflags = 0; c01 = 0; c23 = 0;
                    if (g.herr || g.n > 4 || ngrp >= kStreamGroups) st = 1;
                    else {
                        if (g.modbase == 'N') st = 1;   // the tile pipeline has the direct groups
                        if (!kDot && g.flag == '.') { st = 1; saw_dot = true; }   // ... and, for this instantiation, the implicit calls
                        if (kDot && kIns && g.flag == '.' && dot_ins_far()) st = 1;
                        hp_.err = 0;
                        hp_.lookup_codes(g);
                        if (__ballot(hp_.err != 0)) st = 1;
                        const int16_t a0 = S.g_code[0], a1 = S.g_code[1], a2 = S.g_code[2], a3 = S.g_code[3];
                        const bool unwanted = a0 < 0 && a1 < 0 && a2 < 0 && a3 < 0;
                        gflags = (unwanted ? 64u : 0u) | (g.flag == '.' ? 4u : 0u) | ((uint32_t)g.n << 12);
                        c01 = (uint32_t)(uint16_t)a0 | ((uint32_t)(uint16_t)a1 << 16);
                        c23 = (uint32_t)(uint16_t)a2 | ((uint32_t)(uint16_t)a3 << 16);
                        // what a call needs of its code's table entries: t_hi | (t_lo + 1) << 9 | ctx_is_star << 18 | context class << 19 (five bits) |
                        // (slot of the code's counters among its context class's + 1, 0: no dense counters) << 24
                        const int cix = lane == 0u ? a0 : (lane == 1u ? a1 : (lane == 2u ? a2 : a3));
                        int kcls = -1;
                        if ((int)lane < g.n && lane < 4u && cix >= 0) {
                            const DevCode& dc = p.codes[cix];
                            const int req = dc.req, plane = dc.plane;
                            const DevMod& dm = p.mods[req];
                            kcls = p.cls_of_mod[req];
                            ci_w = (uint32_t)dm.t_hi | ((uint32_t)(dm.t_lo + 1) << 9) | (dm.ctx_is_star ? (1u << 18) : 0u) | ((uint32_t)p.cls_of_mod[req] << 19) |
                                   ((uint32_t)(plane >= 0 ? dc.slot + 1 : 0) << 24);
                        }
                        // the requested codes of a group share one context class here (one site word answers for all of them)
                        const uint64_t wl = __ballot(kcls >= 0);
                        const int cls0 = wl ? lane_val(kcls, __ffsll((unsigned long long)wl) - 1) : 0;
                        if (__ballot(kcls >= 0 && kcls != cls0)) st = 1;
                        gflags |= (uint32_t)cls0 << 16;
                        wave_sync();
                        if (st == 0 && ngrp < kStreamMemo && hlen <= 16u && lstart == mpos + hlen) {
                            if (lane < 16u) S.memo_hdr[slot][lane] = (uint8_t)ch;
                            if (lane < 4u) S.memo_ci[slot][lane] = ci_w;
                            if (lane == 0u) { S.memo_len[slot] = hlen; S.memo_flags[slot] = gflags; S.memo_c01[slot] = c01; S.memo_c23[slot] = c23; }
                        }
                    }
                }
                if (st == 0) {
                    const int mb = rev ? complement_char(c0 == 'U' ? 'T' : c0) : (c0 == 'U' ? 'T' : c0);
                    const int c = base_class_of_char(mb);
                    if (first_cls < 0) first_cls = c;
                    else if (c != first_cls) st = 1;
                    // the group's end: in the staged characters (from the list's start on), or further on in the string
                    const uint32_t skip = lstart - mpos;
                    uint32_t endp = 0xFFFFFFFFu;
#pragma unroll
                    for (int sc = 0; sc < (int)(kStreamChunk / 64); sc++) {
                        const uint32_t xs = mb8[64 * sc + (int)lane];
                        uint64_t sm = __ballot(xs == (uint32_t)';');
                        if (sc == 0) sm &= ~low_bits((int)skip);
                        if (sm && endp == 0xFFFFFFFFu) endp = mpos + 64u * (uint32_t)sc + (uint32_t)__ffsll((unsigned long long)sm) - 1u;
                    }
                    if (endp == 0xFFFFFFFFu) endp = find_semicolon(mm, mlen, mpos + kStreamChunk);
                    if (endp > mlen) endp = mlen;
                    wave_sync();
                    if (lane == 0u) {
                        S.g_mpos[ngrp] = mpos; S.g_lstart[ngrp] = lstart; S.g_flags[ngrp] = gflags; S.g_c01[ngrp] = c01; S.g_c23[ngrp] = c23;
                        S.g_end[ngrp] = endp;
                    }
                    if (lane < 4u) S.g_ci[ngrp][lane] = ci_w;
                    ngrp++;
                    mpos = endp + 1u;
                }
            }
            cpat = class_pattern(first_cls);
            wave_sync();
            if (kView) {
                uint32_t ncodes = 0, dots = 0;
                for (uint32_t gi = 0; gi < ngrp; gi++) {
                    const uint32_t f = uniu(S.g_flags[gi]), c01 = uniu(S.g_c01[gi]), c23 = uniu(S.g_c23[gi]);
                    if (f & 64u) continue;
                    dots |= f & 4u;
                    const uint32_t n = (f >> 12) & 7u;
                    ncodes += (n > 0u && (c01 & 0xFFFFu) != 0xFFFFu) + (n > 1u && (c01 >> 16) != 0xFFFFu) + (n > 2u && (c23 & 0xFFFFu) != 0xFFFFu) +
                              (n > 3u && (c23 >> 16) != 0xFFFFu);
                }
                v_sorted = (ncodes == 1u && !dots) ? 0x80000000u : 0u;
            }
        }
        KFT_LAP(1);
        if (st == 0 && ngrp > 0u) {
            seg_lo32 = (uint32_t)seg_begin; seg_len32 = (uint32_t)seg_len;
            rwb = RefLoad<RefWord>::from(p.refw, ref_base);
            ref_base_g = ref_base; has_dense = seg_len > 0; tid_cur = tid;
            if (kIns) {
                side_fast = ref_base >= 0 && ref_base + ctg_len < (1ll << 35) && hp <= 29;
                side_kbase = ((unsigned long long)ref_base << 28) | ((unsigned long long)(rev & 1) << 27) | (hp < 0 ? 31ull : (unsigned long long)hp);
            }
            gcb = nullptr; gsite = nullptr; gnp = 1;
            ml_start = 0;
            uint32_t unw_last = 0;   // 1 + the largest last rank of the groups nobody asked for (0: none)
            uint32_t prev_ntok = 0;
            bool prev_done = false;   // the group before this one was requested and walked (its tokens are prev_ntok)
            for (uint32_t gi = 0; gi < ngrp && st == 0; gi++) {
                const uint32_t gmpos = uniu(S.g_mpos[gi]), lstart = uniu(S.g_lstart[gi]), gflags = uniu(S.g_flags[gi]), c01 = uniu(S.g_c01[gi]), c23 = uniu(S.g_c23[gi]);
                ncg = (int)((gflags >> 12) & 7u);
                dot_group = (gflags & 4u) != 0u;
                v_gord = gi;
                const bool wanted = !(gflags & 64u);
                if (wanted) {
                    gc01 = c01; gc23 = c23;
                    ci0 = uniu(S.g_ci[gi][0]); ci1 = uniu(S.g_ci[gi][1]); ci2 = uniu(S.g_ci[gi][2]); ci3 = uniu(S.g_ci[gi][3]);
                    // the counters of the group's context class for this read: its strand and haplotype plane, its contig's sites
                    const int kc = (int)((gflags >> 16) & 31u);
                    const DevClass kd = scalar_load(p.classes + kc);
                    const int64_t adjv = scalar_load(p.adj + ((int64_t)tid * p.n_classes + kc) * 2 + rev);
                    gnp = (uint32_t)kd.np;
                    gsite = kd.dense ? nullptr : (rev ? kd.site[1] : kd.site[0]);   // (a select: an index would put the pair into scratch)
                    const int hpl = kIns ? (hpi >= 0 ? hpi : 0) : 0;
                    gcb = has_dense ? p.counters + kd.base + (((int64_t)(hpl * 2 + rev) * kd.nsites) + adjv + (kd.dense ? ref_base : 0)) * kd.np : nullptr;
                }
                uint32_t ntok = 0;
                tw = 0;
                const uint32_t endA = uniu(S.g_end[gi]);
                // A group nobody asked for only matters for its token count (where the next group's ML bytes begin) and for the
                // check of its last rank (mod.c:1116).  5mC + 5hmC callers write the SAME list twice (C+h?,<list>;C+m?,<list>;):
                // when the list is, character for character, that of a requested neighbour -- the one behind it, or the one in
                // front that has just been walked -- and nothing but ",digits,digits,...", its tokens are its commas and its ranks
                // are the neighbour's, whose walk checks them: the group costs one comparison of the two texts.
                if (!wanted && endA > lstart) {
                    uint32_t same = 0;
                    if (gi + 1u < ngrp) {
                        const uint32_t f2 = uniu(S.g_flags[gi + 1u]), lstartB = uniu(S.g_lstart[gi + 1u]), endB = uniu(S.g_end[gi + 1u]);
                        if (!(f2 & 64u) && endA - lstart == endB - lstartB) same = twin_lists(lstart, lstartB, endA - lstart);
                    }
                    if (!same && prev_done && gi > 0u) {
                        const uint32_t lstartB = uniu(S.g_lstart[gi - 1u]), endB = uniu(S.g_end[gi - 1u]);
                        if (endA - lstart == endB - lstartB) { same = twin_lists(lstart, lstartB, endA - lstart); if (same != prev_ntok) same = 0; }
                    }
                    if (same) { ml_start += same * (uint32_t)ncg; prev_done = false; KFT_LAP(8); continue; }
                }
                // TWIN groups: this group and the next one are `?` lists of one requested code each over the same tokens -- the
                // same text.  One pass does both codes on every call; the second list is never parsed and the merge with the
                // sequence and the CIGAR is not made twice.  The reference walks the groups one after the other
                // (mod.c:1003-1370): the same updates in another order.
                if (kTwinOK && wanted && !dot_group && ncg == 1 && gi + 1u < ngrp) {
                    const uint32_t f2 = uniu(S.g_flags[gi + 1u]);
                    const uint32_t lstartB = uniu(S.g_lstart[gi + 1u]), endB = uniu(S.g_end[gi + 1u]);
                    if (!(f2 & (64u | 4u)) && ((f2 >> 12) & 7u) == 1u && endA > lstart && endA - lstart == endB - lstartB) {
                        if (((f2 >> 16) & 31u) == ((gflags >> 16) & 31u)) tw = twin_lists(lstart, lstartB, endA - lstart);   // (one context class for both codes)
                        if (tw) {
                            ncg = 2;
                            gc01 = (c01 & 0xFFFFu) | (uniu(S.g_c01[gi + 1u]) << 16);
                            ci1 = uniu(S.g_ci[gi + 1u][0]);
                        }
                    }
                }
                KFT_LAP(8);
                st = run_group(gmpos, lstart, wanted, ntok);
                KFT_LAP(2);
                prev_done = wanted && st == 0; prev_ntok = ntok;
                if (kTwinOK && tw) {
                    if (st == 0 && ntok != tw) st = 2;   // (cannot happen: a list that plain has as many tokens as commas)
                    ml_start += 2u * ntok; gi++; prev_done = false; continue;
                }
                if (st == 0 && !wanted && Rcarry != 0u) unw_last = max(unw_last, Rcarry);
                ml_start += ntok * (uint32_t)ncg;
            }
            // groups nobody asked for: their last listed rank must exist (mod.c:1116).  A rank the requested groups' walks have
            // found exists; else the class is counted (the whole read's directory, when one segment holds it, has the count)
            if (st == 0 && unw_last > rank_ok) {
                const uint32_t nb_all = (d_n != 0u && d_t0 == 0u && d_n == nblk) ? S_hi : count_all();
                if (unw_last > nb_all) st = 2;
            }
        }
        return st;
    }

    __device__ void flush_stats(uint32_t stat_slot) {
        uint32_t v[4] = {st_look, st_ml, st_dense, st_side};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t x = wave_incl_scan(v[i]);
            uint32_t tt = lane_valu(x, 63);
            if (lane_id() == 0 && tt) p.stats[16 + 4 * (size_t)stat_slot + i] += (unsigned long long)tt;   // this wave owns the row: no atomics
        }
    }
};

// kStats: the tally pass (work counts per wave, routing counts); the timed launches run the instantiation without them
template <typename RefWord, bool kStats, bool kDot, bool kView, bool kIns>
__global__ __launch_bounds__(256, (kDot ? MM_STREAM_WAVES_DOT : (kIns ? MM_STREAM_WAVES_INS : MM_STREAM_WAVES))) void k_stream_reads(const TileParams P) {
    __shared__ StreamLdsT<kDot> lds[kWavesPerBlock];
    __shared__ uint32_t ptab[kSumTabWords];
    fill_sum_table(ptab);
    __syncthreads();
    KF<RefWord, kStats, kDot, kView, kIns> k(P, lds[threadIdx.x >> 6], ptab);
    const DevParams& p = P.d;
    if (lane_id() < (int)kStreamMemo) lds[threadIdx.x >> 6].memo_len[lane_id()] = 0u;   // no header remembered yet
    if (lane_id() == 0) { lds[threadIdx.x >> 6].sres_at = 0u; lds[threadIdx.x >> 6].sres_end = 0u; }   // no side-list chunk yet
    if (P.reset_in_stream && blockIdx.x == 0) {   // the other control set (this launch uses its own until it ends)
        if (p.ctl_next && threadIdx.x < kCtlWords) p.ctl_next[threadIdx.x] = threadIdx.x == 1 ? 0xFFFFFFFFu : 0u;
        if (p.queue_next && threadIdx.x < 192) p.queue_next[threadIdx.x * kQueueStride] = 0u;
    }
    // items costliest first; a wave's first item is fixed, the following ones are handed out by 64 padded counters (as in
    // k_scan_reads)
    const int n_waves = (int)gridDim.x * kWavesPerBlock;
    const int n = (int)scalar_load(P.stream_count);
    const int g = uni((int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6));
    const bool dynamic = n_waves >= (int)kTileRegions;
    k.v_region = (uint32_t)g % kViewRegions;
    // a SLICED launch (freq_tiles.hip.h, stream_bucket): this workgroup's XCD works through its own position slice front to back
    // (eight interleaved queues per slice), then through the slices behind it; otherwise items costliest first, a wave's first item
    // fixed, the following ones handed out by 64 padded counters (as in k_scan_reads)
    const bool sliced = P.stream_slices != nullptr && dynamic;
    const uint32_t xcc = sliced ? (uniu(__builtin_amdgcn_s_getreg((3 << 11) | 20)) & (kStreamSlices - 1u)) : 0u;    // HW_REG_XCC_ID, bits 3:0
    const uint32_t sub = (((uint32_t)blockIdx.x >> 3) * (uint32_t)kWavesPerBlock + (threadIdx.x >> 6)) & 7u;   // (workgroups b and b + 8 share an XCD: its wavefronts spread over the slice's eight queues)
    uint32_t hop = 0;
    int lo = 0, hi = 0;
    if (sliced) { lo = (int)scalar_load(P.stream_slices + xcc); hi = (int)scalar_load(P.stream_slices + xcc + 1u); }
    for (int r = g;;) {
        int r_cur;
        if (sliced) {
            bool got = false;
            while (hop < kStreamSlices) {
                const uint32_t sl = (xcc + hop) & (kStreamSlices - 1u);
                unsigned int c = 0;
                if (lane_id() == 0) c = atomicAdd(P.stream_queue + (sl * 8u + sub) * kQueueStride, 1u);
                r_cur = lo + (int)(uniu(c) * 8u + sub);
                if (r_cur < hi) { got = true; break; }
                hop++;
                if (hop < kStreamSlices) { const uint32_t s2 = (xcc + hop) & (kStreamSlices - 1u); lo = (int)scalar_load(P.stream_slices + s2); hi = (int)scalar_load(P.stream_slices + s2 + 1u); }
            }
            if (!got) break;
        } else {
            if (r >= n) break;
            r_cur = r;
            if (dynamic) {
                unsigned int c = 0;
                if (lane_id() == 0) c = atomicAdd(P.stream_queue + (unsigned int)(g % (int)kTileRegions) * kQueueStride, 1u);
                r = n_waves + (int)(uniu(c) * kTileRegions) + g % (int)kTileRegions;
            } else {
                r += n_waves;
            }
        }
        const int ridx = uni((int)scalar_load(P.stream_items + r_cur));
        const int st = uni(k.run(ridx));
        if (kStats && p.stats && lane_id() == 0) atomicAdd(p.stats + 4 + st, 1ull);   // stats pass: reads done here / handed to the tiles / to the fused kernel
        if (st == 1 && lane_id() == 0) {   // not this kernel's kind of read: one more item for k_scan_reads
            const unsigned int at = atomicAdd(P.tile_plan_count, 1u);
            P.tile_items[at] = ridx;
            if (P.host_tile_flag) *P.host_tile_flag = 1u;
            if (!kDot && k.saw_dot && P.host_dot_flag) *P.host_dot_flag = 1u;
        }
        if (st == 2 && lane_id() == 0) {   // an input error somewhere in the read: the fused kernel names it
            const unsigned int at = atomicAdd(P.fb_count, 1u);
            P.fb_list[at] = ridx;
            if (P.host_fb_flag) *P.host_fb_flag = 1u;
        }
    }
    if (kIns) k.side_fill_rest();
    if (kStats && p.stats) k.flush_stats((uint32_t)g & (kStatSlots - 1));
#ifdef MM_STREAM_TIMING
    if (p.stats && lane_id() == 0) for (int i = 0; i < 9; i++) atomicAdd(p.stats + 7 + i, k.ftacc[i]);
#endif
}

}  // namespace mmhip
