// freq_tiles.hip.h -- the balanced two-kernel form of the freq hot path (reference src/mod.c:776-1370).
//
// The fused kernel of freq_kernels.hip.h gives one wavefront a whole read and keeps the CIGAR prefix arrays and the
// rank directory in LDS: at -K 4096 that is only 4096 waves, 13 KB of LDS each (3 waves per SIMD), and a launch lasts
// as long as its longest read.  Here the same work is cut differently:
//
//   KA  k_scan_reads   three independent waves per read, light and streaming: (1) CIGAR prefix arrays and (2) the rank
//                      directory go to a global scratch (L2-resident, indexed like the pools they come from); (3) the
//                      MM group headers are parsed and every group's skip list is cut into TILES of kTileChars characters
//                      (plus tail tiles for '.' groups) -- nothing sequential over the text itself.
//   KS  k_sum_tiles    one wave per tile: tokens and rank sum of the tile (8 bytes per tile).
//   KC  k_call_tiles   one wave per tile, any order, 1.3 KB of LDS: prefix over the summaries of the tiles in front of
//                      it gives the carries (tokens, ranks, ML index); then parse the tile's tokens, turn ranks into
//                      read positions (directory search + in-block select), project through the CIGAR arrays, test the
//                      reference word, threshold the ML byte, one 64-bit atomic per call.
//
// Reads the tile form does not cover (more than 4 code letters in a group, groups with different canonical bases) are
// put on a list and run through the fused kernel afterwards: same results, just slower.
#pragma once
#include "freq_kernels.hip.h"

namespace mmhip {

#ifdef MM_PHASE_TIMING
#define KAT_DECL unsigned long long _k0 = __builtin_amdgcn_s_memrealtime(), _k1
#define KAT_LAP(slot) do { _k1 = __builtin_amdgcn_s_memrealtime(); tacc[(slot) - 8] += _k1 - _k0; _k0 = _k1; } while (0)
#else
#define KAT_DECL do {} while (0)
#define KAT_LAP(slot) do {} while (0)
#endif

#ifndef MM_TILE_CHARS
#define MM_TILE_CHARS 320
#endif
constexpr uint32_t kTileChars = MM_TILE_CHARS;   // a multiple of 64, at most 448 (the descriptor keeps 9 bits of "characters left")
constexpr uint32_t kTileSub = kTileChars / 64;   // 64-character sub-chunks k_sum_tiles parses a tile in
constexpr uint32_t kTileTok = kTileChars / 2;    // most tokens a tile can hold (a token is at least two characters)
static_assert(kTileChars % 64 == 0 && kTileChars >= 128 && kTileChars <= 448, "tile size");
constexpr uint32_t kTailRanks = 16384;  // implicit-call ranks per tail tile
constexpr uint32_t kTileRegions = 64;   // independent reservation counters / tile regions

struct TileRec {  // 32 bytes
    uint32_t ridx;
    uint32_t cpos;         // list tile: offset of its first character in the MM string; tail tile: its index j in the group's tail
    uint32_t read_first;   // index (inside the region) of the read's first tile
    uint32_t group_first;  // index of the group's first tile
    uint32_t flags;        // bit0 valid, bit1 tail, bit2 dot, bit3 direct, bit4 mb_is_N, bit5 first tile of its list, bit6 no requested code,
                           // bit7 last tile of its list, 8-10 cls, 12-14 n_codes
    int16_t g_code[4];
    uint32_t gord;         // ordinal of the MM group in the read (view mode: which of two entries with one key came first)
};
static_assert(sizeof(TileRec) == 32, "TileRec must be 32 bytes");

constexpr int kStatusHandedOver = -1;   // DevParams.status of a read the tile kernels leave to the fused kernel

struct TileParams {
    DevParams d;              // batch, reference, counters, options
    uint32_t* g_cq;           // [n_cigar_words] query offset at the start of each op (indexed like the cigar pool)
    uint32_t* g_cr;           // [n_cigar_words] reference offset | op << 28
    uint32_t* g_dir;          // [n_seq_bytes / 16] rank directory (indexed like the seq pool, one entry per 16 bytes)
    uint32_t* g_qtot;         // [n_reads] query length of the CIGAR
    uint32_t* g_nb;           // [n_reads] bases of the read's class
    uint32_t* g_qdir;         // coarse: CIGAR op containing read position 256*k  (base (seq_off >> 7) + 2*ridx)
    uint32_t* g_rdir;         // coarse: directory block containing rank 64*k       (base (seq_off >> 5) + 2*ridx)
    TileRec* tiles;           // kTileRegions regions of tile_cap records each
    uint2* g_sum;             // per tile: x = tokens | n_codes << 16, y = sum(skip+1)
    uint32_t* g_tok;          // the listed tokens of every tile of a requested group, as k_sum_tiles parsed them: running sum of
                              // (skip+1) inside the tile.  A token is at least two characters, so the tokens of the tile at
                              // character c of the MM pool fit from index c/2 on: [n_mm_bytes / 2 + 128]
    unsigned int* tile_count; // [kTileRegions] tiles reserved so far in each region (one shared counter would serialise)
    unsigned int tile_cap;    // records per region
    unsigned int* scan_queue; // [kTileRegions * kQueueStride] hand-out counters of k_scan_reads (items behind the static first round)
    unsigned int* tile_queue; // [kTileRegions * kQueueStride] next tile of each region for k_call_tiles (dynamic hand-out behind the static first round)
    int32_t* fb_list;         // reads left to the fused kernel
    unsigned int* fb_count;
    unsigned int* host_fb_flag;  // pinned host word: set when a read went on the fallback list (the host then runs the fused
                                 // kernel for it when it waits for the batch; otherwise that launch is saved)
    int32_t reset_in_call;       // 1: k_call_tiles is the launch's last kernel and resets the next launch's control words
    // work items planned on the device (k_plan_items): d.order points at them, their number is read from *plan_count
    const unsigned int* plan_count;   // null: the caller's plan (d.n_items entries)
    // k_stream_reads (freq_stream.hip.h): its own item list (read indices, costliest first), the hand-out counters behind
    // the static first round, and where it appends the reads it leaves to the tile pipeline (which runs after it)
    const int32_t* stream_items;
    const unsigned int* stream_count;
    const unsigned int* stream_slices;   // sliced launches (k_plan_items with long_cut): first item of every position slice, [8] = the items; else null
    unsigned int* stream_queue;       // [kTileRegions * kQueueStride]
    int32_t* tile_items;              // == d.order, writable
    unsigned int* tile_plan_count;    // == plan_count, writable
    unsigned int* host_tile_flag;     // pinned host word: set when k_stream_reads hands a read to the tile pipeline (when every read of
                                      // the launch is a stream item, the tile kernels are only launched if this says so)
    unsigned int* host_dot_flag;      // pinned host word: set when the lean k_stream_reads handed a read on because of a '.' group
    int32_t reset_in_stream;          // 1: k_stream_reads is the launch's last kernel and resets the next launch's control words
};
constexpr uint32_t kPartSlots = 16;                       // parts per read (4 bits in a work item)
constexpr int kPlanBuckets = 256;                         // cost buckets of 256 bases per part (a part is at most `split` bases: with the
                                                          // default 24 576 fewer than a hundred buckets are ever used; longer ones share the last)
constexpr uint32_t kSplitBases = 24576;                   // default: a read longer than this is cut into parts of about this many bases
                                                          // (measured on C2: 8192 479, 16384 507, 24576 519, 49152 517, 131072 456 Gbases/s)

// g_sum between k_scan_reads and k_sum_tiles: x = offset of the tile's first character in the MM pool, y = these bits |
// n_codes << 16 | characters left in the read's MM string from there (capped at 511)
constexpr uint32_t kSumParse = 0x80000000u, kSumFirst = 0x40000000u, kSumKeep = 0x20000000u;
constexpr uint32_t kGroupEnds = 32;
struct ScanLds {
    uint32_t mmw[260];   // 1024 MM characters + 16 of look-ahead
    char hdr[16];
    int16_t g_code[16];
    // run_mm: what the record pass needs of every group, written once by the header pass
    uint32_t g_lstart[kGroupEnds], g_end[kGroupEnds], g_flags[kGroupEnds], g_c01[kGroupEnds], g_c23[kGroupEnds];
    uint32_t g_first[kGroupEnds + 1];   // index (inside the read) of the group's first tile; [n_groups] = the read's tile count
    uint32_t g_nlist[kGroupEnds];
};
constexpr uint32_t kSliceD = 384;   // rank-directory entries staged in LDS per tile (12 kb of read)
#ifndef MM_SLICE_C
#define MM_SLICE_C 704
#endif
constexpr uint32_t kSliceC = MM_SLICE_C;   // CIGAR ops staged in LDS per tile, one packed word each (with 384-character tiles
                                         // 704 keeps the workgroup at 22.5 KB of LDS: seven per CU)
constexpr uint32_t kSliceSpan = 16384;   // ... when the slice spans fewer read and reference positions than this
struct CallLds {
    uint32_t tok[kTileTok];
    uint32_t gap[64];
    uint32_t gstart[64];
    uint32_t ds[kSliceD];    // slice of the rank directory covering the tile's listed ranks
    uint32_t cs[kSliceC];    // slice of the CIGAR prefix arrays covering the tile's read positions, relative to its first op:
                             // query offset << 18 | reference offset << 4 | op (ordered like the query offsets)
};

// 0x80 in every byte of w that equals ';' (exact for the lowest such byte, which is all that is used)
__device__ __forceinline__ uint32_t semi_bytes(uint32_t w) {
    uint32_t y = w ^ 0x3B3B3B3Bu;
    return (y - 0x01010101u) & ~y & 0x80808080u;
}
// position of the first ';' at or after `from` (or mlen): 4096 characters per trip (four 16-byte loads per lane in
// flight: a long read's group is tens of kilobytes and every trip is a full memory round trip), no LDS.  Loads may run up
// to 15 bytes past the string: inside the pool (16-byte padding per read, 64 bytes of slack at the end).
__device__ __forceinline__ uint32_t find_semicolon(const uint8_t* mm, uint32_t mlen, uint32_t from) {
    const int lane = lane_id();
    uint32_t pos = from;
    uint32_t found = mlen;
    bool hit = false;
    {   // most groups end within a kilobyte: one 16-byte load per lane first
        uint4 w = make_uint4(0, 0, 0, 0);
        const uint32_t off = pos + 16u * lane;
        if (off < mlen) __builtin_memcpy(&w, mm + off, 16);
        uint32_t first = 0xFFFFFFFFu;
        const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int q = 3; q >= 0; q--) {
            uint32_t z = semi_bytes(d[q]);
            if (z) first = 4u * q + ((uint32_t)__ffs((int)z) - 1u) / 8u;
        }
        if (first != 0xFFFFFFFFu && off + first >= mlen) first = 0xFFFFFFFFu;
        uint64_t b = __ballot(first != 0xFFFFFFFFu);
        if (b) {
            int l = __ffsll((unsigned long long)b) - 1;
            found = pos + 16u * (uint32_t)l + lane_valu(first, l);
            hit = true;
        }
        pos += 1024u;
    }
    while (pos < mlen && !hit) {
        uint4 w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t off = pos + 64u * lane + 16u * j;
            w[j] = make_uint4(0, 0, 0, 0);
            if (off < mlen) __builtin_memcpy(&w[j], mm + off, 16);
        }
        // first ';' among this lane's 64 characters (characters at or past mlen do not count)
        uint32_t first = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const uint32_t d[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
            for (int q = 3; q >= 0; q--) {
                uint32_t z = semi_bytes(d[q]);
                if (z) first = 16u * j + 4u * q + ((uint32_t)__ffs((int)z) - 1u) / 8u;
            }
        }
        if (first != 0xFFFFFFFFu && pos + 64u * lane + first >= mlen) first = 0xFFFFFFFFu;
        uint64_t b = __ballot(first != 0xFFFFFFFFu);
        if (b) {
            int l = __ffsll((unsigned long long)b) - 1;
            found = pos + 64u * (uint32_t)l + lane_valu(first, l);
            hit = true;
        }
        pos += 4096u;
    }
    return found < mlen ? found : mlen;
}

// four MM characters starting at byte offset `off`; characters at or past the end of the string read as ';'
__device__ __forceinline__ uint32_t mm_dword(const uint8_t* mm, uint32_t mlen, uint32_t off) {
    uint32_t w = 0x3B3B3B3Bu;
    if (off < mlen) {
        uint32_t raw;
        __builtin_memcpy(&raw, mm + off, 4);
        uint32_t left = mlen - off;
        uint32_t keep = left >= 4u ? 0xFFFFFFFFu : ((1u << (8u * left)) - 1u);
        w = (raw & keep) | (0x3B3B3B3Bu & ~keep);
    }
    return w;
}

// ---- skip lists as sums over characters (k_sum_tiles)
// A token "123" is 1*100 + 2*10 + 3*1: every digit knows its weight from the distance to the next delimiter, so the
// running sum of (skip + 1) at a token's LAST digit is an inclusive scan over characters of
//     digit * 10^(distance to the delimiter - 1)  +  (1 on a token's first character)
// with no per-token loop, no multiply (the product comes from a 11 x 16 table in LDS) and the delimiter bitmaps held in
// scalar registers.  kSumTabWords words: row e (distance 0..10), column d (digit value & 15).
constexpr int kSumTabWords = 11 * 16;
__device__ __forceinline__ void fill_sum_table(uint32_t* tab) {
    for (int i = threadIdx.x; i < kSumTabWords; i += blockDim.x) {
        int e = i >> 4, d = i & 15;
        uint32_t w = 1;
        for (int k = 1; k < e; k++) w *= 10u;
        tab[i] = (e >= 1 && e <= 9 && d <= 9) ? (uint32_t)d * w : 0u;
    }
}
// 32 bits of the 128-bit value (hi:lo) from bit `lane` on
__device__ __forceinline__ uint32_t window32(uint64_t lo, uint64_t hi, int lane) {
    uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = (uint32_t)hi;
    uint32_t a = lane < 32 ? w0 : w1, b = lane < 32 ? w1 : w2;
    return __builtin_amdgcn_alignbit(b, a, (uint32_t)lane & 31u);
}
__device__ __forceinline__ uint64_t low_bits(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }


// ------------------------------------------------------------------------------------------------ plan
// Work items for k_scan_reads, costliest first (mm_freq_plan_batch on the device): read index | part << 24 |
// (parts - 1) << 28, long reads cut into up to kPartSlots parts of about `split` bases: a counting sort over cost buckets
// of 256 bases per part.  The order inside a bucket is whatever the atomics give: results do not depend on it (counters
// add up; view rows are ordered per read afterwards).
__device__ __forceinline__ uint32_t plan_parts(uint32_t L, uint32_t split) {
    uint32_t w = (L + split - 1u) / split;
    return w < 1u ? 1u : (w > kPartSlots ? kPartSlots : w);
}
__device__ __forceinline__ uint32_t plan_bucket(uint32_t L, uint32_t w) {
    uint32_t k = (L / w) >> 8;
    if (k >= (uint32_t)kPlanBuckets) k = kPlanBuckets - 1;
    return (uint32_t)(kPlanBuckets - 1) - k;
}
// Stream items: costliest first like the tile items.  (Tried: four length classes and, inside a class, the reads of one 64th of
// the batch together, so that wavefronts running side by side meet the same reference words in L2 -- C2 58.4 against 56.9 us
// per batch, L2-miss traffic 185 against 192 MB per batch: not kept.)
//
// Round 4, SLICED launches (mean_len != 0: gathered launches of thousands of reads): the chip has eight XCDs with an L2 of 4 MB
// each, and reads come in coordinate order.  The launch's reads are cut into eight POSITION slices (read index * 8 / n), and
// k_stream_reads hands slice x to the workgroups on XCD x (s_getreg XCC_ID), which work through it front to back and then help
// with the slices behind it.  Inside a slice, 32 buckets in three zones:
//   0 - 7    the long reads (more than twice the launch's mean length), costliest first: a 200 kb read lasts as long as the launch;
//   8 - 27   everything within a factor of two of the mean, in FILE ORDER: at any moment an XCD's wavefronts sit on a few hundred
//            consecutive reads -- a few hundred kilobases of reference whose site words, bases and counters stay in that XCD's L2
//            for all thirty reads over a site, instead of every L2 seeing every position once per read;
//   28 - 31  the short reads (less than half the mean), costliest first: what a launch ends with decides how long its last
//            wavefront runs alone (file order to the end: C2 41.6 against 32.0 us per batch).
constexpr uint32_t kStreamSlices = 8, kSliceBuckets = kPlanBuckets / kStreamSlices;
__device__ __forceinline__ uint32_t stream_bucket(uint32_t L, uint32_t i, uint32_t n, uint32_t mean_len) {
    if (mean_len == 0u) return plan_bucket(L, 1u);
    const uint32_t sl = min(kStreamSlices - 1u, (uint32_t)(((uint64_t)i * kStreamSlices) / n));
    const uint32_t lo = (uint32_t)(((uint64_t)sl * n + kStreamSlices - 1u) / kStreamSlices), hi = (uint32_t)(((uint64_t)(sl + 1u) * n + kStreamSlices - 1u) / kStreamSlices);
    uint32_t cls;
    if (L > 2u * mean_len) cls = 7u - min(7u, L / (2u * mean_len) - 1u);
    else if (2u * L < mean_len) cls = 31u - min(3u, (8u * L) / mean_len);
    else cls = 8u + min(19u, (uint32_t)(((uint64_t)(i - lo) * 20u) / max(hi - lo, 1u)));
    return sl * kSliceBuckets + cls;
}
// State the planning workgroups share (one per slot, zeroed once at creation; every launch leaves it zeroed again).
// Two lists are planned at once: class 0 = the tile pipeline's items (parts), class 1 = reads for k_stream_reads.
struct PlanState {
    unsigned int hist[2 * kPlanBuckets];     // parts per bucket, summed over the workgroups ([class][bucket])
    unsigned int cursor[2 * kPlanBuckets];   // first free item of every bucket (each class counts from 0: two item arrays)
    unsigned int done;                       // workgroups that have added their histogram
    unsigned int ready;                      // the launch serial, once the cursors are valid
    unsigned int slices[kStreamSlices + 4];  // sliced launches: first stream item of every slice, [kStreamSlices] = their number
};
constexpr int kPlanThreads = 256;
constexpr int kPlanReadsPerBlock = 512;
static_assert(kPlanBuckets == kPlanThreads, "one bucket of each class per thread in the scan");
// One workgroup per kPlanReadsPerBlock reads (at most 64, all resident at once): a single workgroup is bound by what one
// CU can gather -- 4096 read records 64 bytes apart and as many scattered item stores took it 11 us, 70 us for a group of
// eight batches.  Histograms in LDS, added to the shared one; the last workgroup to arrive scans it into cursors and
// raises `ready`; every workgroup then reserves its share of each bucket with one atomic and hands out places from LDS.
// stream_max: reads of at most this many bases go on the stream list (one item each) instead of the tile list; 0 = none.
__global__ __launch_bounds__(kPlanThreads) void k_plan_items(const mm_read_t* __restrict__ reads, int n, uint32_t split, int32_t* __restrict__ items,
                                                             unsigned int* __restrict__ n_items_out, PlanState* __restrict__ st, unsigned int serial,
                                                             unsigned int* __restrict__ err_summary, unsigned int* __restrict__ host_flag,
                                                             uint32_t stream_max, int32_t* __restrict__ items_stream,
                                                             unsigned int* __restrict__ n_stream_out, unsigned int* __restrict__ host_tile_flag, uint32_t mean_len) {
    __shared__ uint32_t hist[2 * kPlanBuckets];
    __shared__ uint32_t base[2 * kPlanBuckets];
    __shared__ uint32_t wsum[2][kPlanThreads / 64];
    __shared__ uint32_t s_last;
    const int t = threadIdx.x, nb = (int)gridDim.x;
    const int per = ((n + nb - 1) / nb + kPlanThreads - 1) / kPlanThreads * kPlanThreads;
    const int lo = (int)blockIdx.x * per, hi = min(n, lo + per);
    for (int b = t; b < 2 * kPlanBuckets; b += kPlanThreads) hist[b] = 0u;
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += 4 * kPlanThreads) {   // four record loads in flight per thread
        uint32_t L[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + kPlanThreads * u + t; L[u] = i < hi ? reads[i].l_qseq : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + kPlanThreads * u + t;
            if (i < hi) {
                const bool stream = stream_max != 0u && L[u] <= stream_max;
                const uint32_t w = stream ? 1u : plan_parts(L[u], split);
                atomicAdd(&hist[stream ? kPlanBuckets + stream_bucket(L[u], (uint32_t)i, (uint32_t)n, mean_len) : plan_bucket(L[u], w)], w);
            }
        }
    }
    __syncthreads();
    for (int b = t; b < 2 * kPlanBuckets; b += kPlanThreads) if (hist[b]) atomicAdd(&st->hist[b], hist[b]);
    __threadfence();   // a few dozen per launch: the shared histogram is complete before the ticket is drawn
    __syncthreads();
    if (t == 0) s_last = atomicAdd(&st->done, 1u) == (unsigned int)(nb - 1) ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        // exclusive scan of the shared histogram (read and cleared in one exchange), one bucket of each class per thread
        uint32_t v[2], incl[2];
#pragma unroll
        for (int c = 0; c < 2; c++) { v[c] = atomicExch(&st->hist[c * kPlanBuckets + t], 0u); incl[c] = wave_incl_scan(v[c]); }
        if ((t & 63) == 63) { wsum[0][t >> 6] = incl[0]; wsum[1][t >> 6] = incl[1]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t before = incl[c] - v[c];
            for (int w = 0; w < (t >> 6); w++) before += wsum[c][w];
            atomicExch(&st->cursor[c * kPlanBuckets + t], before);
            if (c == 1 && (t % (int)kSliceBuckets) == 0) st->slices[t / (int)kSliceBuckets] = before;
            if (c == 1 && t == kPlanThreads - 1) st->slices[kStreamSlices] = before + v[c];
            if (t == kPlanThreads - 1) {
                if (c == 0) {
                    *n_items_out = before + v[c];
                    // (the host may have left the tile kernels out of the launch because the batch's longest read is a stream
                    // item by its reckoning: if anything was planned for them after all, it has to run them when it waits)
                    if (before + v[c] != 0u && host_tile_flag) *host_tile_flag = 1u;
                } else if (n_stream_out) *n_stream_out = before + v[c];
            }
        }
        __threadfence();
        __syncthreads();
        if (t == 0) { atomicExch(&st->done, 0u); atomicExch(&st->ready, serial); }
    }
    if (t == 0) {
        uint32_t polls = 0;
        while (__hip_atomic_load(&st->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != serial && ++polls < (1u << 24)) __builtin_amdgcn_s_sleep(4);
        s_last = polls < (1u << 24) ? 1u : 0u;   // seconds without the cursors: the launch must still end, and not quietly
        if (!s_last) { atomicMin(err_summary, (unsigned int)MM_E_HIP); if (host_flag) *host_flag = 0u; }
    }
    __syncthreads();
    if (!s_last) return;
    // this workgroup's share of every bucket: one reservation per bucket it has parts in, then places from LDS
    for (int b = t; b < 2 * kPlanBuckets; b += kPlanThreads) {
        const uint32_t cnt = hist[b];
        if (cnt) base[b] = atomicAdd(&st->cursor[b], cnt);
        hist[b] = 0u;
    }
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += 4 * kPlanThreads) {
        uint32_t L[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + kPlanThreads * u + t; L[u] = i < hi ? reads[i].l_qseq : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + kPlanThreads * u + t;
            if (i < hi) {
                const bool stream = stream_max != 0u && L[u] <= stream_max;
                const uint32_t w = stream ? 1u : plan_parts(L[u], split), b = stream ? kPlanBuckets + stream_bucket(L[u], (uint32_t)i, (uint32_t)n, mean_len) : plan_bucket(L[u], w);
                const uint32_t at = base[b] + atomicAdd(&hist[b], w);
                if (stream) items_stream[at] = (int32_t)i;
                else for (uint32_t j = 0; j < w; j++) items[at + j] = (int32_t)((uint32_t)i | (j << 24) | ((w - 1u) << 28));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ KA
struct GroupHdr {
    int herr;
    int modbase, n, ncode, hl, e, flag;
    bool has_nums;
    uint32_t lstart;   // first character of the skip list
};

// Lds: ScanLds, or any struct with hdr[16] and g_code[16] for the header functions alone (k_stream_reads)
template <typename RefWord, typename Lds = ScanLds>
struct KA {
    const TileParams& P;
    const DevParams& p;
    Lds& S;
    int err;
    unsigned long long tacc[5] = {0, 0, 0, 0, 0};   // diagnostic builds: time per phase, flushed once per wave
    __device__ KA(const TileParams& tp, Lds& s) : P(tp), p(tp.d), S(s), err(0) {}

    // group header at mpos (mod.c:1003-1062); also leaves the code characters in S.hdr
    __device__ GroupHdr parse_header(const uint8_t* mm, uint32_t mlen, uint32_t mpos) {
        const uint32_t ci = mpos + (uint32_t)lane_id();
        return parse_header_ch(ci < mlen ? (int)mm[ci] : 0, mlen, mpos);
    }
    // ... with the lane's character (0 past the end of the string) already at hand
    __device__ GroupHdr parse_header_ch(int ch, uint32_t mlen, uint32_t mpos) {
        const int lane = lane_id();
        GroupHdr g;
        uint32_t ci = mpos + lane;
        int c0 = lane_val(ch, 0), c1 = lane_val(ch, 1);
        int herr = 0;
        if (!valid_base_char(c0)) herr = MM_E_MMBASE;
        g.modbase = c0 == 'U' ? 'T' : c0;
        int hl = 1;
        if (mpos + 1 < mlen) {
            if (c1 != '+' && c1 != '-') herr = herr ? herr : MM_E_MMSTRAND;
            hl = 2;
        }
        bool stop = lane >= hl && (ci >= mlen || ch == ',' || ch == ';' || ch == '?' || ch == '.');
        uint64_t sb = __ballot(stop);
        int e = sb ? __ffsll((unsigned long long)sb) - 1 : 64;
        int ncode = e - hl;
        bool iscode = lane >= hl && lane < e;
        bool dig = ch >= '0' && ch <= '9';
        bool alp = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z');
        g.has_nums = __ballot(iscode && dig) != 0;
        bool has_alpha = __ballot(iscode && alp) != 0;
        int n = g.has_nums ? 1 : ncode;
        if (!herr && __ballot(iscode && !dig && !alp)) herr = MM_E_MMCODE;
        if (!herr && (e == 64 || ncode >= MM_CODE_LEN)) herr = MM_E_MMCODE;
        if (!herr && n <= 0) herr = MM_E_MMEMPTY;
        if (!herr && g.has_nums && has_alpha) herr = MM_E_MMMIXED;
        g.herr = uni(herr);
        g.n = n; g.ncode = ncode; g.hl = hl; g.e = e;
        g.flag = '.';
        uint32_t cpos = mpos + (uint32_t)e;
        if (!g.herr && cpos < mlen) {
            int ce = lane_val(ch, e);
            if (ce == '?' || ce == '.') { g.flag = ce; cpos++; }
        }
        g.lstart = cpos;
        wave_sync();
        if (iscode && !g.herr) S.hdr[(lane - hl) & 15] = (char)ch;
        wave_sync();
        return g;
    }

    // required-code lookup per code letter (mod.c:1146-1160) -> S.g_code[m]
    __device__ void lookup_codes(const GroupHdr& g) {
        const int lane = lane_id();
        if (lane < 16) S.g_code[lane] = -1;
        wave_sync();
        int pairs = g.n * p.n_codes;
        for (int p0 = 0; p0 < pairs; p0 += 64) {
            int pi = p0 + lane;
            if (pi < pairs) {
                int m = pi / p.n_codes, t = pi - m * p.n_codes;
                int slen = g.has_nums ? g.ncode : g.ncode - m;
                const DevCode& dc = p.codes[t];
                bool eq = dc.len == slen;
                for (int j = 0; j < slen; j++) eq = eq && (dc.str[j & (MM_CODE_LEN - 1)] == S.hdr[(m + j) & 15]);
                if (eq) S.g_code[m] = (int16_t)t;
            }
        }
        wave_sync();
        if (p.wildcard && lane < g.n && S.g_code[lane] < 0) err = MM_E_NOCODE;
    }

    __device__ __forceinline__ int any_err() const {
        uint64_t eb = __ballot(err != 0);
        int l = eb ? __ffsll((unsigned long long)eb) - 1 : 0;
        int e = lane_val(err, l);
        return eb ? e : 0;
    }

    __device__ void write_tile(TileRec* tiles_base, uint32_t idx, const TileRec& t) {
        // lanes 0..7 store one dword each
        const int lane = lane_id();
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&t);
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) if (lane == i) v = src[i];
        if (lane < 8) reinterpret_cast<uint32_t*>(tiles_base + idx)[lane] = v;
    }

    // The read is the fused kernel's: on the fallback list ONCE (its CIGAR item, its MM item and, for a long read, several parts
    // may all come to that conclusion), and status -1 tells k_sum_tiles / k_call_tiles that its tiles are nobody's.
    __device__ void hand_over(int ridx) {
        if (lane_id() == 0 && atomicCAS(reinterpret_cast<int*>(p.status + ridx), 0, kStatusHandedOver) == 0) {
            const unsigned int k = atomicAdd(P.fb_count, 1u);
            P.fb_list[k] = ridx;
            if (P.host_fb_flag) *P.host_fb_flag = 1u;
        }
    }

    // ---------------- item kind 0: CIGAR prefix arrays -> global (mod.c:776-881 as scans)
    // A long read's scan is cut into `nparts` chunks of ops, one wave each: a chunk first SUMS the op lengths in
    // front of it (cheap: no scans, no stores) to get its carries, then scans only its own ops.
    __device__ int run_cigar(int ridx, uint32_t part, uint32_t nparts) {
        const int lane = lane_id();
        const mm_read_t rd = scalar_load(p.reads + ridx);
        err = 0;
        const int tid = uni(rd.tid), pos = uni(rd.pos);
        const uint32_t L = uniu(rd.l_qseq), ncig = uniu(rd.n_cigar);
        const uint64_t cig_off = rd.cigar_off;
        bool have_ref = tid >= 0 && tid < p.n_contigs;
        if (have_ref) have_ref = scalar_load(p.ref_base + tid) >= 0;
        int result = have_ref ? 0 : MM_E_NOCONTIG;
        if (have_ref) {
            const uint32_t* cg = p.cigar + cig_off;
            const int64_t ctg_len = scalar_load(p.ctg_len + tid);
            uint32_t* const qdir = P.g_qdir + (rd.seq_off >> 7) + 2u * (uint32_t)ridx;
            uint32_t carry_q = 0, carry_r = 0;
            bool over = false;   // an op reaches past the sequence's end
            const uint32_t op_lo = (uint32_t)(((uint64_t)ncig * part) / nparts) & ~63u;   // chunk starts on a 64-op boundary
            const uint32_t op_hi = part + 1u >= nparts ? ncig : ((uint32_t)(((uint64_t)ncig * (part + 1u)) / nparts) & ~63u);
            {   // carries of the chunk: sums over ops [0, op_lo)
                uint32_t sq = 0, sr = 0;
                // 2048 ops per trip (eight 16-byte loads per lane in flight): this pass is nothing but memory round
                // trips, and the last part of a 200 kb read has ten thousand ops in front of it.  op_lo is a multiple
                // of 64 ops, so whole 4-op groups never straddle it.
                for (uint32_t i0 = 0; i0 < op_lo; i0 += 2048) {
                    uint4 wv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        uint32_t i = i0 + 256u * u + 4u * lane;
                        wv[u] = i < op_lo ? *reinterpret_cast<const uint4*>(cg + i) : make_uint4(6u, 6u, 6u, 6u);   // pad op (consumes nothing)
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t w4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            uint32_t op = w4[k] & 15u, len = w4[k] >> 4;
                            sq += ((0x193u >> op) & 1u) ? len : 0u;
                            sr += ((0x18Du >> op) & 1u) ? len : 0u;
                        }
                    }
                }
                carry_q = lane_valu(wave_incl_scan(sq), 63);
                carry_r = lane_valu(wave_incl_scan(sr), 63);
            }
            for (uint32_t i0 = op_lo; i0 < op_hi; i0 += 512) {
                uint32_t wv[8];   // eight loads in flight
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t i = i0 + 64u * u + lane;
                    wv[u] = i < op_hi ? cg[i] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t i = i0 + 64u * u + lane;
                    bool act = i < op_hi;
                    uint32_t w = wv[u], op = w & 15u, len = w >> 4;
                    uint32_t qinc = (act && ((0x193u >> op) & 1u)) ? len : 0u;
                    uint32_t rinc = (act && ((0x18Du >> op) & 1u)) ? len : 0u;
                    if (act && op == 5u) err = MM_E_HARDCLIP;
                    else if (act && (op == 6u || op > 8u)) err = MM_E_CIGAROP;
                    uint32_t qs = wave_incl_scan(qinc), rs = wave_incl_scan(rinc);
                    uint32_t qtot = lane_valu(qs, 63), rtot = lane_valu(rs, 63);
                    qs = carry_q + qs - qinc;
                    rs = carry_r + rs - rinc;
                    bool aligned = act && ((0x181u >> op) & 1u) && len > 0;
                    if (aligned) {
                        if ((uint64_t)qs + len > L) over = true;
                        int64_t r0 = (int64_t)pos + rs;
                        if (r0 < 0 || r0 + (int64_t)len > ctg_len) err = err ? err : MM_E_REFPOS;
                    }
                    if (p.insertions && act && op == 1u && len > 0 && (uint64_t)qs + len > L) over = true;
                    if ((uint64_t)carry_r + rtot >= (1u << 28)) err = err ? err : MM_E_REFPOS;
                    if (act) { P.g_cq[cig_off + i] = qs; P.g_cr[cig_off + i] = (rs & 0x0FFFFFFFu) | (op << 28); }
                    {   // coarse directory: the op that holds every read position that is a multiple of 256.  Most ops
                        // cross no multiple or one (written by their own lane); the few long ones (soft clips, long
                        // matches) are filled by the whole wave so that one lane never loops alone.
                        uint32_t k_lo = (qs + 255u) >> 8, k_hi = (act && qinc) ? (qs + qinc - 1u) >> 8 : 0u;
                        const uint32_t k_max = (L - 1u) >> 8;   // the read's share of g_qdir ends here (a CIGAR longer than the sequence is an error, not a reason to write past it)
                        k_hi = k_hi < k_max ? k_hi : k_max;
                        bool any = act && qinc && L > 0u && k_lo <= k_hi;
                        if (any) qdir[k_lo] = i;
                        uint64_t more = __ballot(any && k_hi > k_lo);
                        while (more) {
                            int l = __ffsll((unsigned long long)more) - 1;
                            uint32_t a = lane_valu(k_lo, l) + 1u, b = lane_valu(k_hi, l), idx = lane_valu(i, l);
                            for (uint32_t k = a + (uint32_t)lane; k <= b; k += 64u) qdir[k] = idx;
                            more &= more - 1ull;
                        }
                    }
                    carry_q += qtot; carry_r += rtot;
                }
            }
            if (lane == 0 && part + 1u >= nparts) P.g_qtot[ridx] = carry_q;
            // a CIGAR that consumes more than the sequence has: whether that is an error depends on where the excess lies and on
            // the strand (get_aln only looks at aligned bases, in the order it walks them, mod.c:813-860): the fused kernel judges
            if (part + 1u >= nparts && carry_q > L) over = true;
            result = any_err();
            if (result == 0 && __ballot(over)) hand_over(ridx);
        }
        return result;
    }

    // the base class the read's groups select from (complement class for reverse reads), -1 if none / unparsable
    __device__ int first_class(const uint8_t* mm, uint32_t mlen, int rev) {
        int cls = -1;
        uint32_t mpos = 0;
        int guard = 0;
        bool stop = false;
        while (mpos < mlen && !stop) {
            GroupHdr g = parse_header(mm, mlen, mpos);
            if (g.herr || ++guard > 4096) stop = true;
            else {
                int mb = rev ? complement_char(g.modbase) : g.modbase;
                bool direct = g.modbase == 'N', dot = g.flag == '.';
                if (!direct || dot) { cls = base_class_of_char(mb); stop = true; }
                else mpos = find_semicolon(mm, mlen, g.lstart) + 1u;
            }
        }
        return cls;
    }

    // class members among the 32 bases of block b (the read's last block may be partial)
    __device__ __forceinline__ uint32_t block_count(uint4 v, int cls, uint32_t b, uint32_t L) const {
        if (cls == 0) {
            int valid = (int)min(32u, L - b * 32u);
            uint32_t o = __popc(nib_eq(v.x, 2) | nib_eq(v.x, 4) | nib_eq(v.x, 8) | nib_eq(v.x, 15)) +
                         __popc(nib_eq(v.y, 2) | nib_eq(v.y, 4) | nib_eq(v.y, 8) | nib_eq(v.y, 15)) +
                         __popc(nib_eq(v.z, 2) | nib_eq(v.z, 4) | nib_eq(v.z, 8) | nib_eq(v.z, 15)) +
                         __popc(nib_eq(v.w, 2) | nib_eq(v.w, 4) | nib_eq(v.w, 8) | nib_eq(v.w, 15));
            return (uint32_t)valid - o;
        }
        return __popc(class_bits(v.x, cls)) + __popc(class_bits(v.y, cls)) + __popc(class_bits(v.z, cls)) + __popc(class_bits(v.w, cls));
    }

    // ---------------- item kind 2: rank directory of the read's one class -> global (mod.c:972-981)
    // Like the CIGAR scan, a long read's directory is cut into `nparts` chunks of blocks, one wave each: a chunk first
    // COUNTS the class members in front of it (no scans, no stores), then scans only its own blocks.
    __device__ int run_dir(int ridx, uint32_t part, uint32_t nparts) {
        const int lane = lane_id();
        const mm_read_t rd = scalar_load(p.reads + ridx);
        const uint32_t L = uniu(rd.l_qseq), mlen = uniu(rd.mm_len);
        const int rev = (uni(rd.flag) & 0x10) ? 1 : 0;
        const uint64_t dir_off = rd.seq_off >> 4;
        const uint32_t nblk = (L + 31u) >> 5;
        const int cls = first_class(p.mm + rd.mm_off, mlen, rev);
        uint32_t nb = 0;
        if (cls >= 0) {
            const uint4* sq = reinterpret_cast<const uint4*>(p.seq + rd.seq_off);
            uint32_t* const rdir = P.g_rdir + (rd.seq_off >> 5) + 2u * (uint32_t)ridx;
            const uint32_t b_lo = (uint32_t)(((uint64_t)nblk * part) / nparts) & ~63u;   // chunks start on a 64-block boundary
            const uint32_t b_hi = part + 1u >= nparts ? nblk : ((uint32_t)(((uint64_t)nblk * (part + 1u)) / nparts) & ~63u);
            uint32_t carry = 0;
            {   // class members in blocks [0, b_lo)
                uint32_t sum = 0;
                for (uint32_t b0 = 0; b0 < b_lo; b0 += 512) {
                    uint4 vv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        uint32_t b = b0 + 64u * u + lane;
                        vv[u] = b < b_lo ? sq[b] : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        uint32_t b = b0 + 64u * u + lane;
                        if (b < b_lo) sum += block_count(vv[u], cls, b, L);
                    }
                }
                carry = lane_valu(wave_incl_scan(sum), 63);
            }
            for (uint32_t b0 = b_lo; b0 < b_hi; b0 += 512) {
                uint4 vv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t b = b0 + 64u * u + lane;
                    vv[u] = b < b_hi ? sq[b] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    uint32_t b = b0 + 64u * u + lane;
                    uint32_t cnt = b < b_hi ? block_count(vv[u], cls, b, L) : 0u;
                    uint32_t incl = wave_incl_scan(cnt);
                    if (b < b_hi) {
                        uint32_t ex = carry + incl - cnt;
                        P.g_dir[dir_off + b] = ex;
                        uint32_t k = (ex + 63u) >> 6;          // a 32-base block holds at most one multiple of 64 ranks
                        if ((k << 6) < ex + cnt) rdir[k] = b;
                    }
                    carry += lane_valu(incl, 63);
                }
            }
            nb = carry;
        }
        if (lane == 0 && part + 1u >= nparts) P.g_nb[ridx] = nb;
        return 0;
    }

    // ---------------- item kind 1: MM group headers -> tiles (or the fallback list)
    __device__ int run_mm(int ridx, uint32_t region) {
        const int lane = lane_id();
        const mm_read_t rd = scalar_load(p.reads + ridx);
        err = 0;
        const int tid = uni(rd.tid);
        const uint32_t L = uniu(rd.l_qseq), mlen = uniu(rd.mm_len);
        const int rev = (uni(rd.flag) & 0x10) ? 1 : 0;
        const uint8_t* mm = p.mm + rd.mm_off;
        bool have_ref = tid >= 0 && tid < p.n_contigs;
        if (have_ref) have_ref = scalar_load(p.ref_base + tid) >= 0;
        int result = 0;   // a missing contig is reported by the CIGAR item
        // pass 1: the group headers (mod.c:1003-1062) -> regular or not, and a table of what the records need.  Reads with
        // more than kGroupEnds groups go the fused kernel's way like the other irregular ones.
        int first_cls = -1;
        bool irregular = false;
        uint32_t need = 0, ngrp = 0;
        if (have_ref) {
            uint32_t mpos = 0;
            while (mpos < mlen && !irregular) {
                GroupHdr g = parse_header(mm, mlen, mpos);
                if (g.herr || g.n > 4 || ngrp >= kGroupEnds) { irregular = true; }   // errors are reported by the fused kernel
                else {
                    int mb = rev ? complement_char(g.modbase) : g.modbase;
                    bool direct = g.modbase == 'N', dot = g.flag == '.';
                    int cls = base_class_of_char(mb);
                    if (!direct || dot) {
                        if (first_cls < 0) first_cls = cls;
                        else if (cls != first_cls) irregular = true;
                    }
                    lookup_codes(g);
                    const int16_t gc0 = S.g_code[0], gc1 = S.g_code[1], gc2 = S.g_code[2], gc3 = S.g_code[3];
                    const bool unwanted = gc0 < 0 && gc1 < 0 && gc2 < 0 && gc3 < 0;   // none of the group's codes was asked for with -c
                    const uint32_t gflags = 1u | (dot ? 4u : 0u) | (direct ? 8u : 0u) | (mb == 'N' ? 16u : 0u) | (unwanted ? 64u : 0u) |
                                            ((uint32_t)cls << 8) | ((uint32_t)g.n << 12);
                    const uint32_t endp = find_semicolon(mm, mlen, g.lstart);
                    const uint32_t nlist = (endp - g.lstart) / kTileChars + 1u;
                    if (lane == 0) {
                        S.g_lstart[ngrp] = g.lstart; S.g_end[ngrp] = endp; S.g_flags[ngrp] = gflags; S.g_first[ngrp] = need; S.g_nlist[ngrp] = nlist;
                        S.g_c01[ngrp] = (uint32_t)(uint16_t)gc0 | ((uint32_t)(uint16_t)gc1 << 16);
                        S.g_c23[ngrp] = (uint32_t)(uint16_t)gc2 | ((uint32_t)(uint16_t)gc3 << 16);
                    }
                    ngrp++;
                    need += nlist + (dot ? L / kTailRanks + 1u : 0u);
                    mpos = endp + 1u;
                }
            }
            if (lane == 0 && ngrp <= kGroupEnds) S.g_first[ngrp < kGroupEnds ? ngrp : kGroupEnds] = need;
            if (p.view && ngrp > kViewMaxGroup + 1u) err = MM_E_TOOMANY;
            wave_sync();
        }
        if (__ballot(err != 0)) irregular = true;   // a code the host did not intern (wildcard runs): reported by the fused kernel in order
        uint32_t tbase = 0;
        bool reserved = false;
        if (have_ref && !irregular && need > 0) {
            if (lane == 0) tbase = atomicAdd(P.tile_count + region, need);
            tbase = uniu(tbase);
            reserved = true;
            if ((uint64_t)tbase + need > P.tile_cap) irregular = true;   // reserved slots are marked invalid below
        }
        uint32_t tcur = tbase;
        TileRec* const rtiles = P.tiles + (size_t)region * P.tile_cap;
        uint2* const rdesc = P.g_sum + (size_t)region * P.tile_cap;
        const uint32_t mm_abs0 = (uint32_t)rd.mm_off;   // the MM pool of a batch is below 4 GiB (checked at submit)
        if (have_ref && irregular) {
            err = 0;
            hand_over(ridx);
        }
        // pass 2: the tile records of all groups in one sweep, a lane per record (no text is read here)
        if (have_ref && !irregular) {
            for (uint32_t j0 = 0; j0 < need; j0 += 64) {
                const uint32_t j = j0 + lane;
                if (j < need) {
                    uint32_t gi = 0;
                    while (gi + 1u < ngrp && S.g_first[gi + 1u] <= j) gi++;
                    const uint32_t gfirst = S.g_first[gi], nlist = S.g_nlist[gi], lstart = S.g_lstart[gi], gflags = S.g_flags[gi];
                    const uint32_t k = j - gfirst;           // record k of its group
                    const bool tail = k >= nlist;
                    const uint32_t n_codes = (gflags >> 12) & 7u;
                    TileRec t;
                    t.ridx = (uint32_t)ridx;
                    t.cpos = tail ? k - nlist : lstart + kTileChars * k;
                    t.read_first = tbase; t.group_first = tbase + gfirst;
                    t.flags = gflags | (tail ? 2u : 0u) | ((!tail && k == 0) ? 32u : 0u) | ((!tail && k + 1u == nlist) ? 128u : 0u);
                    const uint32_t c01 = S.g_c01[gi], c23 = S.g_c23[gi];
                    t.g_code[0] = (int16_t)(c01 & 0xFFFFu); t.g_code[1] = (int16_t)(c01 >> 16); t.g_code[2] = (int16_t)(c23 & 0xFFFFu); t.g_code[3] = (int16_t)(c23 >> 16);
                    t.gord = gi;
                    rtiles[tbase + j] = t;
                    // what k_sum_tiles needs of a list tile, so that it goes from here straight to the text (not record ->
                    // read -> text); it overwrites this with the tile's summary.  Tail tiles get their summary here.
                    const uint32_t rem = mlen - t.cpos;
                    rdesc[tbase + j] = tail ? make_uint2(n_codes << 16, 0u)
                                            : make_uint2(mm_abs0 + t.cpos, kSumParse | (k == 0 ? kSumFirst : 0u) | ((gflags & 64u) ? 0u : kSumKeep) |
                                                                           (n_codes << 16) | (rem < 511u ? rem : 511u));
                }
            }
            tcur = tbase + need;
            result = any_err();
        }
        // reserved slots this read did not fill are marked invalid (flags = 0).  Only a read that RESERVED slots: one that was
        // irregular from its headers on has none, and tbase = 0 would make it wipe the region's first records -- another read's
        if (reserved) {
            uint32_t hi = tbase + need;
            if (hi > P.tile_cap) hi = P.tile_cap;
            for (uint32_t i = tcur + lane; i < hi; i += 64) { rtiles[i].flags = 0u; rdesc[i] = make_uint2(0u, 0u); }
        }
        return result;
    }
};

template <typename RefWord>
__global__ __launch_bounds__(256) void k_scan_reads(const TileParams P) {
    __shared__ ScanLds lds[kWavesPerBlock];
    KA<RefWord> k(P, lds[threadIdx.x >> 6]);
    const DevParams& p = P.d;
    // three items per read (CIGAR scan, MM headers -> tiles, rank directory) over the costliest-first item list.  A wave's
    // first item is fixed (item g); the items behind the first round are handed out by 64 padded counters (item j of the
    // remainder belongs to counter j % 64): the wave that drew the longest read then takes nothing else, and a single
    // shared counter would serialise ~12k dequeues.
    const int n_waves = (int)gridDim.x * kWavesPerBlock;
    const int n = P.plan_count ? (int)scalar_load(P.plan_count) : p.n_items;
    const int g = uni((int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6));   // the wave's index, as a scalar
    const bool dynamic = n_waves >= (int)kTileRegions && P.scan_queue != nullptr;
    for (int r = g; r < 3 * n;) {
        const int r_cur = r;
        if (dynamic) {
            unsigned int c = 0;
            if (lane_id() == 0) c = atomicAdd(P.scan_queue + (unsigned int)(g % (int)kTileRegions) * kQueueStride, 1u);
            r = n_waves + (int)(uniu(c) * kTileRegions) + g % (int)kTileRegions;
        } else {
            r += n_waves;
        }
        const int r_use = r_cur;
        const int ri = r_use / 3, kind = r_use - 3 * ri;   // kinds interleaved: the three items of the costliest reads all start at once
        uint32_t item = p.order ? (uint32_t)scalar_load(p.order + ri) : (uint32_t)ri;
        item = uniu(item);
        const uint32_t part = (item >> 24) & 15u, nparts = ((item >> 28) & 15u) + 1u;
        if (kind == 1 && part != 0u) continue;   // a long read's parts split its CIGAR and directory scans; its MM headers are visited once
        int ridx = (int)(item & 0xFFFFFFu);
        int e = 0;
#ifdef MM_PHASE_TIMING
        unsigned long long kt0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (kind == 0) e = k.run_cigar(ridx, part, nparts);
        else if (kind == 1) e = k.run_mm(ridx, (uint32_t)ri % kTileRegions);
        else e = k.run_dir(ridx, part, nparts);
        e = uni(e);
#ifdef MM_PHASE_TIMING
        if (lane_id() == 0 && p.stats) {   // this wave's own row: longest item of each kind, and when the wave went idle
            unsigned long long kt1 = __builtin_amdgcn_s_memrealtime(), dt = kt1 - kt0;
            unsigned long long* row = p.stats + 16 + 4 * (size_t)(g & (int)(kStatSlots - 1));
            if (dt > row[kind]) row[kind] = dt;
            row[3] = kt1;
        }
#endif
        if (e != 0 && lane_id() == 0) {
            p.status[ridx] = e;
            report_error(p, (unsigned int)ridx, e);
        }
    }
}

// first-character state of a list tile: its first character starts a token iff the previous one is a delimiter
// ------------------------------------------------------------------------------------------------ KS
template <typename RefWord>
__global__ __launch_bounds__(256) void k_sum_tiles(const TileParams P) {
    __shared__ uint32_t lds[kWavesPerBlock][kTileChars / 4 + 4];
    __shared__ uint32_t ptab[kSumTabWords];
    fill_sum_table(ptab);
    __syncthreads();
    const DevParams& p = P.d;
    const int lane = lane_id();
    uint32_t* mmw = lds[threadIdx.x >> 6];
    const unsigned int g = uniu(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // the wave's index, as a scalar
    const unsigned int n_waves = gridDim.x * kWavesPerBlock;
    const unsigned int region = g % kTileRegions;
    unsigned int n_tiles = scalar_load(P.tile_count + region);
    if (n_tiles > P.tile_cap) n_tiles = P.tile_cap;
    const TileRec* const rtiles = P.tiles + (size_t)region * P.tile_cap;
    uint2* const rsum = P.g_sum + (size_t)region * P.tile_cap;
    const unsigned int stride = n_waves / kTileRegions;
    unsigned int ti = g / kTileRegions;
    // descriptors through the scalar cache: the slots this wave reads are written by this wave only
    uint2 d = ti < n_tiles ? scalar_load(rsum + ti) : make_uint2(0u, 0u);
    for (; ti < n_tiles; ti += stride) {
        // the next tile's descriptor is requested before this tile's text
        const unsigned int tn = ti + stride;
        uint2 dn = tn < n_tiles ? scalar_load(rsum + tn) : make_uint2(0u, 0u);
        const uint32_t dx = d.x, dy = d.y;
        d = dn;
        if (!(dy & kSumParse)) continue;   // tail and unused slots already hold their summary
        const uint32_t rem = dy & 511u;
        const uint8_t* mm = p.mm + dx;
        const bool keep = (dy & kSumKeep) != 0;   // a group nobody asked for is only counted
        uint32_t* const tok_out = P.g_tok + (dx >> 1);
        // the tile's characters as dwords: kTileChars / 4 of them + four of look-ahead (two loads per lane at most)
        constexpr uint32_t kDw = kTileChars / 4 + 4;
        uint32_t wd = mm_dword(mm, rem, 4u * lane);
        uint32_t wd2 = (uint32_t)lane + 64u < kDw ? mm_dword(mm, rem, 256u + 4u * lane) : 0u;
        const bool prev_delim = (dy & kSumFirst) ? true : (*(mm - 1) == ',');
        wave_sync();
        mmw[lane] = wd;
        if ((uint32_t)lane + 64u < kDw) mmw[64 + lane] = wd2;
        wave_sync();
        const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(mmw);
        // characters: lane l holds 64*s + l of the sub-chunks, lanes 0..15 the look-ahead; delimiter bitmaps
        uint32_t x[kTileSub];
#pragma unroll
        for (uint32_t sc = 0; sc < kTileSub; sc++) x[sc] = mb8[64 * sc + lane];
        const uint32_t x4 = mb8[kTileChars + (lane & 15)];
        uint64_t D[kTileSub + 1], Sm[kTileSub];
#pragma unroll
        for (uint32_t sc = 0; sc < kTileSub; sc++) {
            Sm[sc] = __ballot(x[sc] == ';');
            D[sc] = Sm[sc] | __ballot(x[sc] == ',');
        }
        D[kTileSub] = __ballot(lane < 16 && (x4 == ',' || x4 == ';')) | ~0xFFFFull;   // past the look-ahead: as if delimited
        uint32_t nends = 0, rsum_v = 0;
        uint64_t bad = 0;
        bool closed = false, open_tail = false;
#pragma unroll
        for (int sc = 0; sc < (int)kTileSub; sc++) {
            if (closed) break;
            // the tile owns characters [lo, hi) of this sub-chunk: up to the group's ';', and not the rest of a token
            // that began in the tile before
            int lo = 0, hi = 64;
            if (Sm[sc]) { hi = __ffsll((unsigned long long)Sm[sc]) - 1; closed = true; }
            if (sc == 0 && !prev_delim) lo = D[0] ? __ffsll((unsigned long long)D[0]) - 1 : 64;
            const uint64_t pd = sc == 0 ? (prev_delim ? 1ull : 0ull) : (D[sc > 0 ? sc - 1 : 0] >> 63);
            // bit 0 of a lane's window: the character before it is a delimiter; bit 1: it is one; bit 2: the next is ...
            const uint32_t w = window32((D[sc] << 1) | pd, (D[sc + 1] << 1) | (D[sc] >> 63), lane);
            const bool own = (uint32_t)(lane - lo) < (uint32_t)(hi - lo) && lo < hi;
            const bool start = own && (w & 3u) == 1u;
            const bool end = own && (w & 6u) == 4u;
            uint32_t e = (uint32_t)__ffs((int)(w >> 1)) - 1u;   // characters up to the delimiter (0: this is one)
            e = e < 10u ? e : 10u;
            const uint32_t dv = x[sc] - (uint32_t)'0';
            uint32_t c = ptab[(e << 4) | (dv & 15u)];
            c = own ? c : 0u;
            c += start ? 1u : 0u;
            const uint32_t run = wave_incl_scan(c);
            const uint64_t eb = __ballot(end);
            if (keep && end) tok_out[nends + (uint32_t)__popcll(eb & lanemask_lt())] = rsum_v + run;   // k_call_tiles does not parse again
            bad |= __ballot(own && !(w & 2u) && dv > 9u) | __ballot(start && e >= 10u);
            rsum_v += lane_valu(run, 63);
            nends += (uint32_t)__popcll(eb);
            if (sc == (int)kTileSub - 1 && !closed) open_tail = ((D[kTileSub - 1] >> 63) == 0) && ((D[kTileSub] & 1ull) == 0);
        }
        uint32_t ntok = nends;
        if (open_tail) {
            // the tile's last token runs into the look-ahead: its remaining digits
            const int k = __ffsll((unsigned long long)D[kTileSub]) - 1;   // 1..16
            const uint32_t dv = x4 - (uint32_t)'0';
            uint32_t e = (uint32_t)(k - lane);
            e = e < 10u ? e : 10u;
            uint32_t c = lane < k ? ptab[(e << 4) | (dv & 15u)] : 0u;
            bad |= __ballot(lane < k && dv > 9u);
            rsum_v += lane_valu(wave_incl_scan(c), 15);
            if (keep && lane == 0) tok_out[nends] = rsum_v;
            ntok = nends + 1u;
        }
        if (bad) {
            // rare: name the error as a sequential reader would (mod.c:1074-1081: a non-digit is seen before the
            // tenth character is counted); one lane walks the tile's characters
            int e = 0;
            if (lane == 0) {
                uint32_t i = 0;
                if (!prev_delim) while (i < kTileChars + 16u && mb8[i] != ',' && mb8[i] != ';') i++;
                while (i < kTileChars && !e) {
                    uint32_t ch = mb8[i];
                    if (ch == ';') break;
                    if (ch == ',') { i++; continue; }
                    int l = 0;
                    while (i < kTileChars + 16u && mb8[i] != ',' && mb8[i] != ';') {
                        if ((uint32_t)mb8[i] - (uint32_t)'0' > 9u) { e = MM_E_SKIPVAL; break; }
                        i++; l++;
                        if (l >= 10) { e = MM_E_SKIPLEN; break; }
                    }
                }
                if (!e) e = MM_E_SKIPVAL;
            }
            e = lane_val(e, 0);
            const uint32_t ridx = uniu(rtiles[ti].ridx);
            if (lane == 0 && p.status[ridx] != kStatusHandedOver) { p.status[ridx] = e; report_error(p, ridx, e); }   // (a handed-over read's errors are the fused kernel's to name)
        }
        if (lane == 0) rsum[ti] = make_uint2(ntok | (dy & 0x70000u), rsum_v);
    }
}

// ------------------------------------------------------------------------------------------------ KC
// kPlain: neither --insertions nor --haplotypes (the usual run): those paths and the values they keep alive are compiled out
template <typename RefWord, bool kView, bool kPlain>
struct KC {
    __device__ __forceinline__ bool opt_insertions() const { return !kPlain && p.insertions; }
    __device__ __forceinline__ bool opt_haplotypes() const { return !kPlain && p.haplotypes; }
    const TileParams& P;
    const DevParams& p;
    CallLds& S;
    int err;
    uint32_t st_look, st_ml, st_dense, st_side;
    // per-tile context (wave-uniform)
    const uint8_t* seq;
    const uint8_t* ml;
    const uint32_t* gq;
    const uint32_t* gr;
    const uint32_t* gd;
    const uint32_t* qdir;
    const uint32_t* rdir;
    int64_t ref_base, seg_begin, seg_len, cnt_base;
    uint32_t L, ncig, nblk, q_total, ml_len, nb, ml_start;
    uint32_t q_shift;   // reverse read with a CIGAR shorter than its sequence: the aligned part lies at BAM positions [q_shift, L) (mod.c:813-860)
    int32_t tid, pos, rev, hp, hpi, cls, direct, mb_is_N, ncg;
    int32_t gc0, gc1, gc2, gc3;
    // what a call needs of its code's table entries, fetched once per tile (wave-uniform; read per call they were
    // three dependent global loads in every round): t_hi | (t_lo + 1) << 9 | ctx_is_star << 18 | context class << 19 (five bits) | (plane + 1) << 24
    uint32_t ci0, ci1, ci2, ci3;
    uint32_t v_ridx, v_gord, v_region;   // view mode: read index, group ordinal, append region
    // LDS slices (wave-uniform): directory blocks [ds_lo, ds_lo+ds_cnt) answer ranks in [ds_rr_lo, ds_rr_hi);
    // CIGAR ops [cs_lo, cs_lo+cs_cnt) answer read positions in [cs_q_lo, cs_q_hi)
    uint32_t ds_lo, ds_cnt, ds_rr_lo, ds_rr_hi, cs_lo, cs_cnt, cs_q_lo, cs_q_hi, cs_q_base, cs_r_base;
    unsigned long long tacc[5] = {0, 0, 0, 0, 0};   // diagnostic builds: time per phase, flushed once per wave

    __device__ __forceinline__ int gcode_at(int m) const { return m == 0 ? gc0 : (m == 1 ? gc1 : (m == 2 ? gc2 : gc3)); }
    __device__ __forceinline__ uint32_t cinfo_at(int m) const { return m == 0 ? ci0 : (m == 1 ? ci1 : (m == 2 ? ci2 : ci3)); }

    __device__ KC(const TileParams& tp, CallLds& s) : P(tp), p(tp.d), S(s), err(0), st_look(0), st_ml(0), st_dense(0), st_side(0) {}

    // Cooperative 64-ary search, two targets at once (t1 <= t2): largest i in [0,n) with arr[i] <= t.  arr is
    // non-decreasing with arr[0] <= t1.  Each round is ONE gather per target (64 lanes sample the current window).
    __device__ __forceinline__ void coop_find2(const uint32_t* arr, uint32_t n, uint32_t t1, uint32_t t2, uint32_t& i1, uint32_t& i2) const {
        const uint32_t lane = (uint32_t)lane_id();
        uint32_t lo1 = 0, len1 = n, lo2 = 0, len2 = n;
        while (len1 > 1u || len2 > 1u) {
            uint32_t st1 = (len1 + 63u) / 64u, st2 = (len2 + 63u) / 64u;
            uint32_t a1 = lo1 + lane * st1, a2 = lo2 + lane * st2;
            bool in1 = lane * st1 < len1, in2 = lane * st2 < len2;
            uint32_t v1 = in1 ? arr[a1] : 0xFFFFFFFFu, v2 = in2 ? arr[a2] : 0xFFFFFFFFu;
            uint64_t b1 = __ballot(in1 && v1 <= t1) | 1ull, b2 = __ballot(in2 && v2 <= t2) | 1ull;
            uint32_t top1 = 63u - (uint32_t)__clzll((unsigned long long)b1), top2 = 63u - (uint32_t)__clzll((unsigned long long)b2);
            uint32_t nlo1 = lo1 + top1 * st1, nlo2 = lo2 + top2 * st2;
            len1 = min(st1, lo1 + len1 - nlo1); lo1 = nlo1;
            len2 = min(st2, lo2 + len2 - nlo2); lo2 = nlo2;
        }
        i1 = lo1; i2 = lo2;
    }

    // stage the directory blocks that cover ranks [rr_a, rr_b] (rr_a <= rr_b < nb) in LDS; the block bounds come from
    // the coarse directory the prepass left (block of every 64th rank): one load instead of a search
    __device__ void setup_dir_slice(uint32_t rr_a, uint32_t rr_b, uint32_t& b1, uint32_t& b2) {
        const uint32_t lane = (uint32_t)lane_id();
        const uint32_t ka = uniu(rr_a >> 6), kb = uniu(rr_b >> 6) + 1u;
        b1 = scalar_load(rdir + ka);
        b2 = kb <= ((nb - 1u) >> 6) ? scalar_load(rdir + kb) : nblk - 1u;
        uint32_t cnt = b2 - b1 + 1u;
        ds_cnt = 0;
        if (cnt <= kSliceD) {
            uint32_t v[kSliceD / 64];
#pragma unroll
            for (uint32_t j = 0; j < kSliceD / 64; j++) { uint32_t i = lane + 64u * j; v[j] = i < cnt ? gd[b1 + i] : 0u; }
#pragma unroll
            for (uint32_t j = 0; j < kSliceD / 64; j++) { uint32_t i = lane + 64u * j; if (i < cnt) S.ds[i] = v[j]; }
            wave_sync();
            ds_lo = b1; ds_cnt = cnt;
            ds_rr_lo = rr_a; ds_rr_hi = rr_b + 1u;   // every rank in [rr_a, rr_b] lies in blocks b1..b2
        }
    }
    // stage the CIGAR ops that cover read positions [q_a, q_b] (q_a <= q_b < q_total) in LDS (bounds: op of every
    // 256th read position, from the prepass)
    __device__ void setup_cig_slice(uint32_t q_a, uint32_t q_b) {
        const uint32_t lane = (uint32_t)lane_id();
        const uint32_t ka = uniu(q_a >> 8), kb = uniu(q_b >> 8) + 1u;
        uint32_t i1 = scalar_load(qdir + ka);
        uint32_t i2 = kb <= ((q_total - 1u) >> 8) ? scalar_load(qdir + kb) : ncig - 1u;
        uint32_t cnt = i2 - i1 + 1u;
        cs_cnt = 0;
        if (cnt <= kSliceC) {
            const uint32_t qb0 = scalar_load(gq + i1), rb0 = scalar_load(gr + i1) & 0x0FFFFFFFu;
            bool ok = q_b - qb0 < kSliceSpan;
            for (uint32_t c0 = 0; c0 < cnt; c0 += 320u) {   // ten loads in flight per trip
                uint32_t vq[5], vr[5];
#pragma unroll
                for (uint32_t j = 0; j < 5; j++) { uint32_t i = c0 + lane + 64u * j; vq[j] = i < cnt ? gq[i1 + i] : 0u; vr[j] = i < cnt ? gr[i1 + i] : 0u; }
#pragma unroll
                for (uint32_t j = 0; j < 5; j++) {
                    uint32_t i = c0 + lane + 64u * j;
                    if (i < cnt) {
                        const uint32_t dq = vq[j] - qb0, dr = (vr[j] & 0x0FFFFFFFu) - rb0;
                        ok = ok && dq < kSliceSpan && dr < kSliceSpan;
                        S.cs[i] = (dq << 18) | (dr << 4) | (vr[j] >> 28);
                    }
                }
            }
            wave_sync();
            if (!__ballot(!ok)) {   // a slice that spans more (a long deletion or intron) is searched in global memory
                cs_lo = i1; cs_cnt = cnt;
                cs_q_lo = q_a; cs_q_hi = q_b + 1u;
                cs_q_base = qb0; cs_r_base = rb0;
            }
        }
    }

    // rank (BAM orientation) -> block and the block's first rank; LDS slice when it covers the rank
    __device__ __forceinline__ uint32_t find_block(uint32_t rr, uint32_t& base) const {
        uint32_t lo = 0, step = 1;
        if (ds_cnt && rr >= ds_rr_lo && rr < ds_rr_hi) {
            while (step < ds_cnt) step <<= 1;
            for (step >>= 1; step; step >>= 1) {
                uint32_t cand = lo + step;
                if (cand < ds_cnt && S.ds[cand] <= rr) lo = cand;
            }
            base = S.ds[lo];
            asm volatile("" : "+v"(base));   // keeps this an LDS read: merged with the global read below it becomes a flat load
            return ds_lo + lo;
        }
        while (step < nblk) step <<= 1;
        for (step >>= 1; step; step >>= 1) {
            uint32_t cand = lo + step;
            if (cand < nblk && gd[cand] <= rr) lo = cand;
        }
        base = gd[lo];
        return lo;
    }
    // read position -> CIGAR op (largest i with cq[i] <= q), its query offset and reference word
    __device__ __forceinline__ uint32_t find_op(uint32_t q, uint32_t& qs, uint32_t& rv) const {
        uint32_t lo = 0, step = 1;
        if (cs_cnt && q >= cs_q_lo && q < cs_q_hi) {
            const uint32_t target = ((q - cs_q_base) << 18) | 0x3FFFFu;
            while (step < cs_cnt) step <<= 1;
            for (step >>= 1; step; step >>= 1) {
                uint32_t cand = lo + step;
                if (cand < cs_cnt && S.cs[cand] <= target) lo = cand;
            }
            const uint32_t w = S.cs[lo];
            qs = cs_q_base + (w >> 18); rv = (cs_r_base + ((w >> 4) & 0x3FFFu)) | (w << 28);
            asm volatile("" : "+v"(qs), "+v"(rv));   // likewise
            return cs_lo + lo;
        }
        while (step < ncig) step <<= 1;
        for (step >>= 1; step; step >>= 1) {
            uint32_t cand = lo + step;
            if (cand < ncig && gq[cand] <= q) lo = cand;
        }
        qs = gq[lo]; rv = gr[lo];
        return lo;
    }
    __device__ __forceinline__ uint32_t select_in_block(uint4 v, uint32_t blk, uint32_t k, uint32_t& code) const {
        int valid = (int)min(32u, L - blk * 32u);
        uint32_t w0 = base_order(v.x), w1 = base_order(v.y), w2 = base_order(v.z), w3 = base_order(v.w);
        uint32_t m0 = class_bits(w0, cls) & valid_bits(valid);
        uint32_t m1 = class_bits(w1, cls) & valid_bits(valid - 8);
        uint32_t m2 = class_bits(w2, cls) & valid_bits(valid - 16);
        uint32_t m3 = class_bits(w3, cls) & valid_bits(valid - 24);
        uint32_t c0 = __popc(m0), c1 = __popc(m1), c2 = __popc(m2);
        uint32_t word = 0, mk = m0, wv = w0;
        if (k >= c0) { k -= c0; word = 1; mk = m1; wv = w1;
            if (k >= c1) { k -= c1; word = 2; mk = m2; wv = w2;
                if (k >= c2) { k -= c2; word = 3; mk = m3; wv = w3; } } }
        uint32_t n = 0, cn = __popc(mk & 0xFFFFu);
        if (k >= cn) { k -= cn; n += 4; mk >>= 16; }
        cn = __popc(mk & 0xFFu);
        if (k >= cn) { k -= cn; n += 2; mk >>= 8; }
        cn = __popc(mk & 0xFu);
        if (k >= cn) { n += 1; }
        code = (wv >> (4 * n)) & 15u;
        return blk * 32u + word * 8u + n;
    }
    __device__ __forceinline__ void side_append(int32_t spos, uint32_t ins_off, int is_mod, int code) {
        unsigned long long key;
        if (side_key(ref_base + spos, rev, code, ins_off, hp, key)) {
            if (side_insert(p.stab, p.smask, p.scur, key, is_mod ? 0x100000001ull : 1ull)) err = MM_E_SIDEFULL;
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
            r.is_mod = (uint8_t)is_mod; r.code = (int16_t)code; r.hp = (int16_t)hp;
            p.side[idx] = r;
        } else {
            err = MM_E_SIDEFULL;
        }
    }

    // J calls per lane as a staged pipeline (same stages as K1::process_calls, prefix arrays in global memory)
    // J calls per lane as a staged pipeline: locate (rank -> read position + base) ...
    template <int J>
    __device__ __forceinline__ void locate(const uint32_t (&rank)[J], bool (&live)[J], uint32_t (&q)[J], uint32_t (&code)[J]) {
        uint32_t blk[J], kk[J];
        uint4 sv[J];
#pragma unroll
        for (int u = 0; u < J; u++) {
            blk[u] = 0; kk[u] = 0; q[u] = 0; code[u] = 0;
            if (!live[u]) continue;
            if (direct) {
                if (rank[u] >= L) { err = MM_E_READPOS; live[u] = false; continue; }
                q[u] = rev ? L - 1 - rank[u] : rank[u];
            } else {
                if (rank[u] >= nb) { err = MM_E_READPOS; live[u] = false; continue; }
                uint32_t rr = rev ? nb - 1 - rank[u] : rank[u];
                uint32_t base;
                blk[u] = find_block(rr, base);
                kk[u] = rr - base;
            }
        }
#pragma unroll
        for (int u = 0; u < J; u++) {
            sv[u] = make_uint4(0, 0, 0, 0);
            if (live[u]) sv[u] = direct ? make_uint4(seq[q[u] >> 1], 0, 0, 0) : reinterpret_cast<const uint4*>(seq)[blk[u]];
        }
#pragma unroll
        for (int u = 0; u < J; u++) {
            if (!live[u]) continue;
            if (direct) { uint32_t b = sv[u].x; code[u] = (q[u] & 1u) ? (b & 15u) : (b >> 4); }
            else q[u] = select_in_block(sv[u], blk[u], kk[u], code[u]);
        }
    }
    // ... then finish (read position -> reference position -> filters -> counter)
    template <int J>
    __device__ __forceinline__ void finish(const uint32_t (&q)[J], const uint32_t (&code)[J], const uint32_t (&kidx)[J], bool (&live)[J], bool is_explicit) {
        uint32_t ins_off[J];
        int64_t ref_pos[J];
#pragma unroll
        for (int u = 0; u < J; u++) {
            ins_off[u] = 0; ref_pos[u] = -1;
            if (!live[u]) continue;
            int64_t rp = -1, anchor = -1;
            const uint32_t qe = q[u] - q_shift;   // (wraps past q_total for the bases in front of the aligned part)
            if (qe < q_total) {
                uint32_t qs, rv;
                (void)find_op(qe, qs, rv);
                uint32_t op = rv >> 28;
                if ((0x181u >> op) & 1u) {
                    rp = (int64_t)pos + (rv & 0x0FFFFFFFu) + (qe - qs);
                } else if (op == 1u && opt_insertions()) {
                    ins_off[u] = (qe - qs + 1u) & 0xFFFFu;
                    anchor = (int64_t)pos + (rv & 0x0FFFFFFFu) - 1;
                }
            }
            if (rp < 0 && opt_insertions()) {
                if (is_explicit || !rev) {
                    rp = anchor;
                } else {   // mod.c:1234,1314 quirk: the mirrored base's insertion anchor
                    uint32_t q2 = L - 1u - q[u] - q_shift;
                    if (q2 < q_total) {
                        uint32_t qs2, rv2;
                        (void)find_op(q2, qs2, rv2);
                        if ((rv2 >> 28) == 1u) rp = (int64_t)pos + (rv2 & 0x0FFFFFFFu) - 1;
                    }
                }
            }
            ref_pos[u] = rp;
            if (rp < 0) live[u] = false;
        }
        uint32_t w[J], ml0[J];
        const typename RefLoad<RefWord>::Base rw = RefLoad<RefWord>::from(p.refw, ref_base);
#pragma unroll
        for (int u = 0; u < J; u++) {
            w[u] = 0; ml0[u] = 0;
            if (!live[u]) continue;
            w[u] = RefLoad<RefWord>::at(rw, ref_pos[u]);
            st_look++;
            if (is_explicit) {
                uint64_t mi = (uint64_t)ml_start + (uint64_t)kidx[u] * ncg;
                if (mi < ml_len) ml0[u] = ml[mi];
            }
        }
#pragma unroll
        for (int u = 0; u < J; u++) {
            if (!live[u]) continue;
            uint32_t refcode = w[u] & 31u;
            for (int m = 0; m < ncg; m++) {
                int ci = gcode_at(m);
                if (ci < 0) continue;
                const uint32_t cinfo = cinfo_at(m);
                const int cls_c = (int)((cinfo >> 19) & 31u), dc_plane = (int)((cinfo >> 24) & 127u) - 1;
                const int t_hi = (int)(cinfo & 511u), t_lo = (int)((cinfo >> 9) & 511u) - 1;
                if (!opt_insertions()) {
                    const bool in_ctx = class_context_bit(p, w[u], cls_c, rev, ref_base + ref_pos[u]);   // (the context class's bits; classes 13 and up: the site word)
                    bool matches = ((cinfo >> 18) & 1u) || mb_is_N || refcode == code[u];
                    if (!(in_ctx && matches)) continue;
                }
                int is_mod = 0;
                if (is_explicit) {
                    uint64_t ml_idx = (uint64_t)ml_start + (uint64_t)kidx[u] * ncg + m;
                    if (ml_idx >= ml_len) { err = MM_E_MLIDX; break; }
                    int mv = m == 0 ? (int)ml0[u] : (int)ml[ml_idx];
                    st_ml++;
                    if (kView) {   // mod.c:1194-1196: no threshold, the ML byte itself
                        view_append(p, v_region, v_ridx, (uint32_t)(ref_pos[u] - pos + 1), rev ? L - 1u - q[u] : q[u], ins_off[u],
                                    (uint32_t)ci, v_gord, 0u, (uint32_t)mv);
                        continue;
                    }
                    if (mv >= t_hi) is_mod = 1;
                    else if (mv <= t_lo) is_mod = 0;
                    else continue;
                } else if (kView) {   // mod.c:1281-1283, :1361-1363: implicit calls carry probability 0
                    view_append(p, v_region, v_ridx, (uint32_t)(ref_pos[u] - pos + 1), rev ? L - 1u - q[u] : q[u], ins_off[u],
                                (uint32_t)ci, v_gord, 1u, 0u);
                    continue;
                }
                int64_t off = ref_pos[u] - seg_begin;
                if (ins_off[u] == 0 && dc_plane >= 0 && hpi >= 0 && off >= 0 && off < seg_len) {
                    atomicAdd(counter_word(p, p.codes[ci], kPlain ? 0 : hpi, rev, tid, ref_base + ref_pos[u]), is_mod ? 0x100000001ull : 1ull);
                    st_dense++;
                } else {
                    side_append((int32_t)ref_pos[u], ins_off[u], is_mod, ci);
                    st_side++;
                }
            }
        }
    }

    template <int J>
    __device__ __forceinline__ void process_calls(const uint32_t (&rank)[J], const uint32_t (&kidx)[J], const bool (&live_in)[J], bool is_explicit) {
        bool live[J];
        uint32_t q[J], code[J];
#pragma unroll
        for (int u = 0; u < J; u++) live[u] = live_in[u];
        locate<J>(rank, live, q, code);
        finish<J>(q, code, kidx, live, is_explicit);
    }

    struct TileArgs { uint32_t ridx, cpos, read_first, group_first, flags, index, gord, region; };
    __device__ __forceinline__ int run(const TileArgs t, uint32_t gc01, uint32_t gc23, const uint2* rsum) {
        const int lane = lane_id();
        err = 0;
        KAT_DECL;
        const int ridx = (int)t.ridx;
        v_ridx = t.ridx; v_gord = t.gord; v_region = t.region;
        // Everything below is a chain of dependent memory round trips, a microsecond each under load; what does not depend
        // on the read record is requested before it: the summaries of the read's tiles in front of this one (first 64)
        // and the tile's own.
        const bool list_tile = !(t.flags & 2u);
        uint2 sv0 = make_uint2(0u, 0u), own_sv = make_uint2(0u, 0u);
        if (t.read_first + (uint32_t)lane < t.index) sv0 = rsum[t.read_first + lane];
        if (list_tile) own_sv = scalar_load(rsum + t.index);
        const mm_read_t rd = scalar_load(p.reads + ridx);
        tid = uni(rd.tid); pos = uni(rd.pos);
        L = uniu(rd.l_qseq); ncig = uniu(rd.n_cigar); ml_len = uniu(rd.ml_len);
        rev = (uni(rd.flag) & 0x10) ? 1 : 0;
        seq = p.seq + rd.seq_off; ml = p.ml + rd.ml_off;
        // ... and what only needs the read record goes out together: the tile's tokens, the per-contig bases, the read's totals
        uint32_t tv[(kTileTok + 63) / 64];
#pragma unroll
        for (uint32_t j = 0; j < (kTileTok + 63) / 64; j++) tv[j] = 0;
        if (list_tile && !(t.flags & 64u)) {
            const uint32_t* const tok_in = P.g_tok + ((rd.mm_off + t.cpos) >> 1);   // kTileTok words belong to the tile: no bound needed
#pragma unroll
            for (uint32_t j = 0; j < (kTileTok + 63) / 64; j++) if (64u * j + (uint32_t)lane < kTileTok) tv[j] = tok_in[64u * j + lane];
        }
        gq = P.g_cq + rd.cigar_off; gr = P.g_cr + rd.cigar_off; gd = P.g_dir + (rd.seq_off >> 4);
        qdir = P.g_qdir + (rd.seq_off >> 7) + 2u * (uint32_t)ridx; rdir = P.g_rdir + (rd.seq_off >> 5) + 2u * (uint32_t)ridx;
        nblk = (L + 31u) >> 5;
        q_total = scalar_load(P.g_qtot + ridx);
        q_shift = (rev && q_total < L) ? L - q_total : 0u;
        nb = scalar_load(P.g_nb + ridx);
        hp = opt_haplotypes() ? (int)rd.hp : -1;
        hpi = opt_haplotypes() ? ((int)rd.hp < p.n_hp ? (int)rd.hp : -1) : 0;
        ref_base = scalar_load(p.ref_base + tid); seg_begin = scalar_load(p.seg_begin + tid);
        seg_len = scalar_load(p.seg_len + tid); cnt_base = scalar_load(p.cnt_base + tid);
        // carries = prefix over the summaries of the read's tiles in front of this one:
        //   ml_start  = sum over earlier groups of tokens * n_codes      (mod.c:1200)
        //   k_carry   = tokens of this group in front of the tile
        //   rank_carry= sum(skip+1) of this group in front of the tile
        uint32_t a_ml = 0, a_k = 0, a_r = 0;
        for (uint32_t i0 = t.read_first; i0 < t.index; i0 += 64) {
            uint32_t i = i0 + lane;
            if (i < t.index) {
                uint2 sv = i0 == t.read_first ? sv0 : rsum[i];
                uint32_t nt = sv.x & 0xFFFFu, nc = (sv.x >> 16) & 7u;
                if (i < t.group_first) a_ml += nt * nc;
                else { a_k += nt; a_r += sv.y; }
            }
        }
        ml_start = lane_valu(wave_incl_scan(a_ml), 63);
        const uint32_t k_carry0 = lane_valu(wave_incl_scan(a_k), 63);
        const uint32_t rank_carry0 = lane_valu(wave_incl_scan(a_r), 63);
        ds_cnt = 0; cs_cnt = 0;
        KAT_LAP(8);
        if (t.flags & 64u) {
            // No code of this group was requested: every call would be discarded (mod.c:1157).  The only thing the
            // reference still does with such a group is assert its read positions (mod.c:1116): the last listed rank
            // must exist.  (Its tokens still count towards ML indices: that is k_sum_tiles' job.)
            // (this is the last tile of the group's list; it may be empty when the list ends on a tile boundary, the
            // running rank in front of it then is the group's last one)
            if (list_tile) {
                const uint2 own = own_sv;
                if (rank_carry0 + own.y != 0u) {
                    uint32_t r_last = rank_carry0 + own.y - 1u;
                    if (r_last >= (((t.flags >> 3) & 1u) ? L : nb)) err = MM_E_READPOS;
                }
            }
            uint64_t eb0 = __ballot(err != 0);
            return eb0 ? MM_E_READPOS : 0;
        }
        const uint32_t fl = t.flags;
        const bool tail = fl & 2u, dot = fl & 4u;
        direct = (fl >> 3) & 1; mb_is_N = (fl >> 4) & 1; cls = (int)((fl >> 8) & 7u); ncg = (int)((fl >> 12) & 7u);
        gc0 = (int16_t)(gc01 & 0xFFFFu); gc1 = (int16_t)(gc01 >> 16); gc2 = (int16_t)(gc23 & 0xFFFFu); gc3 = (int16_t)(gc23 >> 16);
        {
            const int ci = lane == 0 ? gc0 : (lane == 1 ? gc1 : (lane == 2 ? gc2 : gc3));
            uint32_t info = 0;
            if (lane < ncg && lane < 4 && ci >= 0) {
                const DevCode& dc = p.codes[ci];
                const int req = dc.req, plane = dc.plane;
                const DevMod& dm = p.mods[req];
                info = (uint32_t)dm.t_hi | ((uint32_t)(dm.t_lo + 1) << 9) | (dm.ctx_is_star ? (1u << 18) : 0u) | ((uint32_t)P.d.cls_of_mod[req] << 19) |
                       ((uint32_t)(plane + 1) << 24);
            }
            ci0 = lane_valu(info, 0); ci1 = lane_valu(info, 1); ci2 = lane_valu(info, 2); ci3 = lane_valu(info, 3);
        }
        if (tail) {
            // bases after the last listed one (mod.c:1289-1365): this tile's slice of [rank_carry0, nb)
            const uint64_t lo64 = (uint64_t)rank_carry0 + (uint64_t)kTailRanks * t.cpos;
            const uint32_t tlo = lo64 < nb ? (uint32_t)lo64 : nb;
            const uint32_t thi = (lo64 + kTailRanks) < nb ? (uint32_t)(lo64 + kTailRanks) : nb;
            for (uint32_t r0 = tlo; r0 < thi; r0 += 64u) {
                uint32_t r2[1] = {r0 + lane}, k2[1] = {0};
                bool l2[1] = {r2[0] < thi};
                process_calls<1>(r2, k2, l2, false);
            }
        } else {
            // the tile's listed tokens as k_sum_tiles left them (running sums of skip+1 inside the tile) -> ranks in tok[]
            // (the skip of token j is rank[j] - rank[j-1] - 1 again when needed)
            const uint32_t ntok = uniu(own_sv.x) & 0xFFFFu;
#pragma unroll
            for (uint32_t j = 0; j < (kTileTok + 63) / 64; j++) if (64u * j + (uint32_t)lane < ntok) S.tok[64u * j + lane] = rank_carry0 + tv[j] - 1u;
            wave_sync();
            KAT_LAP(9);
            if (ntok > 0) {
                // listed ranks rise with the token index: the first and last token bound everything this tile touches;
                // both slices are staged before the first call ('.' groups also visit the gap in front of the first token)
                const uint32_t r_first = dot ? rank_carry0 : S.tok[0], r_last = S.tok[ntok - 1u];
                if (r_last < (direct ? L : nb)) {
                    uint32_t qa, qb;
                    if (direct) {
                        qa = rev ? L - 1u - r_last : r_first; qb = rev ? L - 1u - r_first : r_last;
                    } else {
                        uint32_t rr_a = rev ? nb - 1u - r_last : r_first, rr_b = rev ? nb - 1u - r_first : r_last;
                        uint32_t b1, b2;
                        setup_dir_slice(rr_a, rr_b, b1, b2);
                        KAT_LAP(10);
                        qa = 32u * b1; qb = min(L - 1u, 32u * b2 + 31u);
                    }
                    // (positions of the aligned query: BAM positions less q_shift)
                    if (qb >= q_shift) {
                        const uint32_t qa_e = qa > q_shift ? qa - q_shift : 0u, qb_e = qb - q_shift;
                        if (q_total > 0 && qa_e < q_total) setup_cig_slice(qa_e, min(qb_e, q_total - 1u));
                    }
                }
            }
            KAT_LAP(11);
#pragma unroll 1
            for (uint32_t t64 = 0; t64 < ntok; t64 += 64u) {
                uint32_t ti = t64 + lane;
                bool lv = ti < ntok;
                uint32_t rk = lv ? S.tok[ti] : 0u;
                uint32_t prev = ti == 0 ? rank_carry0 - 1u : (lv ? S.tok[ti - 1u] : 0u);
                uint32_t r1[1] = {rk}, k1[1] = {k_carry0 + ti};
                bool l1[1] = {lv};
                process_calls<1>(r1, k1, l1, true);
                if (dot) {   // implicit calls in the gap in front of each listed rank (mod.c:1206-1287)
                    uint32_t su = lv ? rk - prev - 1u : 0u;
                    uint32_t gi = wave_incl_scan(su);
                    uint32_t T = lane_valu(gi, 63);
                    wave_sync();
                    S.gap[lane] = gi - su;
                    S.gstart[lane] = rk - su;
                    wave_sync();
                    for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
                        uint32_t tt = t0 + lane;
                        bool l2[1] = {tt < T};
                        uint32_t r2[1] = {0}, k2[1] = {0};
                        if (l2[0]) {
                            uint32_t lo = 0;
#pragma unroll
                            for (uint32_t step = 32; step; step >>= 1) {
                                uint32_t cand = lo + step;
                                if (cand < 64u && S.gap[cand] <= tt) lo = cand;
                            }
                            r2[0] = S.gstart[lo] + (tt - S.gap[lo]);
                        }
                        process_calls<1>(r2, k2, l2, false);
                    }
                }
            }
        }
        KAT_LAP(12);
        uint64_t eb = __ballot(err != 0);
        int l = eb ? __ffsll((unsigned long long)eb) - 1 : 0;
        int e = lane_val(err, l);
        return eb ? e : 0;
    }

    __device__ void flush_stats(uint32_t stat_slot) {
        uint32_t v[4] = {st_look, st_ml, st_dense, st_side};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t x = wave_incl_scan(v[i]);
            uint32_t tt = lane_valu(x, 63);
            if (lane_id() == 0 && tt) p.stats[16 + 4 * (size_t)stat_slot + i] += (unsigned long long)tt;   // this wave owns the row: no atomics
        }
        st_look = st_ml = st_dense = st_side = 0;
    }
};

template <typename RefWord, bool kView, bool kPlain>
__global__ __launch_bounds__(256, kPlain ? 7 : 5) void k_call_tiles(const TileParams P) {
    __shared__ CallLds lds[kWavesPerBlock];
    KC<RefWord, kView, kPlain> k(P, lds[threadIdx.x >> 6]);
    const DevParams& p = P.d;
    if (P.reset_in_call && blockIdx.x == 0) {   // the other control set (this launch uses its own until it ends)
        if (p.ctl_next && threadIdx.x < kCtlWords) p.ctl_next[threadIdx.x] = threadIdx.x == 1 ? 0xFFFFFFFFu : 0u;
        if (p.queue_next && threadIdx.x < 192) p.queue_next[threadIdx.x * kQueueStride] = 0u;
    }
    // static round-robin: wave g serves region g % kTileRegions, striding over that region's tiles with the other
    // waves of the same residue (tiles cost about the same; no shared work counter to serialise on)
    const unsigned int g = uniu(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // the wave's index, as a scalar
    const unsigned int n_waves = gridDim.x * kWavesPerBlock;
    const unsigned int region = g % kTileRegions;
    unsigned int n_tiles = P.tile_count[region];
    if (n_tiles > P.tile_cap) n_tiles = P.tile_cap;
    n_tiles = uniu(n_tiles);
    const TileRec* const rtiles = P.tiles + (size_t)region * P.tile_cap;
    // A wave's first tile is fixed (its index among the region's waves); the following ones are handed out by the region's
    // counter: tiles differ a lot in cost (a group nobody asked for is skipped in a microsecond, a full tile of wanted
    // calls takes twenty) and with a fixed stride the launch lasted twice as long as its average wave.  64 counters, a few
    // hundred increments each: nothing like the single shared work queue that serialised the first version.
    const unsigned int region_waves = n_waves / kTileRegions;
    unsigned int ti = g / kTileRegions;
    while (ti < n_tiles) {
        // the tile record as wave-uniform scalars
        const kptr<uint32_t> src = scalar_ptr(reinterpret_cast<const uint32_t*>(rtiles + ti));
        typename KC<RefWord, kView, kPlain>::TileArgs t;
        t.ridx = src[0]; t.cpos = src[1]; t.read_first = src[2]; t.group_first = src[3];
        t.flags = src[4]; t.index = ti; t.gord = src[7]; t.region = region;
        uint32_t gc01 = src[5], gc23 = src[6];
        // a group nobody asked for only has its last listed rank checked (mod.c:1116): the last tile of its list does that
        if ((t.flags & 1u) && (!(t.flags & 64u) || (t.flags & 130u) == 128u) && scalar_load(p.status + t.ridx) != kStatusHandedOver) {
            int e = uni(k.run(t, gc01, gc23, P.g_sum + (size_t)region * P.tile_cap));
            if (e != 0 && lane_id() == 0) {
                p.status[t.ridx] = e;
                report_error(p, t.ridx, e);
            }
        }
        unsigned int nxt = 0;
        if (lane_id() == 0) nxt = atomicAdd(P.tile_queue + region * kQueueStride, 1u);
        ti = region_waves + uniu(nxt);
    }
#ifndef MM_PHASE_TIMING
    if (p.stats) k.flush_stats(g & (kStatSlots - 1));
#else   // diagnostic builds keep the per-wave rows for k_scan_reads' timings
    if (lane_id() == 0 && p.stats) for (int i = 0; i < 5; i++) atomicAdd(p.stats + 8 + i, k.tacc[i]);
#endif
}

}  // namespace mmhip
