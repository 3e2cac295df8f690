// freq_kernels.hip.h -- hand-written gfx950 kernels of the `freq` hot path (device code only).
//
//   K0  k_build_refwords   load_ref normalisation + load_ref_contexts      reference src/ref.c:73-78,92-229
//   K1  k_freq_reads       get_aln + freq_view_single + update_freq_map    reference src/mod.c:776-1370
//   K2  k_count_nonzero / k_emit_rows   collect step of print_freq_output  reference src/mod.c:644-664
//   slab kernels           halo exchange helpers (no reference counterpart)
//
// Design (MI355X first, not a translation): one 64-lane wavefront owns one read.  Nothing of size O(l_qseq)
// is ever materialised: the CIGAR becomes two exclusive prefix arrays (query / reference offsets per op) in
// LDS, the packed sequence becomes a per-32-base rank directory for the one base class an MM group needs,
// and MM text is parsed 64 characters per step with ballots; skip counts are compacted in LDS so that the
// expensive per-call work (rank -> read position -> reference position -> context test -> ML threshold ->
// counter) runs with all 64 lanes busy.  Counters are one 64-bit word per (plane, strand, position) holding
// {n_called (low 32), n_mod (high 32)}, updated with ONE global_atomic_add_x2 per call.  This is integer
// select/scan/scatter work: no MFMA, HBM- and latency-bound.
#pragma once
#ifdef MM_PHASE_TIMING
#define MMT_DECL unsigned long long _t0 = __builtin_readcyclecounter(), _t1
#define MMT_LAP(slot) do { _t1 = __builtin_readcyclecounter(); if (lane_id() == 0 && p.stats) atomicAdd(p.stats + (slot), _t1 - _t0); _t0 = _t1; } while (0)
#else
#define MMT_DECL do {} while (0)
#define MMT_LAP(slot) do {} while (0)
#endif
#ifdef MM_DEBUG
#define MMDBG(...) do { unsigned long long _m = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll(_m) - 1) { printf("[exec %llx] ", _m); printf(__VA_ARGS__); } } while (0)
#else
#define MMDBG(...) do {} while (0)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minimod_hip.h"

namespace mmhip {

constexpr int kWavesPerBlock = 4;
constexpr uint32_t kStatSlots = 16384;   // per-wave tally rows (power of two)
constexpr int kCigCap = 1024;  // CIGAR ops whose prefix sums live in LDS (the rest spill to a global scratch)
constexpr int kDirCap = 512;   // 32-base blocks whose rank directory lives in LDS (16 kb of read)
constexpr int kTokCap = 512;   // compacted skip counts: one flush takes 256, one 256-character trip adds <= 128
constexpr int kCallsPerLane = 2;  // calls a lane carries through the staged call pipeline (memory-level parallelism)

struct DevCode {
    char str[MM_CODE_LEN];
    int16_t len;
    int16_t req;    // index of the requested mod whose context/threshold applies
    int16_t plane;  // dense counter plane, or -1 (side list only)
    int16_t slot;   // the plane's place among the planes of its context class (DevClass::np of them lie side by side per site)
};

// Counters are kept per SITE, not per position.  The requested mods are grouped into context classes (mods with the same
// context string; with --insertions, where no context is looked at, and for the context `*` every position is a site: a
// `dense` class).  For a class and a strand the positions inside a context match are numbered through the whole reference-word
// space: site[strand][g >> 5] = {bits: which of the 32 positions are sites, rank: sites in front of the block}.  A read's
// consecutive calls then update consecutive counters -- 64-byte lines shared by up to eight calls of one wave instruction,
// which the memory side takes as ONE atomic request each (scattered, it takes 20 G requests a second: tools/atomic_calib.hip) --
// and the planes shrink from 8 bytes a position to 8 bytes a site (CpG: 1 % of the positions).
//   counter of (code plane, haplotype plane hp, strand s, contig t, position pos), g = ref_base[t] + pos:
//       base + (((hp * 2 + s) * nsites) + adj[(t * n_classes + class) * 2 + s] + rank_s(g)) * np + slot
struct DevClass {
    const uint2* site[2];   // null for a dense class (rank(g) = g)
    int64_t base;           // first counter word of the class
    int64_t nsites;         // sites per (haplotype plane, strand) block: the larger of the two strands' counts
    int32_t np;             // code planes of the class, side by side per site
    int32_t dense;
    int32_t stride;         // uint2 units per site word: 1, or 2 when the word also carries its 32 positions' bases, two bits each
                            // (handles with four-bit reference words: k_stream_reads then needs ONE gather per call, not two)
    int32_t pad;
};
__device__ __forceinline__ uint32_t site_rank(uint2 w, uint32_t bit) { return w.y + (uint32_t)__popc(w.x & ((1u << bit) - 1u)); }
constexpr int kPassContexts = 13;   // == MM_MAX_CONTEXTS: context classes whose two bits a reference word holds (bits 5 ... 30)

struct DevMod {
    uint8_t klass[256];
    int32_t t_lo, t_hi;   // klass is monotone in the ML value: <= t_lo -> called, >= t_hi -> called+modified
    int32_t ctx_is_star;
    int32_t ctx_len;
    char ctx_fwd[MM_CODE_LEN];
    char ctx_rev[MM_CODE_LEN];
};

struct SideRec {  // one counter update that does not fit the dense planes (16 bytes)
    int32_t tid;
    int32_t pos;
    uint16_t ins_off;
    uint8_t strand;
    uint8_t is_mod;
    int16_t code;
    int16_t hp;
};

struct DevParams {
    // batch
    const mm_read_t* reads;
    const uint32_t* cigar;
    const uint8_t* seq;
    const uint8_t* mm;
    const uint8_t* ml;
    const int32_t* order;  // optional work items: read index | part << 24 | (parts - 1) << 28
    int32_t n_reads;
    int32_t n_items;       // entries of order[] (== n_reads when order is null)
    const unsigned int* n_items_dev;  // when set, the item count is read from device memory (fallback list)
    // reference
    const void* refw;            // uint16 or uint32 per base: bits 0-4 base code, bit 5+2i fwd ctx, 6+2i rev ctx; one mod: four bits a base (RefNib)
    const int64_t* ref_base;     // per tid: offset into refw, -1 = contig absent
    const int64_t* ctg_len;      // per tid
    const int64_t* seg_begin;    // per tid: first position with dense counters
    const int64_t* seg_len;      // per tid: number of positions with dense counters (0 = none)
    const int64_t* cnt_base;     // per tid: offset of the segment inside a plane
    int32_t n_contigs;
    // counters
    unsigned long long* counters;  // per context class: [hp][strand][site][code plane of the class] (DevClass above)
    int64_t plane_len;             // positions with dense counters (all segments, each padded to 64)
    int32_t n_hp;                  // dense haplotype planes (1 when haplotypes are off)
    int32_t n_classes;
    const DevClass* classes;       // [n_classes]
    const int32_t* cls_of_mod;     // [n_mods]
    const int64_t* adj;            // [(tid * n_classes + class) * 2 + strand]: what added to rank_strand(ref_base + pos) numbers the contig's
                                   // sites inside the class's (haplotype, strand) blocks
    // options
    int32_t n_mods, n_codes, insertions, haplotypes, wildcard;
    const DevMod* mods;
    const DevCode* codes;
    // outputs
    int32_t* status;               // per read
    unsigned int* err_summary;     // min over failing reads of (read index << 8 | code), 0xFFFFFFFF = none
    unsigned int* host_flag;       // pinned host word, set to 0 by any failing read: the host copies err_summary back only then
    SideRec* side;                 // the rare updates that do not even fit a side key (haplotype tags above 29, positions past 2^35)
    unsigned long long* side_count;
    unsigned long long side_cap;
    // updates that do not fit the dense planes (inside an insertion, haplotype or code without a plane, outside the shard)
    // are appended to lists as (64-bit key, increment) records: side_insert above
    unsigned long long* stab;      // kSideRegions lists of `smask` records of 16 bytes: key, n_called | n_mod << 32 like a dense counter
    unsigned long long smask;      // records per region
    unsigned int* scur;            // [kSideRegions * kSideCurStride] records appended to each region (more than smask: it ran full)
    unsigned long long* stats;     // optional: [0..15] diagnostic timers, then kStatSlots rows of {reference-word lookups, ML bytes
                                   // read, dense updates, side updates}, one row per wave slot (summed by the host)
    // scheduling / scratch
    unsigned int* queue;
    uint32_t* spill;               // per wave slot: [max_cig] q, [max_cig] r, [max_blk] dir
    uint32_t spill_cig, spill_blk;
    // `minimod view` (add_view_entry, mod.c:931-946): every call that passes the context test becomes a 16-byte record
    // instead of a counter update.  Records are appended to kViewRegions independent regions (one append counter each,
    // 128 bytes apart: a single counter would serialise every wave of the launch on one address).
    int32_t view;
    unsigned int view_cap;           // records per region
    unsigned long long* view_keys;   // [kViewRegions][view_cap]  prob << 56 | read << 28 | (ref_pos - read.pos + 1)
    unsigned long long* view_vals;   // [kViewRegions][view_cap]  code << 56 | ins_offset << 40 | group << 29 | implicit << 28 | fastq read_pos
    unsigned int* view_count;        // [kViewRegions * kViewCountStride]; counts past view_cap mean "grow and run again"
    unsigned int* view_read_count;   // [n_reads] records per read (sizes the per-read segments of the ordering pass)
    unsigned int* view_seq;          // [kViewRegions][view_cap] the record's place among its read's records when the kernel that made it knows
                                     // (k_stream_reads: a wavefront makes a read's records in order), else 0xFFFFFFFF
    // the control words (queue, error summary, tile counters) the slot's NEXT launch will use: reset by this launch's
    // last kernel, which saves a host-to-device copy per batch (a slot alternates between two sets)
    unsigned int* ctl_next;
    unsigned int* queue_next;      // likewise: the 64 tile-queue + 64 scan-queue + 64 stream-queue counters (kQueueStride words apart) of the next launch
};
// Is position g of the reference-word space inside a match of context class `cls` on the read's strand?  The reference word's own bit for the thirteen
// classes of the words' pass (kPassContexts), the class's site word for the classes behind them (round 6: a run may name more than thirteen different
// contexts); every position of a dense class -- the context `*` -- is one.  (k_stream_reads asks the site word for every class.)
__device__ __forceinline__ bool class_context_bit(const DevParams& p, uint32_t w, int cls, int rev, int64_t g) {
    if (cls < kPassContexts) return (w >> (5 + 2 * cls + (rev ? 1 : 0))) & 1u;
    const DevClass k = p.classes[cls];
    if (k.dense) return true;
    const uint2* site = rev ? k.site[1] : k.site[0];
    return (site[(g >> 5) * k.stride].x >> ((uint32_t)g & 31u)) & 1u;
}
constexpr int kCtlWords = 80;    // words of a control set that are reset: [0..8) scalars, [8..72) tile counts
constexpr int kCtlSetWords = 128;
constexpr int kQueueStride = 32; // the 64 tile-queue counters lie 128 bytes apart (atomics on one line serialise)
constexpr uint32_t kViewRegions = 64;
constexpr uint32_t kViewCountStride = 32;
constexpr uint32_t kViewMaxGroup = 2047;   // group ordinals that fit the record; a read with more MM groups fails loudly

constexpr unsigned long long kSideEmpty = ~0ull;
// key of a side update, ordered like the output inside one contig: position in the reference-word space (35 bits) |
// strand | code (6) | ins_offset (16) | haplotype (5: the tag's value, 31 = none / '*')
__device__ __forceinline__ bool side_key(int64_t rpos, int rev, int code, uint32_t ins, int hp, unsigned long long& key) {
    if (rpos < 0 || rpos >= (1ll << 35) || code < 0 || code >= 64 || hp > 29) return false;
    const unsigned long long h5 = hp < 0 ? 31ull : (unsigned long long)hp;
    key = ((unsigned long long)rpos << 28) | ((unsigned long long)(rev & 1) << 27) | ((unsigned long long)code << 21) |
          ((unsigned long long)(ins & 0xFFFFu) << 5) | h5;
    return true;
}
// One counter update that has no dense counter: a 16-byte record (key, packed {n_called, n_mod} increment) appended to one of
// kSideRegions lists -- the lanes of a wave instruction that have one reserve their records with ONE atomic on the region's
// cursor and store them side by side (plain stores of whole lines run several times the rate of scattered atomics, and a
// hash table costs two of those per update: a CAS on the key and an add).  Equal keys are added up when the lists are
// compacted: sorted by key and reduced (sort_kernels.hip.h), at finalize or when a region runs full.  Returns 0, or
// MM_E_SIDEFULL when the region has no room left.
constexpr uint32_t kSideRegions = 64;
constexpr uint32_t kSideCurStride = 32;   // cursors 128 bytes apart
__device__ __forceinline__ int side_insert(unsigned long long* tab, unsigned long long cap_r, unsigned int* cur, unsigned long long key, unsigned long long inc) {
    // (readfirstlane: the region is the wavefront's, and as a scalar the cursor's and the list's addresses are scalar too -- as
    // per-lane pointers the compiler made them at the kernel's start and kept them in scratch: three scratch loads per insert)
    const uint32_t region = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)blockIdx.x * 4u + ((uint32_t)threadIdx.x >> 6)) & (kSideRegions - 1u)));
    const uint64_t m = __ballot(1);
    const int leader = __ffsll((unsigned long long)m) - 1;
    unsigned int base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(cur + region * kSideCurStride, (unsigned int)__popcll(m));
    base = (unsigned int)__shfl((int)base, leader, 64);
    const unsigned long long idx = (unsigned long long)base + (unsigned long long)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    if (idx >= cap_r) return MM_E_SIDEFULL;
    ulonglong2 rec; rec.x = key; rec.y = inc;
    *reinterpret_cast<ulonglong2*>(tab + 2ull * ((unsigned long long)region * cap_r + idx)) = rec;
    return 0;
}

// ---------------------------------------------------------------------------------- wave primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Inclusive prefix sum over the 64 lanes with DPP row shifts + row broadcasts (no LDS traffic): 4 row_shr steps
// inside each 16-lane row, then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)x;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// A read failed: the first failing read of the batch is the minimum of err_summary; the host learns that there is
// something to fetch from a plain store into pinned host memory (complete when the launch's event is).
__device__ __forceinline__ void report_error(const DevParams& p, unsigned int ridx, int e) {
    atomicMin(p.err_summary, (ridx << 8) | (unsigned int)e);
    if (p.host_flag) *p.host_flag = 0u;
}

// Append one view record per active lane (called under divergence: the ballot is the set of lanes with a record).
// The value orders the records of one (read, position): by key (code, ins_offset), then in the order the reference
// would have met them -- MM group, listed calls before implicit ones, rising position in the read as sequenced -- so
// that the smallest value of a key is the entry add_view_entry keeps.
__device__ __forceinline__ void view_append(const DevParams& p, uint32_t region, uint32_t ridx, uint32_t rel_pos, uint32_t fq_pos,
                                            uint32_t ins_off, uint32_t code, uint32_t group, uint32_t implicit, uint32_t prob) {
    uint64_t m = __ballot(1);
    int leader = __ffsll((unsigned long long)m) - 1;
    unsigned int base = 0;
    if (lane_id() == leader) {
        const unsigned int cnt = (unsigned int)__popcll(m);
        base = atomicAdd(p.view_count + region * kViewCountStride, cnt);
        // all lanes of a wave work on one read; only records that found room are counted for the ordering pass
        const unsigned int room = base < p.view_cap ? p.view_cap - base : 0u;
        atomicAdd(p.view_read_count + ridx, cnt < room ? cnt : room);
    }
    base = __shfl(base, leader, 64);
    unsigned int idx = base + (unsigned int)__popcll(m & lanemask_lt());
    if (idx < p.view_cap) {
        size_t at = (size_t)region * p.view_cap + idx;
        p.view_keys[at] = ((unsigned long long)prob << 56) | ((unsigned long long)ridx << 28) | (unsigned long long)rel_pos;
        p.view_vals[at] = ((unsigned long long)code << 56) | ((unsigned long long)(ins_off & 0xFFFFu) << 40) |
                          ((unsigned long long)group << 29) | ((unsigned long long)implicit << 28) | (unsigned long long)fq_pos;
        p.view_seq[at] = 0xFFFFFFFFu;
    }
}
// The same from a wavefront that makes ALL records of a read, in the order the reference meets the calls (k_stream_reads): called
// by every lane (emit: this lane has a record), `seq` = records of the read so far.  A record carries its place in the read, so
// the ordering pass can put it there without cursor atomics and finds the read's records in call order.  Returns the records made.
__device__ __forceinline__ uint32_t view_append_seq(const DevParams& p, uint32_t region, uint32_t ridx, bool emit, uint32_t rel_pos, uint32_t fq_pos,
                                                    uint32_t ins_off, uint32_t code, uint32_t group, uint32_t implicit, uint32_t prob, uint32_t seq) {
    const uint64_t m = __ballot(emit);
    if (!m) return 0u;
    const unsigned int cnt = (unsigned int)__popcll(m);
    unsigned int base = 0;
    if (lane_id() == 0) {
        base = atomicAdd(p.view_count + region * kViewCountStride, cnt);
        const unsigned int room = base < p.view_cap ? p.view_cap - base : 0u;
        atomicAdd(p.view_read_count + ridx, cnt < room ? cnt : room);
    }
    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned int rank = (unsigned int)__popcll(m & lanemask_lt());
    if (emit && base + rank < p.view_cap) {
        const size_t at = (size_t)region * p.view_cap + base + rank;
        p.view_keys[at] = ((unsigned long long)prob << 56) | ((unsigned long long)ridx << 28) | (unsigned long long)rel_pos;
        p.view_vals[at] = ((unsigned long long)code << 56) | ((unsigned long long)(ins_off & 0xFFFFu) << 40) |
                          ((unsigned long long)group << 29) | ((unsigned long long)implicit << 28) | (unsigned long long)fq_pos;
        p.view_seq[at] = seq + rank;
    }
    return cnt;
}
// A wave-uniform address of data that no wave writes during the launch, read through the scalar cache: the value lands in
// scalar registers (no vector register per load in flight, no readfirstlane) and several loads overlap freely.  The
// compiler cannot prove either property for a plain global pointer and would issue vector loads.
template <typename T> using kptr = const __attribute__((address_space(4))) T*;
template <typename T> __device__ __forceinline__ kptr<T> scalar_ptr(const T* p) { return (kptr<T>)(uintptr_t)p; }
template <typename T> __device__ __forceinline__ T scalar_load(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    T out;
    const kptr<uint32_t> s = scalar_ptr(reinterpret_cast<const uint32_t*>(p));
    uint32_t* d = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) d[i] = s[i];
    return out;
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the counter word of (code, haplotype plane, strand) at position g of the reference-word space of contig tid (the general
// form: one lookup of the class's site word; k_stream_reads keeps the per-read parts in scalar registers instead)
__device__ __forceinline__ unsigned long long* counter_word(const DevParams& p, const DevCode& dc, int hpi, int rev, int tid, int64_t g) {
    const int c = p.cls_of_mod[dc.req];
    const DevClass k = p.classes[c];
    int64_t r = g;
    if (!k.dense) { const uint2 w = (rev ? k.site[1] : k.site[0])[(g >> 5) * k.stride]; r = (int64_t)site_rank(w, (uint32_t)g & 31u); }
    return p.counters + k.base + (((int64_t)(hpi * 2 + rev) * k.nsites) + p.adj[((int64_t)tid * p.n_classes + c) * 2 + rev] + r) * k.np + dc.slot;
}
// value of lane `l` (wave-uniform l) as a wave-uniform scalar
__device__ __forceinline__ int lane_val(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ uint32_t lane_valu(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(l)); }

// ---------------------------------------------------------------------------------- nibble helpers
// bit 4n+3 of the result is set iff nibble n of x equals nib
__device__ __forceinline__ uint32_t nib_eq(uint32_t x, uint32_t nib) {
    uint32_t y = x ^ (nib * 0x11111111u);
    uint32_t t = (y & 0x77777777u) + 0x77777777u;
    return ~(t | y) & 0x88888888u;
}
// match bits (bit 4n+3) of the bases of class cls in one word; mod.c:97 base_idx_lookup on seq_nt16_str:
// A(1)->0 C(2)->1 G(4)->2 T(8)->3 N(15)->4, every other code -> 0.
__device__ __forceinline__ uint32_t class_bits(uint32_t x, int cls) {
    if (cls == 1) return nib_eq(x, 2);
    if (cls == 2) return nib_eq(x, 4);
    if (cls == 3) return nib_eq(x, 8);
    if (cls == 4) return nib_eq(x, 15);
    return 0x88888888u & ~(nib_eq(x, 2) | nib_eq(x, 4) | nib_eq(x, 8) | nib_eq(x, 15));
}
// BAM packs base 2j in the HIGH nibble of byte j: swap nibbles inside each byte so nibble n == base n
__device__ __forceinline__ uint32_t base_order(uint32_t w) { return ((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu); }
// mask with bit 4n+3 set for the first `valid` (0..8) bases of a word
__device__ __forceinline__ uint32_t valid_bits(int valid) {
    return valid >= 8 ? 0x88888888u : (valid <= 0 ? 0u : (0x88888888u & ((1u << (4 * valid)) - 1u)));
}
__device__ __forceinline__ int base_class_of_char(int c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        case 'N': case 'n': return 4;
        default: return 0;
    }
}
__device__ __forceinline__ int complement_char(int c) {  // mod.c:98 base_complement_lookup
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'U': return 'A'; case 'N': return 'N'; case 'a': return 't'; case 'c': return 'g';
        case 'g': return 'c'; case 't': return 'a'; case 'u': return 'a'; case 'n': return 'n';
        default: return 0;
    }
}
__device__ __forceinline__ bool valid_base_char(int c) {  // mod.c:95 valid_bases
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'U': case 'N':
        case 'a': case 'c': case 'g': case 't': case 'u': case 'n': return true;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------- K0
// Per-base reference word.  bits 0-4: index of the (upper-cased, U->T) letter in "=ACMGRSVTWYHKDBN" or 16;
// bit 5+2i: position lies inside a forward-context match of mod i; bit 6+2i: inside a reverse-context match.
__device__ __forceinline__ int norm_ref_char(int c) {
    if (c >= 'a' && c <= 'z') c -= 32;
    return c == 'U' ? 'T' : c;
}
__device__ __forceinline__ int nt16_code(int c) {
    switch (c) {
        case '=': return 0; case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4;
        case 'R': return 5; case 'S': return 6; case 'V': return 7; case 'T': return 8; case 'W': return 9;
        case 'Y': return 10; case 'H': return 11; case 'K': return 12; case 'D': return 13; case 'B': return 14;
        case 'N': return 15; default: return 16;
    }
}
// (mods: one entry per context CLASS -- the class's first requested entry; bit 5 + 2c / 6 + 2c: inside a forward / reverse match of class c's context)
__device__ __forceinline__ uint32_t ref_word_bits(const uint8_t* __restrict__ raw, int64_t len, int64_t p, const DevMod* __restrict__ mods, int n_mods) {
    uint32_t w = (uint32_t)nt16_code(norm_ref_char(raw[p]));
    for (int i = 0; i < n_mods; i++) {
        const DevMod& m = mods[i];
        if (m.ctx_is_star) { w |= 3u << (5 + 2 * i); continue; }
        int L = m.ctx_len;
        if (L <= 0) continue;
        bool f = false, r = false;
        for (int64_t s = p - L + 1; s <= p; s++) {
            if (s < 0 || s + L > len) continue;
            bool mf = true, mr = true;
            for (int j = 0; j < L; j++) {
                int c = norm_ref_char(raw[s + j]);
                mf = mf && (c == m.ctx_fwd[j]);
                mr = mr && (c == m.ctx_rev[j]);
            }
            f = f || mf; r = r || mr;
        }
        w |= (f ? 1u : 0u) << (5 + 2 * i);
        w |= (r ? 1u : 0u) << (6 + 2 * i);
    }
    return w;
}
template <typename RefWord>
__global__ __launch_bounds__(256) void k_build_refwords(const uint8_t* __restrict__ raw, int64_t len,
                                                        RefWord* __restrict__ out, const DevMod* __restrict__ mods,
                                                        int n_mods) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < len; p += (int64_t)gridDim.x * blockDim.x)
        out[p] = (RefWord)ref_word_bits(raw, len, p, mods, n_mods);
}

// One requested mod (the usual run): FOUR bits a position, two positions a byte -- a read's calls gather from a quarter of
// the lines 16-bit words take, and those gathers are two thirds of the path's HBM traffic.  Bits 0-1: the base (A C G T);
// bit 2: inside a forward-context match; bit 3: inside a reverse-context match.  Any other reference letter is stored as A:
// the base is only ever compared with the read's where the position lies in a context match, which a letter that is none
// of A C G T never does -- unless the context is `*`, and then the comparison is not made (mod.c:1139-1152).
struct RefNib {};
__device__ __forceinline__ uint32_t ref_nibble(uint32_t w) {
    const uint32_t b = w & 31u;
    return (b == 2u ? 1u : (b == 4u ? 2u : (b == 8u ? 3u : 0u))) | (((w >> 5) & 3u) << 2);
}
// Round 4: SIXTEEN positions a thread (round 3: one byte of output a thread, every letter fetched with byte loads once per window
// that covers it: 485 us for 50 Mb, 0.02 of the HBM roofline).  A thread loads the 48 raw letters around its 16 positions with
// three 16-byte loads (a context is at most 15 letters: the windows that cover its positions begin at most 15 in front of them),
// normalises them once, and works on 48-bit masks: bit k of eq_j = "letter k is the context's j-th"; a window matches where all
// its eq_j, shifted by j, are set; a position lies in a match if one begins at it or up to L - 1 in front of it.  Eight bytes out.
// kLmax: the longest context the instantiation takes (4: CG, CHG, CHH, A ... -- the letters further than three from the thread's
// positions are not looked at; 15: any)
// bit 8k+7 of the result: byte k of x equals the letter
__device__ __forceinline__ uint32_t bytes_eq(uint32_t x, uint32_t letter) {
    const uint32_t y = x ^ (letter * 0x01010101u);
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
// Contexts of up to four letters, a chunk whose 48 letters all lie inside the contig (every chunk but a contig's first and its last
// two): four letters an instruction.  Letters are folded to upper case by clearing bit 5 (only a lower-case letter can become a
// letter that way) and U becomes T; e_j = "the letter is the context's j-th" as a flag in bit 7 of the letter's byte; a window
// begins where all e_j, moved j letters down, are set (v_alignbyte moves letters across the dwords); a position lies in a match
// where a window begins at it or up to L - 1 letters in front of it.
__device__ __forceinline__ uint2 refnibs_chunk_fast(const uint32_t (&w)[12], const DevMod& m, int L) {
    uint32_t f[6];   // letters [p0 - 4, p0 + 20): dwords 3 .. 8 of the 48
#pragma unroll
    for (int d = 0; d < 6; d++) {
        const uint32_t u = w[3 + d] & 0xDFDFDFDFu;
        f[d] = u ^ (bytes_eq(u, 'U') >> 7);   // 'U' ^ 1 = 'T'
    }
    uint32_t inf[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u}, inr[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
    if (!m.ctx_is_star) {
        uint32_t sf[5], sr[5];   // windows beginning at the letters of dwords 0 .. 4 of f (the last one's would need letters this thread has not got)
#pragma unroll
        for (int d = 0; d < 5; d++) { sf[d] = 0x80808080u; sr[d] = 0x80808080u; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < L) {
                const uint32_t cf = (uint32_t)(uint8_t)m.ctx_fwd[j], cr = (uint32_t)(uint8_t)m.ctx_rev[j];
                uint32_t ef[6], er[6];
#pragma unroll
                for (int d = 0; d < 6; d++) { ef[d] = bytes_eq(f[d], cf); er[d] = bytes_eq(f[d], cr); }
#pragma unroll
                for (int d = 0; d < 5; d++) {
                    sf[d] &= j == 0 ? ef[d] : __builtin_amdgcn_alignbyte(ef[d + 1], ef[d], (uint32_t)j);
                    sr[d] &= j == 0 ? er[d] : __builtin_amdgcn_alignbyte(er[d + 1], er[d], (uint32_t)j);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 4; d++) {   // the thread's positions are dwords 1 .. 4 of f
            uint32_t a = sf[d + 1], b2 = sr[d + 1];
#pragma unroll
            for (int k = 1; k < 4; k++) {
                if (k < L) {
                    a |= __builtin_amdgcn_alignbyte(sf[d + 1], sf[d], (uint32_t)(4 - k));
                    b2 |= __builtin_amdgcn_alignbyte(sr[d + 1], sr[d], (uint32_t)(4 - k));
                }
            }
            inf[d] = a; inr[d] = b2;
        }
    }
    uint32_t o[2] = {0u, 0u};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t eC = bytes_eq(f[d + 1], 'C'), eG = bytes_eq(f[d + 1], 'G'), eT = bytes_eq(f[d + 1], 'T');
        // a nibble in the low half of every byte: base (A C G T = 0 1 2 3, any other letter 0), forward match, reverse match
        uint32_t x = ((eC | eT) >> 7) | ((eG | eT) >> 6) | (inf[d] >> 5) | (inr[d] >> 4);
        x = (x | (x >> 4)) & 0x00FF00FFu;
        x = (x | (x >> 8)) & 0xFFFFu;
        o[d >> 1] |= x << (16 * (d & 1));
    }
    return make_uint2(o[0], o[1]);
}

template <int kLmax>
__global__ __launch_bounds__(256) void k_build_refnibs(const uint8_t* __restrict__ raw, int64_t len, uint8_t* __restrict__ out,
                                                       const DevMod* __restrict__ mods) {
    const DevMod& m = mods[0];
    const int L = m.ctx_is_star ? 0 : min(m.ctx_len, kLmax);
    const int64_t n_chunks = (len + 15) >> 4;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = c << 4;
        if (kLmax <= 4 && p0 >= 16 && p0 + 32 <= len) {
            uint32_t w[12];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const uint4 v = *reinterpret_cast<const uint4*>(raw + p0 + 16 * (q - 1));
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
            *reinterpret_cast<uint2*>(out + (p0 >> 1)) = refnibs_chunk_fast(w, m, L);
            continue;
        }
        // letters [p0 - 16, p0 + 32): three aligned 16-byte loads (the staging buffer is allocated with 64 spare bytes; outside the
        // contig a letter is 0, which matches nothing and is no base)
        uint32_t w[12];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int64_t a = p0 + 16 * (q - 1);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a >= 0 && a < len) v = *reinterpret_cast<const uint4*>(raw + a);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        uint64_t in_f = 0, in_r = 0;
        uint32_t base2 = 0;   // two bits a position: A C G T = 0 1 2 3 (any other letter 0)
        uint64_t st_f = ~0ull, st_r = ~0ull;
        for (int j = 0; j < L; j++) {
            const uint32_t cf = (uint32_t)(uint8_t)m.ctx_fwd[j], cr = (uint32_t)(uint8_t)m.ctx_rev[j];
            uint64_t ef = 0, er = 0;
#pragma unroll
            for (int k = 16 - (kLmax - 1); k < 32 + (kLmax - 1); k++) {
                const int64_t pos = p0 - 16 + k;
                uint32_t ch = (w[k >> 2] >> (8 * (k & 3))) & 255u;
                if (ch >= 'a' && ch <= 'z') ch -= 32u;
                if (ch == 'U') ch = 'T';
                if (pos >= len) ch = 0;
                ef |= (uint64_t)(ch == cf) << k; er |= (uint64_t)(ch == cr) << k;
            }
            st_f &= ef >> j; st_r &= er >> j;
        }
        if (L > 0) {
            // a window that begins at letter k must end inside the contig: k + L <= len - (p0 - 16)
            const int64_t room = len - (p0 - 16) - L;          // last letter index a window may begin at
            const uint64_t ok = room < 0 ? 0ull : (room >= 63 ? ~0ull : ((2ull << room) - 1ull));
            st_f &= ok; st_r &= ok;
            for (int k = 0; k < L; k++) { in_f |= st_f << k; in_r |= st_r << k; }
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t ch = (w[4 + (k >> 2)] >> (8 * (k & 3))) & 255u;
            if (ch >= 'a' && ch <= 'z') ch -= 32u;
            const uint32_t b = ch == 'C' ? 1u : (ch == 'G' ? 2u : ((ch == 'T' || ch == 'U') ? 3u : 0u));
            base2 |= b << (2 * k);
        }
        uint32_t o[2] = {0u, 0u};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t nib = (base2 >> (2 * k)) & 3u;
            if (m.ctx_is_star) nib |= 12u;
            else nib |= (uint32_t)((in_f >> (16 + k)) & 1ull) << 2 | (uint32_t)((in_r >> (16 + k)) & 1ull) << 3;
            if (p0 + k >= len) nib = 0u;
            o[k >> 3] |= nib << (4 * (k & 7));
        }
        const int64_t ob = p0 >> 1, nbytes = (len + 1) >> 1;
        if (ob + 8 <= nbytes) *reinterpret_cast<uint2*>(out + ob) = make_uint2(o[0], o[1]);
        else for (int k = 0; ob + k < nbytes; k++) out[ob + k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
    }
}
// a contig's words: from(refw, ref_base) once per read, at(base, position) per call -> the word in the 16 / 32-bit layout
template <typename RefWord>
struct RefLoad {
    typedef const RefWord* Base;
    static __device__ __forceinline__ Base from(const void* refw, int64_t ref_base) { return reinterpret_cast<const RefWord*>(refw) + ref_base; }
    static __device__ __forceinline__ uint32_t at(Base b, int64_t pos) { return (uint32_t)b[pos]; }
};
template <>
struct RefLoad<RefNib> {
    typedef const uint8_t* Base;
    static __device__ __forceinline__ Base from(const void* refw, int64_t ref_base) { return reinterpret_cast<const uint8_t*>(refw) + (ref_base >> 1); }   // (ref_base: a multiple of 64)
    static __device__ __forceinline__ uint32_t at(Base b, int64_t pos) {
        const uint32_t nib = ((uint32_t)b[pos >> 1] >> (4u * ((uint32_t)pos & 1u))) & 15u;
        return (1u << (nib & 3u)) | ((nib >> 2) << 5);
    }
};

// ---------------------------------------------------------------------------------- K1
struct WaveLds {
    uint32_t cig_q[kCigCap];  // query offset at the start of op i
    uint32_t cig_r[kCigCap];  // reference offset at the start of op i | op << 28
    uint32_t dir[kDirCap];    // # bases of the current class before 32-base block b
    uint32_t tok[kTokCap];    // compacted skip counts
    uint32_t gap[64];         // exclusive prefix of skip counts (implicit-call expansion)
    uint32_t gstart[64];      // first rank of each gap
    uint32_t mmw[68];         // 256 MM characters + 16 of look-ahead, written as dwords
    char hdr[16];             // code characters of the current MM group
    int16_t g_code[16];       // per code letter: device code index or -1
};

struct ReadCtx {
    // wave-uniform
    const uint8_t* seq;
    const uint8_t* ml;
    const void* refw;
    uint32_t* spill_q; uint32_t* spill_r; uint32_t* spill_d;
    int64_t ref_base, seg_begin, seg_len, cnt_base;
    uint32_t L, ncig, nblk, q_total, ml_len;
    uint32_t q_shift;   // reverse read whose CIGAR is shorter than its sequence: get_aln walks the ops back to front from read position 0
                        // of the ORIGINAL orientation (mod.c:813-860), so the aligned part lies at BAM positions [q_shift, L)
    int32_t tid, pos, rev, hp, hpi;
    // current MM group
    int32_t cls, direct, mb_is_N, n_codes_grp;
    int32_t code_off;   // letters of the group in front of the 16 whose codes S.g_code holds (0 unless the group has more than 16)
    uint32_t nb, ml_start;
    uint32_t ridx, gord, vregion;   // view mode: read index, ordinal of the current MM group, append region
};

template <typename RefWord, bool kView>
struct K1 {
    const DevParams& p;
    WaveLds& S;
    ReadCtx c;
    int err;
    uint32_t st_look, st_ml, st_dense, st_side;  // per-lane tallies, flushed once per read when p.stats is set

    __device__ K1(const DevParams& p_, WaveLds& s_) : p(p_), S(s_), err(0), st_look(0), st_ml(0), st_dense(0), st_side(0) {}

    __device__ __forceinline__ uint32_t cq(uint32_t i) const { return i < (uint32_t)kCigCap ? S.cig_q[i] : c.spill_q[i - kCigCap]; }
    __device__ __forceinline__ uint32_t cr(uint32_t i) const { return i < (uint32_t)kCigCap ? S.cig_r[i] : c.spill_r[i - kCigCap]; }
    __device__ __forceinline__ uint32_t dr(uint32_t b) const { return b < (uint32_t)kDirCap ? S.dir[b] : c.spill_d[b - kDirCap]; }

    // ---- a4 get_aln (mod.c:776-881) as prefix sums: no aln[] array
    __device__ void scan_cigar(const uint32_t* cg, int64_t ctg_len) {
        const int lane = lane_id();
        uint32_t carry_q = 0, carry_r = 0;
        // 256 ops per trip: the four loads are issued before any of them is consumed (memory-level parallelism;
        // one wave alone otherwise pays a full HBM round trip per 64 ops)
        for (uint32_t i0 = 0; i0 < c.ncig; i0 += 256) {
            uint32_t wv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t i = i0 + 64u * u + lane;
                wv[u] = i < c.ncig ? cg[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t i = i0 + 64u * u + lane;
                bool act = i < c.ncig;
                uint32_t w = wv[u];
                uint32_t op = w & 15u, len = w >> 4;
                // ops MIDNSHP=X = 0..8 ; query-consuming {M,I,S,=,X}, reference-consuming {M,D,N,=,X}
                uint32_t qinc = (act && ((0x193u >> op) & 1u)) ? len : 0u;
                uint32_t rinc = (act && ((0x18Du >> op) & 1u)) ? len : 0u;
                if (act && op == 5u) err = MM_E_HARDCLIP;                     // mod.c:841-844
                else if (act && (op == 6u || op > 8u)) err = MM_E_CIGAROP;    // mod.c:845-848
                uint32_t qs = wave_incl_scan(qinc), rs = wave_incl_scan(rinc);
                uint32_t qtot = lane_valu(qs, 63), rtot = lane_valu(rs, 63);
                qs = carry_q + qs - qinc;
                rs = carry_r + rs - rinc;
                bool aligned = act && ((0x181u >> op) & 1u) && len > 0;
                // (read_pos < seq_len is asserted for aligned bases -- and inserted ones with --insertions -- as get_aln MEETS them: a
                // forward read's ops front to back, so the op's end in stored order decides; a reverse read's back to front: below)
                if (aligned) {
                    if (!c.rev && (uint64_t)qs + len > c.L) err = err ? err : MM_E_QOVER;              // mod.c:853
                    int64_t r0 = (int64_t)c.pos + rs;
                    if (r0 < 0 || r0 + (int64_t)len > ctg_len) err = err ? err : MM_E_REFPOS;          // mod.c:860
                }
                if (!c.rev && p.insertions && act && op == 1u && len > 0 && (uint64_t)qs + len > c.L) err = err ? err : MM_E_QOVER;  // mod.c:865
                if ((uint64_t)carry_r + rtot >= (1u << 28)) err = err ? err : MM_E_REFPOS;
                if (act) {
                    uint32_t rv = (rs & 0x0FFFFFFFu) | (op << 28);
                    if (i < (uint32_t)kCigCap) { S.cig_q[i] = qs; S.cig_r[i] = rv; }
                    else { c.spill_q[i - kCigCap] = qs; c.spill_r[i - kCigCap] = rv; }
                }
                carry_q += qtot; carry_r += rtot;
            }
        }
        c.q_total = carry_q;
        // a reverse read's BAM position q is query position q - (L - q_total) of the CIGAR (get_aln walks its ops back to front from
        // read position 0 of the original orientation, mod.c:813-860); the difference wraps when the CIGAR is the longer one
        c.q_shift = c.rev ? c.L - carry_q : 0u;
        wave_sync();
        if (c.rev && carry_q > c.L) {
            // A reverse read whose CIGAR consumes more than the sequence has: walked back to front, an aligned (or, with
            // --insertions, inserted) base is out of range when its position counted from the CIGAR's END reaches seq_len -- the op
            // that starts at stored query position qs ends at q_total - qs in that walk.  Excess in the ops the walk meets last
            // (the stored CIGAR's first: a leading soft clip) is never looked at (mod.c:813-860).
            for (uint32_t i0 = 0; i0 < c.ncig; i0 += 64) {
                const uint32_t i = i0 + (uint32_t)lane;
                if (i < c.ncig) {
                    const uint32_t op = cr(i) >> 28, qs = cq(i);
                    const uint32_t len = cg[i] >> 4;
                    const bool counts = (((0x181u >> op) & 1u) || (p.insertions && op == 1u)) && len > 0;
                    if (counts && carry_q - qs > c.L) err = err ? err : MM_E_QOVER;
                }
            }
        }
    }

    // ---- a5 base directory (mod.c:972-981) as a rank directory over 32-base blocks
    __device__ __forceinline__ uint32_t count_block(uint4 v, int cls, uint32_t b) const {
        int valid = (int)min(32u, c.L - b * 32u);
        if (cls == 0) {
            uint32_t o = 0;
            o += __popc(nib_eq(v.x, 2) | nib_eq(v.x, 4) | nib_eq(v.x, 8) | nib_eq(v.x, 15));
            o += __popc(nib_eq(v.y, 2) | nib_eq(v.y, 4) | nib_eq(v.y, 8) | nib_eq(v.y, 15));
            o += __popc(nib_eq(v.z, 2) | nib_eq(v.z, 4) | nib_eq(v.z, 8) | nib_eq(v.z, 15));
            o += __popc(nib_eq(v.w, 2) | nib_eq(v.w, 4) | nib_eq(v.w, 8) | nib_eq(v.w, 15));
            return (uint32_t)valid - o;  // zero padding nibbles never match 2/4/8/15
        }
        return __popc(class_bits(v.x, cls)) + __popc(class_bits(v.y, cls)) + __popc(class_bits(v.z, cls)) +
               __popc(class_bits(v.w, cls));
    }
    __device__ void build_dir(int cls) {
        const int lane = lane_id();
        const uint4* sq = reinterpret_cast<const uint4*>(c.seq);
        uint32_t carry = 0;
        // 256 blocks (8192 bases) per trip, four 16-byte loads in flight per lane
        for (uint32_t b0 = 0; b0 < c.nblk; b0 += 256) {
            uint4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t b = b0 + 64u * u + lane;
                vv[u] = b < c.nblk ? sq[b] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t b = b0 + 64u * u + lane;
                uint32_t cnt = b < c.nblk ? count_block(vv[u], cls, b) : 0u;
                uint32_t incl = wave_incl_scan(cnt);
                uint32_t tot = lane_valu(incl, 63);
                if (b < c.nblk) {
                    uint32_t ex = carry + incl - cnt;
                    if (b < (uint32_t)kDirCap) S.dir[b] = ex; else c.spill_d[b - kDirCap] = ex;
                }
                carry += tot;
            }
        }
        c.nb = carry;
        c.cls = cls;
        wave_sync();
    }

    // rank (0-based, BAM orientation) of a base of the current class -> its 32-base block (largest b with dir[b] <= rr)
    __device__ __forceinline__ uint32_t find_block(uint32_t rr) const {
        uint32_t lo = 0, step = 1;
        while (step < c.nblk) step <<= 1;
        for (step >>= 1; step; step >>= 1) {
            uint32_t cand = lo + step;
            if (cand < c.nblk && dr(cand) <= rr) lo = cand;
        }
        return lo;
    }
    // k-th base of the current class inside block `blk` (16 bytes v) -> BAM index q, nt16 code
    __device__ __forceinline__ uint32_t select_in_block(uint4 v, uint32_t blk, uint32_t k, uint32_t& code) const {
        int valid = (int)min(32u, c.L - blk * 32u);
        uint32_t w0 = base_order(v.x), w1 = base_order(v.y), w2 = base_order(v.z), w3 = base_order(v.w);
        uint32_t m0 = class_bits(w0, c.cls) & valid_bits(valid);
        uint32_t m1 = class_bits(w1, c.cls) & valid_bits(valid - 8);
        uint32_t m2 = class_bits(w2, c.cls) & valid_bits(valid - 16);
        uint32_t m3 = class_bits(w3, c.cls) & valid_bits(valid - 24);
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

    // BAM index q -> CIGAR op containing it (largest i with cq(i) <= q); only valid for q < q_total
    __device__ __forceinline__ uint32_t find_op(uint32_t q) const {
        uint32_t lo = 0, step = 1;
        while (step < c.ncig) step <<= 1;
        for (step >>= 1; step; step >>= 1) {
            uint32_t cand = lo + step;
            if (cand < c.ncig && cq(cand) <= q) lo = cand;
        }
        return lo;
    }

    __device__ __forceinline__ void side_append(int32_t pos, uint32_t ins_off, int is_mod, int code) {
        unsigned long long key;
        if (side_key(c.ref_base + pos, c.rev, code, ins_off, c.hp, key)) {
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
            r.tid = c.tid; r.pos = pos; r.ins_off = (uint16_t)ins_off; r.strand = (uint8_t)c.rev;
            r.is_mod = (uint8_t)is_mod; r.code = (int16_t)code; r.hp = (int16_t)c.hp;
            p.side[idx] = r;
        } else {
            err = MM_E_SIDEFULL;
        }
    }

    // ---- a7/a8/a9: J candidate calls per lane, run as a staged pipeline so that the J dependent chains
    //      rank -> seq block (global) -> read position -> CIGAR op (LDS) -> reference word (global) -> ML (global) -> atomic
    //      overlap their memory latency.  (mod.c:1097-1199 explicit, :1203-1367 implicit, update_freq_map :883-929)
    template <int J>
    __device__ __forceinline__ void process_calls(const uint32_t (&rank)[J], const uint32_t (&kidx)[J], const bool (&live_in)[J], bool is_explicit) {
        bool live[J];
        uint32_t blk[J], kk[J], q[J], code[J], ins_off[J];
        int64_t ref_pos[J];
        uint4 sv[J];
        // stage 1: block of the rank directory (LDS), or the direct position for canonical base N (mod.c:1102-1107)
#pragma unroll
        for (int u = 0; u < J; u++) {
            live[u] = live_in[u];
            blk[u] = 0; kk[u] = 0; q[u] = 0; code[u] = 0; ins_off[u] = 0; ref_pos[u] = -1;
            if (!live[u]) continue;
            if (c.direct) {
                if (rank[u] >= c.L) { err = MM_E_READPOS; live[u] = false; continue; }
                q[u] = c.rev ? c.L - 1 - rank[u] : rank[u];
            } else {
                if (rank[u] >= c.nb) { err = MM_E_READPOS; live[u] = false; continue; }  // the reference reads out of bounds here
                uint32_t rr = c.rev ? c.nb - 1 - rank[u] : rank[u];
                blk[u] = find_block(rr);
                kk[u] = rr - dr(blk[u]);
            }
        }
        // stage 2: the 16 sequence bytes of each block (global; all J loads in flight together)
#pragma unroll
        for (int u = 0; u < J; u++) {
            sv[u] = make_uint4(0, 0, 0, 0);
            if (live[u]) sv[u] = c.direct ? make_uint4(c.seq[q[u] >> 1], 0, 0, 0) : reinterpret_cast<const uint4*>(c.seq)[blk[u]];
        }
        // stage 3: read position + base, then projection through the CIGAR prefix arrays (LDS)
#pragma unroll
        for (int u = 0; u < J; u++) {
            if (!live[u]) continue;
            if (c.direct) { uint32_t b = sv[u].x; code[u] = (q[u] & 1u) ? (b & 15u) : (b >> 4); }
            else q[u] = select_in_block(sv[u], blk[u], kk[u], code[u]);
            // (a') of SURVEY.md: proj(q) for aligned bases; with --insertions the anchor insL() left of the insertion
            int64_t rp = -1, anchor = -1;
            const uint32_t qe = q[u] - c.q_shift;   // (wraps past q_total for the bases in front of the aligned part)
            if (qe < c.q_total) {
                uint32_t i = find_op(qe);
                uint32_t rv = cr(i), op = rv >> 28, qs = cq(i);
                if ((0x181u >> op) & 1u) {
                    rp = (int64_t)c.pos + (rv & 0x0FFFFFFFu) + (qe - qs);
                } else if (op == 1u && p.insertions) {
                    ins_off[u] = (qe - qs + 1u) & 0xFFFFu;  // ins_offset, truncated like make_key's uint16 (mod.c:428)
                    anchor = (int64_t)c.pos + (rv & 0x0FFFFFFFu) - 1;
                }
            }
            if (rp < 0 && p.insertions) {
                if (is_explicit || !c.rev) {
                    rp = anchor;                                             // mod.c:1124
                } else {
                    // quirk (mod.c:1234,1314): the implicit path indexes ins[] with the BAM-orientation position,
                    // i.e. for reverse reads it takes the insertion anchor of the MIRRORED base L-1-q.
                    uint32_t q2 = c.L - 1u - q[u] - c.q_shift;
                    if (q2 < c.q_total) {
                        uint32_t i2 = find_op(q2);
                        uint32_t rv2 = cr(i2);
                        if ((rv2 >> 28) == 1u) rp = (int64_t)c.pos + (rv2 & 0x0FFFFFFFu) - 1;
                    }
                }
            }
            ref_pos[u] = rp;
            if (rp < 0) live[u] = false;
        }
        // stage 4: reference words and the first code's ML byte (global, J + J loads in flight)
        uint32_t w[J], ml0[J];
        const typename RefLoad<RefWord>::Base rw = RefLoad<RefWord>::from(c.refw, c.ref_base);
        const int ncg = c.n_codes_grp;
#pragma unroll
        for (int u = 0; u < J; u++) {
            w[u] = 0; ml0[u] = 0;
            if (!live[u]) continue;
            w[u] = RefLoad<RefWord>::at(rw, ref_pos[u]);
            st_look++;
            if (is_explicit) {
                uint64_t mi = (uint64_t)c.ml_start + (uint64_t)kidx[u] * ncg;
                if (mi < c.ml_len) ml0[u] = c.ml[mi];
            }
        }
        // stage 5: per code: context + base test, threshold class, counter update
#pragma unroll
        for (int u = 0; u < J; u++) {
            if (!live[u]) continue;
            uint32_t refcode = w[u] & 31u;
            for (int m = 0; m < ncg; m++) {
                int ci = m >= c.code_off ? S.g_code[m - c.code_off] : -1;
                if (ci < 0) continue;
                const DevCode& dc = p.codes[ci];
                int req = dc.req;
                const DevMod& dm = p.mods[req];
                if (!p.insertions) {
                    const bool in_ctx = class_context_bit(p, w[u], p.cls_of_mod[req], c.rev, c.ref_base + ref_pos[u]);   // (the bits are the context class's: entries with one context share them; classes 13 and up: the site word)
                    bool matches = dm.ctx_is_star || c.mb_is_N || refcode == code[u];
                    if (!(in_ctx && matches)) continue;
                }
                int is_mod = 0;
                if (is_explicit) {
                    uint64_t ml_idx = (uint64_t)c.ml_start + (uint64_t)kidx[u] * ncg + m;
                    if (ml_idx >= c.ml_len) { err = MM_E_MLIDX; break; }  // mod.c:1174
                    int mv = m == 0 ? (int)ml0[u] : (int)c.ml[ml_idx];
                    st_ml++;
                    if (kView) {                              // mod.c:1194-1196: no threshold, the ML byte itself
                        view_append(p, c.vregion, c.ridx, (uint32_t)(ref_pos[u] - c.pos + 1), c.rev ? c.L - 1u - q[u] : q[u], ins_off[u],
                                    (uint32_t)ci, c.gord, 0u, (uint32_t)mv);
                        continue;
                    }
                    if (mv >= dm.t_hi) is_mod = 1;            // mod.c:1184
                    else if (mv <= dm.t_lo) is_mod = 0;        // mod.c:1187
                    else continue;                             // ambiguous
                } else if (kView) {                           // mod.c:1281-1283, :1361-1363: implicit calls carry 0
                    view_append(p, c.vregion, c.ridx, (uint32_t)(ref_pos[u] - c.pos + 1), c.rev ? c.L - 1u - q[u] : q[u], ins_off[u],
                                (uint32_t)ci, c.gord, 1u, 0u);
                    continue;
                }
                int64_t off = ref_pos[u] - c.seg_begin;
                if (ins_off[u] == 0 && dc.plane >= 0 && c.hpi >= 0 && off >= 0 && off < c.seg_len) {
                    atomicAdd(counter_word(p, dc, c.hpi, c.rev, c.tid, c.ref_base + ref_pos[u]), is_mod ? 0x100000001ull : 1ull);
                    st_dense++;
                } else {
                    side_append((int32_t)ref_pos[u], ins_off[u], is_mod, ci);
                    st_side++;
                }
            }
        }
    }

    // four MM characters starting at byte offset `off`; characters at or past the end of the string read as ';'
    __device__ __forceinline__ uint32_t load_mm_dword(const uint8_t* mm, uint32_t mlen, uint32_t off) const {
        uint32_t w = 0x3B3B3B3Bu;
        if (off < mlen) {
            uint32_t raw;
            __builtin_memcpy(&raw, mm + off, 4);   // unaligned dword; the pool is padded past the string
            uint32_t left = mlen - off;            // valid characters in this dword (>= 1)
            uint32_t keep = left >= 4u ? 0xFFFFFFFFu : ((1u << (8u * left)) - 1u);
            w = (raw & keep) | (0x3B3B3B3Bu & ~keep);
        }
        return w;
    }

    // the rank directory is built lazily, when a group is about to make its first call
    __device__ __forceinline__ void ensure_dir(int cls, bool dot) {
        if ((!c.direct || dot) && cls != c.cls) build_dir(cls);
    }

    // compacted skip counts -> calls; cnt <= 256 tokens of the current group (token t = 64*u + lane)
    __device__ void flush_tokens(uint32_t cnt, uint32_t& rank_carry, uint32_t& k_carry, bool dot) {
        const int lane = lane_id();
        constexpr int J = kCallsPerLane;
        uint32_t s[J], rank[J], kidx[J];
        bool live[J];
        uint32_t carry = rank_carry;
#pragma unroll
        for (int u = 0; u < J; u++) {
            uint32_t t = 64u * u + lane;
            live[u] = t < cnt;
            s[u] = live[u] ? S.tok[t] : 0u;
            uint32_t incl = wave_incl_scan(live[u] ? s[u] + 1u : 0u);
            rank[u] = carry + incl - 1u;
            kidx[u] = k_carry + t;
            carry += lane_valu(incl, 63);
        }
        process_calls<J>(rank, kidx, live, true);
        if (dot) {
            // implicit calls: every rank inside the gap in front of each listed one (mod.c:1206-1287), expanded
            // 64 tokens at a time by a load-balanced search over the exclusive prefix of the gap sizes.  The skip
            // counts are re-read from LDS (they are still in tok[]) so that nothing is indexed dynamically in registers.
            uint32_t carry2 = rank_carry;
#pragma unroll 1
            for (uint32_t t64 = 0; t64 < cnt; t64 += 64u) {
                uint32_t t = t64 + lane;
                bool lv = t < cnt;
                uint32_t su = lv ? S.tok[t] : 0u;
                uint32_t incl = wave_incl_scan(lv ? su + 1u : 0u);
                uint32_t ranku = carry2 + incl - 1u;
                carry2 += lane_valu(incl, 63);
                uint32_t gi = wave_incl_scan(su);
                uint32_t T = lane_valu(gi, 63);
                wave_sync();
                S.gap[lane] = gi - su;
                S.gstart[lane] = ranku - su;
                wave_sync();
                for (uint32_t t0 = 0; t0 < T; t0 += 64u * J) {
                    uint32_t r2[J], k2[J];
                    bool l2[J];
#pragma unroll
                    for (int v = 0; v < J; v++) {
                        uint32_t tt = t0 + 64u * v + lane;
                        l2[v] = tt < T; r2[v] = 0; k2[v] = 0;
                        if (l2[v]) {
                            uint32_t lo = 0;
#pragma unroll
                            for (uint32_t step = 32; step; step >>= 1) {
                                uint32_t cand = lo + step;
                                if (cand < 64u && S.gap[cand] <= tt) lo = cand;
                            }
                            r2[v] = S.gstart[lo] + (tt - S.gap[lo]);
                        }
                    }
                    process_calls<J>(r2, k2, l2, false);
                }
            }
        }
        rank_carry = carry;
        k_carry += cnt;
    }

    // bases after the last listed one of a '.' group (mod.c:1289-1365)
    __device__ void implicit_tail(uint32_t first_rank) {
        const int lane = lane_id();
        constexpr int J = kCallsPerLane;
        for (uint32_t r0 = first_rank; r0 < c.nb; r0 += 64u * J) {
            uint32_t r2[J], k2[J];
            bool l2[J];
#pragma unroll
            for (int v = 0; v < J; v++) { r2[v] = r0 + 64u * v + lane; k2[v] = 0; l2[v] = r2[v] < c.nb; }
            process_calls<J>(r2, k2, l2, false);
        }
    }

    __device__ void flush_stats(uint32_t stat_slot) {
        uint32_t v[4] = {st_look, st_ml, st_dense, st_side};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t x = wave_incl_scan(v[i]);
            uint32_t t = lane_valu(x, 63);
            if (lane_id() == 0 && t) p.stats[16 + 4 * (size_t)stat_slot + i] += (unsigned long long)t;   // this wave owns the row: no atomics
        }
        st_look = st_ml = st_dense = st_side = 0;
    }

    // wave-uniform error code of the read so far (0 = none); no side effects
    __device__ __forceinline__ int any_err() const {
        uint64_t eb = __ballot(err != 0);
        int l = eb ? __ffsll((unsigned long long)eb) - 1 : 0;
        int e = lane_val(err, l);
        return eb ? e : 0;
    }

    // ---- freq_view_single (mod.c:948-1370) for one read
    // Returns the read's status code (wave-uniform).  Single exit, no early returns: the caller reports errors.
    __device__ int run(int ridx, int wave_slot, uint32_t part, uint32_t nparts) {
        const int lane = lane_id();
        const mm_read_t& rd = p.reads[ridx];
        err = 0;
        c.tid = uni(rd.tid); c.pos = uni(rd.pos);
        c.L = uniu(rd.l_qseq); c.ncig = uniu(rd.n_cigar);
        c.rev = (uni(rd.flag) & 0x10) ? 1 : 0;
        c.ml_len = uniu(rd.ml_len);
        const uint32_t mlen = uniu(rd.mm_len);
        c.seq = p.seq + rd.seq_off;
        c.ml = p.ml + rd.ml_off;
        const uint8_t* mm = p.mm + rd.mm_off;
        c.nblk = (c.L + 31u) >> 5;
        c.hp = p.haplotypes ? (int)rd.hp : -1;
        c.hpi = p.haplotypes ? ((int)rd.hp < p.n_hp ? (int)rd.hp : -1) : 0;
        c.refw = p.refw;
        uint32_t* sp = p.spill + (size_t)wave_slot * (2u * p.spill_cig + p.spill_blk);
        c.spill_q = sp; c.spill_r = sp + p.spill_cig; c.spill_d = sp + 2u * p.spill_cig;
        int result = 0;
        bool have_ref = c.tid >= 0 && c.tid < p.n_contigs;
        if (have_ref) have_ref = p.ref_base[c.tid] >= 0;
        if (!have_ref) {  // mod.c:793
            result = MM_E_NOCONTIG;
        } else {
        c.ref_base = p.ref_base[c.tid];
        c.seg_begin = p.seg_begin[c.tid]; c.seg_len = p.seg_len[c.tid]; c.cnt_base = p.cnt_base[c.tid];
        MMT_DECL;
        scan_cigar(p.cigar + rd.cigar_off, p.ctg_len[c.tid]);
        MMT_LAP(4);
        result = any_err();
        c.cls = -1; c.nb = 0; c.ml_start = 0;
        c.ridx = (uint32_t)ridx; c.gord = 0xFFFFFFFFu; c.vregion = (uint32_t)wave_slot % kViewRegions;
        if (result == 0) {
        // Control flow below is kept wave-uniform and free of `break`: every exit condition is folded into
        // `bad` (a ballot over the lanes' err), which the loop headers test.  (A version with divergent-looking
        // breaks out of the nested loops hung on gfx950 when an error was raised inside the MM loop.)
        uint32_t mpos = 0;
        bool bad = false, finished_part = false;
        const uint32_t own_lo = (uint32_t)(((uint64_t)mlen * part) / nparts);
        const uint32_t own_hi = part + 1u >= nparts ? 0xFFFFFFFFu : (uint32_t)(((uint64_t)mlen * (part + 1u)) / nparts);
        while (mpos < mlen && !bad) {
            // ---------------- a6 group header (mod.c:1003-1062)
            c.gord++;
            uint32_t ci = mpos + lane;
            int ch = ci < mlen ? (int)mm[ci] : 0;
            int c0 = lane_val(ch, 0), c1 = lane_val(ch, 1);
            int herr = 0;
            if (!valid_base_char(c0)) herr = MM_E_MMBASE;
            int modbase = c0 == 'U' ? 'T' : c0;
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
            bool has_nums = __ballot(iscode && dig) != 0, has_alpha = __ballot(iscode && alp) != 0;
            int n = has_nums ? 1 : ncode;
            if (!herr && __ballot(iscode && !dig && !alp)) herr = MM_E_MMCODE;      // mod.c:1029-1032
            // More code letters than a code string of the table holds (15): the reference grows its buffer and goes on
            // (mod.c:1034-1038) -- n letters, n ML bytes a token; only the suffixes of at most 15 letters can be a requested code
            // (mod.c:1151 looks up the C string from letter m on), so the last 16 letters are all the lookup needs.  Limits kept:
            // a header must end within the 64 characters a wavefront looks at, and with -c '*' every suffix would have to be a
            // code of the table.
            const bool longcode = ncode >= MM_CODE_LEN;
            const int coff = ncode > 16 ? ncode - 16 : 0;
            if (!herr && (e == 64 || (longcode && p.wildcard))) herr = MM_E_MMCODE;
            if (!herr && n <= 0) herr = MM_E_MMEMPTY;                                // mod.c:1053
            if (!herr && has_nums && has_alpha) herr = MM_E_MMMIXED;                 // mod.c:1054
            if (!herr && kView && c.gord > kViewMaxGroup) herr = MM_E_TOOMANY;
            herr = uni(herr);
            if (herr) {
                err = herr;
                bad = true;
            } else {
                int flag = '.';
                uint32_t cpos = mpos + e;
                if (cpos < mlen) {
                    int ce = lane_val(ch, e);
                    if (ce == '?' || ce == '.') { flag = ce; cpos++; }
                }
                // required-code lookup per code letter (mod.c:1146-1160): the C string starting at letter m
                if (iscode && lane - hl >= coff) S.hdr[lane - hl - coff] = (char)ch;   // (the last 16 code characters)
                if (lane < 16) S.g_code[lane] = -1;
                wave_sync();
                {
                    int pairs = n * p.n_codes;
                    for (int p0 = 0; p0 < pairs; p0 += 64) {
                        int pi = p0 + lane;
                        if (pi < pairs) {
                            int m = pi / p.n_codes, t = pi - m * p.n_codes;
                            int slen = has_nums ? ncode : ncode - m;
                            const DevCode& dc = p.codes[t];
                            bool eq = dc.len == slen && slen < MM_CODE_LEN && m >= coff;
                            if (eq) for (int j = 0; j < slen; j++) eq = eq && (dc.str[j & (MM_CODE_LEN - 1)] == S.hdr[(m + j - coff) & 15]);
                            if (eq) S.g_code[m - coff] = (int16_t)t;
                        }
                    }
                }
                wave_sync();
                if (p.wildcard && lane < n && S.g_code[lane] < 0) err = MM_E_NOCODE;  // the host interns before submit
                bad = __ballot(err != 0) != 0;
                c.n_codes_grp = n; c.code_off = coff;
                int mb = c.rev ? complement_char(modbase) : modbase;
                c.mb_is_N = mb == 'N';
                c.direct = modbase == 'N';
                int cls = base_class_of_char(mb);
                bool dot = flag == '.';
                MMT_LAP(6);

                // ---------------- a6 skip counts (mod.c:1064-1089): 256 characters per trip (one dword per lane, the
                // next trip's dword already in flight), parsed as four 64-character sub-chunks from LDS.
                // A read split into parts owns the sub-chunks whose first character lies in [own_lo, own_hi): chunks
                // before it only advance the token / rank carries, the first chunk after it ends the part.
                uint32_t k_carry = 0, rank_carry = 0, ntok = 0;
                bool prev_delim = true, done = bad, group_end_owned = false;
                uint32_t wd_next = 0;
                bool have_next = false;
                while (!done) {
                    uint32_t wd;
                    if (have_next) wd = wd_next;
                    else wd = load_mm_dword(mm, mlen, cpos + 4u * lane);
                    wd_next = load_mm_dword(mm, mlen, cpos + 256u + 4u * lane);   // look-ahead + next trip
                    wave_sync();
                    S.mmw[lane] = wd;
                    if (lane < 4) S.mmw[64 + lane] = wd_next;   // characters 256..271 = the next trip's first four dwords
                    wave_sync();
                    const uint8_t* mb8 = reinterpret_cast<const uint8_t*>(S.mmw);
                    bool group_closed = false;
#pragma unroll 1
                    for (int sub = 0; sub < 4; sub++) {
                        if (group_closed || done) continue;
                        uint32_t cs = cpos + 64u * sub;
                        if (cs >= own_hi) {          // past this part's share: finish what is pending and stop the read
                            done = true; finished_part = true;
                            continue;
                        }
                        bool own = cs >= own_lo;
                        int x = mb8[64 * sub + lane];
                        uint64_t semi = __ballot(x == ';');
                        int endl = semi ? __ffsll((unsigned long long)semi) - 1 : 64;
                        bool in = lane < endl;
                        int pv = __shfl_up(x, 1, 64);
                        bool pdel = lane == 0 ? prev_delim : (pv == ',');
                        bool tstart = in && x != ',' && pdel;
                        uint32_t v = 0;
                        if (tstart) {
                            // decimal fold over at most 10 look-ahead characters, no early exit (mod.c:1074-1084)
                            bool open = true, nondigit = false;
                            int len = 0;
#pragma unroll
                            for (int j = 0; j < 10; j++) {
                                int d = mb8[64 * sub + lane + j];
                                bool delim = d == ',' || d == ';';
                                open = open && !delim;
                                if (open) {
                                    if (d < '0' || d > '9') nondigit = true;
                                    v = v * 10u + (uint32_t)(d - '0');
                                    len++;
                                }
                            }
                            // a sequential reader meets a non-digit among the first ten characters before it has counted ten
                            if (nondigit) err = MM_E_SKIPVAL;
                            else if (len == 10) err = MM_E_SKIPLEN;                    // assert(l < 10), mod.c:1080
                        }
                        uint64_t tb = __ballot(tstart);
                        uint32_t nt = (uint32_t)__popcll(tb);
                        if (own) {
                            if (tstart) S.tok[ntok + __popcll(tb & lanemask_lt())] = v;
                            ntok += nt;
                        } else {
                            uint32_t sm = wave_incl_scan(tstart ? v + 1u : 0u);
                            rank_carry += lane_valu(sm, 63);
                            k_carry += nt;
                        }
                        if (endl < 64) { group_closed = true; done = true; group_end_owned = own; cpos = cs + (uint32_t)endl + 1u; }
                        else prev_delim = lane_val(x, 63) == ',';
                    }
                    wave_sync();
                    if (!group_closed && !done) { cpos += 256u; have_next = true; }
                    bad = __ballot(err != 0) != 0;
                    if (bad) { done = true; ntok = 0; }
                    MMT_LAP(6);
                    constexpr uint32_t kFlush = 64u * kCallsPerLane;
                    while (ntok >= kFlush || (done && ntok > 0)) {
                        uint32_t cnt = ntok < kFlush ? ntok : kFlush;
                        ensure_dir(cls, dot);
                        MMT_LAP(5);
                        flush_tokens(cnt, rank_carry, k_carry, dot);
                        uint32_t rem = ntok - cnt;   // < 64*J + 128 - 64*J... at most 255
                        uint32_t y0 = (uint32_t)lane < rem ? S.tok[cnt + lane] : 0u;
                        uint32_t y1 = (uint32_t)lane + 64u < rem ? S.tok[cnt + 64u + lane] : 0u;
                        uint32_t y2 = (uint32_t)lane + 128u < rem ? S.tok[cnt + 128u + lane] : 0u;
                        uint32_t y3 = (uint32_t)lane + 192u < rem ? S.tok[cnt + 192u + lane] : 0u;
                        wave_sync();
                        if ((uint32_t)lane < rem) S.tok[lane] = y0;
                        if ((uint32_t)lane + 64u < rem) S.tok[64u + lane] = y1;
                        if ((uint32_t)lane + 128u < rem) S.tok[128u + lane] = y2;
                        if ((uint32_t)lane + 192u < rem) S.tok[192u + lane] = y3;
                        wave_sync();
                        ntok = rem;
                        if (__ballot(err != 0)) { bad = true; done = true; ntok = 0; }
                        MMT_LAP(7);
                    }
                }
                if (!bad && !finished_part) {
                    if (k_carry > 0) c.ml_start += k_carry * (uint32_t)n;                // mod.c:1200
                    if (dot && group_end_owned) {   // bases after the last listed one (mod.c:1289-1365)
                        ensure_dir(cls, dot);
                        implicit_tail(rank_carry);
                    }
                    bad = __ballot(err != 0) != 0;
                    mpos = cpos;
                }
                if (finished_part) bad = bad || true;   // leaves the group loop; `result` comes from any_err()
            }
        }
        result = any_err();
        }  // cigar ok
        }  // have_ref
        return result;
    }
};

template <typename RefWord, bool kView>
__global__ __launch_bounds__(256, 3) void k_freq_reads(const DevParams p) {
    __shared__ WaveLds lds[kWavesPerBlock];
    const int wv = threadIdx.x >> 6;
    const int wave_slot = blockIdx.x * kWavesPerBlock + wv;
    K1<RefWord, kView> k(p, lds[wv]);
    if (p.ctl_next && blockIdx.x == 0 && threadIdx.x < kCtlWords) p.ctl_next[threadIdx.x] = threadIdx.x == 1 ? 0xFFFFFFFFu : 0u;
    if (p.queue_next && blockIdx.x == 0 && threadIdx.x < 192) p.queue_next[threadIdx.x * kQueueStride] = 0u;   // tile queues, scan queues, stream queues
    if (p.n_items_dev && *p.n_items_dev == 0u) return;   // empty fallback list: do not even touch the work counter
    for (;;) {
        int r = 0;
        if (lane_id() == 0) r = (int)atomicAdd(p.queue, 1u);
        r = uni(r);
        if (r >= (p.n_items_dev ? (int)*p.n_items_dev : p.n_items)) break;
        uint32_t item = p.order ? (uint32_t)p.order[r] : (uint32_t)r;
        item = uniu(item);
        int ridx = (int)(item & 0xFFFFFFu);
        uint32_t part = (item >> 24) & 15u, nparts = ((item >> 28) & 15u) + 1u;
        int e = uni(k.run(ridx, wave_slot, part, nparts));
        if (p.stats) k.flush_stats((uint32_t)wave_slot & (kStatSlots - 1));
        if (e != 0 && lane_id() == 0) {
            p.status[ridx] = e;
            report_error(p, (unsigned int)ridx, e);
        }
    }
}

// ---------------------------------------------------------------------------------- site index (K0, second half)
// which positions of the reference-word space are sites of a context class: bit (5 + 2 * class) / (6 + 2 * class) of the reference
// words (bits 2 / 3 of the four-bit words), 32 positions a word, and how many a block holds
template <typename RefWord>
__global__ __launch_bounds__(256) void k_site_bits(const void* __restrict__ refw, int64_t n_blocks, int mod, int stride, uint2* __restrict__ fwd,
                                                   uint2* __restrict__ rev, uint32_t* __restrict__ cnt_fwd, uint32_t* __restrict__ cnt_rev) {
    const typename RefLoad<RefWord>::Base rw = RefLoad<RefWord>::from(refw, 0);
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (int64_t)gridDim.x * blockDim.x) {
        uint32_t f = 0, r = 0;
        unsigned long long bases = 0;   // A C G T = 0 1 2 3 (the words' base bits 0-3 are one-hot; other letters never lie in a match)
        for (int k = 0; k < 32; k++) {
            const uint32_t w = RefLoad<RefWord>::at(rw, b * 32 + k);
            f |= ((w >> (5 + 2 * mod)) & 1u) << k;
            r |= ((w >> (6 + 2 * mod)) & 1u) << k;
            const uint32_t b2 = (w & 2u) ? 1u : ((w & 4u) ? 2u : ((w & 8u) ? 3u : 0u));
            bases |= (unsigned long long)b2 << (2 * k);
        }
        fwd[b * stride].x = f; rev[b * stride].x = r;
        if (stride == 2) {
            fwd[b * 2 + 1] = make_uint2((uint32_t)bases, (uint32_t)(bases >> 32));
            rev[b * 2 + 1] = make_uint2((uint32_t)bases, (uint32_t)(bases >> 32));
        }
        cnt_fwd[b] = (uint32_t)__popc(f); cnt_rev[b] = (uint32_t)__popc(r);
    }
}
// exclusive prefix sums of the blocks' counts into the site words: tiles of 2048 blocks -- (1) a tile's total, (2) the totals
// scanned by one workgroup (k_radix_scan), (3) every block's rank from its tile's offset
constexpr int kScanTile = 2048;
__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t* __restrict__ cnt, int64_t n, uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t part[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    uint32_t c = 0;
    for (int j = threadIdx.x; j < kScanTile; j += 256) { const int64_t i = base + j; if (i < n) c += cnt[i]; }
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ cnt, int64_t n, const uint32_t* __restrict__ tile_offsets, uint2* __restrict__ site, int stride) {
    __shared__ uint32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (a vector load at agent scope, not the s_load a wave-uniform address would be: the standalone probe found nothing wrong with the scalar cache --
    // tools/scache_probe.hip, 0 stale words in 288 processes -- but the words were written by the kernel in front, and the load is one of 2 048 a workgroup)
    uint32_t run = __hip_atomic_load(&tile_offsets[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j0 = 0; j0 < kScanTile; j0 += 256) {
        const int64_t i = base + j0 + threadIdx.x;
        const uint32_t c = i < n ? cnt[i] : 0u;
        const uint32_t incl = wave_incl_scan(c);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = run + incl - c;
        for (int w = 0; w < wv; w++) before += wsum[w];
        if (i < n) site[i * stride].y = before;
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}
// the guard behind the index (nothing counts into an index that does not hold): every block's rank plus its own sites is the next block's rank, the first
// rank is 0, the last block ends at the strand's total, and a block's count is its bits' -- bad[0] counts the blocks that break one of these
__global__ __launch_bounds__(256) void k_site_check(const uint2* __restrict__ site, int stride, const uint32_t* __restrict__ cnt, int64_t n_blocks, const uint32_t* __restrict__ total,
                                                    uint32_t* __restrict__ bad) {
    uint32_t wrong = 0;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (int64_t)gridDim.x * blockDim.x) {
        const uint2 w = site[b * stride];
        const uint32_t next = b + 1 < n_blocks ? site[(b + 1) * stride].y : __hip_atomic_load(total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wrong += (w.y + (uint32_t)__popc(w.x) != next) || (b == 0 && w.y != 0u) || (cnt && cnt[b] != (uint32_t)__popc(w.x));
    }
    if (wrong) atomicAdd(bad, wrong);
}
// rank of a few positions (segment boundaries): out[i] = sites of the strand in front of g[i]
__global__ void k_rank_at(const uint2* __restrict__ site, int stride, const int64_t* __restrict__ g, int n, int64_t n_blocks, uint32_t total, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = g[i] >> 5;
    out[i] = b >= n_blocks ? total : site_rank(site[b * stride], (uint32_t)g[i] & 31u);
}

// ---------------------------------------------------------------------------------- K2
// print_freq_output's collect + sort (src/mod.c:644-664 with cmp_key_fast :59-93) as a compaction of the non-zero counters
// that walks POSITIONS: each position's counters in output order (strand, then code plane, then haplotype plane), contigs in
// strcmp order -- finished mm_row_t rows in the order the reference prints them (ties in this build's canonical order).
// Without haplotype planes and side rows the host copies them and is done; otherwise it merges in the side rows and forms
// the `*` aggregates.
constexpr int kTile = 2048;
struct K2Params {
    const unsigned long long* cnt;
    const DevClass* classes;
    const int64_t* adj;
    const int64_t* ref_base;
    int32_t n_classes, n_hp, n_planes, haplotypes;
    int8_t plane_cls[MM_MAX_CODES], plane_slot[MM_MAX_CODES];
};
struct SiteSeg {        // one contig segment, listed in output order
    int64_t seg_begin, seg_len;
    int64_t tile_start; // first tile (of kTile positions) of the segment
    int32_t tid, pad;
};
__device__ __forceinline__ int site_segment(const SiteSeg* segs, int n_seg, int64_t tile) {
    int lo = 0, hi = n_seg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (segs[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}
constexpr int kSitePerThread = kTile / 256;   // consecutive positions per thread
// the rows of position `pos` of contig tid: counted, and with kEmit written from rows[out] on
template <bool kEmit>
__device__ __forceinline__ uint32_t site_rows(const K2Params& P, int tid, int64_t pos, mm_row_t* __restrict__ rows, unsigned long long out) {
    const int64_t g = P.ref_base[tid] + pos;
    uint32_t n = 0;
    for (int strand = 0; strand < 2; strand++) {
        int last_c = -1;
        bool is_site = false;
        int64_t rk = 0;
        for (int pl = 0; pl < P.n_planes; pl++) {
            const int c = P.plane_cls[pl];
            const DevClass& k = P.classes[c];
            if (c != last_c) {
                last_c = c;
                if (k.dense) { is_site = true; rk = g; }
                else { const uint2 w = k.site[strand][(g >> 5) * k.stride]; is_site = (w.x >> ((uint32_t)g & 31u)) & 1u; rk = (int64_t)site_rank(w, (uint32_t)g & 31u); }
            }
            if (!is_site) continue;
            const int64_t a = P.adj[((int64_t)tid * P.n_classes + c) * 2 + strand] + rk;
            for (int hp = 0; hp < P.n_hp; hp++) {
                const unsigned long long v = P.cnt[k.base + (((int64_t)(hp * 2 + strand) * k.nsites) + a) * k.np + P.plane_slot[pl]];
                if (v == 0ull) continue;
                if (kEmit) {
                    mm_row_t r;
                    r.tid = tid; r.pos = (int32_t)pos; r.strand = (uint8_t)strand; r.rsvd = 0; r.ins_offset = 0;
                    r.code = (int16_t)pl; r.hp = (int16_t)(P.haplotypes ? hp : -1); r.n_called = (uint32_t)v; r.n_mod = (uint32_t)(v >> 32);
                    rows[out + n] = r;
                }
                n++;
            }
        }
    }
    return n;
}
__global__ __launch_bounds__(256) void k_site_count(const K2Params P, const SiteSeg* __restrict__ segs, int n_seg, uint32_t* __restrict__ tile_counts) {
    __shared__ uint32_t part[4];
    const SiteSeg sg = segs[site_segment(segs, n_seg, blockIdx.x)];
    const int64_t p0 = ((int64_t)blockIdx.x - sg.tile_start) * kTile + (int64_t)threadIdx.x * kSitePerThread;
    uint32_t c = 0;
    for (int k = 0; k < kSitePerThread; k++) { const int64_t q = p0 + k; if (q < sg.seg_len) c += site_rows<false>(P, sg.tid, sg.seg_begin + q, nullptr, 0ull); }
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_site_emit(const K2Params P, const SiteSeg* __restrict__ segs, int n_seg,
                                                   const unsigned long long* __restrict__ tile_offsets, mm_row_t* __restrict__ rows) {
    __shared__ uint32_t wsum[4];
    const SiteSeg sg = segs[site_segment(segs, n_seg, blockIdx.x)];
    const int64_t p0 = ((int64_t)blockIdx.x - sg.tile_start) * kTile + (int64_t)threadIdx.x * kSitePerThread;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t c = 0;
    for (int k = 0; k < kSitePerThread; k++) { const int64_t q = p0 + k; if (q < sg.seg_len) c += site_rows<false>(P, sg.tid, sg.seg_begin + q, nullptr, 0ull); }
    // exclusive scan of the threads' row counts in position order
    uint32_t incl = wave_incl_scan(c);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (int w = 0; w < wv; w++) before += wsum[w];
    unsigned long long out = tile_offsets[blockIdx.x] + before;
    if (c == 0) return;
    for (int k = 0; k < kSitePerThread; k++) {
        const int64_t q = p0 + k;
        if (q >= sg.seg_len) break;
        out += site_rows<true>(P, sg.tid, sg.seg_begin + q, rows, out);
    }
}

// ---------------------------------------------------------------------------------- halo slabs
// A slab is position-dense whatever the counters' layout: [run][len] words with run = (plane * n_hp + hp) * 2 + strand, zero
// where a position is not a site of the plane's class.  op 0: export to buf, 1: add buf into the counters (two independent
// 32-bit halves; a count on a position that is no site here -- the two sides disagree about the reference -- raises *flag),
// 2: clear.
__global__ void k_slab_op(const K2Params P, int op, int tid, int64_t begin, int64_t len, unsigned long long* __restrict__ cnt,
                          unsigned long long* __restrict__ buf, unsigned int* __restrict__ flag) {
    const int64_t total = (int64_t)P.n_planes * P.n_hp * 2 * len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t run = i / len, j = i - run * len;
        const int strand = (int)(run & 1), hp = (int)((run >> 1) % P.n_hp), pl = (int)((run >> 1) / P.n_hp);
        const int c = P.plane_cls[pl];
        const DevClass& k = P.classes[c];
        const int64_t g = P.ref_base[tid] + begin + j;
        bool is_site = true;
        int64_t rk = g;
        if (!k.dense) { const uint2 w = k.site[strand][(g >> 5) * k.stride]; is_site = (w.x >> ((uint32_t)g & 31u)) & 1u; rk = (int64_t)site_rank(w, (uint32_t)g & 31u); }
        unsigned long long* word = cnt + k.base + (((int64_t)(hp * 2 + strand) * k.nsites) + P.adj[((int64_t)tid * P.n_classes + c) * 2 + strand] + rk) * k.np + P.plane_slot[pl];
        if (op == 0) buf[i] = is_site ? *word : 0ull;
        else if (op == 2) { if (is_site) *word = 0ull; }
        else {
            const unsigned long long v = buf[i];
            if (v) {
                if (!is_site) { if (flag) *flag = 1u; }
                else {
                    const unsigned long long a = *word;
                    const unsigned long long lo = (a & 0xFFFFFFFFull) + (v & 0xFFFFFFFFull), hi = (a >> 32) + (v >> 32);
                    *word = (lo & 0xFFFFFFFFull) | (hi << 32);
                }
            }
        }
    }
}

}  // namespace mmhip
