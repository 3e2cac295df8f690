// ingest_kernels.hip.h -- the decoded BAM stays in HBM: record framing and load_db's flattening on the device (SURVEY section
// 8(f) row 2, "block inflate + record framing into the SoA directly"; the reference does both on the host: sam_read1 behind
// hts_set_threads, src/minimod.c:73-90,250, and load_db's filters and tag extraction, src/minimod.c:235-333 with
// get_mm_tag_ptr / get_ml_tag / get_hp_tag, src/mod.c:123-202).
//
// A GROUP is a run of consecutive BGZF blocks inflated by k_bgzf_inflate (bgzf_kernels.hip.h) into one contiguous piece of the
// decoded stream in device memory, `out[H, H + obytes)`, with H bytes of head room in front for the unfinished record the group
// before it ended in (the TAIL).  Behind the inflate, on one stream, in the order of the file:
//
//   k_frame_spec   a wavefront per BGZF block.  The records of a BAM form a chain -- every record begins with its own length --
//                  so where records begin inside a block is only known once every record in front of it has been walked.  The
//                  walk is made speculative: the wavefront looks for the first offset of its block whose 36 bytes pass for a
//                  record's fixed fields (length, reference ids inside the header's range, positions >= -1, name / CIGAR /
//                  sequence sizes that fit the length) AND from which the chain stays plausible up to the block's end: the
//                  block's CANDIDATE entry, with the number of records the chain found and the offset it left the block at.
//   k_frame_chain  one wavefront strings the blocks together: the true entry of block b is where the chain left block b - 1; it
//                  equals the candidate (always, for files any writer makes), or the block is walked again from the true entry,
//                  one record after the other (the rare path, and the only one that can be wrong about nothing).  Result: per
//                  block its entry, its record count and its first record's index; the group's tail.
//   k_frame_fill   a lane per block writes the offsets of its records.
//   k_rec_parse    a wavefront per record: the fixed fields checked as the host reader checks them (csrc/host/bamio.c
//                  mm_bam_next), load_db's filters, one walk over the tags for the first MM (Z/H), ML (B:C) and HP -- a
//                  malformed tag ends the walk like bam_aux_get's -- into a 32-byte descriptor.
//   k_rec_scan     one workgroup: which records count (a share's range, src/minimod.c has none: csrc/host/loader.c), which are
//                  accepted, their places in the batch's pools (exclusive sums of the 16-byte-aligned sizes, continuing where
//                  the arena's batch ended), the mm_read_t records, the totals the host prints.
//   k_rec_copy     a wavefront per accepted record: CIGAR, sequence, MM text and ML bytes from the decoded stream (any
//                  alignment) into the pools (16-byte aligned items, zero padding, the unused nibble of an odd sequence cleared)
//                  -- byte for byte what csrc/host/loader.c copy_range writes.
//
// The batch that mm_freq_submit_device gets is therefore the one the host loader would have made of the same records, and the
// decoded bytes never leave the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minimod_bgzf.h"
#include "minimod_hip.h"

namespace mmingest {

typedef mm_bgzf_block_t Block;   // c_off, c_len, o_off (decoded bytes of the group's blocks in front), isize, crc

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kStop = 0x80000000u;      // in a block's record count: the chain ended at a record the group does not hold whole
enum { IE_OK = 0, IE_RECORD = 1 /* a record the host reader refuses */, IE_ARENA = 2 /* the batch's pools are full */, IE_TAIL = 3 /* a tail longer than the head room */,
       IE_RECORDS = 4 /* more records than the group's tables hold */, IE_HEADER = 5 /* the BAM header does not end inside the first group */ };

struct Carry {               // what a group leaves for the next one
    uint32_t tail_len;       // bytes of the unfinished record at its end (in d_tail)
    uint32_t skip;           // decoded bytes the next group still has to skip (never with a header that ends in the first group)
    int32_t done, seen;      // a share's range: its end has been passed / an alignment inside it has been seen
    int32_t err, err_at;     // sticky: IE_*, and the record (counted from the file's first) it was met at
    uint64_t n_records;      // records framed so far
};
struct Cursor {              // where the arena's batch ends
    uint64_t n_reads, cigar_bytes, seq_bytes, mm_bytes, ml_bytes, qname_bytes, bases;
    uint32_t max_n_cigar, max_l_qseq;
};
struct Result {              // a group's outcome, copied to the host
    Carry carry;             // as left for the next group
    Cursor cursor;           // the arena's batch including this group
    uint32_t n_records;      // records framed in this group
    uint32_t n_accepted;
    uint32_t n_bad_blocks, first_bad_block;
    uint32_t n_slow_blocks;  // blocks whose candidate entry was not the true one (walked again, one record after the other)
    uint32_t bad_record;     // the group's first record the host reader would refuse (kNone: none)
    uint64_t total_reads, total_bytes, processed_bytes;   // this group's share of the loader's totals (csrc/host/loader.c)
};
struct Desc {                // one framed record as k_rec_parse leaves it
    uint32_t mm_src, mm_len; // MM text in the decoded stream
    uint32_t ml_src, ml_len;
    uint32_t l_data;         // htslib's bam1_t.l_data
    uint32_t n_cigar_lname;  // n_cigar | l_read_name << 16
    uint32_t l_qseq;
    uint32_t flags;          // bit 0 passes the filters, 1 placed (mapped, tid >= 0), 2 in front of the share, 3 at or behind its end; bits 8-15 hp; 16-31 flag
};
struct Params {
    uint8_t* out;            // the group's decoded stream: tail at [H - tail, H), blocks at [H, H + obytes)
    const Block* blocks;
    const int32_t* status;   // per block, from the inflate and CRC kernels
    uint32_t n_blocks, H, obytes;
    int32_t n_ref;
    uint32_t first_skip;     // decoded bytes in front of the file's first record (the header; the offset inside the block of a .bai's virtual offset)
    int32_t is_first;        // the file's (or share's) first group
    // framing tables: n_blocks + 1 entries, entry 0 = the tail's range [H - tail, H)
    uint32_t *cand, *exit_, *cnt;          // per block: candidate entry, where its chain leaves the block, records (| kStop)
    uint32_t *entry, *nrec, *base;         // per block, true
    uint32_t* rec_off; uint32_t max_records;
    Desc* desc;
    uint32_t* acc_rec;       // accepted records of the group, in order: their index in rec_off
    uint32_t* info;          // per record: l_data | counted << 30 | accepted << 31  (the host's batch accounting)
    const Carry* carry_in; Carry* carry_out;
    const Cursor* cursor_in; Cursor* cursor_out;
    uint8_t* tail_out;       // H bytes: the next group's tail
    Result* result;
    // load_db's filters
    int32_t allow_secondary, skip_supplementary;
    int32_t ranged, first, last, lo_tid, hi_tid;
    int64_t lo_pos, hi_pos;
    // the arena
    mm_read_t* reads; uint8_t *cigar, *seq, *mm, *ml;
    uint64_t cap_reads, cap_cigar, cap_seq, cap_mm, cap_ml;   // reads / bytes
    int32_t new_arena;       // the group starts a batch (cursor_in is not looked at)
    // the read names too (view prints them: src/mod.c print_view_output's first column) -- names == nullptr: not kept
    uint8_t* names; uint64_t* name_off; uint64_t cap_names;
};

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }   // any alignment (one dword load)
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t w; __builtin_memcpy(&w, p, 2); return w; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }

// could the 36 bytes at p be a record's length and fixed fields?  (Necessary for every record a BAM writer makes; a record
// that fails it -- a reference id outside the header -- is still framed, by the chain kernel's walk.)
__device__ __forceinline__ bool plausible(const uint8_t* p, int32_t n_ref, uint32_t* bs_out) {
    const uint32_t bs = ld32(p);
    const int32_t tid = (int32_t)ld32(p + 4), pos = (int32_t)ld32(p + 8);
    const uint32_t w3 = ld32(p + 12), w4 = ld32(p + 16);
    const int32_t l_seq = (int32_t)ld32(p + 20), mtid = (int32_t)ld32(p + 24), mpos = (int32_t)ld32(p + 28);
    const uint32_t l_name = w3 & 255u, n_cigar = w4 & 0xFFFFu;
    *bs_out = bs;
    if (bs < 32u || bs > (1u << 29)) return false;
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || l_seq < 0 || l_name == 0u) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)(uint32_t)l_seq + 1ull) / 2ull + (uint64_t)(uint32_t)l_seq;
    return need <= (uint64_t)bs;
}

// ---- the tail the group before left: into the head room, right in front of the group's first block
__global__ __launch_bounds__(256) void k_tail_in(Params P, const uint8_t* __restrict__ tail_in) {
    const uint32_t n = P.is_first ? 0u : P.carry_in->tail_len;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) P.out[P.H - n + i] = tail_in[P.H - n + i];
}

// ---- framing, speculative: a wavefront per block (index b of the tables = block b - 1; entry 0 is the tail's)
__global__ __launch_bounds__(256) void k_frame_spec(Params P) {
    const uint32_t wave = uni(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    const uint32_t end = P.H + P.obytes;     // the group's stream ends here
    for (uint32_t b = wave; b < P.n_blocks; b += n_waves) {
        const uint32_t lo = P.H + uni(P.blocks[b].o_off), hi = lo + uni(P.blocks[b].isize);
        uint32_t cand = kNone, ex = 0, cnt = 0;
        uint32_t p = lo;
        while (p < hi) {
            const uint32_t q = p + (uint32_t)lane();
            uint32_t bs = 0;
            const bool ok = q < hi && q + 36u <= end && plausible(P.out + q, P.n_ref, &bs);
            const uint64_t m = __ballot(ok);
            if (!m) { p += 64u; continue; }
            const uint32_t c = p + (uint32_t)__builtin_ctzll(m);
            // the chain from c: plausible up to the block's end?
            uint32_t w = c, n = 0, stop = 0;
            bool good = true;
            for (;;) {
                if (w >= hi) break;
                if (w + 36u > end) { stop = 1; break; }           // not even the fixed fields: the group's tail
                uint32_t wbs = 0;
                if (!plausible(P.out + w, P.n_ref, &wbs)) { good = false; break; }
                if ((uint64_t)w + 4ull + wbs > (uint64_t)end) { stop = 1; break; }
                n++; w += 4u + wbs;
            }
            if (good) { cand = c; ex = w; cnt = n | (stop ? kStop : 0u); break; }
            p = c + 1u;
        }
        if (lane() == 0) { P.cand[b + 1] = cand; P.exit_[b + 1] = ex; P.cnt[b + 1] = cnt; }
    }
}

// a range walked one record after the other, as the host reader does (only the length is looked at: mm_bam_next refuses a
// record shorter than its fixed fields); the lanes of the wavefront walk together
struct Walk { uint32_t n, exit_, stop, err; };
__device__ __forceinline__ Walk walk_range(const uint8_t* out, uint32_t from, uint32_t hi, uint32_t end) {
    Walk r = {0u, from, 0u, 0u};
    uint32_t w = from;
    while (w < hi) {
        if (w + 4u > end) { r.stop = 1; break; }
        const uint32_t bs = uni(ld32(out + w));
        if (bs < 32u) { r.err = 1; break; }
        if ((uint64_t)w + 4ull + bs > (uint64_t)end) { r.stop = 1; break; }
        r.n++; w += 4u + bs;
    }
    r.exit_ = w;
    return r;
}

// ---- framing, the chain: one wavefront.  The blocks' tables are taken 64 at a time (a lane each) and walked with readlane.
__global__ __launch_bounds__(64) void k_frame_chain(Params P) {
    Carry cin = *P.carry_in;
    if (P.is_first) { cin.tail_len = 0; cin.skip = 0; cin.done = cin.seen = 0; cin.err = cin.err_at = 0; cin.n_records = 0; }
    const uint32_t end = P.H + P.obytes;
    const int l = lane();
    Carry co = cin;
    uint32_t cur = P.H - cin.tail_len;
    if (P.is_first) cur = P.H + P.first_skip; else cur += cin.skip;
    co.skip = 0;
    uint32_t total = 0, n_slow = 0, n_bad = 0, first_bad = kNone;
    bool stopped = false;
    if (cur > end) { if (!co.err) { co.err = IE_HEADER; co.err_at = 0; } stopped = true; cur = end; }
    // blocks the inflate refused: nothing can be framed (the host decodes them and runs the group again)
    for (uint32_t b = (uint32_t)l; b < P.n_blocks; b += 64u) if (P.status[b] != 0) { n_bad++; first_bad = min(first_bad, b); }
    for (int d = 32; d; d >>= 1) { n_bad += (uint32_t)__shfl_xor((int)n_bad, d); first_bad = min(first_bad, (uint32_t)__shfl_xor((int)first_bad, d)); }
    n_bad = uni(n_bad); first_bad = uni(first_bad);
    if (n_bad) stopped = true;
    for (uint32_t t0 = 0; t0 <= P.n_blocks; t0 += 64u) {
        const uint32_t t = t0 + (uint32_t)l;
        uint32_t my_lo = 0, my_hi = 0, my_cand = kNone, my_cnt = 0, my_exit = 0;
        if (t == 0u) { my_lo = P.H - cin.tail_len; my_hi = P.H; }                 // entry 0: the tail's bytes [H - tail, H)
        else if (t <= P.n_blocks) { my_lo = P.H + P.blocks[t - 1].o_off; my_hi = my_lo + P.blocks[t - 1].isize; my_cand = P.cand[t]; my_cnt = P.cnt[t]; my_exit = P.exit_[t]; }
        uint32_t my_entry = kNone, my_n = 0, my_base = 0;
        const uint32_t kmax = min(64u, P.n_blocks + 1u - t0);
        for (uint32_t k = 0; k < kmax; k++) {
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)my_hi, (int)k);
            uint32_t e = kNone, n = 0;
            const uint32_t before = total;
            if (!stopped && cur < hi) {
                e = cur;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)my_cand, (int)k);
                if (c == cur) {
                    const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)my_cnt, (int)k);
                    n = kc & ~kStop; stopped = (kc & kStop) != 0u; cur = (uint32_t)__builtin_amdgcn_readlane((int)my_exit, (int)k);
                } else {
                    const Walk w = walk_range(P.out, cur, hi, end);
                    n = w.n; stopped = w.stop != 0u; cur = w.exit_;
                    if (t0 + k) n_slow++;
                    if (w.err) { if (!co.err) { co.err = IE_RECORD; co.err_at = (int32_t)(cin.n_records + total + n); } stopped = true; cur = end; }
                }
                total += n;
            }
            if ((uint32_t)l == k) { my_entry = n ? e : kNone; my_n = n; my_base = before; }
        }
        if (t <= P.n_blocks) { P.entry[t] = my_entry; P.nrec[t] = my_n; P.base[t] = my_base; }
    }
    if (n_bad) { cur = end; total = 0; }   // (the group will be run again)
    if (total > P.max_records) { if (!co.err) { co.err = IE_RECORDS; co.err_at = (int32_t)cin.n_records; } total = 0; cur = end; }
    // the tail: [cur, end) moves to the next group's head room
    uint32_t tail = end - cur;
    if (tail > P.H) { if (!co.err) { co.err = IE_TAIL; co.err_at = (int32_t)(cin.n_records + total); } tail = 0; }
    for (uint32_t i = (uint32_t)l; i < tail; i += 64u) P.tail_out[P.H - tail + i] = P.out[cur + i];
    co.tail_len = tail;
    co.n_records = cin.n_records + total;
    if (l == 0) {
        *P.carry_out = co;   // (done / seen and a refused record: k_rec_scan)
        P.result->n_records = total; P.result->n_bad_blocks = n_bad; P.result->first_bad_block = first_bad; P.result->n_slow_blocks = n_slow;
        P.result->bad_record = kNone; P.result->n_accepted = 0;
    }
}

// ---- framing: the offsets of every block's records
__global__ __launch_bounds__(256) void k_frame_fill(Params P) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t > P.n_blocks) return;
    const uint32_t n = P.nrec[t];
    if (!n || P.result->n_records == 0u) return;
    uint32_t w = P.entry[t], i = P.base[t];
    for (uint32_t k = 0; k < n; k++) { P.rec_off[i + k] = w; w += 4u + ld32(P.out + w); }
}

// ---- load_db on one record: a wavefront per record
__global__ __launch_bounds__(256) void k_rec_parse(Params P) {
    const uint32_t n_rec = P.result->n_records;
    const uint32_t wave = uni(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    const int l = lane();
    for (uint32_t i = wave; i < n_rec; i += n_waves) {
        const uint32_t ro = uni(P.rec_off[i]);
        const uint8_t* p = P.out + ro + 4u;
        const uint32_t bs = uni(ld32(P.out + ro));
        const int32_t tid = (int32_t)uni(ld32(p)), pos = (int32_t)uni(ld32(p + 4));
        const uint32_t w3 = uni(ld32(p + 8)), w4 = uni(ld32(p + 12));
        const int32_t l_qseq = (int32_t)uni(ld32(p + 16));
        const uint32_t l_name = w3 & 255u, n_cigar = w4 & 0xFFFFu, flag = w4 >> 16;
        Desc d;
        d.mm_src = d.mm_len = d.ml_src = d.ml_len = 0u;
        d.l_data = bs - 32u + ((4u - (l_name & 3u)) & 3u);    // htslib pads the name to a multiple of 4
        d.n_cigar_lname = n_cigar | (l_name << 16);
        d.l_qseq = (uint32_t)l_qseq;
        d.flags = flag << 16;
        // mm_bam_next's checks
        uint64_t o = 32ull + l_name;
        bool bad = l_qseq < 0 || l_name == 0u || o > (uint64_t)bs;
        if (!bad) bad = uni((uint32_t)p[32u + l_name - 1u]) != 0u;    // the name ends in a NUL inside l_read_name
        o += 4ull * n_cigar + ((uint64_t)(uint32_t)l_qseq + 1ull) / 2ull + (uint64_t)(uint32_t)l_qseq;
        if (!bad && o > (uint64_t)bs) bad = true;
        if (bad) {
            if (l == 0) { atomicMin((unsigned int*)&P.result->bad_record, i); P.desc[i] = d; }
            continue;
        }
        const bool placed = !(flag & 4u) && tid >= 0;
        if (placed) {
            d.flags |= 2u;
            if (P.ranged) {
                if (tid < P.lo_tid || (tid == P.lo_tid && (int64_t)pos < P.lo_pos)) d.flags |= 4u;
                else if (!P.last && (tid > P.hi_tid || (tid == P.hi_tid && (int64_t)pos >= P.hi_pos))) d.flags |= 8u;
            }
        }
        bool pass = !(flag & 4u) && (P.allow_secondary || !(flag & 0x100u)) && !(P.skip_supplementary && (flag & 0x800u)) && l_qseq != 0;   // src/minimod.c:260-275
        if (pass) {
            // one walk over the tags (bam_aux_get's rules, csrc/host/bamio.c mm_aux_get): the first MM, ML and HP; a tag whose payload
            // does not fit ends the walk
            const uint32_t a0 = ro + 4u + (uint32_t)o, ae = ro + 4u + bs;
            uint32_t a = a0;
            uint32_t mm_t = kNone, ml_t = kNone, hp_t = kNone;   // offsets of the TYPE bytes
            while (a + 3u <= ae) {
                const uint32_t hd = uni(ld32(P.out + a));         // tag[2], type, first payload byte
                const uint32_t ty = (hd >> 16) & 255u, t = a + 2u;
                uint64_t sz = 0;
                bool okt = true;
                switch (ty) {
                    case 'A': case 'c': case 'C': sz = 1; break;
                    case 's': case 'S': sz = 2; break;
                    case 'i': case 'I': case 'f': sz = 4; break;
                    case 'd': sz = 8; break;
                    case 'Z': case 'H': {
                        // memchr(t + 1, 0, ae - (t + 1)): 512 bytes a trip
                        uint32_t z = kNone;
                        for (uint32_t s = t + 1u; s < ae && z == kNone; s += 512u) {
                            const uint32_t at = s + 8u * (uint32_t)l;
                            const uint64_t v = at < ae ? ld64(P.out + at) : ~0ull;
                            const uint64_t zb = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull;    // its lowest set bit: the first zero byte
                            uint32_t mine = kNone;
                            if (zb) { const uint32_t k = at + ((uint32_t)__builtin_ctzll(zb) >> 3); if (k < ae) mine = k; }
                            const uint64_t m = __ballot(mine != kNone);
                            if (m) z = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)__builtin_ctzll(m));
                        }
                        if (z == kNone) okt = false; else sz = (uint64_t)(z - (t + 1u)) + 1ull;
                        break;
                    }
                    case 'B': {
                        if (t + 6u > ae) { okt = false; break; }
                        const uint32_t sub = (hd >> 24) & 255u;
                        const uint32_t n = uni(ld32(P.out + t + 2u));
                        uint32_t es = 0;
                        switch (sub) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: okt = false; }
                        sz = 5ull + (uint64_t)n * es;
                        break;
                    }
                    default: okt = false;
                }
                if (!okt || sz > (uint64_t)(ae - (t + 1u))) break;
                const uint32_t name = hd & 0xFFFFu;
                if (name == (uint32_t)('M' | ('M' << 8)) && mm_t == kNone) mm_t = t;
                else if (name == (uint32_t)('M' | ('L' << 8)) && ml_t == kNone) ml_t = t;
                else if (name == (uint32_t)('H' | ('P' << 8)) && hp_t == kNone) hp_t = t;
                a = t + 1u + (uint32_t)sz;
            }
            // get_mm_tag_ptr: a Z (or H) string; get_ml_tag: B:C; get_hp_tag: bam_aux2i
            const uint32_t mm_ty = mm_t != kNone ? (uint32_t)uni((uint32_t)P.out[mm_t]) : 0u;
            if (mm_t == kNone || (mm_ty != 'Z' && mm_ty != 'H')) pass = false;    // src/minimod.c:280-284
            else {
                // strlen: the walk above stopped at the string's NUL when it passed the tag; find it again (one trip for nearly all)
                uint32_t z = kNone;
                for (uint32_t s = mm_t + 1u; z == kNone; s += 512u) {
                    const uint32_t at = s + 8u * (uint32_t)l;
                    const uint64_t v = at < ae ? ld64(P.out + at) : 0ull;
                    const uint64_t zb = (v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull;
                    const uint64_t m = __ballot(zb != 0ull);
                    if (m) { const int src = (int)__builtin_ctzll(m); const uint32_t k = at + ((uint32_t)__builtin_ctzll(zb | (zb == 0ull)) >> 3); z = (uint32_t)__builtin_amdgcn_readlane((int)k, src); }
                }
                d.mm_src = mm_t + 1u; d.mm_len = z - (mm_t + 1u);
                if (ml_t != kNone) {
                    const uint32_t h2 = uni(ld16(P.out + ml_t));
                    if (h2 == (uint32_t)('B' | ('C' << 8))) {
                        d.ml_len = uni(ld32(P.out + ml_t + 2u)); d.ml_src = ml_t + 6u;
                        if ((uint64_t)d.ml_len > (uint64_t)(ae - d.ml_src)) pass = false;
                    }
                }
                if (hp_t != kNone) {
                    const uint32_t ht = uni((uint32_t)P.out[hp_t]);
                    const uint32_t v = uni(ld32(P.out + hp_t + 1u));
                    uint32_t hp = 0;
                    switch (ht) { case 'c': case 'C': hp = v & 255u; break; case 's': case 'S': hp = v & 255u; break; case 'i': case 'I': hp = v & 255u; break; default: hp = 0; }
                    d.flags |= hp << 8;   // (uint8_t)bam_aux2i: its low byte, whatever the width
                }
            }
        }
        if (pass) d.flags |= 1u;
        if (l == 0) P.desc[i] = d;
    }
}

// ---- exclusive sums over a workgroup of kScanThreads threads.  (Four wavefronts that need few registers: the kernel runs beside the
// inflate's workgroups, which hold a CU's LDS and most of its registers for milliseconds -- a workgroup of 1024 threads and 112
// registers waited for a whole CU to drain, 3.8 ms a group.)
constexpr int kScanThreads = 256, kScanWaves = kScanThreads / 64;
__device__ __forceinline__ uint64_t wg_scan(uint64_t v, uint64_t* sh /*[kScanWaves + 1]*/, uint64_t* total) {
    const int l = lane(), w = (int)(threadIdx.x >> 6);
    uint64_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d, 64); if (l >= d) x += y; }
    __syncthreads();
    if (l == 63) sh[w] = x;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < kScanWaves; k++) { const uint64_t t = sh[k]; if (k < w) before += t; all += t; }
    *total = all;
    return before + x - v;
}

// ---- which records count, which are accepted, where they go: one workgroup
__global__ __launch_bounds__(kScanThreads, 4) void k_rec_scan(Params P) {
    __shared__ uint64_t sh[kScanWaves + 1];
    const uint32_t n_rec = P.result->n_records;
    Carry co = *P.carry_out;               // (k_frame_chain's; done / seen and a record's error are added here)
    Cursor cu;
    if (P.new_arena) { cu.n_reads = cu.cigar_bytes = cu.seq_bytes = cu.mm_bytes = cu.ml_bytes = cu.qname_bytes = cu.bases = 0; cu.max_n_cigar = cu.max_l_qseq = 0; }
    else cu = *P.cursor_in;
    const uint32_t bad_rec = P.result->bad_record;                     // first record the parse refused (kNone: none)
    const uint32_t n_use = bad_rec < n_rec ? bad_rec : n_rec;    // records in front of it still count (the host reader fails when it gets there)
    uint64_t total_reads = 0, total_bytes = 0, proc_bytes = 0;
    uint32_t n_acc = 0;
    uint32_t seen = (uint32_t)co.seen, done = (uint32_t)co.done;
    uint32_t mx_c = cu.max_n_cigar, mx_l = cu.max_l_qseq;
    bool full = false;
    for (uint32_t base = 0; base < n_use; base += (uint32_t)kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        Desc d;
        d.flags = 0; d.l_data = 0; d.l_qseq = 0; d.n_cigar_lname = 0; d.mm_len = d.ml_len = 0; d.mm_src = d.ml_src = 0;
        if (i < n_use) d = P.desc[i];
        const bool placed = (d.flags & 2u) != 0u, before = (d.flags & 4u) != 0u, behind = (d.flags & 8u) != 0u;
        bool counted = i < n_use;
        if (P.ranged) {
            uint64_t tot;
            const uint64_t hi_in = wg_scan(behind ? 1ull : 0ull, sh, &tot) + (behind ? 1ull : 0ull);   // inclusive: the record behind the end stops the share
            const uint64_t hi_tot = tot;
            const bool inside = placed && !before && !behind;
            const uint64_t in_ex = wg_scan(inside ? 1ull : 0ull, sh, &tot);
            const bool stop = done || hi_in > 0;
            const bool sn = seen || in_ex > 0;
            counted = counted && !stop && (placed ? inside : (sn || P.first));
            done |= hi_tot > 0 ? 1u : 0u; seen |= tot > 0 ? 1u : 0u;
        }
        const bool acc = counted && (d.flags & 1u);
        const uint32_t n_cigar = d.n_cigar_lname & 0xFFFFu;
        uint64_t t_n, t_c, t_s, t_m, t_l, t_x;
        const uint64_t r_n = wg_scan(acc ? 1ull : 0ull, sh, &t_n);
        const uint64_t r_c = wg_scan(acc ? (4ull * n_cigar + 15ull) & ~15ull : 0ull, sh, &t_c);
        const uint64_t r_s = wg_scan(acc ? (((uint64_t)d.l_qseq + 1ull) / 2ull + 15ull) & ~15ull : 0ull, sh, &t_s);
        const uint64_t r_m = wg_scan(acc ? ((uint64_t)d.mm_len + 1ull + 15ull) & ~15ull : 0ull, sh, &t_m);
        const uint64_t r_l = wg_scan(acc ? ((uint64_t)d.ml_len + 3ull) & ~3ull : 0ull, sh, &t_l);
        (void)wg_scan(counted ? ((uint64_t)d.l_data << 24) + 1ull : 0ull, sh, &t_x);    // counted records (24 bits of count a tile) and their bytes
        total_reads += t_x & 0xFFFFFFull; total_bytes += t_x >> 24;
        (void)wg_scan(acc ? ((uint64_t)d.l_data << 24) | 0ull : 0ull, sh, &t_x);
        proc_bytes += t_x >> 24;
        uint64_t t_b;
        (void)wg_scan(acc ? (uint64_t)d.l_qseq : 0ull, sh, &t_b);
        uint64_t t_q = 0, r_q = 0;
        if (P.names) r_q = wg_scan(acc ? (uint64_t)(d.n_cigar_lname >> 16) : 0ull, sh, &t_q);   // l_read_name counts the NUL
        // room for the tile?  (+ 64 bytes of zero slack behind every pool)
        if (cu.n_reads + t_n > P.cap_reads || cu.cigar_bytes + t_c + 64 > P.cap_cigar || cu.seq_bytes + t_s + 64 > P.cap_seq ||
            cu.mm_bytes + t_m + 64 > P.cap_mm || cu.ml_bytes + t_l + 64 > P.cap_ml || cu.mm_bytes + t_m >= 0xFFFFF000ull || cu.n_reads + t_n >= (1ull << 24) ||
            (P.names && cu.qname_bytes + t_q > P.cap_names)) { full = true; break; }
        if (i < n_use) P.info[i] = d.l_data | (counted ? 1u << 30 : 0u) | (acc ? 1u << 31 : 0u);
        if (acc) {
            mm_read_t rd;
            rd.cigar_off = (cu.cigar_bytes + r_c) / 4ull; rd.seq_off = cu.seq_bytes + r_s; rd.mm_off = cu.mm_bytes + r_m; rd.ml_off = cu.ml_bytes + r_l;
            const uint8_t* p = P.out + P.rec_off[i] + 4u;
            rd.tid = (int32_t)ld32(p); rd.pos = (int32_t)ld32(p + 4);
            rd.l_qseq = d.l_qseq; rd.n_cigar = n_cigar; rd.mm_len = d.mm_len; rd.ml_len = d.ml_len;
            rd.flag = (uint16_t)(d.flags >> 16); rd.hp = (uint8_t)(d.flags >> 8); rd.rsvd = 0; rd.rsvd2 = 0;
            P.reads[cu.n_reads + r_n] = rd;
            if (P.names) P.name_off[cu.n_reads + r_n] = cu.qname_bytes + r_q;
            P.acc_rec[n_acc + (uint32_t)r_n] = i;
        }
        // the largest CIGAR / read of the batch
        uint32_t c2 = acc ? n_cigar : 0u, l2 = acc ? d.l_qseq : 0u;
        for (int dd = 32; dd; dd >>= 1) { c2 = max(c2, (uint32_t)__shfl_xor((int)c2, dd)); l2 = max(l2, (uint32_t)__shfl_xor((int)l2, dd)); }
        __syncthreads();
        if (lane() == 0) { sh[threadIdx.x >> 6] = ((uint64_t)c2 << 32) | l2; }
        __syncthreads();
        for (int k = 0; k < kScanWaves; k++) { mx_c = max(mx_c, (uint32_t)(sh[k] >> 32)); mx_l = max(mx_l, (uint32_t)sh[k]); }
        __syncthreads();
        cu.n_reads += t_n; cu.cigar_bytes += t_c; cu.seq_bytes += t_s; cu.mm_bytes += t_m; cu.ml_bytes += t_l; cu.bases += t_b; cu.qname_bytes += t_q;
        n_acc += (uint32_t)t_n;
    }
    cu.max_n_cigar = mx_c; cu.max_l_qseq = mx_l;
    if (full && !co.err) { co.err = IE_ARENA; co.err_at = (int32_t)(co.n_records - n_rec); }
    if (bad_rec < n_rec && !co.err) { co.err = IE_RECORD; co.err_at = (int32_t)(co.n_records - n_rec + bad_rec); }
    co.seen = (int32_t)seen; co.done = (int32_t)done;
    // 64 zero bytes behind every pool's end (the next group of the arena writes over them)
    if (!full && threadIdx.x < 64u) {
        uint8_t* ends[4] = {P.cigar + cu.cigar_bytes, P.seq + cu.seq_bytes, P.mm + cu.mm_bytes, P.ml + cu.ml_bytes};
        for (int k = 0; k < 4; k++) ends[k][threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        *P.carry_out = co; *P.cursor_out = cu;
        Result* R = P.result;
        R->carry = co; R->cursor = cu; R->n_accepted = full ? 0u : n_acc;
        R->total_reads = total_reads; R->total_bytes = total_bytes; R->processed_bytes = proc_bytes;
    }
}

typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));   // four dwords at a dword-aligned address: one global_load_dwordx4
// ---- the copies.  dst: 16-byte aligned; `len` bytes from src (any alignment), zeros up to `extent` (a multiple of 16)
__device__ __forceinline__ void copy_item16(uint8_t* dst, const uint8_t* src, uint32_t len, uint32_t extent, bool clear_low_nibble) {
    const uint32_t sh = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src - sh);
    for (uint32_t c = (uint32_t)lane(); 16u * c < extent; c += 64u) {
        uint32_t w[5];
        const uint32_t* q = s4 + 4u * c;
        const bool any = 16u * c < len;
        if (any) { const u32x4a v = *reinterpret_cast<const u32x4a*>(q); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; w[4] = sh ? q[4] : 0u; }
        else { w[0] = w[1] = w[2] = w[3] = w[4] = 0u; }
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
            const int rem = (int)len - (int)(16u * c + 4u * (uint32_t)k);
            o[k] &= rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
            if (clear_low_nibble && rem >= 1 && rem <= 4) o[k] &= ~(0x0Fu << (8 * (rem - 1)));
        }
        *reinterpret_cast<uint4*>(dst + 16u * c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// the same with 4-byte items (ML)
__device__ __forceinline__ void copy_item4(uint8_t* dst, const uint8_t* src, uint32_t len, uint32_t extent) {
    const uint32_t sh = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src - sh);
    for (uint32_t c = (uint32_t)lane(); 4u * c < extent; c += 64u) {
        const uint32_t w0 = s4[c], w1 = sh ? s4[c + 1] : 0u;
        uint32_t o = __builtin_amdgcn_alignbyte(w1, w0, sh);
        const int rem = (int)len - (int)(4u * c);
        o &= rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        *reinterpret_cast<uint32_t*>(dst + 4u * c) = o;
    }
}

__global__ __launch_bounds__(256) void k_rec_copy(Params P) {
    const uint32_t n_acc = P.result->n_accepted;
    const uint64_t r0 = P.cursor_out->n_reads - n_acc;    // the group's first read in the arena's batch
    const uint32_t wave = uni(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    for (uint32_t j = wave; j < n_acc; j += n_waves) {
        const uint32_t i = uni(P.acc_rec[j]);
        const uint32_t ro = uni(P.rec_off[i]);
        const Desc* dp = P.desc + i;
        const uint32_t ncl = uni(dp->n_cigar_lname), l_qseq = uni(dp->l_qseq), mm_src = uni(dp->mm_src), mm_len = uni(dp->mm_len), ml_src = uni(dp->ml_src), ml_len = uni(dp->ml_len);
        const uint32_t n_cigar = ncl & 0xFFFFu, l_name = ncl >> 16;
        const mm_read_t* rd = P.reads + (r0 + j);
        const uint64_t o_c = 4ull * rd->cigar_off, o_s = rd->seq_off, o_m = rd->mm_off, o_l = rd->ml_off;
        const uint8_t* cig = P.out + ro + 36u + l_name;
        const uint8_t* sq = cig + 4u * n_cigar;
        const uint32_t sb = (l_qseq + 1u) / 2u;
        copy_item16(P.cigar + o_c, cig, 4u * n_cigar, (4u * n_cigar + 15u) & ~15u, false);
        copy_item16(P.seq + o_s, sq, sb, (sb + 15u) & ~15u, (l_qseq & 1u) != 0u);   // the unused low nibble must be zero for the base counts
        copy_item16(P.mm + o_m, P.out + mm_src, mm_len, (mm_len + 1u + 15u) & ~15u, false);
        copy_item4(P.ml + o_l, P.out + ml_src, ml_len, (ml_len + 3u) & ~3u);
        if (P.names) {
            uint8_t* nd = P.names + P.name_off[r0 + j];
            const uint8_t* ns = P.out + ro + 36u;
            for (uint32_t c = (uint32_t)lane(); c < l_name; c += 64u) nd[c] = ns[c];
        }
    }
}

// ---- the codes a batch's MM tags carry, for a wildcard run (-c '*': the reference counts whatever code a read names, src/mod.c's
// req_all path; the host interns a code the first time it meets one, csrc/host/freq_main.c intern_batch_codes).  A wavefront per read
// finds the groups of its MM text (the text's start and every byte behind a ';'), takes the code behind the base and the strand --
// digits: one ChEBI code; letters: the string from each letter on -- and leaves, per distinct code of up to 8 characters, the SMALLEST (read, text
// offset) it was seen at: sorted by that, the codes come out in the order the host's walk over the batch meets them.
constexpr uint32_t kCodeSlots = 1024;
struct CodeTab { unsigned long long key[kCodeSlots]; unsigned long long stamp[kCodeSlots]; uint32_t flags; uint32_t pad; };   // flags: 1 a code longer than 8 characters, 2 table full
__device__ __forceinline__ void code_put(CodeTab* T, uint64_t key, uint64_t stamp) {
    uint32_t s = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 54) & (kCodeSlots - 1u);
    for (uint32_t probes = 0; probes < kCodeSlots; probes++, s = (s + 1u) & (kCodeSlots - 1u)) {
        unsigned long long k = __atomic_load_n(&T->key[s], __ATOMIC_RELAXED);
        if (k == 0ull) { k = atomicCAS(&T->key[s], 0ull, (unsigned long long)key); if (k == 0ull) k = key; }
        if (k == key) {
            if (__atomic_load_n(&T->stamp[s], __ATOMIC_RELAXED) > stamp) atomicMin(&T->stamp[s], (unsigned long long)stamp);
            return;
        }
    }
    atomicOr(&T->flags, 2u);
}
__global__ __launch_bounds__(256) void k_batch_codes(const mm_read_t* __restrict__ reads, const uint8_t* __restrict__ mm, uint32_t n_reads, CodeTab* T) {
    const uint32_t wave = uni(blockIdx.x * 4u + (threadIdx.x >> 6)), n_waves = gridDim.x * 4u;
    const int l = lane();
    for (uint32_t r = wave; r < n_reads; r += n_waves) {
        const uint32_t off = uni(reads[r].mm_off), n = uni(reads[r].mm_len);
        const uint8_t* t = mm + off;                      // (16-byte aligned, zeros behind the text up to the next multiple of 16 and 64 more behind the pool)
        for (uint32_t base = 0; base < n + 1u; base += 1024u) {
            // group starts in [base, base + 1024): position 0, and p + 1 for every ';' at p
            const uint32_t at = base + 16u * (uint32_t)l;
            uint32_t semis = 0;                           // bit k: a group starts at at + k
            if (at < n + 16u) {
                const uint4 v = *reinterpret_cast<const uint4*>(t + (at >= 16u ? at - 16u : 0u));
                const uint4 w = at < n ? *reinterpret_cast<const uint4*>(t + at) : make_uint4(0, 0, 0, 0);
                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
                if (at == 0u) semis |= 1u; else if ((v.w >> 24) == (uint32_t)';') semis |= 1u;
#pragma unroll
                for (int k = 0; k < 15; k++) if (((ws[k >> 2] >> (8 * (k & 3))) & 255u) == (uint32_t)';') semis |= 2u << k;
            }
            uint64_t m = __ballot(semis != 0u);
            while (m) {
                const int src = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                uint32_t bits = (uint32_t)__builtin_amdgcn_readlane((int)semis, src);
                const uint32_t a0 = base + 16u * (uint32_t)src;
                while (bits) {
                    const uint32_t p = a0 + (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1u;
                    if (p >= n) continue;
                    const uint32_t s = p + 2u;
                    if (s >= n) continue;
                    // the code: [s, e), e = the first of , ; ? . or the text's end -- lanes look at a byte each (codes of 16 and more are not interned)
                    const uint32_t c = s + (uint32_t)l < n ? (uint32_t)t[s + (uint32_t)l] : (uint32_t)';';
                    const uint64_t stop = __ballot(c == ',' || c == ';' || c == '?' || c == '.') | (1ull << 63);
                    const uint32_t len = (uint32_t)__builtin_ctzll(stop);
                    if (len == 0u || len >= 16u) continue;
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                    const uint64_t stamp = ((uint64_t)r << 32) | s;
                    // digits: the whole string is one code; letters: the string from every letter on is one (mod.c:1146-1160 looks a letter's code
                    // up as the C string that starts at it) -- lane m holds the code that starts at character m
                    const bool digits = c0 >= '0' && c0 <= '9';
                    uint64_t key = 0;
                    for (uint32_t k = 0; k < len; k++) {
                        const uint64_t ck = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c, (int)k);
                        if (k >= (uint32_t)l && k - (uint32_t)l < 8u) key |= ck << (8u * (k - (uint32_t)l));
                    }
                    if ((uint32_t)l < (digits ? 1u : len)) {
                        if (len - (uint32_t)l > 8u) atomicOr(&T->flags, 1u);
                        else code_put(T, key, stamp + (uint32_t)l);
                    }
                }
            }
        }
    }
}

}  // namespace mmingest
