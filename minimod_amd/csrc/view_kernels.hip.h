// view_kernels.hip.h -- ordering the records of `minimod view` on the device (reference print_view_output,
// src/mod.c:560-626: per read, rows sorted by reference position; add_view_entry, src/mod.c:931-946: the first entry
// of a key wins).
//
// The call kernels append (key, value) records to kViewRegions regions in no particular order and count them per
// read.  A read's rows are independent of every other read's, so the ordering is a counting sort by read followed by
// one small sort per read -- no global sort:
//   1. k_view_offsets  exclusive scan of the per-read record counts                       (one block)
//   2. k_view_scatter  every record to its read's segment (k_stream_reads' records carry their place; the tile kernels' take one
//                      cursor atomic per read and wave); reads whose records were made in row order get their ROWS here
//   3. k_view_sort     bitonic sort of every read's segment on (position, code, ins_offset, the order the reference met
//                      the calls in), later entries of a key marked, rows expanded to the 16-byte form of the C ABI: one
//                      wavefront per read up to 512 records (LDS), one workgroup per bigger read (LDS up to 2048 records,
//                      in place in global memory beyond), both kinds of workers in the same launch
//   4. k_view_compact  only when some read had duplicate keys: rows of every read moved down over the dropped ones.
#pragma once
#include "freq_kernels.hip.h"

namespace mmhip {

struct ViewRow {   // == mm_view_row_t
    uint32_t read;
    int32_t pos;
    uint32_t read_pos;
    uint16_t ins_offset;
    uint8_t code;
    uint8_t prob;
};
static_assert(sizeof(ViewRow) == 16, "ViewRow must be 16 bytes");

constexpr uint32_t kViewDropped = 0xFFFFFFFFu;   // ViewRow.read of a dropped duplicate (before k_view_compact)
constexpr uint32_t kViewWaveRecs = 512;          // records one wavefront sorts in its slice of the LDS (8 KB)
constexpr uint32_t kViewLdsRecs = 4096;          // records a workgroup sorts in LDS (one packed 64-bit word each: 32 KB)

// offsets[r] = number of records of reads < r; offsets[n_reads] = total.  One block; thread t owns the reads
// [t*chunk, (t+1)*chunk): sum them, scan the 256 sums, write the offsets (cursors are zeroed on the way).
__global__ __launch_bounds__(256) void k_view_offsets(const unsigned int* __restrict__ counts, uint32_t n_reads,
                                                      unsigned int* __restrict__ offsets, unsigned int* __restrict__ cursor) {
    // one workgroup; 256 consecutive reads per round, sixteen rounds' counts requested together (a thread walking its own
    // chunk of reads waited for one load after the other)
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_reads; base += 256u * 16u) {
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { uint32_t i = base + 256u * k + threadIdx.x; v[k] = i < n_reads ? counts[i] : 0u; }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t i = base + 256u * k + threadIdx.x;
            uint32_t incl = wave_incl_scan(v[k]);
            __syncthreads();
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            uint32_t before = carry + incl - v[k];
            for (int w = 0; w < wv; w++) before += wsum[w];
            if (i < n_reads) { offsets[i] = before; cursor[i] = 0u; }
            carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        }
    }
    if (threadIdx.x == 0) offsets[n_reads] = carry;
}

// the same as a tiled scan for the big (gathered) launches -- k_scan_tile_sums over the counts, k_radix_scan over the tile sums
// (freq_kernels.hip.h / sort_kernels.hip.h), then this: every read's offset from its tile's, cursors zeroed, the total behind
// the last read (a single workgroup took 104 us for the 81 920 reads of a 20-batch launch)
__global__ __launch_bounds__(256) void k_view_offsets_apply(const unsigned int* __restrict__ counts, uint32_t n_reads, const uint32_t* __restrict__ tile_off, uint32_t n_tiles,
                                                            unsigned int* __restrict__ offsets, unsigned int* __restrict__ cursor) {
    __shared__ uint32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t run = tile_off[blockIdx.x];
    for (int j0 = 0; j0 < kScanTile; j0 += 256) {
        const int64_t i = base + j0 + threadIdx.x;
        const uint32_t c = i < (int64_t)n_reads ? counts[i] : 0u;
        const uint32_t incl = wave_incl_scan(c);
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        uint32_t before = run + incl - c;
        for (int w = 0; w < wv; w++) before += wsum[w];
        if (i < (int64_t)n_reads) { offsets[i] = before; cursor[i] = 0u; }
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_reads] = tile_off[n_tiles];
}

// opts.view == 2 (rows for the tie-order replay of the host, csrc/host/tieorder.c): a row also says in which MM group the
// call was made (its ordinal, at most kViewMaxGroup = 2047, in the top eleven bits of `read`: a batch of such a handle has
// fewer than 2^21 reads) and whether it was an implicit call of a '.' group (bit 31 of read_pos): with the position in the
// read as sequenced that is the order the reference met the calls in.  Later entries of a key are KEPT there (the replay
// wants every call: update_freq_map may meet a key whose first entry was ambiguous, src/mod.c:886-904).
__device__ __forceinline__ uint32_t view_read_word(uint32_t r, unsigned long long v, uint32_t ordinal) {
    if (!ordinal) return r;
    const uint32_t g = (uint32_t)((v >> 29) & 0x7FFull);
    return r | (g << 21);
}
__device__ __forceinline__ uint32_t view_read_pos_word(unsigned long long v, uint32_t ordinal) {
    const uint32_t q = (uint32_t)(v & 0x0FFFFFFFull);
    return ordinal ? (q | ((uint32_t)((v >> 28) & 1ull) << 31)) : q;
}

// regions -> per-read segments
// A record of k_stream_reads knows its place among its read's records (rseq), and when the read made them in the order of its
// rows (bit 31 of rseq: one requested code in one '?' group -- positions rise with the calls of a forward read, fall with those
// of a reverse one, and no key comes twice) the ROW is written here, at its final place: such a read has nothing left for
// k_view_sort (told so by its cursor word, kViewPresorted).
constexpr unsigned int kViewPresorted = 0xFFFFFFFFu;
__global__ __launch_bounds__(256) void k_view_scatter(const unsigned long long* __restrict__ rk, const unsigned long long* __restrict__ rv,
                                                      const unsigned int* __restrict__ rseq,
                                                      const unsigned int* __restrict__ region_counts, unsigned int cap, uint32_t read_mask,
                                                      const unsigned int* __restrict__ offsets, unsigned int* __restrict__ cursor,
                                                      unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals,
                                                      const mm_read_t* __restrict__ reads, ViewRow* __restrict__ rows, uint32_t ordinal) {
    for (uint32_t region = blockIdx.y; region < kViewRegions; region += gridDim.y) {
        unsigned int n = region_counts[region * kViewCountStride];
        if (n > cap) n = cap;
        const size_t base = (size_t)region * cap;
        for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
            unsigned long long k = rk[base + i], v = rv[base + i];
            uint32_t read = (uint32_t)(k >> 28) & read_mask;
            const unsigned int sq = rseq[base + i];
            if (sq != 0xFFFFFFFFu && (sq >> 31)) {
                const unsigned int off = offsets[read], nrec = offsets[read + 1u] - off, at = sq & 0x7FFFFFFFu;
                const mm_read_t& rd = reads[read];
                ViewRow o;
                o.read = view_read_word(read, v, ordinal);
                o.pos = rd.pos + (int32_t)((uint32_t)k & 0x0FFFFFFFu) - 1;
                o.read_pos = view_read_pos_word(v, ordinal);
                o.ins_offset = (uint16_t)((v >> 40) & 0xFFFFull); o.code = (uint8_t)(v >> 56); o.prob = (uint8_t)(k >> 56);
                if (at < nrec) rows[off + ((rd.flag & 0x10) ? nrec - 1u - at : at)] = o;
                if (at == 0u) cursor[read] = kViewPresorted;
                continue;
            }
            // The others of k_stream_reads: no cursor, and the read's records arrive in the order they were made.  The tile
            // kernels': neighbours in a region mostly belong to one read (a tile appends its records
            // together): one cursor atomic per distinct read of the wave instead of one per record
            unsigned int slot = sq != 0xFFFFFFFFu ? offsets[read] + sq : 0u;
            bool pending = sq == 0xFFFFFFFFu;
            while (pending) {
                const uint64_t pm = __ballot(pending);
                if (!pm) break;
                const uint32_t first = lane_valu(read, __ffsll((unsigned long long)pm) - 1);   // the first still-pending lane's read
                bool same = pending && read == first;
                uint64_t m = __ballot(same);
                unsigned int at = 0;
                int leader = __ffsll((unsigned long long)m) - 1;
                if (lane_id() == leader) at = atomicAdd(cursor + first, (unsigned int)__popcll(m));
                at = __shfl(at, leader, 64);
                if (same) { slot = offsets[first] + at + (unsigned int)__popcll(m & lanemask_lt()); pending = false; }
            }
            keys[slot] = k;
            vals[slot] = v;
        }
    }
}

// (position, value) order of two records of one read; position = the key's low 28 bits
__device__ __forceinline__ bool view_less(unsigned long long ka, unsigned long long va, unsigned long long kb, unsigned long long vb) {
    uint32_t pa = (uint32_t)ka & 0x0FFFFFFFu, pb = (uint32_t)kb & 0x0FFFFFFFu;
    return pa != pb ? pa < pb : va < vb;
}

// Bitonic network in its normalised form (every compare-exchange puts the smaller record at the lower index), which
// sorts any n: the positions from n up to the next power of two behave as +infinity and never have to exist.
// kThreads cooperating threads (one wavefront, or a 256-thread workgroup), `tid` this thread's index among them.
template <int kThreads, typename Ptr>
__device__ __forceinline__ void view_bitonic(Ptr K, Ptr V, uint32_t n, uint32_t tid) {
    uint32_t l2 = 0;
    while ((1u << l2) < n) l2++;
    const uint32_t half = (1u << l2) >> 1;
    for (uint32_t lk = 1; lk <= l2; lk++) {
        const uint32_t k = 1u << lk;
        for (uint32_t lj = lk; lj-- > 0;) {
            const uint32_t j = 1u << lj;
            const bool flip = lj + 1u == lk;
            for (uint32_t t = tid; t < half; t += kThreads) {
                // t-th pair of this step
                uint32_t lo = ((t >> lj) << (lj + 1u)) | (t & (j - 1u));
                uint32_t hi = flip ? (lo ^ (k - 1u)) : (lo + j);   // first step of a stage: the mirror position inside the k-block
                if (hi < n) {
                    unsigned long long ka = K[lo], va = V[lo], kb = K[hi], vb = V[hi];
                    if (view_less(kb, vb, ka, va)) { K[lo] = kb; V[lo] = vb; K[hi] = ka; V[hi] = va; }
                }
            }
            if (kThreads == 64) wave_sync(); else __syncthreads();
        }
    }
}

// sorted records -> rows; a record with the key (position, code, ins_offset) of the one in front is a later entry of that
// key and is dropped.  Returns this thread's number of dropped rows.
template <int kThreads, typename Ptr>
__device__ __forceinline__ uint32_t view_emit_rows(Ptr K, Ptr V, uint32_t n, uint32_t tid, uint32_t r, int32_t rpos, ViewRow* __restrict__ out, uint32_t ordinal) {
    uint32_t dropped = 0;
    for (uint32_t i = tid; i < n; i += kThreads) {
        unsigned long long k = K[i], v = V[i];
        bool dup = false;
        if (i > 0 && !ordinal) {
            unsigned long long kp = K[i - 1], vp = V[i - 1];
            dup = ((uint32_t)kp & 0x0FFFFFFFu) == ((uint32_t)k & 0x0FFFFFFFu) && (vp >> 40) == (v >> 40);
        }
        ViewRow o;
        o.read = dup ? kViewDropped : view_read_word(r, v, ordinal);
        o.pos = rpos + (int32_t)((uint32_t)k & 0x0FFFFFFFu) - 1;
        o.read_pos = view_read_pos_word(v, ordinal);
        o.ins_offset = (uint16_t)((v >> 40) & 0xFFFFull); o.code = (uint8_t)(v >> 56); o.prob = (uint8_t)(k >> 56);
        out[i] = o;
        dropped += dup;
    }
    return dropped;
}

// ---- the same two steps on ONE 64-bit word per record, for segments of up to 4096 records (nearly all reads): the word
// packs what the order depends on -- position << 36 | code << 28 | ins_offset << 12 | index of the record in the segment
// -- so a compare-exchange moves 8 bytes instead of 16 and compares one integer.  The records themselves stay where they
// are in global memory and are fetched once, in sorted order, when the rows are written.  Entries of one key then stand
// next to each other in arbitrary order: the first of them looks through the run and keeps the one the reference met
// first (smallest value word).
__device__ __forceinline__ unsigned long long view_pack(unsigned long long k, unsigned long long v, uint32_t idx) {
    return ((k & 0x0FFFFFFFull) << 36) | ((v >> 56) << 28) | (((v >> 40) & 0xFFFFull) << 12) | (unsigned long long)idx;
}

template <int kThreads>
__device__ __forceinline__ void view_bitonic1(unsigned long long* P, uint32_t n, uint32_t tid) {
    uint32_t l2 = 0;
    while ((1u << l2) < n) l2++;
    const uint32_t half = (1u << l2) >> 1;
    for (uint32_t lk = 1; lk <= l2; lk++) {
        const uint32_t k = 1u << lk;
        for (uint32_t lj = lk; lj-- > 0;) {
            const uint32_t j = 1u << lj;
            const bool flip = lj + 1u == lk;
            for (uint32_t t = tid; t < half; t += kThreads) {
                uint32_t lo = ((t >> lj) << (lj + 1u)) | (t & (j - 1u));
                uint32_t hi = flip ? (lo ^ (k - 1u)) : (lo + j);
                if (hi < n) {
                    unsigned long long a = P[lo], b = P[hi];
                    if (b < a) { P[lo] = b; P[hi] = a; }
                }
            }
            if (kThreads == 64) wave_sync(); else __syncthreads();
        }
    }
}

template <int kThreads>
__device__ __forceinline__ uint32_t view_emit_rows1(const unsigned long long* P, const unsigned long long* __restrict__ gk,
                                                    const unsigned long long* __restrict__ gv, uint32_t n, uint32_t tid, uint32_t r,
                                                    int32_t rpos, ViewRow* __restrict__ out, uint32_t ordinal) {
    uint32_t dropped = 0;
    for (uint32_t i = tid; i < n; i += kThreads) {
        const unsigned long long pi = P[i];
        const bool head = ordinal || i == 0 || (P[i - 1] >> 12) != (pi >> 12);
        ViewRow o;
        if (head) {
            uint32_t best = (uint32_t)(pi & 0xFFFull);
            unsigned long long bv = gv[best];
            for (uint32_t j = i + 1; !ordinal && j < n && (P[j] >> 12) == (pi >> 12); j++) {   // other entries of the same key: rare, short
                uint32_t cand = (uint32_t)(P[j] & 0xFFFull);
                unsigned long long cv = gv[cand];
                if (cv < bv) { bv = cv; best = cand; }
            }
            const unsigned long long k = gk[best];
            o.read = view_read_word(r, bv, ordinal);
            o.pos = rpos + (int32_t)((uint32_t)k & 0x0FFFFFFFu) - 1;
            o.read_pos = view_read_pos_word(bv, ordinal);
            o.ins_offset = (uint16_t)((bv >> 40) & 0xFFFFull); o.code = (uint8_t)(bv >> 56); o.prob = (uint8_t)(k >> 56);
        } else {
            o.read = kViewDropped; o.pos = 0; o.read_pos = 0; o.ins_offset = 0; o.code = 0; o.prob = 0;
            dropped++;
        }
        out[i] = o;
    }
    return dropped;
}

// One launch, two kinds of workers, so that the few big reads do not serialise behind the many small ones:
//   blocks [0, n_big_blocks)   take the reads with MORE than kViewWaveRecs records, one workgroup per read (packed
//                              sort words in the workgroup's 32 KB of LDS up to kViewLdsRecs, whole records in place in
//                              global memory beyond);
//   the other blocks           take the reads with up to kViewWaveRecs records (nearly all of them), one WAVEFRONT per
//                              read in its own slice of the same LDS, no workgroup barriers.
__global__ __launch_bounds__(256) void k_view_sort(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals,
                                                   const unsigned int* __restrict__ offsets, uint32_t n_reads, uint32_t n_big_blocks,
                                                   const mm_read_t* __restrict__ reads, ViewRow* __restrict__ rows,
                                                   unsigned int* __restrict__ kept, unsigned int* __restrict__ n_dropped, uint32_t ordinal,
                                                   const unsigned int* __restrict__ cursor) {
    __shared__ unsigned long long sp[kViewLdsRecs];   // packed sort words: a workgroup's segment, or four waves' slices
    __shared__ uint32_t drop_s;
    static_assert(kViewLdsRecs >= kWavesPerBlock * kViewWaveRecs, "one LDS layout for both kinds of workers");
    if (blockIdx.x < n_big_blocks) {
        for (uint32_t r = blockIdx.x; r < n_reads; r += n_big_blocks) {
            const uint32_t off = offsets[r], n = offsets[r + 1] - off;
            if (n <= kViewWaveRecs) continue;   // a wave's job
            if (cursor[r] == kViewPresorted) { if (threadIdx.x == 0) kept[r] = n; continue; }   // its rows are written (k_view_scatter)
            if (threadIdx.x == 0) drop_s = 0;
            __syncthreads();
            unsigned long long* gk = keys + off;
            unsigned long long* gv = vals + off;
            uint32_t dropped;
            if (n <= kViewLdsRecs) {
                for (uint32_t i = threadIdx.x; i < n; i += 256) sp[i] = view_pack(gk[i], gv[i], i);
                __syncthreads();
                view_bitonic1<256>(sp, n, threadIdx.x);
                dropped = view_emit_rows1<256>(sp, gk, gv, n, threadIdx.x, r, reads[r].pos, rows + off, ordinal);
            } else {
                view_bitonic<256>(gk, gv, n, threadIdx.x);
                dropped = view_emit_rows<256>(gk, gv, n, threadIdx.x, r, reads[r].pos, rows + off, ordinal);
            }
            if (dropped) atomicAdd(&drop_s, dropped);
            __syncthreads();
            if (threadIdx.x == 0) {
                kept[r] = n - drop_s;
                if (drop_s) atomicAdd(n_dropped, drop_s);
            }
            __syncthreads();
        }
        return;
    }
    const uint32_t wv = threadIdx.x >> 6, lane = (uint32_t)lane_id();
    unsigned long long* Pw = sp + wv * kViewWaveRecs;
    const uint32_t n_small_waves = (gridDim.x - n_big_blocks) * kWavesPerBlock;
    for (uint32_t r = (blockIdx.x - n_big_blocks) * kWavesPerBlock + wv; r < n_reads; r += n_small_waves) {
        const uint32_t off = uniu(offsets[r]), n = uniu(offsets[r + 1]) - off;
        if (n == 0) { if (lane == 0) kept[r] = 0; continue; }
        if (n > kViewWaveRecs) continue;        // a workgroup's job
        if (uniu(cursor[r]) == kViewPresorted) { if (lane == 0) kept[r] = n; continue; }   // its rows are written (k_view_scatter)
        wave_sync();
        for (uint32_t i = lane; i < n; i += 64) Pw[i] = view_pack(keys[off + i], vals[off + i], i);
        wave_sync();
        // A read whose records arrived in call order (k_stream_reads) with one code a call is sorted already -- by position for a
        // forward read, backwards for a reverse one: the network is only run when neither holds (keys without the index bits)
        bool asc = true, desc = true;
        for (uint32_t i = lane; i + 1u < n; i += 64) { const unsigned long long a = Pw[i] >> 12, b = Pw[i + 1u] >> 12; asc = asc && a < b; desc = desc && a > b; }
        const bool all_asc = !__ballot(!asc), all_desc = !__ballot(!desc);
        if (!all_asc && all_desc) {
            for (uint32_t i = lane; 2u * i + 1u < n; i += 64) { const unsigned long long a = Pw[i], b = Pw[n - 1u - i]; Pw[i] = b; Pw[n - 1u - i] = a; }
            wave_sync();
        } else if (!all_asc) {
            view_bitonic1<64>(Pw, n, lane);
        }
        uint32_t dropped = view_emit_rows1<64>(Pw, keys + off, vals + off, n, lane, r, reads[r].pos, rows + off, ordinal);
        uint32_t tot = lane_valu(wave_incl_scan(dropped), 63);
        if (lane == 0) {
            kept[r] = n - tot;
            if (tot) atomicAdd(n_dropped, tot);
        }
    }
}

// Only when rows were dropped: out[new_offsets[r] ...] = the kept rows of read r, in order (one workgroup per read).
__global__ __launch_bounds__(256) void k_view_compact(const ViewRow* __restrict__ rows, const unsigned int* __restrict__ offsets,
                                                      const unsigned int* __restrict__ new_offsets, uint32_t n_reads,
                                                      ViewRow* __restrict__ out) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint32_t off = offsets[r], n = offsets[r + 1] - off;
        if (threadIdx.x == 0) carry_s = new_offsets[r];
        __syncthreads();
        for (uint32_t base = 0; base < n; base += 256) {
            uint32_t i = base + threadIdx.x;
            ViewRow row;
            bool keep = false;
            if (i < n) { row = rows[off + i]; keep = row.read != kViewDropped; }
            uint64_t b = __ballot(keep);
            if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
            __syncthreads();
            uint32_t before = carry_s;
            for (int w = 0; w < wv; w++) before += wsum[w];
            if (keep) out[before + __popcll(b & lanemask_lt())] = row;
            __syncthreads();
            if (threadIdx.x == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
            __syncthreads();
        }
    }
}

}  // namespace mmhip
