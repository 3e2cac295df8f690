// view_kernels.hip.h -- ordering the records of `minimod view` on the device (reference print_view_output,
// src/mod.c:560-626: per read, rows sorted by reference position; add_view_entry, src/mod.c:931-946: the first entry
// of a key wins).
//
// The call kernels append (key, value) records to kViewRegions regions in no particular order.  Here they are
//   1. packed into one contiguous array                                       (k_view_pack)
//   2. radix-sorted on (read, reference position) -- rocPRIM's device radix sort, a plain library primitive
//   3. put into canonical order inside every (read, position) run -- (code, ins_offset), then the order the reference
//      met the calls in -- duplicates of a key dropped in favour of the earliest, and expanded to the 16-byte rows of the C ABI  (k_view_rows)
//   4. compacted (rocPRIM select on the keep flags).
#pragma once
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

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

// regions -> one array.  Every block recomputes the 64-entry prefix of the region counts (cheaper than another launch).
__global__ __launch_bounds__(256) void k_view_pack(const unsigned long long* __restrict__ rk, const unsigned long long* __restrict__ rv,
                                                   const unsigned int* __restrict__ counts, unsigned int cap,
                                                   unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals) {
    __shared__ unsigned long long start[kViewRegions + 1];
    if (threadIdx.x < 64) {
        unsigned int c = counts[threadIdx.x * kViewCountStride];
        if (c > cap) c = cap;
        uint32_t incl = wave_incl_scan(c);   // fewer than 2^32 records per batch (checked by the host)
        start[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) start[0] = 0;
    }
    __syncthreads();
    const unsigned long long n = start[kViewRegions];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
            uint32_t cand = lo + step;
            if (cand < kViewRegions && start[cand] <= i) lo = cand;
        }
        size_t src = (size_t)lo * cap + (size_t)(i - start[lo]);
        keys[i] = rk[src];
        vals[i] = rv[src];
    }
}

// order of two records of one (read, position) run: (code, ins_offset), then the order the reference met them in
// (group, listed before implicit, position in the read) -- the whole value
__device__ __forceinline__ bool view_before(unsigned long long va, unsigned long long vb) { return va < vb; }

// One thread per record; the first record of every (read, position) run orders the run and writes its rows.
__global__ __launch_bounds__(256) void k_view_rows(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals,
                                                   unsigned long long n, unsigned int key_bits, const mm_read_t* __restrict__ reads,
                                                   ViewRow* __restrict__ rows, uint8_t* __restrict__ keep) {
    const unsigned long long kmask = (1ull << key_bits) - 1ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long k = keys[i] & kmask;
        if (i > 0 && (keys[i - 1] & kmask) == k) continue;   // not the head of its run
        unsigned long long j = i + 1;
        while (j < n && (keys[j] & kmask) == k) j++;
        const unsigned long long len = j - i;
        if (len > 1) {
            if (len <= 16) {   // insertion sort
                for (unsigned long long a = i + 1; a < j; a++) {
                    unsigned long long kv = keys[a], vv = vals[a];
                    unsigned long long b = a;
                    while (b > i && view_before(vv, vals[b - 1])) { keys[b] = keys[b - 1]; vals[b] = vals[b - 1]; b--; }
                    keys[b] = kv; vals[b] = vv;
                }
            } else {           // heap sort in place (a long insertion puts thousands of records on one anchor)
                unsigned long long* K = keys + i;
                unsigned long long* V = vals + i;
                auto sift = [&](unsigned long long root, unsigned long long end) {
                    for (;;) {
                        unsigned long long child = 2 * root + 1;
                        if (child >= end) break;
                        if (child + 1 < end && view_before(V[child], V[child + 1])) child++;
                        if (!view_before(V[root], V[child])) break;
                        unsigned long long tk = K[root], tv = V[root];
                        K[root] = K[child]; V[root] = V[child]; K[child] = tk; V[child] = tv;
                        root = child;
                    }
                };
                for (unsigned long long s = len / 2; s-- > 0;) sift(s, len);
                for (unsigned long long e = len - 1; e > 0; e--) {
                    unsigned long long tk = K[0], tv = V[0];
                    K[0] = K[e]; V[0] = V[e]; K[e] = tk; V[e] = tv;
                    sift(0, e);
                }
            }
        }
        const uint32_t read = (uint32_t)(k >> 28);
        const int32_t pos = reads[read].pos + (int32_t)(k & 0x0FFFFFFFull) - 1;
        unsigned long long prev = ~0ull;
        for (unsigned long long a = i; a < j; a++) {
            const unsigned long long kv = keys[a], vv = vals[a];
            ViewRow r;
            r.read = read; r.pos = pos; r.read_pos = (uint32_t)(vv & 0x0FFFFFFFull);
            r.ins_offset = (uint16_t)((vv >> 40) & 0xFFFFull); r.code = (uint8_t)(vv >> 56); r.prob = (uint8_t)(kv >> 56);
            rows[a] = r;
            keep[a] = (vv >> 40) != prev;   // same (code, ins_offset) as the record in front: a later entry of the same key
            prev = vv >> 40;
        }
    }
}

}  // namespace mmhip
