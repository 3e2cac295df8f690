// sort_kernels.hip.h -- finalize-side helpers for the updates that do not fit the dense counter planes:
// they are counted in a device hash table keyed by one 64-bit word (side_insert, freq_kernels.hip.h); at finalize the
// occupied slots are compacted and ordered by key with an 8-bit LSD radix sort, so the host receives unique, ordered
// (key, counts) pairs instead of one record per call.
#pragma once
#include "freq_kernels.hip.h"

namespace mmhip {

__global__ __launch_bounds__(256) void k_side_clear(unsigned long long* __restrict__ tab, unsigned long long cap) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < cap; i += (unsigned long long)gridDim.x * 256u) {
        tab[2 * i] = kSideEmpty; tab[2 * i + 1] = 0ull;
    }
}

// occupied slots of the table
__global__ __launch_bounds__(256) void k_side_count(const unsigned long long* __restrict__ tab, unsigned long long cap, unsigned long long* __restrict__ counter) {
    unsigned long long c = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < cap; i += (unsigned long long)gridDim.x * 256u)
        c += tab[2 * i] != kSideEmpty ? 1ull : 0ull;
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane_id() == 0 && c) atomicAdd(counter, c);
}

// occupied slots -> dense (key, value) arrays, in any order (one wave-aggregated reservation per wave)
__global__ __launch_bounds__(256) void k_side_compact(const unsigned long long* __restrict__ tab,
                                                      unsigned long long cap, unsigned long long* __restrict__ out_k,
                                                      unsigned long long* __restrict__ out_v, unsigned long long* __restrict__ counter) {
    for (unsigned long long i0 = (unsigned long long)blockIdx.x * 256u; i0 < cap; i0 += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long i = i0 + threadIdx.x;
        const unsigned long long k = i < cap ? tab[2 * i] : kSideEmpty;
        const bool have = k != kSideEmpty;
        const uint64_t m = __ballot(have);
        if (!m) continue;
        unsigned long long base = 0;
        const int leader = __ffsll((unsigned long long)m) - 1;
        if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
        base = __shfl(base, leader, 64);
        if (have) {
            const unsigned long long at = base + (unsigned long long)__popcll(m & lanemask_lt());
            out_k[at] = k; out_v[at] = tab[2 * i + 1];
        }
    }
}

constexpr int kSortTile = 2048;   // keys per workgroup (one wavefront walks them 64 at a time, which keeps the pass stable)

__global__ __launch_bounds__(64) void k_radix_hist(const unsigned long long* __restrict__ keys, unsigned long long n, int shift,
                                                   uint32_t* __restrict__ block_hist, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) h[d] = 0u;
    wave_sync();
    const unsigned long long lo = (unsigned long long)blockIdx.x * kSortTile;
    for (int j = 0; j < kSortTile; j += 64) {
        const unsigned long long i = lo + (unsigned long long)(j + lane);
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    wave_sync();
    for (int d = lane; d < 256; d += 64) block_hist[(size_t)d * n_blocks + blockIdx.x] = h[d];
}

// exclusive scan over the digit-major histogram matrix (256 x n_blocks), one workgroup
__global__ __launch_bounds__(1024) void k_radix_scan(uint32_t* __restrict__ a, unsigned long long n) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0u;
    __syncthreads();
    for (unsigned long long i0 = 0; i0 < n; i0 += 1024u) {
        const unsigned long long i = i0 + (unsigned long long)t;
        const uint32_t v = i < n ? a[i] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if ((t & 63) == 63) wsum[t >> 6] = incl;
        __syncthreads();
        uint32_t before = carry_s + incl - v;
        for (int w = 0; w < (t >> 6); w++) before += wsum[w];
        if (i < n) a[i] = before;
        __syncthreads();
        if (t == 1023) carry_s = before + v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_radix_scatter(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                      unsigned long long n, int shift, const uint32_t* __restrict__ block_off, uint32_t n_blocks,
                                                      unsigned long long* __restrict__ out_k, unsigned long long* __restrict__ out_v) {
    __shared__ uint32_t base[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) base[d] = block_off[(size_t)d * n_blocks + blockIdx.x];
    wave_sync();
    const unsigned long long lo = (unsigned long long)blockIdx.x * kSortTile;
    for (int j = 0; j < kSortTile; j += 64) {
        const unsigned long long i = lo + (unsigned long long)(j + lane);
        const bool have = i < n;
        const unsigned long long k = have ? keys[i] : 0ull;
        const uint32_t d = (uint32_t)(k >> shift) & 255u;
        // lanes with the same digit (eight ballots), this lane's place among them
        uint64_t peers = __ballot(have);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t bb = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bb : ~bb;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lanemask_lt());
        uint32_t dst = 0;
        if (have) dst = base[d] + before;
        wave_sync();
        if (have && before == 0u) base[d] += (uint32_t)__popcll(peers);   // the first lane of every digit moves its cursor
        wave_sync();
        if (have) { out_k[dst] = k; out_v[dst] = vals[i]; }
    }
}

}  // namespace mmhip
