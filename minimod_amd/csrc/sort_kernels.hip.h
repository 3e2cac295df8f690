// sort_kernels.hip.h -- finalize-side helpers for the updates that do not fit the dense counter planes:
// they are appended to regional lists as (64-bit key, increment) records (side_insert, freq_kernels.hip.h); a compaction
// gathers the lists, orders the records by key with an 8-bit LSD radix sort and adds up equal keys, so the host receives
// unique, ordered (key, counts) pairs instead of one record per call.
#pragma once
#include "freq_kernels.hip.h"

namespace mmhip {

// the regions' records -> one (key, value) array pair: region r's first n[r] records go to [off[r], off[r] + n[r])
__global__ __launch_bounds__(256) void k_side_gather(const unsigned long long* __restrict__ tab, unsigned long long cap_r, const unsigned long long* __restrict__ off,
                                                     unsigned long long* __restrict__ out_k, unsigned long long* __restrict__ out_v) {
    const uint32_t r = blockIdx.y;
    const unsigned long long n = off[r + 1] - off[r];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const ulonglong2 rec = *reinterpret_cast<const ulonglong2*>(tab + 2ull * ((unsigned long long)r * cap_r + i));
        out_k[off[r] + i] = rec.x; out_v[off[r] + i] = rec.y;
    }
}
// sorted (key, value) pairs -> one pair per key, values added up: (1) the keys' first entries ("heads") counted per tile of
// 2048, (2) the counts scanned (k_radix_scan), (3) every head writes its key and the sum of its run
constexpr int kReduceTile = 2048;
__global__ __launch_bounds__(256) void k_reduce_count(const unsigned long long* __restrict__ k, unsigned long long n, uint32_t* __restrict__ tile_heads) {
    __shared__ uint32_t part[4];
    const unsigned long long base = (unsigned long long)blockIdx.x * kReduceTile;
    uint32_t c = 0;
    for (int j = threadIdx.x; j < kReduceTile; j += 256) {
        const unsigned long long i = base + (unsigned long long)j;
        if (i < n && (i == 0 || k[i] != k[i - 1])) c++;
    }
    for (int d = 32; d; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_heads[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_reduce_emit(const unsigned long long* __restrict__ k, const unsigned long long* __restrict__ v, unsigned long long n,
                                                     const uint32_t* __restrict__ tile_off, unsigned long long* __restrict__ out_k, unsigned long long* __restrict__ out_v) {
    __shared__ uint32_t wsum[4];
    const unsigned long long base = (unsigned long long)blockIdx.x * kReduceTile;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t run = tile_off[blockIdx.x];
    for (int j0 = 0; j0 < kReduceTile; j0 += 256) {
        const unsigned long long i = base + (unsigned long long)(j0 + (int)threadIdx.x);
        const bool head = i < n && (i == 0 || k[i] != k[i - 1]);
        const uint64_t b = __ballot(head);
        if (lane == 0) wsum[wv] = (uint32_t)__popcll(b);
        __syncthreads();
        uint32_t before = run;
        for (int w = 0; w < wv; w++) before += wsum[w];
        if (head) {
            const unsigned long long key = k[i];
            unsigned long long lo = 0, hi = 0;   // the two 32-bit halves are added on their own (n_called must not carry into n_mod)
            for (unsigned long long j = i; j < n && k[j] == key; j++) { lo += v[j] & 0xFFFFFFFFull; hi += v[j] >> 32; }
            // (a count past 2^32 - 1 becomes {n_called 0, n_mod 1}: more modified than called is what mm_freq_finalize reports as MM_E_OVERFLOW)
            const unsigned long long at = (unsigned long long)before + (unsigned long long)__popcll(b & ((1ull << lane) - 1ull));
            out_k[at] = key;
            out_v[at] = lo > 0xFFFFFFFFull ? (1ull << 32) : (lo | (hi << 32));
        }
        run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
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
