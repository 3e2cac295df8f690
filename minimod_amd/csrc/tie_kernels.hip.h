// tie_kernels.hip.h -- the reference's order of rows that tie on (contig, start), made on the device.
//
// print_freq_output (reference src/mod.c:644-664) walks the core hash table slot by slot and sorts that array with ks_introsort under a
// comparator that looks at contig and start only (cmp_key_fast, src/mod.c:59-87): rows of one (contig, start) come out in an order that
// is a function of every key's slot in the core table -- i.e. of the order of first insertion (merge_freq_maps, src/mod.c:743-774: reads
// in file order, a read's keys in the slot order of the read's own table) and of khash's growth history -- and of what the unstable sort
// does to that array.  csrc/host/tieorder.c restates both serially and is the checker; here they are data-parallel:
//
//   T1  k_tie_reads      a read's calls (view rows of a second handle, mm_freq_opts_t.view = 2) put back into the order
//                        freq_view_single met them, thresholded, turned into keys + X31 hashes of make_key's string
//                        (src/mod.c:428-439, src/khash.h:486-494), run through the read's OWN khash (update_freq_map, src/mod.c:883-929;
//                        kh_put / kh_resize, src/khash.h:242-420) -- a thread per read, serial inside it as the reference is -- and the
//                        table's slot order stamped: (read's serial number << 24 | slot rank)
//   T2  k_stamp_*        every key's SMALLEST stamp kept in an open-addressing table in HBM (CAS on the key, atomicMin on the stamp):
//                        its first insertion into the core table
//   T3  k_place_*, k_grow_*   the core table WITHOUT walking it key by key:
//                        * between two growths keys enter at the first free slot of their probe path (i += ++step) in first-insertion
//                          order.  Every key of such an epoch proposes its rank at its slot with atomicMin; who finds a smaller rank
//                          there moves on, who is displaced by a smaller rank later moves on from there.  The fixpoint is the serial
//                          table: a key sits at the first slot of its path that no EARLIER key holds.
//                        * a growth rehashes in place with kick-outs (kh_resize): bucket j's key goes to its slot in the new table, and
//                          if that slot, as a bucket of the OLD table, holds a key not yet moved, that key goes next.  So the new table
//                          is "insert into an empty table in chain order", and the chain order depends on where keys land.  The chains:
//                          succ(x) = the old bucket x lands on; a key is taken up by the walk that starts at the smallest bucket among
//                          its predecessors.  Landing := home slot gives a first order, the placement under that order new landings;
//                          repeated until nothing moves (the pair is unique, by induction along the order; measured: 2 - 3 passes).
//   T4  k_qs_*           ks_introsort (src/ksort.h:180-230): a partition swaps the k-th element from the left that is not smaller than the
//                        pivot with the k-th from the right that is not bigger while they have not crossed -- ranks by prefix sums, all
//                        swaps at once, all segments of a level side by side; segments of up to 2048 elements are finished by a thread
//                        each running ksort's own loop; the insertion sort that ends it all is a STABLE sort of what the partitions left
//                        (an LSD radix sort here).
// Nothing here is shaped for MFMA: integer scans, gathers and atomics, bound by launch count for small inputs and by HBM for large ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "minimod_hip.h"

namespace mmtie {

typedef unsigned long long u64;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr u64 kNone64 = ~0ull;

__device__ __forceinline__ int lane() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ u64 shfl_up64(u64 v, int d) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_incl_scan64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u64 o = shfl_up64(v, d); if (lane() >= d) v += o; }
    return v;
}

// ---------------------------------------------------------------- inclusive scan of 64-bit words (two 32-bit counts ride in one word)
constexpr int kScanTile = 2048;   // 256 threads x 8 consecutive words
__global__ __launch_bounds__(256) void k_scan_reduce(const u64* __restrict__ a, u64 n, u64* __restrict__ tile_sum) {
    __shared__ u64 part[4];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * 8u;
    u64 s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) if (base + j < n) s += a[base + j];
    s = wave_incl_scan64(s);
    if (lane() == 63) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void k_scan_spine(u64* __restrict__ t, uint32_t n) {   // exclusive, in place, one workgroup
    __shared__ u64 wsum[16];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += 1024u) {
        const uint32_t i = i0 + threadIdx.x;
        const u64 v = i < n ? t[i] : 0ull;
        const u64 incl = wave_incl_scan64(v);
        if (lane() == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        u64 before = carry + incl - v;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < n) t[i] = before;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(u64* __restrict__ a, u64 n, const u64* __restrict__ tile_off) {   // a := inclusive prefix sums
    __shared__ u64 part[4];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * 8u;
    u64 v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = base + j < n ? a[base + j] : 0ull; s += v[j]; }
    const u64 incl = wave_incl_scan64(s);
    if (lane() == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 run = tile_off[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
#pragma unroll
    for (int j = 0; j < 8; j++) { run += v[j]; if (base + j < n) a[base + j] = run; }
}

// ---------------------------------------------------------------- T3: the core table
__global__ __launch_bounds__(256) void k_fill32(uint32_t* __restrict__ a, u64 n, uint32_t v) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = v;
}
__global__ __launch_bounds__(256) void k_fill64(u64* __restrict__ a, u64 n, u64 v) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = v;
}
// an epoch's keys (ranks lo .. hi - 1) at their home slots
__global__ __launch_bounds__(256) void k_place_init(const uint32_t* __restrict__ hash, uint32_t lo, uint32_t hi, uint32_t mask, uint32_t* __restrict__ cur, uint32_t* __restrict__ stp) {
    const uint32_t r = lo + blockIdx.x * 256u + threadIdx.x;
    if (r < hi) { cur[r] = hash[r] & mask; stp[r] = 0u; }
}
// one round: a key that does not hold its slot proposes itself there and moves along its path while it meets smaller ranks
__global__ __launch_bounds__(256) void k_place_round(uint32_t* __restrict__ tab, uint32_t mask, uint32_t lo, uint32_t hi, uint32_t* __restrict__ cur, uint32_t* __restrict__ stp,
                                                     uint32_t* __restrict__ changed) {
    const uint32_t r = lo + blockIdx.x * 256u + threadIdx.x;
    if (r >= hi) return;
    uint32_t i = cur[r];
    if (tab[i] == r) return;
    uint32_t s = stp[r];
    for (;;) {
        const uint32_t w = atomicMin(&tab[i], r);
        if (w >= r) break;
        s++; i = (i + s) & mask;
    }
    cur[r] = i; stp[r] = s;
    *changed = 1u;
}
// growth: where every key of the old table lands first (its home slot in the new one)
__global__ __launch_bounds__(256) void k_grow_home(const uint32_t* __restrict__ told, uint32_t C, const uint32_t* __restrict__ hash, uint32_t mask2, uint32_t* __restrict__ land) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C) return;
    const uint32_t r = told[s];
    land[s] = r == kNone ? kNone : (hash[r] & mask2);
}
// pred[l] = the old bucket whose key lands on old bucket l (landings are distinct: no two writers)
__global__ __launch_bounds__(256) void k_grow_succ(const uint32_t* __restrict__ told, uint32_t C, const uint32_t* __restrict__ land, uint32_t* __restrict__ pred) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C || told[s] == kNone) return;
    const uint32_t l = land[s];
    if (l < C && l != s && told[l] != kNone) pred[l] = s;
}
// a key is taken up by the walk that starts at the smallest bucket among its predecessors (itself included): word = that bucket << 32 | steps from it.
// Also resets the key to its home slot for the placement that follows.
constexpr uint32_t kChainLimit = 4096;
__global__ __launch_bounds__(256) void k_grow_prio(const uint32_t* __restrict__ told, uint32_t C, const uint32_t* __restrict__ pred, const uint32_t* __restrict__ hash, uint32_t mask2,
                                                   u64* __restrict__ word, uint32_t* __restrict__ cur, uint32_t* __restrict__ stp, uint32_t* __restrict__ fail) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C) return;
    const uint32_t r = told[s];
    if (r == kNone) return;
    uint32_t best = s, dist = 0, d = 1, p = pred[s];
    while (p != kNone && p != s) {
        if (p < best) { best = p; dist = d; }
        p = pred[p]; d++;
        if (d > kChainLimit) { *fail = 1u; break; }
    }
    word[s] = ((u64)best << 32) | (u64)dist;
    cur[s] = hash[r] & mask2; stp[s] = 0u;
}
__global__ __launch_bounds__(256) void k_grow_round(const uint32_t* __restrict__ told, uint32_t C, const u64* __restrict__ word, u64* __restrict__ tw, uint32_t mask2,
                                                    uint32_t* __restrict__ cur, uint32_t* __restrict__ stp, uint32_t* __restrict__ changed) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C || told[s] == kNone) return;
    const u64 me = word[s];
    uint32_t i = cur[s];
    if (tw[i] == me) return;
    uint32_t st = stp[s];
    for (;;) {
        const u64 w = atomicMin(&tw[i], me);
        if (w >= me) break;
        st++; i = (i + st) & mask2;
    }
    cur[s] = i; stp[s] = st;
    *changed = 1u;
}
__global__ __launch_bounds__(256) void k_grow_check(const uint32_t* __restrict__ told, uint32_t C, const uint32_t* __restrict__ cur, uint32_t* __restrict__ land, uint32_t* __restrict__ moved) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C || told[s] == kNone) return;
    if (land[s] != cur[s]) { land[s] = cur[s]; *moved = 1u; }
}
__global__ __launch_bounds__(256) void k_grow_commit(const uint32_t* __restrict__ told, uint32_t C, const uint32_t* __restrict__ land, uint32_t* __restrict__ tnew) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= C) return;
    const uint32_t r = told[s];
    if (r != kNone) tnew[land[s]] = r;
}
// ---- T3 for the SMALL tables, in one workgroup (round 5's end).  The host's loop around k_place_* / k_grow_* pays a launch per round and a
// stream synchronisation per fixpoint test, and a table of a few thousand buckets gives a launch nothing to do: of the 1 311 launches a
// 3-million-key run needed, some 700 were for the eleven growths from 4 to 8 192 buckets.  This kernel runs the same steps -- the very bodies
// above, a thread striding over the keys / buckets -- for every epoch up to `Cstop` buckets, with a workgroup barrier where the host had a
// kernel boundary: stores and atomics drained in front of the barrier (release), the vector L1 invalidated behind it (acquire), which is what a
// kernel boundary gives the next launch's loads.  It leaves the table as the host's loop would find it at that point and says where it stopped.
struct SmallState { uint32_t C, tcur, finished, fail; unsigned long long done; };
__device__ __forceinline__ void wg_phase() { __threadfence(); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// a workgroup's flag, read by everybody BEFORE anybody may go on and clear it for the next round: without the second barrier thread 0 -- one wavefront ahead of
// the others -- cleared the flag while a slower wavefront had not read it yet; that wavefront left the loop alone, and the barriers behind paired up wrongly
// (seen only with other processes' kernels on the device: tests/test_hip_fuzz_gpu.py's seven campaigns at once, one run in four)
#ifdef MM_TIE_RACY_FLAGS   // (diagnostic build: the race as it was, to show that tests/test_hip_tie_gpu.py::test_small_tables_with_a_busy_device finds it)
__device__ __forceinline__ bool wg_flag(const uint32_t* f) { return *(const volatile uint32_t*)f != 0u; }
#else
__device__ __forceinline__ bool wg_flag(const uint32_t* f) { const bool v = *(const volatile uint32_t*)f != 0u; __syncthreads(); return v; }
#endif
__global__ __launch_bounds__(1024) void k_small_epochs(const uint32_t* __restrict__ hash, u64 n, int put_after_last, uint32_t Cstop, uint32_t* tab0, uint32_t* tab1,
                                                       uint32_t* cur, uint32_t* stp, uint32_t* land, uint32_t* pred, u64* word, u64* tw, SmallState* out, unsigned long long* counts) {
    __shared__ uint32_t sh_flag[2];   // changed / moved (or failed)
    const uint32_t T = 1024u, t = threadIdx.x;
    uint32_t C = 4u, tcur = 0u, finished = 0u, fail = 0u;
    u64 done = 0;
    unsigned long long n_growths = 0, n_passes = 0, n_rounds = 0;
    for (uint32_t s = t; s < C; s += T) tab0[s] = kNone;
    wg_phase();
    for (;;) {
        uint32_t* tab = tcur ? tab1 : tab0;
        const uint32_t U = (uint32_t)((double)C * 0.77 + 0.5);
        const u64 hi64 = n < (u64)U ? n : (u64)U;
        if (hi64 > done) {
            // an epoch's keys at their home slots, then rounds until nothing moves (k_place_init / k_place_round)
            const uint32_t lo = (uint32_t)done, hi = (uint32_t)hi64, mask = C - 1u;
            for (uint32_t r = lo + t; r < hi; r += T) { cur[r] = hash[r] & mask; stp[r] = 0u; }
            for (;;) {
                if (t == 0) sh_flag[0] = 0u;
                wg_phase();
                for (uint32_t r = lo + t; r < hi; r += T) {
                    uint32_t i = cur[r];
                    if (tab[i] == r) continue;
                    uint32_t st = stp[r];
                    for (;;) {
                        const uint32_t w = atomicMin(&tab[i], r);
                        if (w >= r) break;
                        st++; i = (i + st) & mask;
                    }
                    cur[r] = i; stp[r] = st;
                    sh_flag[0] = 1u;
                }
                n_rounds++;
                wg_phase();
                if (!wg_flag(&sh_flag[0])) break;
            }
            done = hi64;
        }
        bool grow = false;
        if (done == n) { grow = put_after_last && n >= (u64)U; finished = grow ? 2u : 1u; }   // (2: finished behind the growth below)
        else if (C >= Cstop) break;                                                            // the host's loop goes on from here: its next step is this growth
        else grow = true;
        if (!grow) break;
        // ---- the growth C -> 2 C (k_grow_home, then passes of succ / prio / rounds / check until the landings hold, then commit)
        {
            const uint32_t mask2 = 2u * C - 1u;
            uint32_t* tnew = tcur ? tab0 : tab1;
            n_growths++;
            for (uint32_t s = t; s < C; s += T) { const uint32_t r = tab[s]; land[s] = r == kNone ? kNone : (hash[r] & mask2); }
            for (;;) {
                n_passes++;
                for (uint32_t s = t; s < C; s += T) pred[s] = kNone;
                wg_phase();
                for (uint32_t s = t; s < C; s += T) {
                    if (tab[s] == kNone) continue;
                    const uint32_t l = land[s];
                    if (l < C && l != s && tab[l] != kNone) pred[l] = s;
                }
                for (uint32_t s = t; s < 2u * C; s += T) tw[s] = kNone64;
                if (t == 0) sh_flag[1] = 0u;
                wg_phase();
                for (uint32_t s = t; s < C; s += T) {
                    const uint32_t r = tab[s];
                    if (r == kNone) continue;
                    uint32_t best = s, dist = 0, d = 1, p = pred[s];
                    while (p != kNone && p != s) {
                        if (p < best) { best = p; dist = d; }
                        p = pred[p]; d++;
                        if (d > kChainLimit) { sh_flag[1] = 1u; break; }
                    }
                    word[s] = ((u64)best << 32) | (u64)dist;
                    cur[s] = hash[r] & mask2; stp[s] = 0u;
                }
                for (;;) {
                    if (t == 0) sh_flag[0] = 0u;
                    wg_phase();
                    for (uint32_t s = t; s < C; s += T) {
                        if (tab[s] == kNone) continue;
                        const u64 me = word[s];
                        uint32_t i = cur[s];
                        if (tw[i] == me) continue;
                        uint32_t st = stp[s];
                        for (;;) {
                            const u64 w = atomicMin(&tw[i], me);
                            if (w >= me) break;
                            st++; i = (i + st) & mask2;
                        }
                        cur[s] = i; stp[s] = st;
                        sh_flag[0] = 1u;
                    }
                    n_rounds++;
                    wg_phase();
                    if (!wg_flag(&sh_flag[0])) break;
                }
                if (wg_flag(&sh_flag[1])) { fail = 1u; break; }
                if (t == 0) sh_flag[1] = 0u;
                wg_phase();
                for (uint32_t s = t; s < C; s += T) {
                    if (tab[s] == kNone) continue;
                    if (land[s] != cur[s]) { land[s] = cur[s]; sh_flag[1] = 1u; }
                }
                wg_phase();
                const bool moved = wg_flag(&sh_flag[1]);
                if (!moved) break;
            }
            if (fail) break;
            for (uint32_t s = t; s < 2u * C; s += T) tnew[s] = kNone;
            wg_phase();
            for (uint32_t s = t; s < C; s += T) { const uint32_t r = tab[s]; if (r != kNone) tnew[land[s]] = r; }
            wg_phase();
            tcur ^= 1u; C *= 2u;
        }
        if (finished) { finished = 1u; break; }
    }
    if (t == 0) {
        out->C = C; out->tcur = tcur; out->finished = finished ? 1u : 0u; out->fail = fail; out->done = done;
        counts[0] = n_growths; counts[1] = n_passes; counts[2] = n_rounds;
    }
}

// the table's keys in slot order: flags -> (scan) -> gather
__global__ __launch_bounds__(256) void k_slot_flags(const uint32_t* __restrict__ tab, u64 C, u64* __restrict__ f) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < C) f[i] = tab[i] != kNone ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_slot_gather(const uint32_t* __restrict__ tab, u64 C, const u64* __restrict__ incl, const long long* __restrict__ sortkey,
                                                     long long* __restrict__ key, uint32_t* __restrict__ id) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= C) return;
    const uint32_t r = tab[i];
    if (r == kNone) return;
    const u64 at = incl[i] - 1ull;
    key[at] = sortkey[r]; id[at] = r;
}

// ---------------------------------------------------------------- T4: ks_introsort
struct Seg { uint32_t s, t; int32_t d; uint32_t pad; };
// per segment of the level: the depth budget, ksort's pivot (src/ksort.h: k = the middle + 1, then the three-way choice), moved to the right end
__global__ __launch_bounds__(256) void k_qs_pivot(Seg* __restrict__ segs, uint32_t n_seg, long long* __restrict__ key, uint32_t* __restrict__ id, long long* __restrict__ rp, uint32_t* __restrict__ fail) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_seg) return;
    Seg sg = segs[g];
    sg.d -= 1;
    if (sg.d == 0) { *fail = 2u; }   // comb sort over a segment this big: not made on the device (never seen: the budget is twice the binary depth)
    segs[g].d = sg.d;
    const uint32_t s = sg.s, t = sg.t;
    uint32_t k = s + ((t - s) >> 1) + 1u;
    const long long ks = key[s], kt = key[t], kk = key[k];
    if (kk < ks) { if (kk < kt) k = t; }
    else k = kt < ks ? s : t;
    const long long p = key[k];
    rp[g] = p;
    if (k != t) { const uint32_t ik = id[k]; key[k] = kt; id[k] = id[t]; key[t] = p; id[t] = ik; }
}
// per element: is it a stopper of the scan from the left (not smaller than the pivot; positions s+1 .. t) / from the right (not bigger; s+1 .. t-1)
__global__ __launch_bounds__(256) void k_qs_flags(const uint32_t* __restrict__ segof, const Seg* __restrict__ segs, const long long* __restrict__ key, const long long* __restrict__ rp,
                                                  u64 n, u64* __restrict__ f) {
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = segof[p];
    u64 v = 0;
    if (g != kNone) {
        const Seg sg = segs[g];
        const long long k = key[p], r = rp[g];
        if (p > sg.s && !(k < r)) v |= 1ull;
        if (p > sg.s && p < sg.t && !(r < k)) v |= 1ull << 32;
    }
    f[p] = v;
}
// the k-th stopper from the left / right of its segment writes its place at [s + k]
__global__ __launch_bounds__(256) void k_qs_scatter(const uint32_t* __restrict__ segof, const Seg* __restrict__ segs, const long long* __restrict__ key, const long long* __restrict__ rp,
                                                    u64 n, const u64* __restrict__ incl, uint32_t* __restrict__ lpos, uint32_t* __restrict__ rpos) {
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = segof[p];
    if (g == kNone) return;
    const Seg sg = segs[g];
    if (p <= sg.s) return;
    const long long k = key[p], r = rp[g];
    const u64 mine = incl[p];
    if (!(k < r)) lpos[sg.s + ((uint32_t)mine - (uint32_t)incl[sg.s])] = (uint32_t)p;
    if (p < sg.t && !(r < k)) rpos[sg.s + ((uint32_t)(incl[sg.t] >> 32) - (uint32_t)(mine >> 32)) + 1u] = (uint32_t)p;
}
// pair k of a segment swaps while the left stopper lies in front of the right one; the last such k is the segment's swap count
__global__ __launch_bounds__(256) void k_qs_swap(const uint32_t* __restrict__ segof, const Seg* __restrict__ segs, u64 n, const u64* __restrict__ incl,
                                                 const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ rpos, long long* __restrict__ key, uint32_t* __restrict__ id, uint32_t* __restrict__ nswap) {
    const u64 q = (u64)blockIdx.x * 256u + threadIdx.x;
    if (q >= n) return;
    const uint32_t g = segof[q];
    if (g == kNone) return;
    const Seg sg = segs[g];
    if (q <= sg.s) return;
    const uint32_t k = (uint32_t)q - sg.s;
    const u64 a = incl[sg.s], b = incl[sg.t];
    const uint32_t nl = (uint32_t)b - (uint32_t)a, nr = (uint32_t)(b >> 32) - (uint32_t)(a >> 32);
    const uint32_t m = nl < nr ? nl : nr;
    if (k > m) return;
    const uint32_t x = lpos[q], y = rpos[q];
    if (!(x < y)) return;
    const long long kx = key[x]; const uint32_t ix = id[x];
    key[x] = key[y]; id[x] = id[y]; key[y] = kx; id[y] = ix;
    if (k == m || !(lpos[q + 1] < rpos[q + 1])) nswap[g] = k;
}
// per segment: where the pivot goes, the children (those of more than 16 elements: src/ksort.h pushes / continues with nothing smaller),
// handed to the next level or, from 2048 elements down, to the list of segments a thread finishes
#ifndef MM_TIE_SMALL_SEG
#define MM_TIE_SMALL_SEG 64   /* (2 048 until the end of round 5: 3 000 threads then finished a thousand keys each, 37 of a 3-million-key call's 57 ms; 512: 5 ms, 256: 1.8, 64: 0.3, for four more levels of 0.12 ms) */
#endif
constexpr uint32_t kSmallSeg = MM_TIE_SMALL_SEG;
__global__ __launch_bounds__(256) void k_qs_finish(const Seg* __restrict__ segs, uint32_t n_seg, const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ rpos, const uint32_t* __restrict__ nswap,
                                                   long long* __restrict__ key, uint32_t* __restrict__ id, Seg* __restrict__ next, uint32_t* __restrict__ n_next, Seg* __restrict__ small, uint32_t* __restrict__ n_small,
                                                   uint32_t* __restrict__ pivot_at, uint32_t* __restrict__ child) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_seg) return;
    const Seg sg = segs[g];
    const uint32_t K = nswap[g];
    uint32_t i = lpos[sg.s + K + 1u];
    const uint32_t jr = K > 0u ? rpos[sg.s + K] : sg.t;
    if (jr < i) i = jr;
    { const long long ki = key[i]; const uint32_t ii = id[i]; key[i] = key[sg.t]; id[i] = id[sg.t]; key[sg.t] = ki; id[sg.t] = ii; }
    pivot_at[g] = i;
    uint32_t cl = kNone, cr = kNone;
    if (i - sg.s > 16u) {
        Seg c; c.s = sg.s; c.t = i - 1u; c.d = sg.d; c.pad = 0;
        if (i - sg.s <= kSmallSeg) small[atomicAdd(n_small, 1u)] = c;
        else { cl = atomicAdd(n_next, 1u); next[cl] = c; }
    }
    if (sg.t - i > 16u) {
        Seg c; c.s = i + 1u; c.t = sg.t; c.d = sg.d; c.pad = 0;
        if (sg.t - i <= kSmallSeg) small[atomicAdd(n_small, 1u)] = c;
        else { cr = atomicAdd(n_next, 1u); next[cr] = c; }
    }
    child[2u * g] = cl; child[2u * g + 1u] = cr;
}
__global__ __launch_bounds__(256) void k_qs_assign(uint32_t* __restrict__ segof, u64 n, const uint32_t* __restrict__ pivot_at, const uint32_t* __restrict__ child) {
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = segof[p];
    if (g == kNone) return;
    const uint32_t i = pivot_at[g];
    segof[p] = p < i ? child[2u * g] : (p > i ? child[2u * g + 1u] : kNone);
}
// a thread finishes a small segment with ksort's own loop (no final insertion sort: that one runs over everything, as a stable sort)
__device__ inline void qs_swap(long long* key, uint32_t* id, uint32_t a, uint32_t b) {
    const long long k = key[a]; const uint32_t i = id[a];
    key[a] = key[b]; id[a] = id[b]; key[b] = k; id[b] = i;
}
__device__ inline void qs_insertion(long long* key, uint32_t* id, uint32_t s, uint32_t e) {   // [s, e)
    for (uint32_t i = s + 1u; i < e; ++i)
        for (uint32_t j = i; j > s && key[j] < key[j - 1u]; --j) qs_swap(key, id, j, j - 1u);
}
__device__ inline void qs_comb(long long* key, uint32_t* id, uint32_t s, uint32_t n) {   // ks_combsort (src/ksort.h:140-160) on [s, s + n)
    const double shrink = 1.2473309501039786540366528676643;
    int swapped;
    uint32_t gap = n;
    do {
        if (gap > 2u) { gap = (uint32_t)(gap / shrink); if (gap == 9u || gap == 10u) gap = 11u; }
        swapped = 0;
        for (uint32_t i = s; i < s + n - gap; ++i) {
            const uint32_t j = i + gap;
            if (key[j] < key[i]) { qs_swap(key, id, i, j); swapped = 1; }
        }
    } while (swapped || gap > 2u);
    if (gap != 1u) qs_insertion(key, id, s, s + n);
}
__global__ __launch_bounds__(64) void k_qs_small(const Seg* __restrict__ small, uint32_t n_small, long long* __restrict__ key, uint32_t* __restrict__ id) {
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g >= n_small) return;
    Seg stack[40];
    int top = 0;
    long long s = small[g].s, t = small[g].t;   // (signed: t = i - 1 may pass below s)
    int d = small[g].d;
    for (;;) {
        if (s < t) {
            if (--d == 0) { qs_comb(key, id, (uint32_t)s, (uint32_t)(t - s) + 1u); t = s; continue; }
            long long i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (key[k] < key[i]) { if (key[k] < key[j]) k = j; }
            else k = key[j] < key[i] ? i : j;
            const long long rp = key[k];
            if (k != t) qs_swap(key, id, (uint32_t)k, (uint32_t)t);
            for (;;) {
                do ++i; while (key[i] < rp);
                do --j; while (i <= j && rp < key[j]);
                if (j <= i) break;
                qs_swap(key, id, (uint32_t)i, (uint32_t)j);
            }
            qs_swap(key, id, (uint32_t)i, (uint32_t)t);
            if (i - s > t - i) {
                if (i - s > 16) { stack[top].s = (uint32_t)s; stack[top].t = (uint32_t)(i - 1); stack[top].d = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { stack[top].s = (uint32_t)(i + 1); stack[top].t = (uint32_t)t; stack[top].d = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) return;
            --top; s = stack[top].s; t = stack[top].t; d = stack[top].d;
        }
    }
}

// ---------------------------------------------------------------- a stable LSD radix sort, eight bits a pass (64-bit key, 32-bit value)
constexpr int kSortTile = 2048;   // keys per wavefront, walked 64 at a time: that keeps the pass stable
__global__ __launch_bounds__(64) void k_rx_hist(const u64* __restrict__ keys, u64 n, int shift, uint32_t* __restrict__ block_hist, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    const int l = threadIdx.x;
    for (int d = l; d < 256; d += 64) h[d] = 0u;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * kSortTile;
    for (int j = 0; j < kSortTile; j += 64) {
        const u64 i = lo + (u64)(j + l);
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = l; d < 256; d += 64) block_hist[(size_t)d * n_blocks + blockIdx.x] = h[d];
}
__global__ __launch_bounds__(1024) void k_rx_scan(uint32_t* __restrict__ a, u64 n) {   // exclusive, in place, one workgroup
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (u64 i0 = 0; i0 < n; i0 += 1024u) {
        const u64 i = i0 + threadIdx.x;
        const uint32_t v = i < n ? a[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane() >= d) incl += o; }
        if (lane() == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t before = carry + incl - v;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < n) a[i] = before;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void k_rx_scatter(const u64* __restrict__ keys, const uint32_t* __restrict__ vals, u64 n, int shift, const uint32_t* __restrict__ block_off, uint32_t n_blocks,
                                                   u64* __restrict__ out_k, uint32_t* __restrict__ out_v) {
    __shared__ uint32_t base[256];
    const int l = threadIdx.x;
    for (int d = l; d < 256; d += 64) base[d] = block_off[(size_t)d * n_blocks + blockIdx.x];
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * kSortTile;
    for (int j = 0; j < kSortTile; j += 64) {
        const u64 i = lo + (u64)(j + l);
        const bool have = i < n;
        const u64 k = have ? keys[i] : 0ull;
        const uint32_t d = (uint32_t)(k >> shift) & 255u;
        u64 peers = __ballot(have);
#pragma unroll
        for (int b = 0; b < 8; b++) { const u64 bb = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? bb : ~bb; }
        const uint32_t before = (uint32_t)__popcll(peers & ((1ull << l) - 1ull));
        uint32_t dst = 0;
        if (have) dst = base[d] + before;
        __syncthreads();
        if (have && before == 0u) base[d] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (have) { out_k[dst] = k; out_v[dst] = vals[i]; }
    }
}
__global__ __launch_bounds__(256) void k_bias_keys(const long long* __restrict__ key, u64 n, u64* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = (u64)key[i] ^ (1ull << 63);
}
__global__ __launch_bounds__(256) void k_iota32(uint32_t* __restrict__ a, u64 n) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_gather32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, u64 n, uint32_t* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}


// ---------------------------------------------------------------- T1 / T2: a read's own table, every key's first insertion
// X31 (src/khash.h:486-494) carried on: h -> h * 31 + c
__device__ __forceinline__ uint32_t x31_c(uint32_t h, uint32_t c) { return (h << 5) - h + c; }
__device__ inline uint32_t x31_dec(uint32_t h, long long v) {   // the digits of "%d"
    if (v < 0) { h = x31_c(h, (uint32_t)'-'); v = -v; }
    u64 u = (u64)v, p = 1;
    while (u / p >= 10ull) p *= 10ull;
    for (; p; p /= 10ull) h = x31_c(h, (uint32_t)'0' + (uint32_t)((u / p) % 10ull));
    return h;
}
// a key as one word: position in the concatenated genome 35 bits | strand | code 6 | ins_offset 16 | haplotype + 1 (0 = the `-1` of
// make_key: haplotypes off, or the aggregate) 6
__device__ __host__ inline u64 tie_key(u64 gpos, uint32_t strand, uint32_t code, uint32_t ins, uint32_t hp1) {
    return (gpos << 29) | ((u64)(strand & 1u) << 28) | ((u64)(code & 63u) << 22) | ((u64)(ins & 0xFFFFu) << 6) | (u64)(hp1 & 63u);
}
__device__ __forceinline__ u64 mix64(u64 x) { x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }

struct TieTables {
    const uint8_t* klass;       // [n_codes][256]: 0 = ambiguous (never reaches update_freq_map, src/mod.c:1180-1191)
    const uint32_t* ctg_hash;   // X31 of "<contig>\t" per tid
    const u64* ctg_base;        // the contig's first position in the concatenated genome
    const int32_t* ctg_rank;    // rank of the contig's name in strcmp order (equal names share one: cmp_key_fast cannot tell them apart)
    const uint2* mid;           // [2][64]: "\t<strand>\t<code>\t" as h -> h * x + y
    const char* codes;          // [64][MM_CODE_LEN]
    int32_t n_codes, n_contigs, insertions, haplotypes;
};
struct TieLaunch {
    const mm_read_t* reads; const uint8_t* mm; const mm_view_row_t* rows;
    uint32_t n_reads; uint32_t n_rows;
    const uint32_t* beg; const uint32_t* end;   // a read's rows [beg, end)
    u64 serial0;
    u64* sk_a; u64* sk_b;                       // n_rows each: the calls' sort words
    u64* keys; uint32_t* khash;                 // n_rows * per
    u64* tab;   // 4 * per * n_rows + 8 * n_reads words: the reads' own tables (ReadTab)
    uint32_t* rc;   // n_reads (zeroed): what k_tie_reads leaves k_tie_keys / k_tie_puts -- the read's calls that reach the table | their order stands in sk_b << 31
    uint32_t* rt;   // n_reads (zeroed): what k_tie_reads leaves k_tie_stamps -- a read's table size | which flag bit marks its buckets << 31
    u64* gkey; u64* gstamp; u64 gmask;
    u64* last_put; uint32_t* fail;
};
enum { TIE_F_CODE = 1, TIE_F_ROWS = 2, TIE_F_HP = 4, TIE_F_CONTIG = 8, TIE_F_SLOTS = 16, TIE_F_MISSING = 32, TIE_F_INTERNAL = 64 /* a launch was not taken: out of memory, a HIP failure */ };

// rows arrive ordered by read: [beg, end) of every read that has any (the arrays start zeroed)
__global__ __launch_bounds__(256) void k_tie_bounds(const mm_view_row_t* __restrict__ rows, uint32_t n, uint32_t n_reads, uint32_t* __restrict__ beg, uint32_t* __restrict__ end, uint32_t* __restrict__ fail) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rows[i].read & 0x1FFFFFu;
    if (r >= n_reads) { atomicOr(fail, (uint32_t)TIE_F_ROWS); return; }
    if (i == 0u || (rows[i - 1u].read & 0x1FFFFFu) != r) beg[r] = i;
    if (i + 1u == n || (rows[i + 1u].read & 0x1FFFFFu) != r) end[r] = i + 1u;
}

__device__ inline bool mm_multi_letter(const uint8_t* mm, uint32_t len) {   // any group with more than one code letter ("C+hm?")
    uint32_t p = 0;
    while (p + 2u < len) {
        uint32_t s = p + 2u, e = s;
        while (e < len && mm[e] != ',' && mm[e] != ';' && mm[e] != '?' && mm[e] != '.') e++;
        if (e - s > 1u && !(mm[s] >= '0' && mm[s] <= '9')) return true;
        while (p < len && mm[p] != ';') p++;
        p++;
    }
    return false;
}
// which letter of group number gord's header the code is (the code of letter m is the string from m on, src/mod.c:1151)
__device__ inline uint32_t mm_letter_index(const uint8_t* mm, uint32_t len, uint32_t gord, const char* code) {
    uint32_t p = 0, g = 0;
    while (p < len && g < gord) { while (p < len && mm[p] != ';') p++; p++; g++; }
    if (p + 2u >= len) return 0u;
    uint32_t s = p + 2u, e = s;
    while (e < len && mm[e] != ',' && mm[e] != ';' && mm[e] != '?' && mm[e] != '.') e++;
    if (e <= s || (mm[s] >= '0' && mm[s] <= '9')) return 0u;
    uint32_t cl = 0;
    while (cl < MM_CODE_LEN && code[cl]) cl++;
    for (uint32_t m = 0; s + m < e; m++) {
        if (e - s - m != cl) continue;
        bool same = true;
        for (uint32_t q = 0; q < cl; q++) if (mm[s + m + q] != (uint8_t)code[q]) { same = false; break; }
        if (same) return m;
    }
    return 0u;
}
__device__ inline void shell_sort(u64* a, uint32_t n) {
    const uint32_t gaps[12] = {40423u, 17961u, 7983u, 3548u, 1577u, 701u, 301u, 132u, 57u, 23u, 10u, 4u};
    for (int g = 0; g <= 12; g++) {
        const uint32_t gap = g < 12 ? gaps[g] : 1u;
        if (gap >= n) continue;
        for (uint32_t i = gap; i < n; i++) {
            const u64 v = a[i];
            uint32_t j = i;
            for (; j >= gap && a[j - gap] > v; j -= gap) a[j] = a[j - gap];
            a[j] = v;
        }
    }
}

// A read's own table (update_freq_map's khash, src/mod.c:883-929), ONE 64-bit word a bucket: the key's X31 hash (32) | its number among the read's
// keys << 32 (20 bits) | two flag bits -- `fo`: the bucket holds a key; the other: it holds a key the running resize has placed.  (The first form kept
// three arrays -- key numbers, "old" and "new" flag bytes -- and looked the hash and the 64-bit key up through the number at every probe: four
// dependent loads from global memory a probe in a kernel that is one serial walk a thread, bounded by its longest read: 19 ms a launch.)
struct ReadTab { u64* ent; uint32_t nb, size, upper; u64 fo; };
constexpr u64 kRtData = (1ull << 52) - 1ull, kRtA = 1ull << 63, kRtB = 1ull << 62;
__device__ inline void rtab_grow(ReadTab& t) {   // kh_resize to the next power of two, in place, with its kick-outs
    const uint32_t nb2 = t.nb ? t.nb * 2u : 4u;
    const uint32_t mask = nb2 - 1u;
    const u64 fo = t.fo, fn = fo ^ (kRtA | kRtB);       // (occupied before / placed by this resize: the roles swap at its end)
    for (uint32_t i = t.nb; i < nb2; i++) t.ent[i] = 0ull;
    for (uint32_t j = 0; j < t.nb; j++) {
        const u64 e = t.ent[j];
        if (!(e & fo)) continue;
        u64 key = e & kRtData;
        t.ent[j] = 0ull;
        for (;;) {
            uint32_t i = (uint32_t)key & mask, step = 0;
            u64 x;
            while ((x = t.ent[i]) & fn) i = (i + (++step)) & mask;
            t.ent[i] = fn | key;
            if (i < t.nb && (x & fo)) key = x & kRtData;       // (the bucket's old key is kicked out and placed next)
            else break;
        }
    }
    t.fo = fn;
    t.nb = nb2; t.upper = (uint32_t)(nb2 * 0.77 + 0.5);
}
// update_freq_map: kh_get, and kh_put only for a key that is not there (the put looks at the growth bound first)
__device__ inline bool rtab_put(ReadTab& t, uint32_t k, uint32_t h, const u64* keys) {
    const u64 mine = ((u64)k << 32) | (u64)h;
    if (t.nb) {
        const uint32_t mask = t.nb - 1u;
        uint32_t i = h & mask, step = 0;
        u64 e;
        while ((e = t.ent[i]) & t.fo) {
            if ((uint32_t)e == h && keys[(uint32_t)((e & kRtData) >> 32)] == keys[k]) return false;
            i = (i + (++step)) & mask;
        }
        if (t.size < t.upper) { t.ent[i] = t.fo | mine; t.size++; return true; }
    }
    rtab_grow(t);
    const uint32_t mask = t.nb - 1u;
    uint32_t i = h & mask, step = 0;
    while (t.ent[i] & t.fo) i = (i + (++step)) & mask;
    t.ent[i] = t.fo | mine; t.size++;
    return true;
}

__global__ __launch_bounds__(64) void k_tie_reads(TieTables T, TieLaunch L) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= L.n_reads) return;
    const uint32_t a = L.beg[r], b = L.end[r];
    if (b <= a) return;
    const uint32_t n = b - a;
    const mm_read_t rd = L.reads[r];
    const uint8_t* mm = L.mm + rd.mm_off;
    if (n >= (1u << 18)) { atomicOr(L.fail, (uint32_t)TIE_F_ROWS); return; }
    if (rd.tid < 0 || rd.tid >= T.n_contigs) { atomicOr(L.fail, (uint32_t)TIE_F_CONTIG); return; }
    const bool multi = mm_multi_letter(mm, rd.mm_len);
    // 1. the calls that reach the table, as sort words: group 11 | implicit 1 | position in the read as sequenced 28 | letter 6 | row 18
    u64* sa = L.sk_a + a;
    u64* sb = L.sk_b + a;
    uint32_t cnt = 0, gmax = 0;
    for (uint32_t i = 0; i < n; i++) {
        const mm_view_row_t w = L.rows[a + i];
        const uint32_t implicit = w.read_pos >> 31, gord = w.read >> 21, fq = w.read_pos & 0x7FFFFFFFu;
        if ((int)w.code >= T.n_codes) { atomicOr(L.fail, (uint32_t)TIE_F_CODE); continue; }
        if (!implicit && T.klass[(uint32_t)w.code * 256u + w.prob] == 0) continue;
        if (fq >= (1u << 28)) { atomicOr(L.fail, (uint32_t)TIE_F_ROWS); continue; }
        uint32_t m = 0;
        if (multi) { m = mm_letter_index(mm, rd.mm_len, gord, T.codes + (uint32_t)w.code * MM_CODE_LEN); if (m > 63u) m = 63u; }
        sa[cnt++] = ((u64)gord << 53) | ((u64)implicit << 52) | ((u64)fq << 24) | ((u64)m << 18) | (u64)i;
        if (gord > gmax) gmax = gord;
    }
    if (cnt == 0u) return;
    // 2. into the order freq_view_single met them (src/mod.c:1003-1370): group by group, listed calls before implicit ones, along the read
    //    as sequenced.  They arrive by reference position, the groups mixed: a stable distribution by (group, implicit), and a run that
    //    falls instead of rising (a reverse read) turned around; anything else sorted the general way
    u64* ord = sa;
    if (!multi && gmax < 32u && cnt >= 2u) {
        uint32_t at[65];
        for (int q = 0; q < 65; q++) at[q] = 0;
        for (uint32_t i = 0; i < cnt; i++) at[(uint32_t)(sa[i] >> 52) + 1u]++;
        for (int q = 0; q < 64; q++) at[q + 1] += at[q];
        {
            uint32_t cur[64];
            for (int q = 0; q < 64; q++) cur[q] = at[q];
            for (uint32_t i = 0; i < cnt; i++) sb[cur[(uint32_t)(sa[i] >> 52)]++] = sa[i];
        }
        for (int q = 0; q < 64; q++) {
            u64* p = sb + at[q];
            const uint32_t c = at[q + 1] - at[q];
            if (c < 2u) continue;
            bool up = true, down = true;
            for (uint32_t i = 1; i < c; i++) { if (p[i] <= p[i - 1]) up = false; if ((p[i] >> 18) >= (p[i - 1] >> 18)) down = false; }
            if (up) continue;
            if (down) { for (uint32_t i = 0, k = c - 1u; i < k; i++, k--) { const u64 x = p[i]; p[i] = p[k]; p[k] = x; } continue; }
            shell_sort(p, c);
        }
        ord = sb;
    } else shell_sort(sa, cnt);
    // 3. (k_tie_keys, k_tie_puts) keys and their X31 hashes in that order, through the read's own table
    if (T.haplotypes && rd.hp > 61) { atomicOr(L.fail, (uint32_t)TIE_F_HP); return; }
    L.rc[r] = cnt | (ord == sb ? 0x80000000u : 0u);
}
// 3a. the keys and their X31 hashes, in the order step 2 left the calls in: a THREAD A CALL (round 5's end: nothing here depends on the call in
// front -- as part of the read's thread it was a random load and a hundred and fifty instructions a key in the longest read's chain)
__global__ __launch_bounds__(256) void k_tie_keys(TieTables T, TieLaunch L) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= L.n_rows) return;
    const uint32_t r = L.rows[g].read & 0x1FFFFFu;
    if (r >= L.n_reads) return;
    const uint32_t c = L.rc[r], cnt = c & 0x7FFFFFFFu, a = L.beg[r], j = g - a;
    if (j >= cnt) return;
    const u64* ord = ((c >> 31) ? L.sk_b : L.sk_a) + a;
    const mm_read_t rd = L.reads[r];
    const mm_view_row_t w = L.rows[a + (uint32_t)(ord[j] & 0x3FFFFu)];
    const uint32_t per = T.haplotypes ? 2u : 1u;
    const uint32_t strand = (rd.flag & 0x10) ? 1u : 0u;
    const uint32_t hc = T.ctg_hash[rd.tid];
    const u64 gbase = T.ctg_base[rd.tid];
    const uint32_t ins = T.insertions ? w.ins_offset : 0u;
    const uint2 md = T.mid[strand * 64u + w.code];
    uint32_t h0 = x31_dec(hc, (long long)w.pos);
    h0 = h0 * md.x + md.y;
    h0 = x31_dec(h0, (long long)ins);
    h0 = x31_c(h0, (uint32_t)'\t');
    for (uint32_t v = 0; v < per; v++) {   // the key with the haplotype, then the aggregate (src/mod.c:883-929)
        const int hp = T.haplotypes ? (v == 0u ? (int)rd.hp : -1) : -1;
        const u64 at = ((u64)a + j) * per + v;
        L.keys[at] = tie_key(gbase + (u64)(uint32_t)w.pos, strand, w.code, ins, (uint32_t)(hp + 1));
        L.khash[at] = x31_dec(h0, (long long)hp);
    }
}
// 3b. update_freq_map's puts, one after the other: a thread a read -- the chain that is left
__global__ __launch_bounds__(64) void k_tie_puts(TieTables T, TieLaunch L) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= L.n_reads) return;
    const uint32_t cnt = L.rc[r] & 0x7FFFFFFFu;
    if (cnt == 0u) return;
    const uint32_t a = L.beg[r];
    const uint32_t per = T.haplotypes ? 2u : 1u;
    const u64* keys = L.keys + (u64)a * per;
    const uint32_t* kh = L.khash + (u64)a * per;
    ReadTab tab;
    tab.ent = L.tab + (4ull * per * a + 8ull * r); tab.nb = 0; tab.size = 0; tab.upper = 0; tab.fo = kRtA;
    const uint32_t n = cnt * per;
    for (uint32_t k = 0; k < n; k++) (void)rtab_put(tab, k, kh[k], keys);
    L.rt[r] = tab.nb | (tab.fo == kRtB ? 0x80000000u : 0u);
}
// 4. the table's slot order is the order merge_freq_maps offers the keys to the core table (src/mod.c:743-774): stamps, smallest kept per key.
// A WAVEFRONT a read (round 5's end: this walk over up to four buckets a key, with two atomics a key on the global table, was the longest read's
// last eight milliseconds as one thread's chain; the buckets are independent once the table stands): a lane a bucket, a bucket's rank among the
// occupied ones from a ballot.
__global__ __launch_bounds__(256) void k_tie_stamps(TieTables T, TieLaunch L) {
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= L.n_reads) return;
    const uint32_t w = L.rt[r], nb = w & 0x7FFFFFFFu;
    if (nb == 0u) return;
    const u64 fo = (w >> 31) ? kRtB : kRtA;
    const uint32_t a = L.beg[r];
    const uint32_t per = T.haplotypes ? 2u : 1u;
    const u64* keys = L.keys + (u64)a * per;
    const u64* ent = L.tab + (4ull * per * a + 8ull * r);
    const u64 sbase = (L.serial0 + (u64)r) << 24;
    const uint32_t l = (uint32_t)lane();
    uint32_t w2 = 0;
    for (uint32_t s0 = 0; s0 < nb; s0 += 64u) {
        const uint32_t sl = s0 + l;
        const u64 e = sl < nb ? ent[sl] : 0ull;
        const bool occ = (e & fo) != 0ull;
        const u64 m = __ballot(occ);
        if (occ) {
            const u64 key = keys[(uint32_t)((e & kRtData) >> 32)], stamp = sbase | (u64)(w2 + (uint32_t)__popcll(m & ((1ull << l) - 1ull)));
            u64 i = mix64(key) & L.gmask;
            for (;;) {
                u64 k = L.gkey[i];
                if (k == kNone64) { k = atomicCAS(&L.gkey[i], kNone64, key); if (k == kNone64) k = key; }
                if (k == key) { atomicMin(&L.gstamp[i], stamp); break; }
                i = (i + 1ull) & L.gmask;
            }
        }
        w2 += (uint32_t)__popcll(m);
    }
    if (l == 0u) {
        if (w2 >= (1u << 24)) atomicOr(L.fail, (uint32_t)TIE_F_SLOTS);
        if (w2) atomicMax(L.last_put, sbase | (u64)(w2 - 1u));
    }
}
// the stamp table into one of twice the size
__global__ __launch_bounds__(256) void k_stamp_rehash(const u64* __restrict__ ok, const u64* __restrict__ os, u64 ocap, u64* __restrict__ nk, u64* __restrict__ ns, u64 nmask) {
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j >= ocap) return;
    const u64 key = ok[j];
    if (key == kNone64) return;
    u64 i = mix64(key) & nmask;
    for (;;) {
        const u64 k = atomicCAS(&nk[i], kNone64, key);
        if (k == kNone64) { ns[i] = os[j]; break; }
        i = (i + 1ull) & nmask;
    }
}
// (a thread strides over the table and a workgroup adds ONE number: the first form added a wavefront's count at a time -- a million atomics
// on one address for a table of 2^26 slots, 5.9 ms a launch of the replay's second handle)
__global__ __launch_bounds__(256) void k_stamp_count(const u64* __restrict__ gk, u64 cap, u64* __restrict__ count) {
    __shared__ uint32_t part[4];
    uint32_t mine = 0;
    for (u64 j = (u64)blockIdx.x * 256u + threadIdx.x; j < cap; j += (u64)gridDim.x * 256u) mine += gk[j] != kNone64 ? 1u : 0u;
    for (int d = 32; d; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d);
    if (lane() == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t t = part[0] + part[1] + part[2] + part[3]; if (t) atomicAdd(count, (u64)t); }
}
// X31 of make_key's string for a key (src/mod.c:428-439)
__device__ inline uint32_t key_x31(const TieTables& T, int32_t tid, int32_t pos, uint32_t strand, uint32_t code, uint32_t ins, int hp) {
    const uint2 md = T.mid[(strand & 1u) * 64u + code];
    uint32_t h = x31_dec(T.ctg_hash[tid], (long long)pos);
    h = h * md.x + md.y;
    h = x31_dec(h, (long long)ins);
    h = x31_c(h, (uint32_t)'\t');
    return x31_dec(h, (long long)hp);
}
// every key the table holds, in the order of their first insertion: entries -> (flags, scan) -> compact -> (sort by stamp) -> decoded
__global__ __launch_bounds__(256) void k_stamp_flags(const u64* __restrict__ gk, u64 cap, u64* __restrict__ f) {
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j < cap) f[j] = gk[j] != kNone64 ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_stamp_gather(const u64* __restrict__ gk, const u64* __restrict__ gs, u64 cap, const u64* __restrict__ incl, u64* __restrict__ out_k, u64* __restrict__ out_s) {
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j >= cap) return;
    const u64 k = gk[j];
    if (k == kNone64) return;
    const u64 at = incl[j] - 1ull;
    out_k[at] = k; out_s[at] = gs[j];
}
__global__ __launch_bounds__(256) void k_key_decode(TieTables T, const u64* __restrict__ keys, const uint32_t* __restrict__ idx, u64 n, mm_row_t* __restrict__ out, uint32_t* __restrict__ hash) {
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const u64 key = keys[idx[j]];
    const u64 gpos = key >> 29;
    int lo = 0, hi = T.n_contigs - 1;   // the last contig whose first position is not behind gpos
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.ctg_base[mid] <= gpos) lo = mid; else hi = mid - 1; }
    mm_row_t r;
    r.tid = lo; r.pos = (int32_t)(gpos - T.ctg_base[lo]);
    r.strand = (uint8_t)((key >> 28) & 1ull); r.rsvd = 0;
    r.code = (int16_t)((key >> 22) & 63ull);
    r.ins_offset = (uint16_t)((key >> 6) & 0xFFFFull);
    r.hp = (int16_t)((int)(key & 63ull) - 1);
    r.n_called = 0; r.n_mod = 0;
    out[j] = r;
    hash[j] = key_x31(T, r.tid, r.pos, r.strand, (uint32_t)r.code, r.ins_offset, (int)r.hp);
}
// per output row (any order): its stamp, the reference's hash of its key, what the comparator looks at
__global__ __launch_bounds__(256) void k_tie_rows(TieTables T, const mm_row_t* __restrict__ rows, u64 n, const u64* __restrict__ gkey, const u64* __restrict__ gstamp, u64 gmask,
                                                  u64* __restrict__ stamp, uint32_t* __restrict__ hash, long long* __restrict__ sortkey, uint32_t* __restrict__ fail) {
    const u64 j = (u64)blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const mm_row_t w = rows[j];
    if (w.tid < 0 || w.tid >= T.n_contigs || w.code < 0 || w.code >= T.n_codes || w.hp > 61) { atomicOr(fail, (uint32_t)TIE_F_MISSING); stamp[j] = kNone64; hash[j] = 0; sortkey[j] = 0; return; }
    const u64 key = tie_key(T.ctg_base[w.tid] + (u64)(uint32_t)w.pos, w.strand, (uint32_t)w.code, w.ins_offset, (uint32_t)(w.hp + 1));
    u64 i = mix64(key) & gmask, st = kNone64;
    for (;;) {
        const u64 k = gkey[i];
        if (k == key) { st = gstamp[i]; break; }
        if (k == kNone64) break;
        i = (i + 1ull) & gmask;
    }
    if (st == kNone64) atomicOr(fail, (uint32_t)TIE_F_MISSING);
    stamp[j] = st;
    hash[j] = key_x31(T, w.tid, w.pos, w.strand, (uint32_t)w.code, w.ins_offset, (int)w.hp);
    sortkey[j] = ((long long)T.ctg_rank[w.tid] << 32) + (long long)w.pos;
}
__global__ __launch_bounds__(256) void k_gather64(const long long* __restrict__ src, const uint32_t* __restrict__ idx, u64 n, long long* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_gather_rows(const mm_row_t* __restrict__ src, const uint32_t* __restrict__ idx, u64 n, mm_row_t* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
__global__ __launch_bounds__(256) void k_or_diff(const u64* __restrict__ keys, u64 n, u64* __restrict__ out) {   // which bits differ anywhere (passes of the sort that can be left out)
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    u64 v = i < n ? (keys[i] ^ keys[0]) : 0ull;
#pragma unroll
    for (int d = 32; d; d >>= 1) v |= ((u64)(uint32_t)__shfl_down((int)(uint32_t)v, d, 64)) | ((u64)(uint32_t)__shfl_down((int)(uint32_t)(v >> 32), d, 64) << 32);
    if (lane() == 0 && v) atomicOr(out, v);
}

}  // namespace mmtie
