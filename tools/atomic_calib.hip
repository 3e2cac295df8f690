// atomic_calib.hip -- what the counter update of the freq hot path costs on its own: the rate of scattered 64-bit
// `global_atomic_add_x2` (no return value) on an MI355X, at the address distributions the kernels produce.
//
//   random      every lane anywhere in the plane (the worst case: one 32-byte sector per atomic, no reuse)
//   c2          C2's shape: a wavefront works on one read; a round's 64 updates are the CpG sites of ~6 kb of one strand plane
//               (one site every ~100 bases on average, i.e. ~8 bytes x 100 apart: every update its own 128-byte line), the
//               next round continues behind it; reads start anywhere in a 50 Mb plane
//   c2_deep     the same with the reads confined to 1 Mb (1000x depth: the same lines hit again and again)
//   side        C5's side table: a CAS on a random 16-byte slot followed by an add on the slot's second word
//   pairs / quads / octs   lanes 2k and 2k+1 (4k..4k+3, 8k..8k+7) add to ADJACENT 8-byte words of one random 64-byte line: do the
//               lanes of one instruction that fall into one line leave as ONE request?  (two codes counted at one site, if
//               their counters lie side by side)
//   pair_2instr the same two adjacent words from ONE lane in two instructions back to back
//
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/atomic_calib tools/atomic_calib.hip ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// mode 0 random, 1 read-shaped (region = positions the reads start in), 2 side table
__global__ void k_atomics(unsigned long long* plane, uint64_t plane_len, uint64_t region, int rounds, int mode, uint64_t seed) {
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t strand = wave & 1;
    uint64_t start = mix(seed + wave) % region;
    for (int r = 0; r < rounds; r++) {
        uint64_t idx;
        if (mode == 0) idx = mix(seed ^ (wave * 64 + lane) * 0x9E3779B97F4A7C15ull + r) % (2 * plane_len);
        else if (mode == 1) {
            // a site every ~100 positions with a jitter, 64 sites a round
            const uint64_t p = start + (uint64_t)r * 6400 + (uint64_t)lane * 100 + (mix(wave * 977 + r * 64 + lane) & 63);
            idx = strand * plane_len + (p < plane_len ? p : p % plane_len);
        } else if (mode >= 3 && mode <= 5) {
            const int g = mode == 3 ? 2 : (mode == 4 ? 4 : 8);
            idx = (mix(seed ^ (wave * 64 + lane / g) * 0x9E3779B97F4A7C15ull + r) % (2 * plane_len / 8)) * 8 + (lane % g);
        } else if (mode == 6) {
            idx = (mix(seed ^ (wave * 64 + lane) * 0x9E3779B97F4A7C15ull + r) % (2 * plane_len / 8)) * 8;
            atomicAdd(plane + idx, 1ull);
            atomicAdd(plane + idx + 1, 0x100000001ull);
            continue;
        } else {
            idx = (mix(seed ^ (wave * 64 + lane) * 0x9E3779B97F4A7C15ull + r) % plane_len) * 2;   // a 16-byte slot
            unsigned long long key = idx | 1ull;
            atomicCAS(plane + idx, 0ull, key);
            atomicAdd(plane + idx + 1, 0x100000001ull);
            continue;
        }
        atomicAdd(plane + idx, (r & 1) ? 0x100000001ull : 1ull);
    }
}

int main() {
    const uint64_t plane_len = 50331648;   // 48 Mi positions, two strands: 0.8 GB like C2's counters
    unsigned long long* d = nullptr;
    CHK(hipMalloc(&d, 2 * plane_len * sizeof(unsigned long long)));
    CHK(hipMemset(d, 0, 2 * plane_len * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 6, rounds = 40;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct Case { const char* name; int mode; uint64_t region; } cases[] = {
        {"random", 0, plane_len}, {"c2", 1, plane_len - 300000}, {"c2_deep", 1, 1u << 20}, {"side_cas_plus_add", 2, plane_len},
        {"pairs", 3, plane_len}, {"quads", 4, plane_len}, {"octs", 5, plane_len}, {"pair_2instr", 6, plane_len}};
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"waves\": %d, \"rounds\": %d", prop.name, prop.multiProcessorCount, blocks * 4, rounds);
    for (const Case& c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_atomics, dim3(blocks), dim3(256), 0, 0, d, plane_len, c.region, rounds, c.mode, (uint64_t)(1234 + rep));
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double n = (double)blocks * 256 * rounds * (c.mode == 6 ? 2 : 1);
        std::printf(", \"%s\": {\"updates\": %.0f, \"ms\": %.4f, \"G_updates_per_s\": %.2f}", c.name, n, best, n / (best * 1e-3) / 1e9);
    }
    std::printf("}\n");
    return 0;
}
