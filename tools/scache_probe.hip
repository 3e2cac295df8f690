// Does a wave-uniform `const __restrict__` load (s_load: the scalar data cache) ever see what an EARLIER kernel of the same stream left at an address a
// LATER kernel rewrote -- with many short-lived processes sharing the GPU?  The site-index sequence of mm_freq_create without the product around it:
// W (64 workgroups, one vector store each) -> S (one workgroup rewrites the words in place) -> R (64 workgroups read "their" word) -> 4-byte copy to the host.
//   scache_probe <mode> <iterations>     mode 0: R reads with s_load; 1: agent-scope (vector) load; 2: s_load behind s_dcache_inv
// Prints one line: iterations, mismatches, and the first few (iteration, word, seen, expected).  Run 12-16 copies at once: tools/scache_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
constexpr int N = 64;
__global__ void W(uint32_t* __restrict__ t, uint32_t k) { if (threadIdx.x == 0) t[blockIdx.x] = k * 1000u + blockIdx.x; }
__global__ void S(uint32_t* __restrict__ t, uint32_t k) { if (threadIdx.x <= N) t[threadIdx.x] = t[threadIdx.x] * 3u + k; }   // (in place, like k_radix_scan)
template <int MODE> __global__ void R(const uint32_t* __restrict__ t, uint32_t k, uint32_t* __restrict__ bad, uint32_t* __restrict__ log) {
    if (MODE == 2) __builtin_amdgcn_s_dcache_inv();
    const uint32_t v = MODE == 1 ? __hip_atomic_load(&t[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t[blockIdx.x];
    const uint32_t want = (k * 1000u + blockIdx.x) * 3u + k;
    if (threadIdx.x == 0 && v != want) { const uint32_t i = atomicAdd(bad, 1u); if (i < 8) { log[4 * i] = k; log[4 * i + 1] = blockIdx.x; log[4 * i + 2] = v; log[4 * i + 3] = want; } }
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 200;
    hipStream_t s, extra[6];
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
    for (int i = 0; i < 6; i++) (void)hipStreamCreateWithPriority(&extra[i], hipStreamNonBlocking, -(i % 3));   // (the CLI holds about ten hardware queues)
    uint32_t *t, *bad, *log; char* big;
    (void)hipMalloc(&big, 64 << 20); (void)hipMemset(big, 1, 64 << 20); (void)hipFree(big);
    (void)hipMalloc(&t, 4 * (N + 1)); (void)hipMalloc(&bad, 4); (void)hipMalloc(&log, 4 * 32); (void)hipMemset(bad, 0, 4);
    uint32_t total = 0, wrong_total = 0;
    for (uint32_t k = 1; k <= (uint32_t)iters; k++) {
        (void)hipMemsetAsync(t + N, 0, 4, s);
        hipLaunchKernelGGL(W, dim3(N), dim3(256), 0, s, t, k);
        hipLaunchKernelGGL(S, dim3(1), dim3(1024), 0, s, t, k);
        if (mode == 0) hipLaunchKernelGGL(R<0>, dim3(N), dim3(256), 0, s, t, k, bad, log);
        else if (mode == 1) hipLaunchKernelGGL(R<1>, dim3(N), dim3(256), 0, s, t, k, bad, log);
        else hipLaunchKernelGGL(R<2>, dim3(N), dim3(256), 0, s, t, k, bad, log);
        (void)hipMemcpyAsync(&total, t + N, 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) { printf("mode %d: sync failed at %u\n", mode, k); return 3; }
        if (total != k) wrong_total++;
    }
    uint32_t hb = 0, hl[32] = {0};
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hl, log, sizeof hl, hipMemcpyDeviceToHost);
    printf("mode %d: %d iterations, %u stale words, %u wrong totals", mode, iters, hb, wrong_total);
    for (uint32_t i = 0; i < hb && i < 8; i++) printf(" [k=%u word %u: saw %u, want %u]", hl[4 * i], hl[4 * i + 1], hl[4 * i + 2], hl[4 * i + 3]);
    printf("\n");
    return hb || wrong_total ? 1 : 0;
}
