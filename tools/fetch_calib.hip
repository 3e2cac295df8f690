// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes of the freq kernels,
// against byte counts known by construction (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel reads a buffer far larger than the Infinity
// Cache exactly once.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/fetch_calib tools/fetch_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o calib -- tools/bin/fetch_calib   (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_read16(const uint4* __restrict__ p, size_t n, unsigned long long* out) {   // 16 B per lane, streaming
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x123456789ull) *out = acc;
}
__global__ void k_read4(const uint32_t* __restrict__ p, size_t n, unsigned long long* out) {   // 4 B per lane, streaming
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x123456789ull) *out = acc;
}
__global__ void k_read1(const uint8_t* __restrict__ p, size_t n, unsigned long long* out) {   // 1 B per lane, streaming
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x123456789ull) *out = acc;
}
// one 16-byte block per lane out of every `stride` blocks (the in-block select's gather): 64 different 128-byte lines per
// wave instruction when stride >= 8; bytes asked for = n_touched * 16
__global__ void k_gather16(const uint4* __restrict__ p, size_t n_blocks, uint32_t stride, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < n_blocks; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i * stride]; acc += v.x ^ v.w; }
    if (acc == 0x123456789ull) *out = acc;
}
// one 2-byte word per lane out of every `stride` words (the reference-word lookup)
__global__ void k_gather2(const uint16_t* __restrict__ p, size_t n_words, uint32_t stride, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < n_words; i += (size_t)gridDim.x * blockDim.x) acc += p[i * stride];
    if (acc == 0x123456789ull) *out = acc;
}
// one 64-bit atomic add per lane out of every `stride` words (the counter update)
__global__ void k_atomic8(unsigned long long* __restrict__ p, size_t n_words, uint32_t stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < n_words; i += (size_t)gridDim.x * blockDim.x) atomicAdd(p + i * stride, 0x100000001ull);
}
__global__ void k_write4(uint32_t* __restrict__ p, size_t n) {   // 4 B per lane, streaming stores
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

int main() {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: eight times the Infinity Cache
    void* buf = nullptr; unsigned long long* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const dim3 g(256 * 8), b(256);
    printf("kernel,bytes_asked_for,lines_touched_128B\n");
    hipLaunchKernelGGL(k_read16, g, b, 0, 0, (const uint4*)buf, bytes / 16, out); printf("k_read16,%zu,%zu\n", bytes, bytes / 128);
    hipLaunchKernelGGL(k_read4, g, b, 0, 0, (const uint32_t*)buf, bytes / 4, out); printf("k_read4,%zu,%zu\n", bytes, bytes / 128);
    hipLaunchKernelGGL(k_read1, g, b, 0, 0, (const uint8_t*)buf, bytes / 4, out); printf("k_read1,%zu,%zu\n", bytes / 4, bytes / 4 / 128);
    hipLaunchKernelGGL(k_gather16, g, b, 0, 0, (const uint4*)buf, bytes / 16, 8u, out); printf("k_gather16,%zu,%zu\n", bytes / 16 / 8 * 16, bytes / 128);
    hipLaunchKernelGGL(k_gather2, g, b, 0, 0, (const uint16_t*)buf, bytes / 2, 64u, out); printf("k_gather2,%zu,%zu\n", bytes / 2 / 64 * 2, bytes / 128);
    hipLaunchKernelGGL(k_atomic8, g, b, 0, 0, (unsigned long long*)buf, bytes / 8, 16u); printf("k_atomic8,%zu,%zu\n", bytes / 8 / 16 * 8, bytes / 128);
    hipLaunchKernelGGL(k_write4, g, b, 0, 0, (uint32_t*)buf, bytes / 4); printf("k_write4,%zu,%zu\n", bytes, bytes / 128);
    hipDeviceSynchronize();
    hipFree(buf); hipFree(out);
    return 0;
}
