// What a wave64 vector instruction costs a SIMD on gfx950, by kind: W wavefronts a SIMD each run a stream of N independent instructions of one kind
// (eight accumulators, unrolled); cycles of the slowest wave / instructions = cycles an instruction when the SIMD is shared by W waves.
//   valu_rate        -> a table: kind x wavefronts a SIMD (1, 2, 4, 7)
// k_stream_reads issues one vector instruction a SIMD every 4.0 cycles at seven wavefronts: is that the unit's limit for integer work?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int KIND> __device__ __forceinline__ void step(uint32_t (&a)[8], uint32_t k) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 3) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
        if (KIND == 4) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[i]));
        if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(k));
        if (KIND == 6) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 9) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s)); }
        if (KIND == 10) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a[i]) : "s"(k));
        if (KIND == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 12) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 13) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 15) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x6c" : "+v"(a[i]) : "v"(k));
        if (KIND == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(k) : "vcc");
        if (KIND == 17) asm volatile("s_add_u32 %0, %0, 1" : "+s"(k) :: "scc");
        if (KIND == 18) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(k) : "s20", "s21");
        if (KIND == 19) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[(i + 4) & 7]), "v"(k) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(k)); }   // (two instructions a step)
        if (KIND == 20) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 21) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 22) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 23) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k));
        if (KIND == 24) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(k));   // ONE accumulator: a dependent chain
        if (KIND == 25) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[0]));          // a dependent chain of a two-cycle instruction
        if (KIND == 26) { uint32_t s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s)); }
        if (KIND == 27) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(k));
    }
}
template <int KIND> __global__ __launch_bounds__(64) void k(uint32_t* out, unsigned long long* cyc, int iters) {
    uint32_t a[8]; for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
    uint32_t kk = out[0];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) { step<KIND>(a, kk); step<KIND>(a, kk); step<KIND>(a, kk); step<KIND>(a, kk); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = kk; for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345678u) out[1] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> double run(int waves_per_simd, uint32_t* out, unsigned long long* cyc, int n_cu) {
    const int iters = 400, blocks = n_cu * 4 * waves_per_simd;   // one wave a workgroup: the dispatcher spreads them over the SIMDs
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    std::vector<unsigned long long> h(blocks);
    if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed"); return -1.0; }
    (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    return sum / blocks / (iters * 32.0 * (KIND == 19 ? 2.0 : 1.0)) / waves_per_simd;   // cycles of a wave per instruction, divided by the waves sharing the SIMD = cycles of the SIMD per instruction
}
int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    uint32_t* out; unsigned long long* cyc; (void)hipMalloc(&out, 64); (void)hipMemset(out, 0, 64); (void)hipMalloc(&cyc, 8 * 65536);
    const char* names[] = {"v_add_u32", "v_and_b32", "v_xor_b32", "v_lshrrev_b32", "v_bfe_u32", "v_cndmask_b32", "v_bcnt_u32_b32", "v_add_f32", "v_fma_f32", "v_readlane_b32", "v_writelane_b32", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_u32_dpp", "v_lshl_add_u32", "v_bitop3_b32", "v_cmp_lt_u32", "s_add_u32", "v_cndmask_e64 sgpr", "v_cmp + v_cndmask /2", "v_bfi_b32", "v_and_or_b32", "v_min_u32", "v_perm_b32", "v_add_u32 dependent", "v_bfe_u32 dependent", "v_readfirstlane", "v_mov_b32"};
    printf("cycles of a SIMD per wave64 instruction (s_memrealtime-free: __builtin_readcyclecounter = shader clock), %d CUs\n%-18s %8s %8s %8s %8s\n", pr.multiProcessorCount, "instruction", "1 wave", "2 waves", "4 waves", "7 waves");
#define ROW(K) { printf("%-18s", names[K]); fflush(stdout); for (int w : {1, 2, 4, 7}) { printf(" %8.2f", run<K>(w, out, cyc, pr.multiProcessorCount)); fflush(stdout); } printf("\n"); fflush(stdout); }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27)
    return 0;
}
