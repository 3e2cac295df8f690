// which XCD does workgroup b run on?  s_getreg XCC_ID per workgroup of a 2048-workgroup launch: tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
int main() {
    const int n = 2048;
    unsigned* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int hist[16] = {0}, same = 0;
    for (int i = 0; i < n; i++) { hist[h[i] & 15]++; if (i >= 8 && (h[i] & 15) == (h[i - 8] & 15)) same++; }
    printf("XCC_ID histogram:"); for (int i = 0; i < 16; i++) printf(" %d", hist[i]); printf("\nworkgroups b and b+8 on one XCD: %d of %d; first 16:", same, n - 8);
    for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
    return 0;
}
