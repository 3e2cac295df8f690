// where a HIP process's start goes: tools/init_probe  (prints the steps' seconds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k_nothing(int* p) { if (p) *p = 1; }
int main() {
    double t[8]; int i = 0;
    t[i++] = now();
    (void)hipInit(0); t[i++] = now();
    (void)hipSetDevice(0); t[i++] = now();
    (void)hipFree(nullptr); t[i++] = now();
    void* d = nullptr; (void)hipMalloc(&d, 1 << 20); t[i++] = now();
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, 0, nullptr); (void)hipDeviceSynchronize(); t[i++] = now();
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t[i++] = now();
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, nullptr); (void)hipStreamSynchronize(s); t[i++] = now();
    printf("%.6f hipInit %.3f  hipSetDevice %.3f  hipFree(0) %.3f  hipMalloc %.3f  first launch on the null stream %.3f  a stream made %.3f  its first launch %.3f\n", now(), t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
    fflush(stdout);
    _exit(0);
}
