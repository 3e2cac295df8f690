// what a HIP stream costs: made, used once, and at the process's death.  tools/exit_probe_streams <n streams> <0 plain | 1 with priorities spread> [destroy]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void k_nothing(int* p) { if (p) *p = 1; }
static double rss_gb() { long a = 0, b = 0; FILE* f = fopen("/proc/self/statm", "r"); if (f) { if (fscanf(f, "%ld %ld", &a, &b) != 2) b = 0; fclose(f); } return b * 4096.0 / 1e9; }
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 0, prio = argc > 2 ? atoi(argv[2]) : 0, destroy = argc > 3;
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, 0, nullptr);
    (void)hipDeviceSynchronize();
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const double r0 = rss_gb(), t0 = now();
    hipStream_t s[64];
    for (int i = 0; i < n; i++) {
        if (prio) (void)hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i % 3 == 0 ? lo : i % 3 == 1 ? hi : (lo + hi) / 2);
        else (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    }
    const double t1 = now();
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s[i], nullptr);
    (void)hipDeviceSynchronize();
    const double t2 = now(), r1 = rss_gb();
    if (destroy) for (int i = 0; i < n; i++) (void)hipStreamDestroy(s[i]);
    printf("%.6f %.4f %.4f %.3f %.3f %.4f\n", now(), t1 - t0, t2 - t1, r0, r1, now() - t2);
    fflush(stdout);
    _exit(0);
}
