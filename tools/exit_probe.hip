// what a process's death costs by what it holds: tools/exit_probe <device GB> <pinned MB> <host GB touched> [free]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unistd.h>
static double now() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
    double dev_gb = argc > 1 ? atof(argv[1]) : 0, pin_mb = argc > 2 ? atof(argv[2]) : 0, host_gb = argc > 3 ? atof(argv[3]) : 0;
    int do_free = argc > 4;
    void* d[64]; int nd = 0;
    for (double left = dev_gb; left > 0 && nd < 64; left -= 0.5) {
        size_t n = (size_t)((left < 0.5 ? left : 0.5) * 1e9);
        if (hipMalloc(&d[nd], n) != hipSuccess) return 2;
        (void)hipMemset(d[nd], 1, n); nd++;
    }
    void* p = nullptr;
    if (pin_mb > 0) { if (hipHostMalloc(&p, (size_t)(pin_mb * 1e6)) != hipSuccess) return 3; memset(p, 1, (size_t)(pin_mb * 1e6)); }
    char* h = nullptr;
    if (host_gb > 0) { h = (char*)malloc((size_t)(host_gb * 1e9)); memset(h, 1, (size_t)(host_gb * 1e9)); }
    (void)hipDeviceSynchronize();
    double t0 = now();
    if (do_free) { for (int i = 0; i < nd; i++) (void)hipFree(d[i]); if (p) (void)hipHostFree(p); free(h); }
    printf("%.6f %.6f\n", now(), now() - t0);
    fflush(stdout);
    _exit(0);
}
