// diagnostic build of the freq unit (hipcc ... -include tools/crumbs.h): every kernel launch leaves its name on stderr and is waited for, so that the
// last name in front of a "Memory access fault" is the kernel that made it.  MM_CRUMBS=1 at run time switches it on.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
static inline bool mm_crumbs_on() { static const bool on = std::getenv("MM_CRUMBS") != nullptr; return on; }
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                          \
    do {                                                                                                                           \
        if (mm_crumbs_on()) { std::fprintf(stderr, "[crumb] %s\n", #kernelName); std::fflush(stderr); }                            \
        hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__);               \
        if (mm_crumbs_on()) { hipError_t e_ = hipDeviceSynchronize(); if (e_ != hipSuccess) { std::fprintf(stderr, "[crumb] %s failed: %s\n", #kernelName, hipGetErrorString(e_)); std::fflush(stderr); } } \
    } while (0)

// MM_POISON=<byte>: every device allocation of the unit is filled with that byte first -- memory that is read before it is written stops looking like
// the zeroes a quiet machine hands out (hipMalloc promises nothing; with processes starting and dying all the time it gives back other runs' bytes)
static inline hipError_t mm_poison_malloc(void** p, size_t n) {
    const hipError_t e = hipMalloc(p, n);
    static const char* pz = std::getenv("MM_POISON");
    if (e == hipSuccess && pz && n) { (void)hipMemset(*p, std::atoi(pz), n); (void)hipDeviceSynchronize(); }
    return e;
}
#define hipMalloc(p, n) mm_poison_malloc((void**)(p), (n))
