// Device and pinned memory for every unit of the library: blocks are kept, never handed back to the driver while the process lives.
//
// Why (round 6, DESIGN section 4.0 "the allocator", profiles/r6_site_index_root_cause.txt): a virtual address range that hipFree gave back and a later
// hipMalloc handed out again was read -- and written -- through its OLD translation by the workgroups of one or two XCDs when a dozen short-lived
// processes shared the GPU: mm_freq_create freed the raw reference, the site index's block counts got its address, and one run in a hundred summed
// ASCII bases instead of counts (a wrong '-'/'+' strand index: memory faults, "n_called overflowed", wrong rows with exit 0).  A block that is reused
// keeps its pages, so a translation an XCD still holds stays TRUE.  dfree / hfree keep hipFree's meaning otherwise (they wait for the device first).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mmdev {
hipError_t dmalloc(void** p, size_t bytes);                       // instead of hipMalloc
hipError_t dfree(void* p);                                        // instead of hipFree: waits for the device, keeps the block
hipError_t hmalloc(void** p, size_t bytes, unsigned int flags = 0u);// instead of hipHostMalloc
hipError_t hfree(void* p);                                        // instead of hipHostFree
template <typename T> inline hipError_t dmalloc(T** p, size_t bytes) { return dmalloc(reinterpret_cast<void**>(p), bytes); }
template <typename T> inline hipError_t hmalloc(T** p, size_t bytes, unsigned int flags = 0u) { return hmalloc(reinterpret_cast<void**>(p), bytes, flags); }
}  // namespace mmdev

extern "C" {
// [0] device bytes held by callers, [1] device bytes kept for reuse, [2] pinned bytes held, [3] pinned bytes kept, [4] requests served from kept blocks,
// [5] requests that went to the driver, [6] blocks given back to the driver (only when the driver had no memory left, or by mm_devmem_trim)
void mm_devmem_stats(int64_t out[7]);
// gives every kept block back to the driver -- for a long-lived process at a moment when none of its kernels is queued or running; returns the bytes
int64_t mm_devmem_trim(void);
}
