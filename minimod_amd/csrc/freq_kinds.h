// freq_kinds.h -- the freq / view path is compiled THREE times, once per reference-word kind (four bits a position for one requested
// mod, 16 bits up to five, 32 bits beyond), each into a code object of its own: a process only ever uses one kind, and the HIP
// runtime loads a code object when the first of its kernels is asked for -- with all three kinds in one object a `minimod freq`
// run paid 0.12 s for instantiations it never launched (0.25 s against 0.13 s from start to exit on a small file).  A copy sees
// its API names and its kernels' namespace with its kind's suffix (this header, included first); freq_dispatch.cpp holds the public
// names of include/minimod_hip.h and forwards by the kind the handle was made with.
#pragma once
#ifdef MM_KIND
#define MM_K_CAT2(a, b) a##_k##b
#define MM_K_CAT(a, b) MM_K_CAT2(a, b)
#define MM_K(name) MM_K_CAT(name, MM_KIND)
#define mmhip MM_K(mmhip)
#define mm_freq MM_K(mm_freq)
#define mm_abi_version MM_K(mm_abi_version)
#define mm_strerror MM_K(mm_strerror)
#define mm_freq_create MM_K(mm_freq_create)
#define mm_freq_destroy MM_K(mm_freq_destroy)
#define mm_freq_submit MM_K(mm_freq_submit)
#define mm_freq_host_done MM_K(mm_freq_host_done)
#define mm_freq_read_record MM_K(mm_freq_read_record)
#define mm_freq_ticket_batch MM_K(mm_freq_ticket_batch)
#define mm_freq_submit_device MM_K(mm_freq_submit_device)
#define mm_freq_submit_device_now MM_K(mm_freq_submit_device_now)
#define mm_freq_ticket_batches MM_K(mm_freq_ticket_batches)
#define mm_freq_wait MM_K(mm_freq_wait)
#define mm_view_fetch MM_K(mm_view_fetch)
#define mm_view_fetch_device MM_K(mm_view_fetch_device)
#define mm_freq_plan_batch MM_K(mm_freq_plan_batch)
#define mm_freq_intern_code MM_K(mm_freq_intern_code)
#define mm_freq_n_codes MM_K(mm_freq_n_codes)
#define mm_freq_code_name MM_K(mm_freq_code_name)
#define mm_freq_finalize MM_K(mm_freq_finalize)
#define mm_freq_finalize_device MM_K(mm_freq_finalize_device)
#define mm_freq_slab_words MM_K(mm_freq_slab_words)
#define mm_freq_slab_export MM_K(mm_freq_slab_export)
#define mm_freq_slab_add MM_K(mm_freq_slab_add)
#define mm_freq_slab_clear MM_K(mm_freq_slab_clear)
#define mm_freq_slab_export_host MM_K(mm_freq_slab_export_host)
#define mm_freq_slab_add_host MM_K(mm_freq_slab_add_host)
#define mm_freq_slab_export_ipc MM_K(mm_freq_slab_export_ipc)
#define mm_freq_slab_add_ipc MM_K(mm_freq_slab_add_ipc)
#define mm_freq_last_kernel_ms MM_K(mm_freq_last_kernel_ms)
#define mm_freq_stats_enable MM_K(mm_freq_stats_enable)
#define mm_freq_stats_get MM_K(mm_freq_stats_get)
#define mm_freq_device_bytes MM_K(mm_freq_device_bytes)
#define mm_freq_launch_counts MM_K(mm_freq_launch_counts)
#define mm_freq_reset_counters MM_K(mm_freq_reset_counters)
#endif
