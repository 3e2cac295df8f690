/*
 * minimod_summary.h -- C ABI of `minimod summary`'s census on the device.
 *
 * What it stands for (paths under /root/reference): summary_single src/mod.c:1426-1555 (the walk over a read's MM groups: base, strand,
 * codes, status flag, whether any skip count is listed), make_key_summary / add_summary_entry src/mod.c:1402-1424 (the read's own khash
 * of "<base>|<codes>|<flag>" strings) and the key loop of print_summary_output src/mod.c:1389-1394 (keys in the table's slot order, a
 * blank behind each).  The read name and the tab in front of the column are the caller's (it holds the names).
 *
 * Plain C.  mm_summary_batch returns 0 or the first failing read's MM_E_* (include/minimod_hip.h) with its index in *bad_read: the
 * reference exits on a malformed MM tag (src/mod.c:1467-1530), the caller prints its message.
 */
#ifndef MINIMOD_SUMMARY_H
#define MINIMOD_SUMMARY_H

#include <stddef.h>
#include <stdint.h>

#include "minimod_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_summary mm_summary_t;

mm_summary_t *mm_summary_create(int32_t device, char *err, size_t err_len);
/* One batch from HOST memory (reads[].mm_off / mm_len and the mm pool are looked at).  *text + (*off)[i], (*len)[i] bytes: read i's
 * modifications column.  The arrays belong to the handle and stay valid until its next call. */
int32_t mm_summary_batch(mm_summary_t *s, const mm_batch_t *host_batch, const char **text, const uint64_t **off, const uint32_t **len, int32_t *bad_read);
void mm_summary_destroy(mm_summary_t *s);

#ifdef __cplusplus
}
#endif
#endif /* MINIMOD_SUMMARY_H */
