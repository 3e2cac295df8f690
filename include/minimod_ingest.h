/* minimod_ingest.h -- C ABI of the device-side BAM ingestion (libminimod_hip.so; kernels in minimod_amd/csrc/ingest_kernels.hip.h).
 *
 * What it replaces: the reference decodes its BAM on the host -- sam_read1 behind hts_set_threads (src/minimod.c:73-90, :250) -- and
 * load_db (src/minimod.c:235-333) filters the records and pulls MM / ML / HP out of their tags (get_mm_tag_ptr / get_ml_tag /
 * get_hp_tag, src/mod.c:123-202), one record after the other.  Here the host only finds the BGZF blocks of the file and moves their
 * COMPRESSED bytes; the device inflates them (include/minimod_bgzf.h's kernels), finds the records in the decoded stream, applies
 * load_db's filters and writes the flattened batch of include/minimod_hip.h (mm_read_t + four pools) in place, where
 * mm_freq_submit_device picks it up.  The decoded bytes never cross PCIe.
 *
 * Two kinds of buffers:
 *   GROUP SLOTS  a group = up to max_blocks consecutive BGZF blocks, the unit of an inflate launch.  The caller writes the blocks'
 *                bytes into mm_ingest_staging(slot), describes them in mm_ingest_blocks(slot) and calls mm_ingest_inflate (any
 *                thread; asynchronous).  Groups are numbered in the order of their mm_ingest_inflate calls, which must be the order
 *                of the file.
 *   ARENAS       a batch under construction: mm_ingest_flatten(slot, arena, ...) frames the group's records and APPENDS the accepted
 *                ones to the arena's batch (asynchronous, on the handle's chain stream; groups in order).  mm_ingest_result blocks
 *                until that is done and says what the batch holds now; the caller decides when a batch is big enough, takes it
 *                with mm_ingest_arena_batch (device pointers: for mm_freq_submit_device on mm_ingest_stream()) and starts the
 *                next one in another arena.  An arena may be filled again once its batch's ticket has been waited for.
 * A record that ends in the next group (the TAIL) is carried on the device; -K / -B do not cut device batches (the reference's
 * output does not depend on them, SURVEY section 8c "Invariance").
 */
#ifndef MINIMOD_INGEST_H
#define MINIMOD_INGEST_H
#include <stddef.h>
#include <stdint.h>

#include "minimod_bgzf.h"
#include "minimod_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_ingest mm_ingest_t;

typedef struct mm_ingest_opts {
    int32_t device;
    int32_t n_targets;            /* the BAM header's n_ref: reference ids outside [-1, n_targets) mark a false record start */
    int32_t allow_secondary;      /* load_db's filters (src/minimod.c:260-275) */
    int32_t skip_supplementary;
    int32_t group_slots;          /* 0 = 4 */
    int32_t max_blocks;           /* BGZF blocks per group; 0 = 2048 */
    int32_t arenas;               /* 0 = 3 */
    int32_t names;                /* not 0: the read names are kept too (mm_ingest_arena_names; view prints them, src/mod.c:560-626) */
    uint64_t max_cbytes;          /* staging bytes per group; 0 = 48 MiB */
    uint64_t arena_bytes;         /* decoded bytes a batch may be made of (sizes the pools); 0 = 1 GiB */
    uint64_t head_room;           /* longest record tail that can be carried; 0 = 32 MiB */
    /* a worker of a sharded run (csrc/host/loader.c mmh_loader_open_share): only alignments that START inside the share */
    int32_t ranged, first, last;
    int32_t lo_tid, hi_tid;
    int32_t rsvd2;
    int64_t lo_pos, hi_pos;
} mm_ingest_opts_t;

/* what a group left (mm_ingest_result) */
typedef struct mm_ingest_result {
    int32_t err;                  /* 0, or MM_INGEST_E_*: the file cannot be read this way (the caller fails, or reads it with the host loader) */
    int32_t n_bad_blocks;         /* blocks the device inflater refused: nothing of the group was framed.  The caller decodes them
                                   * (status[i] != 0), hands the bytes over with mm_ingest_patch_block and calls mm_ingest_flatten again */
    const int32_t *status;        /* per block of the group */
    uint32_t n_records;           /* records framed in this group */
    uint32_t n_accepted;          /* ... of which passed load_db's filters: appended to the arena's batch */
    uint32_t n_slow_blocks;       /* diagnostics: blocks whose speculative entry was wrong */
    int32_t done;                 /* a share's end has been passed: nothing further of the file is wanted */
    uint64_t total_reads, total_bytes, processed_bytes;   /* this group's part of the loader's totals (core_t counters, src/minimod.h:190-194) */
    uint64_t tail_len;            /* bytes of an unfinished record behind the group (not 0 behind the file's last group: the file is cut off) */
    int64_t err_record;           /* with MM_INGEST_E_RECORD: the record, counted from the first one framed */
    /* the arena's batch INCLUDING this group */
    uint64_t batch_reads, batch_bases;
    uint64_t cigar_bytes, seq_bytes, mm_bytes, ml_bytes;
    uint32_t max_n_cigar, max_l_qseq;
    uint64_t qname_bytes;         /* opts.names: bytes of the batch's names, NULs included */
} mm_ingest_result_t;

enum { MM_INGEST_OK = 0, MM_INGEST_E_RECORD = 1, MM_INGEST_E_ARENA = 2, MM_INGEST_E_TAIL = 3, MM_INGEST_E_RECORDS = 4, MM_INGEST_E_HEADER = 5, MM_INGEST_E_CODES = 6,
       MM_INGEST_E_ARG = 16, MM_INGEST_E_HIP = 17, MM_INGEST_E_ORDER = 18 };

mm_ingest_t *mm_ingest_create(const mm_ingest_opts_t *opts, char *err, size_t err_len);
void mm_ingest_destroy(mm_ingest_t *h);
int32_t mm_ingest_group_slots(const mm_ingest_t *h);
int32_t mm_ingest_max_blocks(const mm_ingest_t *h);
uint64_t mm_ingest_max_cbytes(const mm_ingest_t *h);
uint64_t mm_ingest_arena_bytes(const mm_ingest_t *h);
uint8_t *mm_ingest_staging(mm_ingest_t *h, int32_t slot);          /* pinned, max_cbytes + 64 */
mm_bgzf_block_t *mm_ingest_blocks(mm_ingest_t *h, int32_t slot);   /* pinned, max_blocks records: c_off into the staging, o_off = decoded bytes of the group's blocks in front */
/* compressed bytes to the device, inflate, CRC32: asynchronous.  obytes = the sum of the blocks' ISIZE.  0 or -MM_INGEST_E_* */
int32_t mm_ingest_inflate(mm_ingest_t *h, int32_t slot, int32_t n_blocks, size_t cbytes, size_t obytes);
/* frame + flatten, appended to `arena`'s batch (new_arena != 0: the batch starts with this group).  first_skip: decoded bytes in front
 * of the first record, for the first group only (the BAM header's length; for a reader that starts at a .bai's virtual offset, the
 * offset inside its first block).  Asynchronous; groups in order (the group flattened last may be flattened again: patched blocks, or
 * MM_INGEST_E_ARENA answered with a fresh arena). */
int32_t mm_ingest_flatten(mm_ingest_t *h, int32_t slot, int32_t arena, int32_t new_arena, uint64_t first_skip);
/* blocks until the slot's flatten is done.  The slot's staging may be filled again after it (unless blocks are to be patched). */
int32_t mm_ingest_result(mm_ingest_t *h, int32_t slot, mm_ingest_result_t *out);
/* per record of the slot's group, in file order: l_data | counted << 30 | accepted << 31 (the caller's -K / -B accounting: which read
 * of which batch the reference would name in an error message).  Synchronous copy of n <= n_records words. */
int32_t mm_ingest_group_info(mm_ingest_t *h, int32_t slot, uint32_t *dst, uint32_t n);
/* decoded bytes of a block the device refused (the host's decoder has judged it) */
int32_t mm_ingest_patch_block(mm_ingest_t *h, int32_t slot, int32_t block, const uint8_t *decoded, size_t n);
/* the arena's batch as it stands after the group `res` describes: device pointers, for mm_freq_submit_device(.., mm_ingest_stream(h)) */
int32_t mm_ingest_arena_batch(mm_ingest_t *h, int32_t arena, const mm_ingest_result_t *res, mm_batch_t *out);
/* opts.names: the batch's read names in DEVICE memory -- read i's name (NUL-terminated, bam1_t's qname) at names + name_off[i]; the
 * batch's names take result.qname_bytes bytes.  (What print_view_output's first column is printed from, src/mod.c:560-626.) */
int32_t mm_ingest_arena_names(mm_ingest_t *h, int32_t arena, const uint8_t **names_dev, const uint64_t **name_off_dev);
/* A wildcard run (-c '*') counts whatever code a read's MM tag names (src/mod.c's req_all): the codes of a batch in DEVICE memory, in
 * the order a walk over its reads and their MM text meets them first -- a group of digits is one code (a ChEBI number), a group of letters one
 * code per letter: the string from that letter on, which is what the reference looks up (mod.c:1146-1160).  Writes up to max_codes
 * strings of MM_CODE_LEN bytes (NUL-padded) and returns how many; -MM_INGEST_E_CODES when a code is longer than 8 characters or the
 * batch holds more than 1024 different ones (the caller then reads the MM text itself).  Runs on the chain stream and waits for it. */
int32_t mm_ingest_batch_codes(mm_ingest_t *h, const mm_batch_t *batch_dev, char *codes, int32_t max_codes);
/* n bytes of device memory to the host, behind everything queued on the chain stream so far (tests; callers that want a batch's
 * read records on the host) */
int32_t mm_ingest_copy_to_host(mm_ingest_t *h, void *dst_host, const void *src_dev, size_t n);
void *mm_ingest_stream(mm_ingest_t *h);   /* hipStream_t of the flatten kernels: work submitted to it runs behind them */
/* device milliseconds of the slot's last group: [0] host -> device copy, [1] inflate, [2] CRC, [3] frame + flatten */
int32_t mm_ingest_times(mm_ingest_t *h, int32_t slot, float ms[4]);
const char *mm_ingest_strerror(int32_t code);

#ifdef __cplusplus
}
#endif
#endif
