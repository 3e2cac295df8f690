/* minimod_bgzf.h -- C ABI of the device-side BGZF inflate (libminimod_hip.so; kernels in minimod_amd/csrc/bgzf_kernels.hip.h).
 *
 * What it replaces: the reference reads its BAM through htslib (src/minimod.c:249, sam_read1 behind hts_set_threads), whose
 * thread pool inflates the BGZF blocks on the host cores.  With 16-24 host cores per GPU that inflate is the end-to-end limit
 * (DESIGN.md section 5); a BGZF block is an independent <= 64 KB unit, so thousands of them are one launch.  The host keeps
 * what is sequential -- framing the blocks of the file, framing the records of the decoded stream -- and the CRC32 of every
 * block is checked on the device as well.  A block the device refuses (status != 0) is the host decoder's to judge.
 *
 * Use: the caller owns `slots` launches in flight.  For a slot it writes the blocks' deflate payloads one behind the other into
 * mm_bgzf_staging(), describes them in mm_bgzf_blocks(), and calls mm_bgzf_submit(): compressed bytes to the device, inflate,
 * CRC, decoded bytes back to `out_host` (pinned memory from mm_bgzf_host_alloc() makes that a DMA), all asynchronous;
 * mm_bgzf_wait() blocks until the slot's bytes are in `out_host` and returns the per-block status words. */
#ifndef MINIMOD_BGZF_H
#define MINIMOD_BGZF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mm_bgzf mm_bgzf_t;

typedef struct {
    uint32_t c_off;   /* the block's raw deflate payload (behind the gzip header and its extra field) in the slot's staging */
    uint32_t c_len;
    uint32_t o_off;   /* where its decoded bytes go in the slot's output */
    uint32_t isize;   /* ISIZE of the block's trailer */
    uint32_t crc;     /* CRC32 of the block's trailer */
} mm_bgzf_block_t;

/* status words: 0 = decoded, size and CRC32 as the trailer says; 1-7 malformed deflate stream, 8 size mismatch, 9 CRC mismatch */
#define MM_BGZF_OK 0
#define MM_BGZF_E_SIZE 8
#define MM_BGZF_E_CRC 9

/* NULL on failure (err says why).  A slot holds up to max_blocks blocks, max_cbytes of payload, max_obytes of decoded bytes. */
mm_bgzf_t *mm_bgzf_create(int32_t device, int32_t slots, int32_t max_blocks, size_t max_cbytes, size_t max_obytes, char *err, size_t err_len);
void mm_bgzf_destroy(mm_bgzf_t *h);
void *mm_bgzf_host_alloc(size_t bytes);   /* pinned host memory (NULL: none); for the decoded bytes' destination */
void mm_bgzf_host_free(void *p);
uint8_t *mm_bgzf_staging(mm_bgzf_t *h, int32_t slot);            /* pinned, max_cbytes + 64 */
mm_bgzf_block_t *mm_bgzf_blocks(mm_bgzf_t *h, int32_t slot);     /* pinned, max_blocks entries */
/* 0, or a negative error (arguments out of range, a HIP failure).  out_host receives obytes bytes. */
int32_t mm_bgzf_submit(mm_bgzf_t *h, int32_t slot, int32_t n_blocks, size_t cbytes, size_t obytes, uint8_t *out_host);
int32_t mm_bgzf_wait(mm_bgzf_t *h, int32_t slot, const int32_t **status);   /* status: n_blocks words, valid until the slot's next submit */
/* device milliseconds of the slot's last launch: [0] host -> device copies, [1] inflate kernel, [2] CRC kernel, [3] device -> host */
int32_t mm_bgzf_times(mm_bgzf_t *h, int32_t slot, float ms[4]);

/* sha256 (16 hex digits) over the sources this library was built from -- every .hip / .hip.h under csrc and every header under include, minimod_amd/build.py
 * library_source_hash() -- or "unstamped": build() recompiles a shipped library that was made from other sources, smoke() checks */
const char *mm_build_source_hash(void);
/* brings the HIP runtime up on `device` (its first call takes ~0.2 s): a caller with something else to do meanwhile -- reading the
 * reference, say -- makes it on a thread of its own.  NOT in a process that is going to fork workers.  0 or -4 */
int32_t mm_hip_warm(int32_t device);

/* The library's device and pinned memory (csrc/devmem.h): blocks a handle gives up are KEPT and handed out again, never returned to the driver while the
 * process lives -- an address the driver takes back and hands out again was seen through its old translation by one XCD's workgroups when a dozen processes
 * shared the GPU (round 6).  No reference counterpart (CPU code).  mm_devmem_stats: [0] device bytes held by handles, [1] device bytes kept for reuse,
 * [2] / [3] the same for pinned memory, [4] requests served from kept blocks, [5] requests that went to the driver, [6] blocks given back (out of memory,
 * or mm_devmem_trim).  mm_devmem_trim: gives every kept block back -- for a long-lived process at a moment when none of its kernels is queued; the bytes. */
void mm_devmem_stats(int64_t out[7]);
int64_t mm_devmem_trim(void);

/* The same two kernels on blocks that already lie in DEVICE memory, the decoded bytes left there (include/minimod_ingest.h builds on
 * it): d_c = the payloads (at least 1024 readable bytes behind the last one), d_blocks = n_blocks records, d_out / d_status = where the
 * decoded bytes and the status words go, stream = a hipStream_t, between_event = a hipEvent_t recorded between the inflate and the
 * CRC kernel (or NULL).  Asynchronous.  0 or -4 (a HIP failure). */
int32_t mm_bgzf_inflate_device(int32_t device, void *stream, const uint8_t *d_c, const mm_bgzf_block_t *d_blocks, int32_t n_blocks,
                               uint8_t *d_out, int32_t *d_status, void *between_event);

#ifdef __cplusplus
}
#endif
#endif
