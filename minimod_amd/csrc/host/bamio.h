/* bamio.h -- minimal BGZF/BAM reader for the freq path (own code; replaces the htslib calls the reference makes:
 * sam_open/sam_hdr_read/sam_read1 reference src/minimod.c:73-90,250 and the aux accessors of src/mod.c:123-202).
 * BGZF blocks are inflated by a worker pool, one group ahead of the parser (the reference gets block-level
 * parallelism from hts_set_threads, minimod.c:76-78). */
#ifndef MM_BAMIO_H
#define MM_BAMIO_H
#include <stdint.h>
#include <stdio.h>

typedef struct mm_bam_hdr {
    int32_t n_targets;
    char **target_name;
    uint32_t *target_len;
} mm_bam_hdr_t;

/* a view of one alignment record inside the reader's buffers (valid until mm_bam_release) */
typedef struct mm_bam_rec {
    int32_t tid, pos;
    uint16_t flag;
    uint8_t mapq, l_read_name;
    uint32_t n_cigar;
    int32_t l_qseq;
    const char *qname;
    const uint32_t *cigar;   /* may be unaligned: copy with memcpy */
    const uint8_t *seq;      /* (l_qseq+1)/2 bytes */
    const uint8_t *aux;
    int32_t l_aux;
    int32_t l_data;          /* htslib's bam1_t.l_data for this record (used by the -B rule, minimod.c:249,324) */
} mm_bam_rec_t;

typedef struct mm_bam mm_bam_t;

/* worker pool shared by the reader (block inflate) and the loader (parallel copy into the flattened pools) */
/* A device-side BGZF inflater the reader may hand whole groups of blocks to (include/minimod_bgzf.h has one; this library does
 * not link it: the program that does fills the table in).  Groups the backend has no free slot for -- and blocks it refuses --
 * are inflated by the host pool as without it. */
typedef struct mm_bgzf_backend {
    void *ctx;
    int slots, max_blocks;            /* launches in flight; blocks, payload bytes, decoded bytes per launch */
    size_t max_cbytes, max_obytes;
    void *(*host_alloc)(size_t);      /* pinned host memory for the decoded groups */
    void (*host_free)(void *);
    uint8_t *(*staging)(void *ctx, int slot);
    void *(*blocks)(void *ctx, int slot);   /* records of five uint32: payload offset, payload length, output offset, ISIZE, CRC32 */
    int (*submit)(void *ctx, int slot, int n_blocks, size_t cbytes, size_t obytes, uint8_t *out_host);
    int (*wait)(void *ctx, int slot, const int32_t **status);
} mm_bgzf_backend_t;
void mm_bam_set_backend(const mm_bgzf_backend_t *be);   /* process-wide; NULL: host inflate only.  Before the readers are opened;
                                                         * the table must outlive them */
void mm_bam_backend_stats(unsigned long long out[3]);   /* groups and blocks given to the device, blocks inflated again on the host */

typedef struct mm_pool mm_pool_t;
mm_pool_t *mm_pool_create(int n_threads);
void mm_pool_destroy(mm_pool_t *p);
int mm_pool_threads(const mm_pool_t *p);
void mm_pool_stats(const mm_pool_t *p, double *busy_s, unsigned long long *jobs, double *submit_s);   /* diagnostics: seconds inside jobs (all workers), jobs run, seconds spent queueing */
/* fn(arg, lo, hi) over [0, n) in pieces of `grain` items; returns when every piece is done */
void mm_pool_for(mm_pool_t *p, int64_t n, int64_t grain, void (*fn)(void *, int64_t, int64_t), void *arg);

mm_bam_t *mm_bam_open(const char *path, int n_threads);        /* with its own pool of n_threads workers */
mm_bam_t *mm_bam_open_pool(const char *path, mm_pool_t *pool);  /* on a pool the caller owns */
/* positioned readers: start at the record at virtual offset `voffset` of a .bai (0 = the first record) */
mm_bam_t *mm_bam_open_at(const char *path, int n_threads, uint64_t voffset);
mm_bam_t *mm_bam_open_pool_at(const char *path, mm_pool_t *pool, uint64_t voffset);
/* the linear index of a .bai (the reference never reads an index: its region code is commented out, src/minimod.c:92-130;
 * here it lets every GPU's worker start at its share of the file) */
typedef struct mm_bai { int32_t n_ref; int32_t *n_intv; uint64_t **ioffset; } mm_bai_t;
mm_bai_t *mm_bai_load(const char *bai_path);
void mm_bai_free(mm_bai_t *x);
uint64_t mm_bai_start(const mm_bai_t *x, int32_t tid, int64_t pos);   /* UINT64_MAX: nothing at or after (tid, pos) */
mm_pool_t *mm_bam_pool(mm_bam_t *b);
double mm_bam_wait_seconds(const mm_bam_t *b);   /* seconds the record reader has waited for decoded data so far */
/* record views handed out since the last release are no longer needed: their buffers may be reused */
void mm_bam_release(mm_bam_t *b);
const mm_bam_hdr_t *mm_bam_header(const mm_bam_t *b);
/* the header alone, read with plain file reads on the calling thread (no reader is made): 0 ok (mm_bam_hdr_free it), -1 not a BAM file */
int mm_bam_peek_header(const char *path, mm_bam_hdr_t *hdr);
int mm_bam_peek_header2(const char *path, mm_bam_hdr_t *hdr, uint64_t *hdr_bytes);   /* ... and where the first record begins in the decoded stream */
void mm_bam_hdr_free(mm_bam_hdr_t *hdr);
/* one BGZF block header at h (avail bytes are there): the block's total size, 0 if the header itself is cut off, -1 if it is no
 * BGZF header; *xlen_out = its extra field's length (the deflate payload begins at h + 12 + xlen, the CRC32 / ISIZE trailer is the
 * block's last 8 bytes) */
long mm_bgzf_block_total(const uint8_t *h, size_t avail, uint32_t *xlen_out);
/* one block's payload through the host's decoders (the own one, then zlib), CRC32 checked: 0 ok, -1 a damaged block */
int mm_bgzf_inflate_host(const uint8_t *cdata, uint32_t clen, uint32_t isize, uint32_t crc, uint8_t *out);
/* 1 = record read, 0 = end of file, <0 = error */
int mm_bam_next(mm_bam_t *b, mm_bam_rec_t *rec);
void mm_bam_close(mm_bam_t *b);

/* bam_aux_get: pointer to the TYPE byte of the first tag `tag`, or NULL */
const uint8_t *mm_aux_get(const uint8_t *aux, int32_t l_aux, const char tag[2]);

#endif
