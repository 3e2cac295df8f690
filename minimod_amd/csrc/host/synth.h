/* synth.h -- synthetic inputs for tests and bench.py.  Plain C. */
#ifndef MM_SYNTH_H
#define MM_SYNTH_H
#include <stdint.h>

#include "minimod_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MM_SYNTH_ONT = 0, MM_SYNTH_HIFI = 1 };

typedef struct mm_synth_opts {
    uint64_t seed;
    int64_t contig_len;      /* length of the contig reads align to */
    int64_t region_begin;    /* reads start inside [region_begin, region_begin+region_len) */
    int64_t region_len;      /* 0 = whole contig */
    int64_t n_reads_total;   /* number of reads over the region (sets the stratified start positions) */
    double median_len;       /* ONT log-normal median (0 = 12000) */
    double max_len;          /* ONT length cap (0 = 200000) */
    double dot_fraction;     /* fraction of reads whose MM groups use the '.' flag (implicit calls) */
    int32_t tid;
    int32_t shape;           /* MM_SYNTH_ONT / MM_SYNTH_HIFI */
    int32_t single_code;     /* 1: only "C+m" ; 0: "C+h?...;C+m?...;" */
    int32_t haplotypes;      /* 1: HP in {0 (absent),1,2} */
    int32_t long_insertions; /* 1: longer insertions made of CpGs (config C5) */
    int32_t rsvd;
} mm_synth_opts_t;

/* a host batch that owns its pools */
typedef struct mm_host_batch {
    mm_batch_t b;
    uint64_t n_bases;         /* sum of l_qseq */
    uint64_t n_listed_calls;  /* MM-listed calls over all groups */
} mm_host_batch_t;

void mm_synth_reference(uint64_t seed, int64_t len, uint8_t *out);
/* begin must be a multiple of 1 MiB; out points at position `begin` */
void mm_synth_reference_slice(uint64_t seed, int64_t begin, int64_t len, uint8_t *out);
int mm_synth_batch(const mm_synth_opts_t *o, const uint8_t *ref, int64_t first_read, int32_t n_reads, mm_host_batch_t *out);
void mm_synth_batch_free(mm_host_batch_t *b);
int mm_batch_make_order(mm_host_batch_t *b);

/* any flattened host batch as a BGZF-compressed BAM; optional filter fodder (unmapped / secondary / tag-less copies) */
typedef struct mm_bam_writer mm_bam_writer_t;
mm_bam_writer_t *mm_bam_writer_open(const char *path, int32_t n_contigs, const char *const *names, const int64_t *lens);
/* a big file written in pieces on several threads and concatenated (BGZF members concatenate) */
enum { MM_BAMW_NO_HEADER = 1, MM_BAMW_NO_EOF = 2, MM_BAMW_INDEX = 4 /* also write <path>.bai; a piece's index holds offsets inside the piece (synth.py merge_bai shifts and joins them) */ };
mm_bam_writer_t *mm_bam_writer_open_piece(const char *path, int32_t n_contigs, const char *const *names, const int64_t *lens,
                                          int flags, uint64_t first_serial);
int mm_bam_writer_put_batch(mm_bam_writer_t *bw, const mm_batch_t *b, int with_filter_fodder);
int mm_bam_writer_close(mm_bam_writer_t *bw);
int mm_write_fasta(const char *path, const char *name, const uint8_t *seq, int64_t len);

#ifdef __cplusplus
}
#endif
#endif
