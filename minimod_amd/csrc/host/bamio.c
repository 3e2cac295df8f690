/* bamio.c -- see bamio.h.  BGZF = a series of gzip members, each with a 'BC' extra subfield giving the block
 * size, at most 64 KiB of payload per block (SAM/BAM specification section 4.1).
 *
 * Pipeline: a producer thread reads groups of up to 256 BGZF blocks from the file and has the shared worker pool
 * inflate them (raw deflate, zlib) into a CHUNK -- up to 16 MiB of decoded stream behind 4 MiB of head room -- while
 * the consumer parses BAM records out of the chunks decoded earlier.  A record that straddles two chunks is made
 * contiguous by copying its first part into the head room of the second chunk (or, for a record bigger than that,
 * into a spill buffer).  Record views stay valid until mm_bam_release(): the loader parses a whole batch first and
 * then copies it into the flattened pools with the same worker pool (mm_pool_for). */
#include "bamio.h"
#include "inflate_fast.h"
#include "crc32_fast.h"

#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define GROUP_BLOCKS 256
#define GPU_GROUP_BLOCKS 1024           /* with a device inflater (mm_bam_set_backend) a group is one launch: it wants thousands of
                                         * blocks in flight, three or four launches of this size */
#define GPU_GROUP_CBYTES ((size_t)40 << 20)
#define CHUNK_HEAD ((size_t)4 << 20)
#define CHUNK_PAYLOAD ((size_t)GROUP_BLOCKS * 65536)
#define GROUP_CBYTES ((size_t)8 << 20)   /* compressed bytes read per group (a group is the whole blocks among them) */
#define GROUP_TICKETS 16                 /* workers on one group at a time (several groups are in flight) */
#define CHUNKS_AHEAD 8                   /* chunks the producer may run ahead of the consumer: framed, their blocks being inflated or done */

/* ------------------------------------------------------------------ worker pool */
/* A parallel loop over [0, n) is ONE task: the queue gets a few tickets for it (one lock, not one per piece) and the workers
 * that draw a ticket take pieces of `grain` indices off the task's counter until none is left.  (Queueing every piece by
 * itself cost 16 us a piece with 128 workers on the queue's lock: 7 ms to hand out the 450 copies of a batch.) */
typedef struct pool_group {
    void (*fn)(void *arg, int64_t lo, int64_t hi);
    void *arg;
    int64_t n, grain;
    int64_t next;        /* first index nobody has taken yet (atomic) */
    int pending;         /* tickets not yet handed back */
    int urgent;          /* somebody waits for this loop now (mm_pool_for): its tickets go to the front of the queue, and a worker
                          * in the middle of a read-ahead ticket runs one of them between two of its pieces */
    pthread_mutex_t mu;
    pthread_cond_t cv;
} pool_group_t;

#define POOL_QCAP 4096
struct mm_pool {
    int n_threads;
    pthread_t *th;
    pool_group_t *q[POOL_QCAP];   /* tickets */
    int qhead, qtail, qlen;
    int stop;
    int n_urgent;                 /* urgent tickets in the queue (atomic) */
    pthread_mutex_t mu;
    pthread_cond_t cv_job, cv_room;
    /* diagnostics (MM_LOADER_TIMING): nanoseconds the workers spent inside jobs, pieces run, nanoseconds callers spent queueing */
    unsigned long long busy_ns, jobs, submit_ns;
};

static unsigned long long pool_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (unsigned long long)t.tv_sec * 1000000000ull + (unsigned long long)t.tv_nsec; }

/* one ticket: pieces off the task's counter until none is left, then the ticket is handed back */
static void pool_run_ticket(mm_pool_t *p, pool_group_t *g) {
    const unsigned long long t_j = pool_ns();
    unsigned long long pieces = 0, nested_ns = 0;
    for (;;) {
        const int64_t lo = __atomic_fetch_add(&g->next, g->grain, __ATOMIC_RELAXED);
        if (lo >= g->n) break;
        g->fn(g->arg, lo, lo + g->grain < g->n ? lo + g->grain : g->n);
        pieces++;
        if (!g->urgent && __atomic_load_n(&p->n_urgent, __ATOMIC_RELAXED) > 0) {
            pool_group_t *u = NULL;
            pthread_mutex_lock(&p->mu);
            if (p->qlen > 0 && p->q[p->qhead]->urgent) {
                u = p->q[p->qhead];
                p->qhead = (p->qhead + 1) % POOL_QCAP; p->qlen--;
                __atomic_fetch_sub(&p->n_urgent, 1, __ATOMIC_RELAXED);
                pthread_cond_signal(&p->cv_room);
            }
            pthread_mutex_unlock(&p->mu);
            if (u) { const unsigned long long t_n = pool_ns(); pool_run_ticket(p, u); nested_ns += pool_ns() - t_n; }
        }
    }
    __atomic_fetch_add(&p->busy_ns, pool_ns() - t_j - nested_ns, __ATOMIC_RELAXED);
    __atomic_fetch_add(&p->jobs, pieces, __ATOMIC_RELAXED);
    pthread_mutex_lock(&g->mu);
    if (--g->pending == 0) pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}

static void *pool_worker(void *arg) {
    mm_pool_t *p = (mm_pool_t *)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->qlen == 0 && !p->stop) pthread_cond_wait(&p->cv_job, &p->mu);
        if (p->qlen == 0 && p->stop) { pthread_mutex_unlock(&p->mu); return NULL; }
        pool_group_t *g = p->q[p->qhead];
        p->qhead = (p->qhead + 1) % POOL_QCAP; p->qlen--;
        if (g->urgent) __atomic_fetch_sub(&p->n_urgent, 1, __ATOMIC_RELAXED);
        pthread_cond_signal(&p->cv_room);
        pthread_mutex_unlock(&p->mu);
        pool_run_ticket(p, g);
    }
}

mm_pool_t *mm_pool_create(int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    mm_pool_t *p = (mm_pool_t *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->n_threads = n_threads;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv_job, NULL);
    pthread_cond_init(&p->cv_room, NULL);
    p->th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!p->th) { p->n_threads = 0; mm_pool_destroy(p); return NULL; }
    int started = 0;
    for (int i = 0; i < n_threads; i++) {
        if (pthread_create(&p->th[started], NULL, pool_worker, p) != 0) break;
        started++;
    }
    p->n_threads = started;   /* fewer workers than asked for still work; none at all is a failure */
    if (started == 0) { mm_pool_destroy(p); return NULL; }
    return p;
}

void mm_pool_destroy(mm_pool_t *p) {
    if (!p) return;
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_job);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_threads; i++) pthread_join(p->th[i], NULL);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_job); pthread_cond_destroy(&p->cv_room);
    free(p->th); free(p);
}

int mm_pool_threads(const mm_pool_t *p) { return p ? p->n_threads : 1; }
void mm_pool_stats(const mm_pool_t *p, double *busy_s, unsigned long long *jobs, double *submit_s) {
    *busy_s = p ? 1e-9 * (double)p->busy_ns : 0.0; *jobs = p ? p->jobs : 0ull; *submit_s = p ? 1e-9 * (double)p->submit_ns : 0.0;
}

/* the loop over [0, n) queued as at most `max_tickets` tickets; `g` (initialised, nothing pending) counts them down; the caller
 * waits with group_wait when it needs the results */
static void pool_submit(mm_pool_t *p, int64_t n, int64_t grain, void (*fn)(void *, int64_t, int64_t), void *arg, pool_group_t *g, int max_tickets, int front) {
    const unsigned long long t_s = pool_ns();
    const int64_t pieces = (n + grain - 1) / grain;
    int k = p->n_threads;
    if (k > max_tickets) k = max_tickets;
    if ((int64_t)k > pieces) k = (int)pieces;
    if (k > POOL_QCAP / 2) k = POOL_QCAP / 2;
    pthread_mutex_lock(&g->mu);
    g->fn = fn; g->arg = arg; g->n = n; g->grain = grain; g->urgent = front;
    __atomic_store_n(&g->next, 0, __ATOMIC_RELAXED);
    g->pending = k;
    pthread_mutex_unlock(&g->mu);
    pthread_mutex_lock(&p->mu);
    while (p->qlen + k > POOL_QCAP) pthread_cond_wait(&p->cv_room, &p->mu);
    /* front: somebody is waiting for this loop right now (mm_pool_for): its tickets go in front of the read-ahead's */
    if (front) {
        for (int i = 0; i < k; i++) { p->qhead = (p->qhead + POOL_QCAP - 1) % POOL_QCAP; p->q[p->qhead] = g; p->qlen++; }
        __atomic_fetch_add(&p->n_urgent, k, __ATOMIC_RELAXED);
    }
    else for (int i = 0; i < k; i++) { p->q[p->qtail] = g; p->qtail = (p->qtail + 1) % POOL_QCAP; p->qlen++; }
    if (2 * k >= p->n_threads) pthread_cond_broadcast(&p->cv_job);
    else for (int i = 0; i < k; i++) pthread_cond_signal(&p->cv_job);
    pthread_mutex_unlock(&p->mu);
    __atomic_fetch_add(&p->submit_ns, pool_ns() - t_s, __ATOMIC_RELAXED);
}
static void group_wait(pool_group_t *g) {
    pthread_mutex_lock(&g->mu);
    while (g->pending > 0) pthread_cond_wait(&g->cv, &g->mu);
    pthread_mutex_unlock(&g->mu);
}

void mm_pool_for(mm_pool_t *p, int64_t n, int64_t grain, void (*fn)(void *, int64_t, int64_t), void *arg) {
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    if (!p || n <= grain) { fn(arg, 0, n); return; }
    pool_group_t g;
    memset(&g, 0, sizeof g);
    pthread_mutex_init(&g.mu, NULL);
    pthread_cond_init(&g.cv, NULL);
    pool_submit(p, n, grain, fn, arg, &g, p->n_threads, 1);
    group_wait(&g);
    pthread_mutex_destroy(&g.mu); pthread_cond_destroy(&g.cv);
}

/* ------------------------------------------------------------------ chunks */
typedef struct {
    const uint8_t *cdata;
    uint32_t clen, isize, crc;   /* crc: the block's CRC32 trailer (RFC 1952) */
    uint8_t *out;
    int err;
} blk_t;

typedef struct chunk {
    uint8_t *buf;        /* CHUNK_HEAD + CHUNK_PAYLOAD */
    uint8_t *cbuf;       /* compressed bytes of the group */
    size_t len;          /* decoded bytes at buf + CHUNK_HEAD */
    int n_blk;
    blk_t blk[GPU_GROUP_BLOCKS];
    int gpu_slot;        /* >= 0: the group was given to the device inflater's slot (the consumer waits for the launch) */
    int pinned;          /* buf comes from the backend's allocator ... */
    void (*buf_free)(void *);   /* ... and goes back through this (kept with the chunk: the backend may be gone by then) */
    size_t cap_blocks;
    int err, last;       /* last: the file ended with this group */
    int tail_err;        /* the file is damaged right behind this group's last block (a bad header, a block cut off by the file's end) */
    pool_group_t grp;    /* the group's inflate jobs: the producer queues them and frames the next group, the consumer waits for
                          * them when it takes the chunk (a producer that waited for every group itself kept 64 of 128 workers
                          * busy for one group at a time, with nothing running while it framed the next) */
    struct chunk *next;
} chunk_t;

struct mm_bam {
    FILE *fp;
    mm_pool_t *pool;
    int own_pool;
    mm_bam_hdr_t hdr;
    /* producer */
    pthread_t producer;
    int producer_started;
    pthread_mutex_t mu;
    pthread_cond_t cv_ready, cv_room;
    chunk_t *ready_head, *ready_tail;   /* decoded, not yet taken by the consumer */
    int n_ready;
    chunk_t *free_list;
    int quit;
    uint8_t *carry;                     /* compressed bytes read past the last whole block of a group */
    size_t carry_len, carry_cap;
    const uint8_t *map;                 /* the whole file mapped (regular files): blocks are inflated straight from the page
                                         * cache, the workers take the page faults; NULL = read with fread into cbuf */
    size_t map_len, map_pos;
    /* consumer */
    chunk_t *cur;                       /* chunk b->p points into (NULL while it points into a spill buffer) */
    const uint8_t *p, *end;             /* unread decoded bytes */
    chunk_t *held;                      /* chunks the consumer has left but whose records may still be in use */
    uint8_t **spills; int n_spills;     /* assembled oversized records */
    uint8_t *cur_spill;                 /* the spill buffer b->p points into, if any */
    int eof, failed;
    int fail_next;                      /* the chunk handed out last ends in front of a damaged block: the next request fails */
    double wait_s;                      /* consumer: seconds spent waiting for decoded chunks */
};

static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

/* MM_BAM_VERIFY_ZLIB=1 (debug switch): every block is also inflated by zlib and the two outputs must agree */
static int verify_zlib = -1;

static int zlib_inflate(const blk_t *b, uint8_t *out) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    zs.next_in = (Bytef *)b->cdata; zs.avail_in = b->clen;
    zs.next_out = out; zs.avail_out = b->isize;
    int r = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return (r == Z_STREAM_END && zs.avail_out == 0) ? 0 : -1;
}

static int inflate_block(blk_t *b) {
    if (b->isize == 0) return b->crc == 0 ? 0 : -1;   /* crc32 of nothing */
    /* malformed or not understood by the own decoder: zlib has the last word (and decides what counts as an error) */
    if (mm_inflate_raw(b->cdata, b->clen, b->out, b->isize) != 0 && zlib_inflate(b, b->out) != 0) return -1;
    /* the trailer's CRC32 covers the decoded bytes: a damaged block (or a decoder bug) that still yields ISIZE bytes
     * must not reach the counters -- htslib fails such a file too */
    if (mm_crc32(b->out, b->isize) != b->crc) return -1;
    if (verify_zlib > 0) {
        uint8_t *chk = (uint8_t *)malloc(b->isize);
        int bad = !chk || zlib_inflate(b, chk) != 0 || memcmp(chk, b->out, b->isize) != 0;
        free(chk);
        if (bad) { fprintf(stderr, "[bamio] MM_BAM_VERIFY_ZLIB: own decoder and zlib disagree on a block\n"); return -1; }
    }
    return 0;
}

int mm_bgzf_inflate_host(const uint8_t *cdata, uint32_t clen, uint32_t isize, uint32_t crc, uint8_t *out) {
    blk_t b;
    memset(&b, 0, sizeof b);
    b.cdata = cdata; b.clen = clen; b.isize = isize; b.crc = crc; b.out = out;
    if (verify_zlib < 0) { const char *ev = getenv("MM_BAM_VERIFY_ZLIB"); verify_zlib = ev && atoi(ev) != 0; }
    return inflate_block(&b);
}

static void inflate_range(void *arg, int64_t lo, int64_t hi) {
    chunk_t *c = (chunk_t *)arg;
    for (int64_t i = lo; i < hi; i++) c->blk[i].err = inflate_block(&c->blk[i]);
}

/* ------------------------------------------------------------------ device inflater (bamio.h: mm_bgzf_backend_t) */
static const mm_bgzf_backend_t *g_be = NULL;
static pthread_mutex_t g_be_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_be_busy[16];
static unsigned long long g_be_groups, g_be_blocks, g_be_fallback_blocks;   /* diagnostics */

void mm_bam_set_backend(const mm_bgzf_backend_t *be) {
    pthread_mutex_lock(&g_be_mu);
    g_be = be;
    memset(g_be_busy, 0, sizeof g_be_busy);
    pthread_mutex_unlock(&g_be_mu);
}
void mm_bam_backend_stats(unsigned long long out[3]) { out[0] = g_be_groups; out[1] = g_be_blocks; out[2] = g_be_fallback_blocks; }

static int be_take_slot(void) {
    int got = -1;
    pthread_mutex_lock(&g_be_mu);
    if (g_be) for (int i = 0; i < g_be->slots && i < 16; i++) if (!g_be_busy[i]) { g_be_busy[i] = 1; got = i; break; }
    pthread_mutex_unlock(&g_be_mu);
    return got;
}
static void be_release_slot(int slot) {
    pthread_mutex_lock(&g_be_mu);
    if (slot >= 0 && slot < 16) g_be_busy[slot] = 0;
    pthread_mutex_unlock(&g_be_mu);
}

typedef struct { chunk_t *c; uint8_t *staging; const uint32_t *c_off; } stage_ctx_t;
static void stage_range(void *arg, int64_t lo, int64_t hi) {
    stage_ctx_t *s = (stage_ctx_t *)arg;
    for (int64_t i = lo; i < hi; i++) memcpy(s->staging + s->c_off[i], s->c->blk[i].cdata, s->c->blk[i].clen);
}

/* the group's blocks to the device: payloads one behind the other into the slot's staging (copied by the pool), the block
 * records, the launch.  0 = on its way, -1 = not taken (the host pool inflates the group as usual) */
static int be_submit_group(mm_pool_t *pool, chunk_t *c) {
    const mm_bgzf_backend_t *be = g_be;
    if (!be || !c->pinned || c->n_blk <= 0 || c->n_blk > be->max_blocks || c->len > be->max_obytes || verify_zlib > 0) return -1;
    const int slot = be_take_slot();
    if (slot < 0) return -1;
    uint32_t *rec = (uint32_t *)be->blocks(be->ctx, slot);
    uint8_t *staging = be->staging(be->ctx, slot);
    size_t cb = 0;
    for (int i = 0; i < c->n_blk; i++) {
        const blk_t *k = &c->blk[i];
        rec[5 * i + 0] = (uint32_t)cb; rec[5 * i + 1] = k->clen; rec[5 * i + 2] = (uint32_t)(k->out - (c->buf + CHUNK_HEAD));
        rec[5 * i + 3] = k->isize; rec[5 * i + 4] = k->crc;
        cb += k->clen;
    }
    if (cb > be->max_cbytes) { be_release_slot(slot); return -1; }
    {
        /* (the offsets live in the records: every fifth word) */
        uint32_t *offs = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)c->n_blk);
        if (!offs) { be_release_slot(slot); return -1; }
        for (int i = 0; i < c->n_blk; i++) offs[i] = rec[5 * i];
        stage_ctx_t sc = {c, staging, offs};
        mm_pool_for(pool, c->n_blk, 16, stage_range, &sc);
        free(offs);
    }
    if (be->submit(be->ctx, slot, c->n_blk, cb, c->len, c->buf + CHUNK_HEAD) != 0) { be_release_slot(slot); return -1; }
    c->gpu_slot = slot;
    __atomic_fetch_add(&g_be_groups, 1ull, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_be_blocks, (unsigned long long)c->n_blk, __ATOMIC_RELAXED);
    return 0;
}

static int inflate_block(blk_t *b);
/* the launch's end: a block the device refused (or a failed launch) is inflated here, by the host's decoder -- what counts as an
 * error stays what it is without a device */
static void inflate_range(void *arg, int64_t lo, int64_t hi);
static void be_finish_group(chunk_t *c, mm_pool_t *pool) {
    const mm_bgzf_backend_t *be = g_be;
    const int32_t *st = NULL;
    int r = be ? be->wait(be->ctx, c->gpu_slot, &st) : -1;
    if (r != 0 || !st) {
        /* the launch itself failed (a HIP error): the whole group through the pool, and no further group to a device that has
         * stopped answering -- one warning instead of a file read on one thread */
        mm_pool_for(pool, c->n_blk, 2, inflate_range, c);
        __atomic_fetch_add(&g_be_fallback_blocks, (unsigned long long)c->n_blk, __ATOMIC_RELAXED);
        be_release_slot(c->gpu_slot);
        pthread_mutex_lock(&g_be_mu);
        if (g_be) { g_be = NULL; fprintf(stderr, "[bamio] the device inflater failed: the host threads inflate alone from here on\n"); }
        pthread_mutex_unlock(&g_be_mu);
        c->gpu_slot = -1;
        return;
    }
    for (int i = 0; i < c->n_blk; i++)
        if (st[i] != 0) { c->blk[i].err = inflate_block(&c->blk[i]); __atomic_fetch_add(&g_be_fallback_blocks, 1ull, __ATOMIC_RELAXED); }
    be_release_slot(c->gpu_slot);
    c->gpu_slot = -1;
}

static chunk_t *chunk_new(void) {
    chunk_t *c = (chunk_t *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->gpu_slot = -1;
    {   /* a device inflater: groups of GPU_GROUP_BLOCKS, decoded into pinned memory */
        pthread_mutex_lock(&g_be_mu);
        const mm_bgzf_backend_t *be = g_be;
        pthread_mutex_unlock(&g_be_mu);
        if (be && be->host_alloc) {
            c->buf = (uint8_t *)be->host_alloc(CHUNK_HEAD + (size_t)GPU_GROUP_BLOCKS * 65536);
            c->pinned = c->buf != NULL;
            c->buf_free = be->host_free;
            c->cap_blocks = GPU_GROUP_BLOCKS;
        }
    }
    if (!c->buf) { c->buf = (uint8_t *)malloc(CHUNK_HEAD + CHUNK_PAYLOAD); c->pinned = 0; c->cap_blocks = GROUP_BLOCKS; }
    c->cbuf = (uint8_t *)malloc(GROUP_CBYTES + 65536 + 1024);
    pthread_mutex_init(&c->grp.mu, NULL);
    pthread_cond_init(&c->grp.cv, NULL);
    return c;
}
static void chunk_free(chunk_t *c) {
    if (!c) return;
    group_wait(&c->grp);   /* (a reader closed early: its workers may still be writing into the chunk) */
    if (c->gpu_slot >= 0 && g_be) { const int32_t *st; (void)g_be->wait(g_be->ctx, c->gpu_slot, &st); be_release_slot(c->gpu_slot); }
    pthread_mutex_destroy(&c->grp.mu); pthread_cond_destroy(&c->grp.cv);
    if (c->pinned) { if (c->buf_free) c->buf_free(c->buf); } else free(c->buf);
    free(c->cbuf); free(c);
}

/* one BGZF block header at h (avail bytes are there): total block size, or 0 if the header itself is cut off, -1 if bad */
static long block_total(const uint8_t *h, size_t avail, uint32_t *xlen_out) {
    if (avail < 18) return 0;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
    uint32_t xlen = rd_u16(h + 10);
    if (avail < 12 + (size_t)xlen) return 0;
    const uint8_t *x = h + 12, *xe = x + xlen;
    int bsize = -1;
    while (x + 4 <= xe) {
        uint32_t sl = rd_u16(x + 2);
        if (x + 4 + sl > xe) return -1;   /* a subfield that runs past XLEN */
        if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = rd_u16(x + 4);
        x += 4 + sl;
    }
    if (bsize < 0) return -1;
    long total = (long)bsize + 1;
    if ((size_t)total < 12 + (size_t)xlen + 8) return -1;
    *xlen_out = xlen;
    return total;
}

long mm_bgzf_block_total(const uint8_t *h, size_t avail, uint32_t *xlen_out) { return block_total(h, avail, xlen_out); }

/* the same group layout over the mapped file: no copy of the compressed bytes at all */
static int read_group_mapped(mm_bam_t *b, chunk_t *c) {
    size_t out = 0, cbytes = 0;
    int n = 0;
    const size_t max_blk = c->cap_blocks, max_cb = c->cap_blocks > GROUP_BLOCKS ? GPU_GROUP_CBYTES : GROUP_CBYTES;
    while ((size_t)n < max_blk && cbytes < max_cb && b->map_pos < b->map_len) {
        const uint8_t *h = b->map + b->map_pos;
        uint32_t xlen = 0;
        long total = block_total(h, b->map_len - b->map_pos, &xlen);
        /* a bad header, or the file ends inside a block: the blocks in front of it are good data, the error is the reader's when it
         * gets there (as with htslib, and whatever the group's size) */
        if (total <= 0 || (size_t)total > b->map_len - b->map_pos || rd_u32(h + total - 4) > 65536) {
            if (n == 0) return -1;
            c->tail_err = 1;
            break;
        }
        blk_t *k = &c->blk[n];
        k->cdata = h + 12 + xlen;
        k->clen = (uint32_t)((size_t)total - xlen - 12 - 8);
        k->isize = rd_u32(h + total - 4);
        k->crc = rd_u32(h + total - 8);
        k->out = c->buf + CHUNK_HEAD + out;
        k->err = 0;
        out += k->isize;
        b->map_pos += (size_t)total;
        cbytes += (size_t)total;
        n++;
    }
    c->n_blk = n; c->len = out;
    c->last = c->tail_err || b->map_pos >= b->map_len;
    return 0;
}

/* read one group of whole BGZF blocks into c->cbuf and lay out its blocks; 0 ok, -1 error */
static int read_group(mm_bam_t *b, chunk_t *c) {
    size_t have = 0;
    const size_t ccap = GROUP_CBYTES + 65536 + 1024;
    if (b->carry_len) { memcpy(c->cbuf, b->carry, b->carry_len); have = b->carry_len; b->carry_len = 0; }
    int file_end = 0;
    while (have < GROUP_CBYTES) {
        size_t n = fread(c->cbuf + have, 1, GROUP_CBYTES - have, b->fp);
        if (n == 0) { file_end = 1; break; }
        have += n;
    }
    size_t pos = 0, out = 0;
    int n = 0;
    for (;;) {
        if (n == GROUP_BLOCKS) break;
        if (have - pos < 18) {   /* not even a block header: need more bytes (or the file is over) */
            if (file_end || n > 0) break;
            size_t m = fread(c->cbuf + have, 1, ccap - have, b->fp);
            if (m == 0) { file_end = 1; break; }
            have += m;
            continue;
        }
        const uint8_t *h = c->cbuf + pos;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { if (n == 0) return -1; c->tail_err = 1; break; }
        uint32_t xlen = rd_u16(h + 10);
        size_t total = 0;
        if (have - pos >= 12 + (size_t)xlen) {
            const uint8_t *x = h + 12, *xe = x + xlen;
            int bsize = -1;
            while (x + 4 <= xe) {
                uint32_t sl = rd_u16(x + 2);
                if (x + 4 + sl > xe) { bsize = -2; break; }   /* a subfield that runs past XLEN */
                if (x[0] == 'B' && x[1] == 'C' && sl == 2) bsize = rd_u16(x + 4);
                x += 4 + sl;
            }
            if (bsize < 0 || (size_t)bsize + 1 < 12 + (size_t)xlen + 8) { if (n == 0) return -1; c->tail_err = 1; break; }
            total = (size_t)bsize + 1;
        }
        if (total == 0 || have - pos < total) {   /* the block is cut off by the end of what was read */
            if (n > 0) break;                      /* it starts the next group */
            if (file_end) return -1;               /* truncated file */
            size_t m = fread(c->cbuf + have, 1, ccap - have, b->fp);   /* a first block always fits: ccap > 64 KiB + header */
            if (m == 0) file_end = 1;
            have += m;
            if (file_end && (total == 0 || have - pos < total)) return -1;
            continue;
        }
        blk_t *k = &c->blk[n];
        k->cdata = h + 12 + xlen;
        k->clen = (uint32_t)(total - xlen - 12 - 8);
        k->isize = rd_u32(h + total - 4);
        k->crc = rd_u32(h + total - 8);
        if (k->isize > 65536) { if (n == 0) return -1; c->tail_err = 1; break; }
        k->out = c->buf + CHUNK_HEAD + out;
        k->err = 0;
        out += k->isize;
        pos += total;
        n++;
    }
    size_t rest = have - pos;
    if (c->tail_err) {   /* (a damaged header behind good blocks: they are handed out, the reader fails when it gets here -- as in read_group_mapped) */
        c->n_blk = n; c->len = out; c->last = 1;
        return 0;
    }
    if (rest) {
        if (file_end && n == 0) return -1;   /* trailing bytes that are not a block */
        if (rest > b->carry_cap) { b->carry = (uint8_t *)realloc(b->carry, rest); b->carry_cap = rest; }
        memcpy(b->carry, c->cbuf + pos, rest);
        b->carry_len = rest;
    }
    c->n_blk = n; c->len = out;
    c->last = file_end && rest == 0;
    return 0;
}

static void *producer_main(void *arg) {
    mm_bam_t *b = (mm_bam_t *)arg;
    for (;;) {
        pthread_mutex_lock(&b->mu);
        while (!b->quit && b->n_ready >= CHUNKS_AHEAD) pthread_cond_wait(&b->cv_room, &b->mu);
        if (b->quit) { pthread_mutex_unlock(&b->mu); return NULL; }
        chunk_t *c = b->free_list;
        if (c) b->free_list = c->next;
        pthread_mutex_unlock(&b->mu);
        if (!c) c = chunk_new();
        if (!c) {   /* out of memory: tell the consumer through the failed flag */
            pthread_mutex_lock(&b->mu);
            b->failed = 1; b->n_ready = -1;
            pthread_cond_broadcast(&b->cv_ready);
            pthread_mutex_unlock(&b->mu);
            return NULL;
        }
        c->next = NULL; c->err = 0; c->last = 0; c->len = 0; c->n_blk = 0; c->gpu_slot = -1; c->tail_err = 0;
        if (!c->buf || !c->cbuf || (b->map ? read_group_mapped(b, c) : read_group(b, c)) != 0) c->err = 1;
        else if (c->n_blk > 0) {
            if (b->map && be_submit_group(b->pool, c) == 0) { /* the device has it */ }
            else if (b->pool) pool_submit(b->pool, c->n_blk, 2, inflate_range, c, &c->grp, GROUP_TICKETS, 0);
            else inflate_range(c, 0, c->n_blk);
        }
        int stop = c->err || c->last;
        pthread_mutex_lock(&b->mu);
        if (b->ready_tail) b->ready_tail->next = c; else b->ready_head = c;
        b->ready_tail = c;
        b->n_ready++;
        pthread_cond_broadcast(&b->cv_ready);
        pthread_mutex_unlock(&b->mu);
        if (stop) return NULL;
    }
}

static void hold(mm_bam_t *b, chunk_t *c) { c->next = b->held; b->held = c; }

/* next decoded chunk (blocks until one is ready); NULL at end of file or on error (b->failed) */
static double mono_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static chunk_t *take_chunk(mm_bam_t *b) {
    if (b->fail_next) { b->fail_next = 0; b->failed = 1; b->eof = 1; return NULL; }
    if (b->eof) return NULL;
    const double t_in = mono_s();
    pthread_mutex_lock(&b->mu);
    while (!b->ready_head && b->n_ready >= 0) pthread_cond_wait(&b->cv_ready, &b->mu);
    chunk_t *c = b->ready_head;
    if (c) {
        b->ready_head = c->next;
        if (!b->ready_head) b->ready_tail = NULL;
        b->n_ready--;
        pthread_cond_signal(&b->cv_room);
    }
    pthread_mutex_unlock(&b->mu);
    if (!c) { b->failed = 1; b->eof = 1; return NULL; }
    c->next = NULL;
    group_wait(&c->grp);
    if (c->gpu_slot >= 0) be_finish_group(c, b->pool);
    b->wait_s += mono_s() - t_in;
    /* a damaged block: the blocks in front of it are handed out, the request after them fails (groups are 256 or 1024 blocks: a
     * group failed as a whole lost up to that many good blocks, and which ones depended on the group's size) */
    int first_bad = -1;
    for (int i = 0; i < c->n_blk; i++) if (c->blk[i].err) { first_bad = i; break; }
    if (c->err || first_bad == 0) { b->failed = 1; b->eof = 1; hold(b, c); return NULL; }
    if (first_bad > 0) {
        c->len = (size_t)(c->blk[first_bad].out - (c->buf + CHUNK_HEAD));
        c->n_blk = first_bad;
        b->fail_next = 1;
        return c;
    }
    if (c->tail_err) { b->fail_next = 1; return c; }
    if (c->last) b->eof = 1;
    return c;
}

/* make `need` contiguous bytes available at b->p; 1 ok, 0 clean end of file, -1 error / truncated */
static int want(mm_bam_t *b, size_t need) {
    while ((size_t)(b->end - b->p) < need) {
        const size_t tail = (size_t)(b->end - b->p);
        chunk_t *nx = take_chunk(b);
        if (!nx) return (b->failed || tail) ? -1 : 0;
        if (nx->len == 0 && tail == 0) { hold(b, nx); continue; }   /* a group of empty blocks (the EOF marker) */
        if (tail <= CHUNK_HEAD && tail + nx->len >= need) {
            /* the usual case: the unread tail moves into the head room of the next chunk */
            uint8_t *start = nx->buf + CHUNK_HEAD - tail;
            if (tail) memcpy(start, b->p, tail);
            if (b->cur) hold(b, b->cur);
            b->cur = nx; b->cur_spill = NULL;
            b->p = start; b->end = nx->buf + CHUNK_HEAD + nx->len;
        } else {
            /* a record bigger than the head room, or spanning more than two chunks: assemble it in a spill buffer */
            if (getenv("MM_BAM_DEBUG")) fprintf(stderr, "[bamio] spill: tail %zu need %zu\n", tail, need);
            size_t have = tail + nx->len;
            uint8_t *sp = (uint8_t *)malloc(have + 64);
            if (!sp) { hold(b, nx); return -1; }
            if (tail) memcpy(sp, b->p, tail);
            memcpy(sp + tail, nx->buf + CHUNK_HEAD, nx->len);
            hold(b, nx);   /* fully copied: nothing will point into it */
            while (have < need) {
                chunk_t *more = take_chunk(b);
                if (!more) { free(sp); return -1; }
                uint8_t *grown = (uint8_t *)realloc(sp, have + more->len + 64);
                if (!grown) { free(sp); hold(b, more); return -1; }
                sp = grown;
                memcpy(sp + have, more->buf + CHUNK_HEAD, more->len);
                have += more->len;
                hold(b, more);
            }
            b->spills = (uint8_t **)realloc(b->spills, sizeof(uint8_t *) * (size_t)(b->n_spills + 1));
            b->spills[b->n_spills++] = sp;
            if (b->cur) hold(b, b->cur);
            b->cur = NULL; b->cur_spill = sp;
            b->p = sp; b->end = sp + have;
        }
    }
    return 1;
}

void mm_bam_release(mm_bam_t *b) {
    /* chunks the consumer has left behind go back to the producer; the one b->p points into stays */
    pthread_mutex_lock(&b->mu);
    while (b->held) {
        chunk_t *c = b->held;
        b->held = c->next;
        c->next = b->free_list; b->free_list = c;
    }
    pthread_mutex_unlock(&b->mu);
    int k = 0;
    for (int i = 0; i < b->n_spills; i++) {
        if (b->spills[i] == b->cur_spill) b->spills[k++] = b->spills[i];
        else free(b->spills[i]);
    }
    b->n_spills = k;
}

/* open the file and start the producer at compressed offset `coffset` (the start of a BGZF block) */
static mm_bam_t *reader_start(const char *path, mm_pool_t *pool, uint64_t coffset) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    mm_bam_t *b = (mm_bam_t *)calloc(1, sizeof(*b));
    if (!b) { fclose(fp); return NULL; }
    if (verify_zlib < 0) { const char *ev = getenv("MM_BAM_VERIFY_ZLIB"); verify_zlib = ev && atoi(ev) != 0; }
    b->fp = fp;
    b->pool = pool;
    {   /* regular file: map it (pipes and the like keep the fread path) */
        struct stat st;
        if (fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && !getenv("MM_BAM_NO_MMAP")) {
            void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
            if (m != MAP_FAILED) {
                b->map = (const uint8_t *)m; b->map_len = (size_t)st.st_size; b->map_pos = 0;
                (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            }
        }
    }
    if (coffset) {
        if (b->map) { if (coffset > b->map_len) coffset = b->map_len; b->map_pos = (size_t)coffset; }
        else if (fseeko(fp, (off_t)coffset, SEEK_SET) != 0) { fclose(fp); free(b); return NULL; }
    }
    pthread_mutex_init(&b->mu, NULL);
    pthread_cond_init(&b->cv_ready, NULL);
    pthread_cond_init(&b->cv_room, NULL);
    if (pthread_create(&b->producer, NULL, producer_main, b) != 0) { mm_bam_close(b); return NULL; }
    b->producer_started = 1;
    return b;
}

static int read_header(mm_bam_t *b) {
    if (want(b, 12) != 1 || memcmp(b->p, "BAM\1", 4) != 0) return -1;
    uint32_t l_text = rd_u32(b->p + 4);
    if (want(b, 12 + (size_t)l_text) != 1) return -1;
    b->p += 8 + l_text;
    int32_t n_ref = (int32_t)rd_u32(b->p);
    b->p += 4;
    if (n_ref < 0) return -1;
    b->hdr.n_targets = n_ref;
    b->hdr.target_name = (char **)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(char *));
    b->hdr.target_len = (uint32_t *)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(uint32_t));
    if (!b->hdr.target_name || !b->hdr.target_len) { b->hdr.n_targets = 0; return -1; }
    for (int32_t i = 0; i < n_ref; i++) {
        if (want(b, 4) != 1) return -1;
        uint32_t l_name = rd_u32(b->p);
        if (want(b, 8 + (size_t)l_name) != 1) return -1;
        b->hdr.target_name[i] = (char *)malloc((size_t)l_name + 1);
        if (!b->hdr.target_name[i]) return -1;
        memcpy(b->hdr.target_name[i], b->p + 4, l_name);
        b->hdr.target_name[i][l_name] = 0;
        b->hdr.target_len[i] = rd_u32(b->p + 4 + l_name);
        b->p += 8 + l_name;
    }
    return 0;
}

/* The header alone, with plain file reads and the host decoder (no reader, no threads): what a caller wants to know of the file
 * before its readers exist (the CLI sets up the device while the inflater's pinned buffers are being made).  0 ok, -1 not a BAM
 * file / cut off inside its header. */
static int header_from_bytes(const uint8_t *p, size_t n, mm_bam_hdr_t *hdr, uint64_t *hdr_bytes) {   /* 0 done, 1 more bytes needed, -1 bad */
    if (n < 12) return 1;
    if (memcmp(p, "BAM\1", 4) != 0) return -1;
    if (rd_u32(p + 4) > (1u << 30)) return -1;   /* (a header text of a gigabyte: not a header) */
    size_t pos = 8 + (size_t)rd_u32(p + 4);
    if (n < pos + 4) return 1;
    const int32_t n_ref = (int32_t)rd_u32(p + pos);
    pos += 4;
    if (n_ref < 0) return -1;
    size_t q = pos;
    for (int32_t i = 0; i < n_ref; i++) {
        if (n < q + 4) return 1;
        const size_t l_name = rd_u32(p + q);
        if (l_name > (1u << 20)) return -1;
        if (n < q + 8 + l_name) return 1;
        q += 8 + l_name;
    }
    if (hdr_bytes) *hdr_bytes = (uint64_t)q;   /* the first record's offset in the decoded stream */
    hdr->target_name = (char **)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(char *));
    hdr->target_len = (uint32_t *)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(uint32_t));
    if (!hdr->target_name || !hdr->target_len) return -1;
    hdr->n_targets = n_ref;
    for (int32_t i = 0; i < n_ref; i++) {
        const size_t l_name = rd_u32(p + pos);
        hdr->target_name[i] = (char *)malloc(l_name + 1);
        if (!hdr->target_name[i]) return -1;
        memcpy(hdr->target_name[i], p + pos + 4, l_name);
        hdr->target_name[i][l_name] = 0;
        hdr->target_len[i] = rd_u32(p + pos + 4 + l_name);
        pos += 8 + l_name;
    }
    return 0;
}
void mm_bam_hdr_free(mm_bam_hdr_t *hdr) {
    if (!hdr) return;
    for (int32_t i = 0; i < hdr->n_targets; i++) free(hdr->target_name ? hdr->target_name[i] : NULL);
    free(hdr->target_name); free(hdr->target_len);
    memset(hdr, 0, sizeof *hdr);
}
int mm_bam_peek_header(const char *path, mm_bam_hdr_t *hdr) { return mm_bam_peek_header2(path, hdr, NULL); }
int mm_bam_peek_header2(const char *path, mm_bam_hdr_t *hdr, uint64_t *hdr_bytes) {
    memset(hdr, 0, sizeof *hdr);
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    uint8_t *raw = (uint8_t *)malloc(65536 + 64), *dec = NULL;
    size_t n_dec = 0, cap = 0;
    int rc = raw ? 1 : -1;
    while (rc == 1) {
        /* one more block */
        uint32_t xlen = 0;
        if (fread(raw, 1, 12, fp) != 12) { rc = -1; break; }
        const size_t xl = rd_u16(raw + 10);
        if (xl < 6 || fread(raw + 12, 1, xl, fp) != xl) { rc = -1; break; }
        const long total = block_total(raw, 12 + xl, &xlen);
        if (total <= 0 || total > 65536 || fread(raw + 12 + xl, 1, (size_t)total - 12 - xl, fp) != (size_t)total - 12 - xl) { rc = -1; break; }
        blk_t k;
        memset(&k, 0, sizeof k);
        k.cdata = raw + 12 + xlen;
        k.clen = (uint32_t)((size_t)total - xlen - 12 - 8);
        k.isize = rd_u32(raw + total - 4);
        k.crc = rd_u32(raw + total - 8);
        if (k.isize > 65536) { rc = -1; break; }
        if (n_dec + k.isize > cap) {
            cap = (n_dec + k.isize) * 2 + 65536;
            uint8_t *g = (uint8_t *)realloc(dec, cap);
            if (!g) { rc = -1; break; }
            dec = g;
        }
        k.out = dec + n_dec;
        if (inflate_block(&k) != 0) { rc = -1; break; }
        n_dec += k.isize;
        rc = header_from_bytes(dec, n_dec, hdr, hdr_bytes);
    }
    free(raw); free(dec); fclose(fp);
    if (rc != 0) { mm_bam_hdr_free(hdr); return -1; }
    return 0;
}

mm_bam_t *mm_bam_open_pool(const char *path, mm_pool_t *pool) {
    mm_bam_t *b = reader_start(path, pool, 0);
    if (!b) return NULL;
    if (read_header(b) != 0) { mm_bam_close(b); return NULL; }
    mm_bam_release(b);
    return b;
}

/* the reader positioned on the record at virtual offset `voffset` (compressed block offset << 16 | offset inside the
 * decoded block, as a .bai gives it); 0 = the first record.  The header is read from the start of the file first. */
mm_bam_t *mm_bam_open_pool_at(const char *path, mm_pool_t *pool, uint64_t voffset) {
    if (voffset == 0) return mm_bam_open_pool(path, pool);
    mm_bam_t *h = mm_bam_open_pool(path, pool);
    if (!h) return NULL;
    mm_bam_t *b = reader_start(path, pool, voffset >> 16);
    if (!b) { mm_bam_close(h); return NULL; }
    b->hdr = h->hdr;                       /* the header moves over */
    memset(&h->hdr, 0, sizeof h->hdr);
    mm_bam_close(h);
    const size_t skip = (size_t)(voffset & 0xFFFF);
    if (skip) {
        if (want(b, skip) != 1) { mm_bam_close(b); return NULL; }
        b->p += skip;
    }
    mm_bam_release(b);
    return b;
}

mm_bam_t *mm_bam_open_at(const char *path, int n_threads, uint64_t voffset) {
    mm_pool_t *pool = mm_pool_create(n_threads);
    if (!pool) return NULL;
    mm_bam_t *b = mm_bam_open_pool_at(path, pool, voffset);
    if (!b) { mm_pool_destroy(pool); return NULL; }
    b->own_pool = 1;
    return b;
}
mm_bam_t *mm_bam_open(const char *path, int n_threads) { return mm_bam_open_at(path, n_threads, 0); }

/* ------------------------------------------------------------------ .bai (SAM specification section 5.2): the linear index */
mm_bai_t *mm_bai_load(const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    mm_bai_t *x = (mm_bai_t *)calloc(1, sizeof(*x));
    uint8_t m[8];
    int ok = x && fread(m, 1, 8, fp) == 8 && memcmp(m, "BAI\1", 4) == 0;
    int32_t n_ref = ok ? (int32_t)rd_u32(m + 4) : 0;
    if (ok && (n_ref < 0 || n_ref > (1 << 24))) ok = 0;
    if (ok) {
        x->n_ref = n_ref;
        x->n_intv = (int32_t *)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(int32_t));
        x->ioffset = (uint64_t **)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(uint64_t *));
        ok = x->n_intv && x->ioffset;
    }
    for (int32_t t = 0; ok && t < n_ref; t++) {
        if (fread(m, 1, 4, fp) != 4) { ok = 0; break; }
        int32_t n_bin = (int32_t)rd_u32(m);
        for (int32_t k = 0; ok && k < n_bin; k++) {          /* the bins are skipped: only the linear index is used */
            if (fread(m, 1, 8, fp) != 8) { ok = 0; break; }
            int32_t n_chunk = (int32_t)rd_u32(m + 4);
            if (n_chunk < 0 || fseeko(fp, (off_t)16 * n_chunk, SEEK_CUR) != 0) ok = 0;
        }
        if (!ok || fread(m, 1, 4, fp) != 4) { ok = 0; break; }
        int32_t n_intv = (int32_t)rd_u32(m);
        if (n_intv < 0) { ok = 0; break; }
        x->n_intv[t] = n_intv;
        x->ioffset[t] = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_intv > 0 ? n_intv : 1));
        if (!x->ioffset[t] || fread(x->ioffset[t], 8, (size_t)n_intv, fp) != (size_t)n_intv) { ok = 0; break; }
    }
    fclose(fp);
    if (!ok) { mm_bai_free(x); return NULL; }
    return x;
}
void mm_bai_free(mm_bai_t *x) {
    if (!x) return;
    if (x->ioffset) for (int32_t t = 0; t < x->n_ref; t++) free(x->ioffset[t]);
    free(x->ioffset); free(x->n_intv); free(x);
}
/* where to start reading for alignments that START at or after (tid, pos): the smallest virtual offset the linear index
 * gives for that window or, when nothing overlaps it, for the next window anything overlaps (on this or a later
 * reference).  Records in front of (tid, pos) may follow from there: the caller skips them.  UINT64_MAX = nothing left. */
uint64_t mm_bai_start(const mm_bai_t *x, int32_t tid, int64_t pos) {
    if (tid < 0) tid = 0, pos = 0;
    for (int32_t t = tid; t < x->n_ref; t++) {
        int64_t w0 = t == tid ? pos >> 14 : 0;
        for (int64_t w = w0; w < x->n_intv[t]; w++) if (x->ioffset[t][w]) return x->ioffset[t][w];
    }
    return UINT64_MAX;
}

mm_pool_t *mm_bam_pool(mm_bam_t *b) { return b->pool; }
double mm_bam_wait_seconds(const mm_bam_t *b) { return b->wait_s; }

const mm_bam_hdr_t *mm_bam_header(const mm_bam_t *b) { return &b->hdr; }

int mm_bam_next(mm_bam_t *b, mm_bam_rec_t *r) {
    int w = want(b, 4);
    if (w <= 0) return w;
    uint32_t bs = rd_u32(b->p);
    if (bs < 32) return -1;
    w = want(b, 4 + (size_t)bs);
    if (w <= 0) return -1;
    const uint8_t *p = b->p + 4;
    r->tid = (int32_t)rd_u32(p); r->pos = (int32_t)rd_u32(p + 4);
    r->l_read_name = p[8]; r->mapq = p[9];
    r->n_cigar = rd_u16(p + 12); r->flag = rd_u16(p + 14);
    r->l_qseq = (int32_t)rd_u32(p + 16);
    if (r->l_qseq < 0) return -1;
    size_t o = 32;
    if (r->l_read_name == 0 || o + r->l_read_name > bs || p[o + r->l_read_name - 1] != 0) return -1;   /* NUL-terminated inside l_read_name */
    r->qname = (const char *)(p + o); o += r->l_read_name;
    r->cigar = (const uint32_t *)(p + o); o += 4 * (size_t)r->n_cigar;
    r->seq = p + o; o += ((size_t)r->l_qseq + 1) / 2;
    o += (size_t)r->l_qseq;
    if (o > bs) return -1;
    r->aux = p + o; r->l_aux = (int32_t)(bs - o);
    r->l_data = (int32_t)(bs - 32 + ((4 - (r->l_read_name & 3)) & 3)); /* htslib pads qname to a multiple of 4 */
    b->p += 4 + (size_t)bs;
    return 1;
}

void mm_bam_close(mm_bam_t *b) {
    if (!b) return;
    if (b->producer_started) {
        pthread_mutex_lock(&b->mu);
        b->quit = 1;
        pthread_cond_broadcast(&b->cv_room);
        pthread_mutex_unlock(&b->mu);
        pthread_join(b->producer, NULL);
    }
    /* the chunks first: a reader closed early (the header's reader of mm_bam_open_pool_at) has groups in flight whose workers
     * read the mapping -- chunk_free waits for them */
    chunk_t *lists[3] = {b->ready_head, b->free_list, b->held};
    for (int k = 0; k < 3; k++) for (chunk_t *c = lists[k]; c;) { chunk_t *n = c->next; chunk_free(c); c = n; }
    chunk_free(b->cur);
    if (b->map) munmap((void *)b->map, b->map_len);
    if (b->fp) fclose(b->fp);
    for (int32_t i = 0; i < b->hdr.n_targets; i++) free(b->hdr.target_name ? b->hdr.target_name[i] : NULL);
    free(b->hdr.target_name); free(b->hdr.target_len);
    for (int i = 0; i < b->n_spills; i++) free(b->spills[i]);
    free(b->spills); free(b->carry);
    if (b->own_pool) mm_pool_destroy(b->pool);
    pthread_mutex_destroy(&b->mu); pthread_cond_destroy(&b->cv_ready); pthread_cond_destroy(&b->cv_room);
    free(b);
}

const uint8_t *mm_aux_get(const uint8_t *aux, int32_t l_aux, const char tag[2]) {
    /* like bam_aux_get, a tag whose payload does not fit [aux, aux + l_aux) -- a B array longer than the record, a Z
     * string without its NUL -- ends the walk: the record is treated as not carrying the tag */
    const uint8_t *p = aux, *e = aux + l_aux;
    while (p + 3 <= e) {
        const uint8_t *t = p + 2;
        int match = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
        size_t sz;
        switch (*t) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'd': sz = 8; break;
            case 'Z': case 'H': {
                const uint8_t *z = (const uint8_t *)memchr(t + 1, 0, (size_t)(e - (t + 1)));
                if (!z) return NULL;
                sz = (size_t)(z - (t + 1)) + 1;
                break;
            }
            case 'B': {
                if (t + 6 > e) return NULL;
                uint32_t n = rd_u32(t + 2);
                size_t es;
                switch (t[1]) {
                    case 'c': case 'C': es = 1; break;
                    case 's': case 'S': es = 2; break;
                    case 'i': case 'I': case 'f': es = 4; break;
                    default: return NULL;
                }
                sz = 5 + (size_t)n * es;
                break;
            }
            default: return NULL;
        }
        if (sz > (size_t)(e - (t + 1))) return NULL;
        if (match) return t;
        p = t + 1 + sz;
    }
    return NULL;
}
