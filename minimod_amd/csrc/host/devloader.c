/* devloader.c -- load_db (reference src/minimod.c:235-333) with the decoded BAM kept in GPU memory: the host finds the BGZF
 * blocks of the file and moves their COMPRESSED bytes; include/minimod_ingest.h inflates them, frames the records, applies
 * load_db's filters and writes the flattened batch on the device (loader.c does the same on the host for everything this
 * path does not take: pipes, view, runs that replay the reference's tie order).
 *
 * A producer thread reads windows of the file into the group slots' pinned staging (the worker pool's threads read pieces of a
 * window side by side), walks the windows' BGZF block headers, and starts the group's copy + inflate; the consumer
 * (mmh_devloader_next) takes the groups in file order, has them framed + flattened into the batch under construction and
 * hands the batch out when it is big enough to fill the GPU.  -K and -B do not cut these batches: the reference's output does
 * not depend on them (SURVEY section 8c "Invariance"); the totals the reference prints are kept. */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "minimod_ingest.h"
#include "mmhost.h"

#define DL_MAX_SLOTS 16
#define DL_MAX_ARENAS 16

typedef struct {
    int n_blocks, last, err;     /* last: the file ends with this group; err: the file is damaged at (n_blocks == 0) or right behind this group */
    size_t cbytes, obytes;
} ginfo_t;

struct mmh_devloader {
    mm_ingest_t *ing;
    mm_pool_t *pool;
    int fd;
    uint64_t file_size, file_pos;        /* producer */
    uint64_t first_skip;
    size_t avg_block;                    /* compressed bytes a block of the last group took (sizes the next window) */
    int n_slots, n_arenas;
    /* producer -> consumer: group slots in file order */
    pthread_t producer;
    int producer_started;
    pthread_mutex_t mu;
    pthread_cond_t cv_ready, cv_free;
    int ring[DL_MAX_SLOTS], ring_head, ring_len;   /* slots with a group under way, oldest first */
    int slot_free[DL_MAX_SLOTS];
    ginfo_t ginfo[DL_MAX_SLOTS];
    int quit, producer_done, producer_err;
    /* consumer */
    int arena_busy[DL_MAX_ARENAS];
    int cur_arena;                       /* the batch under construction (-1: none) */
    mm_ingest_result_t cur;              /* ... as the last group left it */
    int rerun_slot;                      /* a group that did not fit its arena any more: first into the next batch */
    int first_group, finished, failed;
    uint64_t target_bases;
    uint64_t groups, slow_blocks, patched_blocks;
    double t_wait, t_stage;
    mmh_devloader_stats_t st;
};

static double dl_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

typedef struct { int fd; uint8_t *dst; uint64_t off; size_t len; int err; } read_ctx_t;
#define DL_PIECE ((int64_t)4 << 20)
static void read_range(void *arg, int64_t lo, int64_t hi) {
    read_ctx_t *c = (read_ctx_t *)arg;
    for (int64_t k = lo; k < hi; k++) {
        size_t a = (size_t)(k * DL_PIECE), e = a + (size_t)DL_PIECE < c->len ? a + (size_t)DL_PIECE : c->len;
        while (a < e) {
            ssize_t n = pread(c->fd, c->dst + a, e - a, (off_t)(c->off + a));
            if (n < 0) { if (errno == EINTR) continue; c->err = 1; return; }
            if (n == 0) { c->err = 1; return; }   /* the file shrank */
            a += (size_t)n;
        }
    }
}

/* one window of the file into the slot's staging, its whole blocks described: the next window begins behind the last of them */
static void stage_group(mmh_devloader_t *dl, int slot, ginfo_t *g) {
    const size_t cap = (size_t)mm_ingest_max_cbytes(dl->ing);
    const int max_blocks = mm_ingest_max_blocks(dl->ing);
    uint8_t *st = mm_ingest_staging(dl->ing, slot);
    mm_bgzf_block_t *rec = mm_ingest_blocks(dl->ing, slot);
    memset(g, 0, sizeof *g);
    if (!st || !rec) { g->err = 2; g->last = 1; return; }   /* (the slot's buffers could not be made) */
    const uint64_t left = dl->file_size - dl->file_pos;
    /* the window: what max_blocks blocks of the last group's size take (+ 2 % and a block), so that little is read twice */
    size_t want = cap;
    if (dl->avg_block) { const size_t w = dl->avg_block * (size_t)max_blocks; want = w + w / 50 + 65536 + 1024; if (want > cap) want = cap; }
    const size_t len = left < want ? (size_t)left : want;
    read_ctx_t rc = {dl->fd, st, dl->file_pos, len, 0};
    mm_pool_for(dl->pool, (int64_t)((len + (size_t)DL_PIECE - 1) / (size_t)DL_PIECE), 1, read_range, &rc);
    if (rc.err) { g->err = 1; g->last = 1; return; }
    size_t pos = 0, out = 0;
    int n = 0;
    while (n < max_blocks && pos < len) {
        uint32_t xlen = 0;
        const long total = mm_bgzf_block_total(st + pos, len - pos, &xlen);
        if (total == 0 || (total > 0 && (size_t)total > len - pos)) {      /* cut off by the window's end ... */
            if (len == left) { g->err = 1; g->last = 1; }                   /* ... which is the file's: truncated */
            break;
        }
        if (total < 0 || rd_u32(st + pos + total - 4) > 65536u) { g->err = 1; g->last = 1; break; }   /* no BGZF block: the blocks in front of it are good data */
        rec[n].c_off = (uint32_t)(pos + 12 + xlen);
        rec[n].c_len = (uint32_t)((size_t)total - xlen - 12 - 8);
        rec[n].o_off = (uint32_t)out;
        rec[n].isize = rd_u32(st + pos + total - 4);
        rec[n].crc = rd_u32(st + pos + total - 8);
        out += rec[n].isize;
        pos += (size_t)total;
        n++;
    }
    if (n == 0 && !g->err && pos < len) { g->err = 1; g->last = 1; }   /* a block bigger than the window cannot be */
    g->n_blocks = n; g->cbytes = pos; g->obytes = out;
    if (n > 0) dl->avg_block = pos / (size_t)n + 1;
    dl->file_pos += pos;
    if (dl->file_pos >= dl->file_size) g->last = 1;
}

static void *producer_main(void *arg) {
    mmh_devloader_t *dl = (mmh_devloader_t *)arg;
    for (;;) {
        int slot = -1;
        pthread_mutex_lock(&dl->mu);
        for (;;) {
            if (dl->quit) break;
            for (int i = 0; i < dl->n_slots; i++) if (dl->slot_free[i]) { slot = i; break; }
            if (slot >= 0) break;
            pthread_cond_wait(&dl->cv_free, &dl->mu);
        }
        if (slot >= 0) dl->slot_free[slot] = 0;
        pthread_mutex_unlock(&dl->mu);
        if (slot < 0) return NULL;
        ginfo_t g;
        const double t0 = dl_now();
        stage_group(dl, slot, &g);
        dl->t_stage += dl_now() - t0;
        int stop = g.last;
        if (g.n_blocks > 0 || !g.err) {
            if (mm_ingest_inflate(dl->ing, slot, g.n_blocks, g.cbytes, g.obytes) != 0) { g.err = 2; g.n_blocks = 0; stop = 1; }
        }
        pthread_mutex_lock(&dl->mu);
        dl->ginfo[slot] = g;
        dl->ring[(dl->ring_head + dl->ring_len) % DL_MAX_SLOTS] = slot;
        dl->ring_len++;
        if (stop) dl->producer_done = 1;
        pthread_cond_broadcast(&dl->cv_ready);
        pthread_mutex_unlock(&dl->mu);
        if (stop) return NULL;
    }
}

mmh_devloader_t *mmh_devloader_open(const char *bam_path, mm_pool_t *pool, const mmh_devloader_opts_t *o, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    mmh_devloader_t *dl = (mmh_devloader_t *)calloc(1, sizeof(*dl));
    if (!dl) return NULL;
    dl->fd = open(bam_path, O_RDONLY);
    struct stat st;
    if (dl->fd < 0 || fstat(dl->fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        if (err) snprintf(err, err_len, "%s: not a regular file that can be opened", bam_path);
        if (dl->fd >= 0) close(dl->fd);
        free(dl);
        return NULL;
    }
    dl->file_size = (uint64_t)st.st_size;
    dl->pool = pool;
    /* where the records begin: behind the header, or at a .bai's virtual offset (block offset << 16 | offset inside the block) */
    if (o->voffset) { dl->file_pos = o->voffset >> 16; dl->first_skip = o->voffset & 0xFFFFu; }
    else { dl->file_pos = 0; dl->first_skip = o->header_bytes; }
    if (dl->file_pos > dl->file_size) dl->file_pos = dl->file_size;
    mm_ingest_opts_t io;
    memset(&io, 0, sizeof io);
    io.device = o->device; io.n_targets = o->n_targets; io.allow_secondary = o->allow_secondary; io.skip_supplementary = o->skip_supplementary;
    io.ranged = o->ranged; io.first = o->first; io.last = o->last; io.lo_tid = o->lo_tid; io.hi_tid = o->hi_tid; io.lo_pos = o->lo_pos; io.hi_pos = o->hi_pos;
    io.group_slots = o->group_slots; io.max_blocks = o->max_blocks; io.arenas = o->arenas; io.max_cbytes = o->max_cbytes; io.arena_bytes = o->arena_bytes; io.head_room = o->head_room; io.names = o->names;
    {   /* tests: small groups and batches through the environment */
        const char *e1 = getenv("MM_INGEST_MAX_BLOCKS"), *e2 = getenv("MM_INGEST_TARGET_BASES");
        if (e1 && atoi(e1) > 0) { io.max_blocks = atoi(e1); if (!io.max_cbytes) io.max_cbytes = (uint64_t)io.max_blocks * 66000 + 65536; }
        if (e2 && atoll(e2) > 0) dl->target_bases = (uint64_t)atoll(e2);
        const char *e3 = getenv("MM_INGEST_HEAD_ROOM");   /* (a head room shorter than a record: the reader gives up on its first group, the CLI goes on with the host reader) */
        if (e3 && atoll(e3) > 0) io.head_room = (uint64_t)atoll(e3);
    }
    const double t_create = dl_now();
    dl->ing = mm_ingest_create(&io, err, err_len);
    if (getenv("MM_TIMELINE")) fprintf(stderr, "[timeline] mm_ingest_create took %.3f s\n", dl_now() - t_create);
    if (!dl->ing) { close(dl->fd); free(dl); return NULL; }
    dl->n_slots = mm_ingest_group_slots(dl->ing);
    dl->n_arenas = io.arenas > 0 ? io.arenas : 3;
    if (dl->n_slots > DL_MAX_SLOTS) dl->n_slots = DL_MAX_SLOTS;
    for (int i = 0; i < dl->n_slots; i++) dl->slot_free[i] = 1;
    dl->cur_arena = -1; dl->rerun_slot = -1; dl->first_group = 1;
    if (!dl->target_bases) dl->target_bases = o->target_bases ? o->target_bases : (uint64_t)600 * 1000 * 1000;
    pthread_mutex_init(&dl->mu, NULL);
    pthread_cond_init(&dl->cv_ready, NULL);
    pthread_cond_init(&dl->cv_free, NULL);
    if (o->range_done_before_start) { dl->finished = 1; return dl; }   /* the index has nothing at or behind the share's start */
    if (pthread_create(&dl->producer, NULL, producer_main, dl) != 0) { mmh_devloader_close(dl); if (err) snprintf(err, err_len, "could not start the reader thread"); return NULL; }
    dl->producer_started = 1;
    return dl;
}

static void free_slot(mmh_devloader_t *dl, int slot) {
    pthread_mutex_lock(&dl->mu);
    dl->slot_free[slot] = 1;
    pthread_cond_signal(&dl->cv_free);
    pthread_mutex_unlock(&dl->mu);
}

/* the oldest group under way (blocks until the producer has one); -1: none will come */
static int take_group(mmh_devloader_t *dl, ginfo_t *g) {
    const double t0 = dl_now();
    pthread_mutex_lock(&dl->mu);
    while (dl->ring_len == 0 && !dl->producer_done) pthread_cond_wait(&dl->cv_ready, &dl->mu);
    int slot = -1;
    if (dl->ring_len > 0) {
        slot = dl->ring[dl->ring_head];
        dl->ring_head = (dl->ring_head + 1) % DL_MAX_SLOTS; dl->ring_len--;
        *g = dl->ginfo[slot];
    }
    pthread_mutex_unlock(&dl->mu);
    dl->t_wait += dl_now() - t0;
    return slot;
}

static int pick_arena(mmh_devloader_t *dl) {
    for (int i = 0; i < dl->n_arenas; i++) if (!dl->arena_busy[i]) return i;
    return -1;
}

/* flatten the slot's group into the batch under construction; blocks the device refused are decoded here and the group runs
 * again.  0 ok (dl->cur updated), 1 the group did not fit the arena (nothing changed), -1 failed */
static int flatten_group(mmh_devloader_t *dl, int slot, const ginfo_t *g, int new_arena, mm_ingest_result_t *r) {
    for (int attempt = 0; attempt < 2; attempt++) {
        if (mm_ingest_flatten(dl->ing, slot, dl->cur_arena, new_arena, dl->first_group ? dl->first_skip : 0) != 0) return -1;
        if (mm_ingest_result(dl->ing, slot, r) != 0) return -1;
        if (r->n_bad_blocks == 0) break;
        if (attempt == 1) return -1;
        const mm_bgzf_block_t *rec = mm_ingest_blocks(dl->ing, slot);
        const uint8_t *st = mm_ingest_staging(dl->ing, slot);
        uint8_t *tmp = (uint8_t *)malloc(65536 + 64);
        if (!tmp) return -1;
        for (int i = 0; i < g->n_blocks; i++) {
            if (r->status[i] == 0) continue;
            /* what counts as a damaged block is the host decoders' to say (as without a device) */
            if (mm_bgzf_inflate_host(st + rec[i].c_off, rec[i].c_len, rec[i].isize, rec[i].crc, tmp) != 0 || mm_ingest_patch_block(dl->ing, slot, i, tmp, rec[i].isize) != 0) { free(tmp); return -1; }
            dl->patched_blocks++;
        }
        free(tmp);
    }
    if (r->err == MM_INGEST_E_ARENA) return 1;
    if (r->err) { dl->st.err = r->err; return -1; }
    return 0;
}

int32_t mmh_devloader_next(mmh_devloader_t *dl, mmh_devbatch_t *out, int *more) {
    memset(out, 0, sizeof *out);
    out->arena = -1;
    *more = 0;
    if (dl->failed) return -1;
    uint64_t b_total_reads = 0, b_total_bytes = 0, b_proc_bytes = 0;
    while (!dl->finished) {
        ginfo_t g;
        int slot, rerun = 0;
        if (dl->rerun_slot >= 0) { slot = dl->rerun_slot; dl->rerun_slot = -1; rerun = 1; pthread_mutex_lock(&dl->mu); g = dl->ginfo[slot]; pthread_mutex_unlock(&dl->mu); }
        else slot = take_group(dl, &g);
        if (slot < 0) { dl->finished = 1; break; }
        if (g.err == 2 || (g.err && g.n_blocks == 0)) { dl->failed = 1; return -1; }   /* nothing readable where a block should be (or the device refused the launch) */
        int new_arena = 0;
        if (dl->cur_arena < 0) {
            dl->cur_arena = pick_arena(dl);
            if (dl->cur_arena < 0) { dl->failed = 1; dl->st.err = MM_INGEST_E_ARENA; return -1; }   /* the caller holds every arena */
            new_arena = 1;
            memset(&dl->cur, 0, sizeof dl->cur);
        }
        mm_ingest_result_t r;
        const int f = flatten_group(dl, slot, &g, new_arena, &r);
        if (f < 0) { dl->failed = 1; return -1; }
        if (f == 1) {
            if (new_arena || rerun) { dl->failed = 1; dl->st.err = MM_INGEST_E_ARENA; return -1; }   /* a single group bigger than an empty arena */
            dl->rerun_slot = slot;      /* the batch is closed without the group, which then begins the next one */
            break;
        }
        dl->first_group = 0;
        dl->groups++; dl->slow_blocks += r.n_slow_blocks;
        { float ms[4]; if (mm_ingest_times(dl->ing, slot, ms) == 0) for (int k = 0; k < 4; k++) dl->st.stage_ms[k] += ms[k]; }
        dl->cur = r;
        b_total_reads += r.total_reads; b_total_bytes += r.total_bytes; b_proc_bytes += r.processed_bytes;
        dl->st.total_reads += r.total_reads; dl->st.total_bytes += r.total_bytes; dl->st.processed_bytes += r.processed_bytes;
        dl->st.processed_reads += r.n_accepted;
        free_slot(dl, slot);
        if (r.done) { dl->finished = 1; break; }
        if (g.last) {
            dl->finished = 1;
            if (g.err || r.tail_len) { dl->failed = 1; dl->st.err = MM_INGEST_E_RECORD; }   /* a damaged block behind the group, or the file ends inside a record */
            break;
        }
        if (r.batch_bases >= dl->target_bases || r.seq_bytes > mm_ingest_arena_bytes(dl->ing) / 3 || r.cigar_bytes > mm_ingest_arena_bytes(dl->ing) / 6 || r.mm_bytes > mm_ingest_arena_bytes(dl->ing) / 6) break;   /* (two thirds of a pool: the next group still fits) */
    }
    if (dl->finished && !dl->quit) { pthread_mutex_lock(&dl->mu); dl->quit = 1; pthread_cond_broadcast(&dl->cv_free); pthread_mutex_unlock(&dl->mu); }
    int32_t n = 0;
    if (dl->cur_arena >= 0) {
        n = (int32_t)dl->cur.batch_reads;
        if (n > 0) {
            if (mm_ingest_arena_batch(dl->ing, dl->cur_arena, &dl->cur, &out->batch) != 0) { dl->failed = 1; return -1; }
            out->arena = dl->cur_arena; out->bases = dl->cur.batch_bases;
            if (mm_ingest_arena_names(dl->ing, dl->cur_arena, &out->names, &out->name_off) == 0) out->names_bytes = dl->cur.qname_bytes;
            dl->arena_busy[dl->cur_arena] = 1;
            dl->st.processed_bases += dl->cur.batch_bases;
        }
        dl->cur_arena = -1;
    }
    out->total_reads = b_total_reads; out->total_bytes = b_total_bytes; out->processed_bytes = b_proc_bytes;
    *more = !dl->finished || dl->rerun_slot >= 0;
    if (dl->failed) return -1;   /* (the reads in front of the damage are not processed: the run fails, as the host reader's does) */
    return n;
}

void mmh_devloader_release(mmh_devloader_t *dl, int arena) { if (dl && arena >= 0 && arena < DL_MAX_ARENAS) dl->arena_busy[arena] = 0; }
int mmh_devloader_fetch(mmh_devloader_t *dl, void *dst_host, const void *src_dev, size_t n) { return dl ? mm_ingest_copy_to_host(dl->ing, dst_host, src_dev, n) : -1; }
int mmh_devloader_codes(mmh_devloader_t *dl, const mm_batch_t *batch, char *codes, int max_codes) { return dl ? mm_ingest_batch_codes(dl->ing, batch, codes, max_codes) : -1; }
void *mmh_devloader_stream(mmh_devloader_t *dl) { return dl ? mm_ingest_stream(dl->ing) : NULL; }
const mmh_devloader_stats_t *mmh_devloader_stats(mmh_devloader_t *dl) {
    dl->st.groups = dl->groups; dl->st.slow_blocks = dl->slow_blocks; dl->st.patched_blocks = dl->patched_blocks;
    dl->st.wait_seconds = dl->t_wait; dl->st.stage_seconds = dl->t_stage;
    return &dl->st;
}

void mmh_devloader_close(mmh_devloader_t *dl) {
    if (!dl) return;
    if (dl->producer_started) {
        pthread_mutex_lock(&dl->mu);
        dl->quit = 1;
        pthread_cond_broadcast(&dl->cv_free);
        pthread_mutex_unlock(&dl->mu);
        pthread_join(dl->producer, NULL);
    }
    mm_ingest_destroy(dl->ing);
    if (dl->fd >= 0) close(dl->fd);
    pthread_mutex_destroy(&dl->mu); pthread_cond_destroy(&dl->cv_ready); pthread_cond_destroy(&dl->cv_free);
    free(dl);
}
