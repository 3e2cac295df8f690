/* tieorder.c -- the reference's order of the rows that tie on (contig, start).
 *
 * print_freq_output (reference src/mod.c:644-664) copies the core hash table slot by slot into an array and sorts it
 * with ks_introsort under a comparator that looks at contig and start only (cmp_key_fast, src/mod.c:59-87).  Rows of one
 * (contig, start) -- several codes, both strands under a `*` context, insertion offsets, haplotypes and their `*`
 * aggregate -- therefore come out in an order that is a function of (a) the slot every key has in the core table and
 * (b) what the unstable sort does to that array.  (a) depends on the order in which keys were first inserted
 * (merge_freq_maps, src/mod.c:743-774: reads in file order, inside a read the SLOT order of the read's own table, which was
 * filled in the order freq_view_single met the calls, update_freq_map src/mod.c:883-929: the key with the haplotype, then
 * the `-1` aggregate) and on the tables' growth history.  None of that carries information about the data; it is
 * reproduced here because "byte-identical" includes it.
 *
 * What is restated (from the behaviour of klib's khash / ksort as the reference instantiates them, not from their text):
 *   - the string key of a row (make_key, src/mod.c:428-439) and its X31 hash (khash.h kh_str_hash_func);
 *   - an insert-only open-addressing table with khash's probe sequence (i += ++step), growth rule (n_occupied >= 0.77 n
 *     at the START of a put: double) and its in-place "kick-out" rehash, whose slot assignment depends on the old slot
 *     order (khash.h kh_resize / kh_put);
 *   - introsort as ksort.h runs it: median of three with that file's pivot choice, partitions down to 16 elements, comb
 *     sort when the depth budget is spent, one insertion sort over everything at the end.
 * The COUNTS never come from here: they are the GPU's.  This file only decides in which order the GPU's rows are printed.
 * Input per batch: the calls of every read as `minimod view` rows with group ordinal and implicit flag (mm_freq_opts_t.view
 * == 2), from a second handle that sees the same batches. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mmhost.h"

/* ---------------------------------------------------------------- keys */
typedef struct { int32_t tid, pos; uint16_t ins; int16_t code, hp; uint8_t strand, pad; } tkey_t;   /* 16 bytes */

static inline int tkey_eq(const tkey_t *a, const tkey_t *b) {
    return a->tid == b->tid && a->pos == b->pos && a->ins == b->ins && a->code == b->code && a->hp == b->hp && a->strand == b->strand;
}
static inline uint64_t tkey_mix(const tkey_t *k) {   /* this file's own hash for its bookkeeping sets */
    uint64_t a = ((uint64_t)(uint32_t)k->tid << 32) | (uint32_t)k->pos;
    uint64_t b = ((uint64_t)k->ins << 40) | ((uint64_t)(uint16_t)k->code << 24) | ((uint64_t)(uint16_t)k->hp << 8) | k->strand;
    uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}

static size_t put_dec(char *dst, long long v) {
    char tmp[24];
    int n = 0, neg = v < 0;
    unsigned long long u = neg ? (unsigned long long)(-v) : (unsigned long long)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    size_t k = 0;
    if (neg) dst[k++] = '-';
    while (n) dst[k++] = tmp[--n];
    return k;
}

/* X31 over "contig \t pos \t strand \t code \t ins_offset \t haplotype" */
static uint32_t key_hash(const tkey_t *k, const char *contig, size_t clen, const char *code, size_t colen) {
    char stack[512], *buf = stack;
    size_t need = clen + colen + 64;
    if (need > sizeof stack) buf = (char *)malloc(need);
    size_t n = 0;
    memcpy(buf, contig, clen); n += clen;
    buf[n++] = '\t'; n += put_dec(buf + n, k->pos);
    buf[n++] = '\t'; buf[n++] = k->strand ? '-' : '+';
    buf[n++] = '\t'; memcpy(buf + n, code, colen); n += colen;
    buf[n++] = '\t'; n += put_dec(buf + n, k->ins);
    buf[n++] = '\t'; n += put_dec(buf + n, k->hp);
    uint32_t h = n ? (uint32_t)(unsigned char)buf[0] : 0;
    if (h) for (size_t i = 1; i < n; i++) h = (h << 5) - h + (uint32_t)(unsigned char)buf[i];   /* signed char in the reference; names are ASCII */
    if (buf != stack) free(buf);
    return h;
}

/* The same hash without the string: X31 is h = h * 31 + c over the characters, so a prefix's hash is carried on (the contig and its tab
 * once per read) and a fixed piece (a code and its tabs) is one multiply-add with 31^length and the piece's own sum. */
static inline uint32_t x31_str(uint32_t h, const char *p, size_t n) { for (size_t i = 0; i < n; i++) h = (h << 5) - h + (uint32_t)(unsigned char)p[i]; return h; }
static inline uint32_t x31_dec(uint32_t h, long long v) {
    char tmp[24];
    int n = 0;
    if (v < 0) { h = (h << 5) - h + (uint32_t)'-'; v = -v; }
    unsigned long long u = (unsigned long long)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) h = (h << 5) - h + (uint32_t)(unsigned char)tmp[--n];
    return h;
}
/* hash of the key whose "contig \t" prefix hashes to hc (x31_prefix): the rest of make_key's string */
static inline uint32_t key_hash_from(uint32_t hc, const tkey_t *k, const char *code, size_t colen) {
    uint32_t h = x31_dec(hc, k->pos);
    h = (h << 5) - h + (uint32_t)'\t'; h = (h << 5) - h + (uint32_t)(k->strand ? '-' : '+'); h = (h << 5) - h + (uint32_t)'\t';
    h = x31_str(h, code, colen);
    h = (h << 5) - h + (uint32_t)'\t'; h = x31_dec(h, k->ins);
    h = (h << 5) - h + (uint32_t)'\t'; h = x31_dec(h, k->hp);
    return h;
}
/* a fixed piece of the string as h -> h * mul + add (mul = 31^length, add = the piece's own sum): "\t+\tm\t" between the position and the
 * insertion offset is one multiply-add per call, and so is the usual end "0\t-1" */
typedef struct { uint32_t mul, add; } x31_piece_t;
static inline x31_piece_t x31_piece(const char *p, size_t n) {
    x31_piece_t q = {1u, 0u};
    for (size_t i = 0; i < n; i++) { q.add = (q.add << 5) - q.add + (uint32_t)(unsigned char)p[i]; q.mul *= 31u; }
    return q;
}
/* ... of "contig \t"; *plain = 0 when the string's first byte is NUL (khash's function stops there: the general routine does those) */
static inline uint32_t x31_prefix(const char *contig, size_t clen, int *plain) {
    *plain = clen > 0 && contig[0] != 0;
    if (!*plain) return 0;
    uint32_t h = (uint32_t)(unsigned char)contig[0];
    h = x31_str(h, contig + 1, clen - 1);
    return (h << 5) - h + (uint32_t)'\t';
}

/* ---------------------------------------------------------------- the table */
typedef struct {
    uint32_t n_buckets, size, n_occupied, upper;
    uint8_t *used;        /* 1 = holds a key */
    uint32_t *id;         /* key id per bucket */
    /* a read's own table lives in its thread's scratch (two `used` arrays taken in turn, one `id` array, scr_cap buckets each): its
     * nine growths from 4 to 1024 buckets were nine calloc / realloc / free a read, half of the replay's time a call */
    uint8_t *scr_used[2]; uint32_t *scr_id; uint32_t scr_cap; int scr_turn;
} ktab_t;

static void ktab_free(ktab_t *t) {
    if (t->scr_cap) { if (t->used != t->scr_used[0] && t->used != t->scr_used[1]) free(t->used); if (t->id != t->scr_id) free(t->id); }
    else { free(t->used); free(t->id); }
    memset(t, 0, sizeof *t);
}

static int ktab_grow(ktab_t *t, uint32_t want, const uint32_t *hash) {
    uint32_t nb = want - 1;
    nb |= nb >> 1; nb |= nb >> 2; nb |= nb >> 4; nb |= nb >> 8; nb |= nb >> 16; nb++;
    if (nb < 4) nb = 4;
    if (t->size >= (uint32_t)(nb * 0.77 + 0.5)) return 0;   /* requested size is too small: nothing happens */
    uint8_t *nused;
    uint32_t *nid;
    const int in_scratch = t->scr_cap && (t->used == NULL || t->used == t->scr_used[0] || t->used == t->scr_used[1]);
    if (in_scratch && nb <= t->scr_cap) {
        t->scr_turn ^= 1;
        nused = t->scr_used[t->scr_turn];
        memset(nused, 0, nb);
        nid = t->scr_id;
    } else {
        nused = (uint8_t *)calloc(nb, 1);
        if (in_scratch && nused) {   /* the table leaves the scratch: its ids move to memory of its own */
            nid = (uint32_t *)malloc(sizeof(uint32_t) * nb);
            if (nid && t->n_buckets) memcpy(nid, t->id, sizeof(uint32_t) * t->n_buckets);
        } else nid = (uint32_t *)realloc(t->id, sizeof(uint32_t) * nb);
        if (!nused || !nid) { free(nused); return -1; }
    }
    t->id = nid;
    /* the rehash works in place: an element taken out of old bucket j goes to its slot in the new table; if that slot (as a
     * bucket of the OLD table) still holds an element not yet moved, the two swap and the evicted one goes next */
    uint8_t *old = t->used;
    const uint32_t mask = nb - 1;
    const int big = t->n_buckets >= 8192;   /* (a read's own table is in the cache anyway) */
    for (uint32_t j = 0; j < t->n_buckets; j++) {
        /* (the walk is in bucket order, its targets are anywhere: the bucket sixteen ahead says where its element will go) */
        if (big && j + 16 < t->n_buckets && old[j + 16]) { const uint32_t pf = hash[t->id[j + 16]] & mask; __builtin_prefetch(&nused[pf], 1); __builtin_prefetch(&t->id[pf], 1); }
        if (!old[j]) continue;
        uint32_t key = t->id[j];
        old[j] = 0;
        for (;;) {
            uint32_t i = hash[key] & mask, step = 0;
            while (nused[i]) i = (i + (++step)) & mask;
            nused[i] = 1;
            if (i < t->n_buckets && old[i]) { uint32_t tmp = t->id[i]; t->id[i] = key; key = tmp; old[i] = 0; }
            else { t->id[i] = key; break; }
        }
    }
    if (!(t->scr_cap && (old == t->scr_used[0] || old == t->scr_used[1]))) free(old);
    t->used = nused; t->n_buckets = nb; t->n_occupied = t->size; t->upper = (uint32_t)(nb * 0.77 + 0.5);
    return 0;
}

/* put: returns 1 when the key was new.  update_freq_map looks the key up first (kh_get) and calls kh_put only for a key that is not there
 * (src/mod.c:885-893): a key met AGAIN at the growth bound does not grow the read's table -- the next new key does. */
static int ktab_put(ktab_t *t, uint32_t key, const uint32_t *hash, const tkey_t *keys) {
    const uint32_t hk = hash[key];
    if (t->n_buckets) {
        const uint32_t mask = t->n_buckets - 1;
        uint32_t i = hk & mask, step = 0;
        const uint32_t last = i;
        /* (an occupied slot's key is first told apart by its hash -- equal keys have equal hashes -- and only then fetched) */
        while (t->used[i]) {
            if (hash[t->id[i]] == hk && tkey_eq(&keys[t->id[i]], &keys[key])) return 0;
            i = (i + (++step)) & mask;
            if (i == last) return -1;   /* cannot happen below the load bound */
        }
        if (t->n_occupied < t->upper) { t->used[i] = 1; t->id[i] = key; t->size++; t->n_occupied++; return 1; }
    }
    if (ktab_grow(t, t->n_buckets > (t->size << 1) ? t->n_buckets - 1 : t->n_buckets + 1, hash)) return -1;
    const uint32_t mask = t->n_buckets - 1;
    uint32_t i = hk & mask, step = 0;
    const uint32_t last = i;
    while (t->used[i]) {
        i = (i + (++step)) & mask;
        if (i == last) return -1;
    }
    t->used[i] = 1; t->id[i] = key; t->size++; t->n_occupied++;
    return 1;
}

/* put of a key that is known not to be in the table (the core table is filled from a sequence of distinct keys): the same walk, without
 * fetching the key of every occupied slot on the way to compare it */
static int ktab_put_new(ktab_t *t, uint32_t key, const uint32_t *hash) {
    if (t->n_occupied >= t->upper) {
        if (ktab_grow(t, t->n_buckets > (t->size << 1) ? t->n_buckets - 1 : t->n_buckets + 1, hash)) return -1;
    }
    const uint32_t mask = t->n_buckets - 1;
    uint32_t i = hash[key] & mask, step = 0;
    const uint32_t last = i;
    while (t->used[i]) {
        i = (i + (++step)) & mask;
        if (i == last) return -1;
    }
    t->used[i] = 1; t->id[i] = key; t->size++; t->n_occupied++;
    return 1;
}

/* ---------------------------------------------------------------- introsort, as ksort.h runs it */
/* An element carries what the comparator looks at -- (rank of the contig's name) << 32 + start -- beside the key's number: the sort
 * walks its array, it does not chase 27 M pointers into the key table (round 3 did, a cache miss a comparison). */
typedef struct { int64_t key; uint32_t id, pad; } sel_t;
#define key_lt(a, b) ((a).key < (b).key)   /* cmp_key_fast(a, b) < 0 */
static void ins_sort(sel_t *s, sel_t *t) {
    for (sel_t *i = s + 1; i < t; ++i)
        for (sel_t *j = i; j > s && key_lt(*j, *(j - 1)); --j) { sel_t tmp = *j; *j = *(j - 1); *(j - 1) = tmp; }
}
static void comb_sort(size_t n, sel_t *a) {
    const double shrink = 1.2473309501039786540366528676643;
    int swapped;
    size_t gap = n;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        swapped = 0;
        for (sel_t *i = a; i < a + n - gap; ++i) {
            sel_t *j = i + gap;
            if (key_lt(*j, *i)) { sel_t tmp = *i; *i = *j; *j = tmp; swapped = 1; }
        }
    } while (swapped || gap > 2);
    if (gap != 1) ins_sort(a, a + n);
}
typedef struct { sel_t *left, *right; int depth; } sstack_t;
static int intro_sort(size_t n, sel_t *a) {
    if (n < 1) return 0;
    if (n == 2) { if (key_lt(a[1], a[0])) { sel_t tmp = a[0]; a[0] = a[1]; a[1] = tmp; } return 0; }
    int d;
    for (d = 2; (1ul << d) < n; ++d) {}
    sstack_t *stack = (sstack_t *)malloc(sizeof(sstack_t) * (sizeof(size_t) * (size_t)d + 2)), *top = stack;
    if (!stack) return -1;
    sel_t *s = a, *t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { comb_sort((size_t)(t - s) + 1, s); t = s; continue; }
            sel_t *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
            if (key_lt(*k, *i)) { if (key_lt(*k, *j)) k = j; }
            else k = key_lt(*j, *i) ? i : j;
            const sel_t rp = *k;
            if (k != t) { sel_t tmp = *k; *k = *t; *t = tmp; }
            for (;;) {
                do ++i; while (key_lt(*i, rp));
                do --j; while (i <= j && key_lt(rp, *j));
                if (j <= i) break;
                sel_t tmp = *i; *i = *j; *j = tmp;
            }
            { sel_t tmp = *i; *i = *t; *t = tmp; }
            if (i - s > t - i) {
                if (i - s > 16) { top->left = s; top->right = i - 1; top->depth = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { top->left = i + 1; top->right = t; top->depth = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == stack) { free(stack); ins_sort(a, a + n); return 0; }
            --top; s = top->left; t = top->right; d = top->depth;
        }
    }
}

/* The core table and the sort by themselves, from the reference hash and the comparator's key of every distinct key in first-insertion
 * order: the checker of the device-side replay (csrc/tie_kernels.hip.h, which computes the same two permutations with parallel
 * kernels) and what a `--devices` parent runs.  put_after_last: some kh_put followed the last NEW key's (merge_freq_maps offers every
 * key of every read, src/mod.c:756: a put of a key that is already there still grows a table that has reached its bound,
 * khash.h kh_put).  slot_order (may be NULL): ids in the core table's slot order; final: ids in the order print_freq_output prints. */
int mmh_tie_order_plain(const uint32_t *hash, const int64_t *sortkey, int64_t n, int put_after_last, uint32_t *slot_order, uint32_t *final) {
    if (n <= 0) return 0;
    if (n >= 0xFFFFFFF0ll) return -1;
    ktab_t core;
    memset(&core, 0, sizeof core);
    for (int64_t i = 0; i < n; i++) {
        if (i + 12 < n && core.n_buckets) { const uint32_t pf = hash[i + 12] & (core.n_buckets - 1); __builtin_prefetch(&core.used[pf], 1); __builtin_prefetch(&core.id[pf], 1); }
        if (ktab_put_new(&core, (uint32_t)i, hash) != 1) { ktab_free(&core); return -1; }
    }
    if (put_after_last && core.n_occupied >= core.upper && ktab_grow(&core, core.n_buckets + 1, hash)) { ktab_free(&core); return -1; }
    sel_t *arr = (sel_t *)malloc(sizeof(sel_t) * (size_t)n);
    if (!arr) { ktab_free(&core); return -1; }
    size_t w = 0;
    for (uint32_t s = 0; s < core.n_buckets; s++)
        if (core.used[s]) { arr[w].key = sortkey[core.id[s]]; arr[w].id = core.id[s]; arr[w].pad = 0; if (slot_order) slot_order[w] = core.id[s]; w++; }
    ktab_free(&core);
    if (intro_sort(w, arr)) { free(arr); return -1; }
    for (size_t i = 0; i < w; i++) final[i] = arr[i].id;
    free(arr);
    return 0;
}

/* ---------------------------------------------------------------- the replay */
/* Round 4: the first-insertion ORDER of the keys is kept as a stamp per key -- (serial number of the read in the file) << 24 | place
 * of the key among the read's own -- in 256 tables of their own lock each, so that the reads of a batch are replayed AND entered by
 * the worker pool side by side (a key met again keeps the smaller stamp); the sequence itself is made once, at the end, by a
 * radix sort on the stamps.  Round 3 entered every key of every read into one table on one thread: 60 M probes for 3 Gbases of
 * HiFi reads with two codes, 2.2 s of a 2.9 s run. */
#define TS_SHARDS 256
typedef struct { tkey_t k; uint64_t stamp; uint32_t hash, pad; } trec_t;   /* a key on its way into its table (32 bytes) */
typedef struct {
    pthread_mutex_t mu;
    tkey_t *keys; uint32_t *hash; uint64_t *stamp; size_t n, cap;
    uint32_t *slot; size_t slot_cap;   /* open addressing on tkey_mix's low bits; 0xFFFFFFFF = free */
    /* the batch's keys of this table, as the threads that replayed the reads left them (one lock per 16 reads and table, not per key);
     * ONE thread enters them afterwards (shard_merge): a table of a 256th of the keys stays in that core's cache while it is worked on,
     * where 16 threads entering keys into all 256 tables at once missed the cache on every probe and passed the locks' lines around */
    trec_t *stg; size_t stg_n, stg_cap;
    char pad[64];
} tshard_t;
struct mmh_tie {
    const mm_bam_hdr_t *hdr;
    int insertions, haplotypes;
    int32_t *rank;                 /* contig tid -> rank of its name in strcmp order */
    tshard_t *shard;               /* [TS_SHARDS] */
    uint64_t reads_seen;           /* serial number of the next batch's first read */
    uint64_t last_put;             /* the largest stamp any put carried, new key or not (atomic max): does a put follow the last new key's? */
    uint64_t top_stamp;            /* the last new key's stamp (seq_build) */
    size_t n_keys;                 /* keys in all (atomic) */
    /* first-insertion sequence of the core table, made from the stamps when somebody asks for it (seq_build) */
    tkey_t *keys; uint32_t *hash; size_t n, cap;
    int seq_valid;
    int failed;                    /* memory ran out, or more keys than max_keys */
    size_t max_keys;               /* the sequence is not kept beyond this many keys (host memory: 28 bytes a key + the tables) */
};

mmh_tie_t *mmh_tie_create(const mm_bam_hdr_t *hdr, int insertions, int haplotypes) {
    mmh_tie_t *t = (mmh_tie_t *)calloc(1, sizeof(*t));
    if (!t) return NULL;
    t->hdr = hdr; t->insertions = insertions; t->haplotypes = haplotypes;
    const int n = hdr->n_targets;
    t->rank = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) idx[i] = i;
    for (int a = 1; a < n; a++) {
        int x = idx[a], b = a - 1;
        while (b >= 0 && strcmp(hdr->target_name[idx[b]], hdr->target_name[x]) > 0) { idx[b + 1] = idx[b]; b--; }
        idx[b + 1] = x;
    }
    for (int r = 0, cur = -1; r < n; r++) {   /* equal names share a rank: the comparator cannot tell them apart */
        if (r == 0 || strcmp(hdr->target_name[idx[r]], hdr->target_name[idx[r - 1]]) != 0) cur = r;
        t->rank[idx[r]] = cur;
    }
    free(idx);
    t->max_keys = (size_t)128 << 20;   /* ~3.5 GB of host memory; MINIMOD_REPLAY_MAX_KEYS sets another bound */
    { const char *e = getenv("MINIMOD_REPLAY_MAX_KEYS"); if (e && atoll(e) > 0) t->max_keys = (size_t)atoll(e); }
    t->shard = (tshard_t *)calloc(TS_SHARDS, sizeof(tshard_t));
    if (!t->shard) { t->failed = 1; return t; }
    for (int i = 0; i < TS_SHARDS; i++) pthread_mutex_init(&t->shard[i].mu, NULL);
    return t;
}

void mmh_tie_destroy(mmh_tie_t *t) {
    if (!t) return;
    if (t->shard) for (int i = 0; i < TS_SHARDS; i++) { tshard_t *s = &t->shard[i]; pthread_mutex_destroy(&s->mu); free(s->keys); free(s->hash); free(s->stamp); free(s->slot); free(s->stg); }
    free(t->shard); free(t->rank); free(t->keys); free(t->hash); free(t);
}

/* enter key k (reference hash h) with its stamp; a key already there keeps the smaller one.  0 or -1.  stamp_add_locked: the caller
 * holds the table's lock or is the only thread at it */
static int stamp_add_locked(mmh_tie_t *t, tshard_t *s, uint64_t mix, const tkey_t *k, uint32_t h, uint64_t stamp);
static int stamp_add(mmh_tie_t *t, const tkey_t *k, uint32_t h, uint64_t stamp) {
    const uint64_t mix = tkey_mix(k);
    tshard_t *s = &t->shard[(mix >> 56) & (TS_SHARDS - 1)];
    pthread_mutex_lock(&s->mu);
    const int rc = stamp_add_locked(t, s, mix, k, h, stamp);
    pthread_mutex_unlock(&s->mu);
    return rc;
}
static int stamp_add_locked(mmh_tie_t *t, tshard_t *s, uint64_t mix, const tkey_t *k, uint32_t h, uint64_t stamp) {
    int rc = 0;
    if ((s->n + 1) * 10 > s->slot_cap * 6) {   /* grow the slots */
        const size_t nc = s->slot_cap ? s->slot_cap * 2 : 1024;
        uint32_t *ns = (uint32_t *)malloc(sizeof(uint32_t) * nc);
        if (!ns) return -1;
        memset(ns, 0xFF, sizeof(uint32_t) * nc);
        for (size_t i = 0; i < s->n; i++) {
            size_t q = (size_t)tkey_mix(&s->keys[i]) & (nc - 1);
            while (ns[q] != 0xFFFFFFFFu) q = (q + 1) & (nc - 1);
            ns[q] = (uint32_t)i;
        }
        free(s->slot); s->slot = ns; s->slot_cap = nc;
    }
    size_t q = (size_t)mix & (s->slot_cap - 1);
    while (s->slot[q] != 0xFFFFFFFFu) {
        const uint32_t i = s->slot[q];
        if (tkey_eq(&s->keys[i], k)) { if (stamp < s->stamp[i]) s->stamp[i] = stamp; return 0; }
        q = (q + 1) & (s->slot_cap - 1);
    }
    if (s->n == s->cap) {
        const size_t nc = s->cap ? s->cap * 2 : 512;
        tkey_t *nk = (tkey_t *)realloc(s->keys, sizeof(tkey_t) * nc);
        if (nk) s->keys = nk;
        uint32_t *nh = (uint32_t *)realloc(s->hash, sizeof(uint32_t) * nc);
        if (nh) s->hash = nh;
        uint64_t *nst = (uint64_t *)realloc(s->stamp, sizeof(uint64_t) * nc);
        if (nst) s->stamp = nst;
        if (!nk || !nh || !nst) rc = -1; else s->cap = nc;
    }
    if (!rc) {
        const size_t all = __atomic_add_fetch(&t->n_keys, 1, __ATOMIC_RELAXED);
        if (all >= 0xFFFFFFF0u || all > t->max_keys) rc = -1;
        else { s->keys[s->n] = *k; s->hash[s->n] = h; s->stamp[s->n] = stamp; s->slot[q] = (uint32_t)s->n; s->n++; }
    }
    if (!rc && t->seq_valid) t->seq_valid = 0;   /* (tested first: a store from every thread on every call kept the handle's cache line travelling) */
    return rc;
}

/* the first-insertion sequence from the stamps: every table's keys, ordered by stamp (LSD radix sort, 11 bits a pass) */
typedef struct { uint64_t stamp; uint32_t shard, idx; } ent_t;
#define SQ_PARTS 64   /* pieces of the array a radix pass is counted and scattered in (any number of threads take them) */
typedef struct { mmh_tie_t *t; ent_t *a, *b; size_t n; const size_t *off; int shift; size_t *cnt; /* [SQ_PARTS][2048] */ uint64_t top[TS_SHARDS]; } seqjob_t;
static void seq_gather(void *arg, int64_t lo, int64_t hi) {   /* a table's stamps behind those of the tables in front of it */
    seqjob_t *q = (seqjob_t *)arg;
    for (int64_t sh = lo; sh < hi; sh++) {
        const tshard_t *s = &q->t->shard[sh];
        ent_t *a = q->a + q->off[sh];
        uint64_t top = 0;
        for (size_t i = 0; i < s->n; i++) { a[i].stamp = s->stamp[i]; a[i].shard = (uint32_t)sh; a[i].idx = (uint32_t)i; if (s->stamp[i] > top) top = s->stamp[i]; }
        q->top[sh] = top;
    }
}
static void seq_count(void *arg, int64_t lo, int64_t hi) {
    seqjob_t *q = (seqjob_t *)arg;
    for (int64_t p = lo; p < hi; p++) {
        size_t *c = q->cnt + (size_t)p * 2048;
        memset(c, 0, sizeof(size_t) * 2048);
        const size_t i0 = q->n * (size_t)p / SQ_PARTS, i1 = q->n * (size_t)(p + 1) / SQ_PARTS;
        for (size_t i = i0; i < i1; i++) c[(q->a[i].stamp >> q->shift) & 2047]++;
    }
}
static void seq_scatter(void *arg, int64_t lo, int64_t hi) {   /* (cnt holds every piece's first place per digit: the pass stays stable) */
    seqjob_t *q = (seqjob_t *)arg;
    for (int64_t p = lo; p < hi; p++) {
        size_t *c = q->cnt + (size_t)p * 2048;
        const size_t i0 = q->n * (size_t)p / SQ_PARTS, i1 = q->n * (size_t)(p + 1) / SQ_PARTS;
        for (size_t i = i0; i < i1; i++) q->b[c[(q->a[i].stamp >> q->shift) & 2047]++] = q->a[i];
    }
}
static void seq_emit(void *arg, int64_t lo, int64_t hi) {
    seqjob_t *q = (seqjob_t *)arg;
    mmh_tie_t *t = q->t;
    for (int64_t i = lo; i < hi; i++) { t->keys[i] = t->shard[q->a[i].shard].keys[q->a[i].idx]; t->hash[i] = t->shard[q->a[i].shard].hash[q->a[i].idx]; }
}
/* the first-insertion sequence from the stamps: every table's keys, ordered by stamp (LSD radix sort, 11 bits a pass; `pool` may be NULL) */
static int seq_build_mt(mmh_tie_t *t, mm_pool_t *pool) {
    if (t->seq_valid) return 0;
    size_t n = 0, off[TS_SHARDS + 1];
    for (int i = 0; i < TS_SHARDS; i++) { off[i] = n; n += t->shard[i].n; }
    off[TS_SHARDS] = n;
    free(t->keys); free(t->hash); t->keys = NULL; t->hash = NULL; t->n = t->cap = 0;
    ent_t *a = (ent_t *)malloc(sizeof(ent_t) * (n ? n : 1)), *b = (ent_t *)malloc(sizeof(ent_t) * (n ? n : 1));
    t->keys = (tkey_t *)malloc(sizeof(tkey_t) * (n ? n : 1)); t->hash = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    seqjob_t *q = (seqjob_t *)calloc(1, sizeof(seqjob_t));
    size_t *cnt = (size_t *)malloc(sizeof(size_t) * 2048 * SQ_PARTS);
    if (!a || !b || !t->keys || !t->hash || !q || !cnt) { free(a); free(b); free(q); free(cnt); return -1; }
    q->t = t; q->a = a; q->b = b; q->n = n; q->off = off; q->cnt = cnt;
    mm_pool_for(pool, TS_SHARDS, 4, seq_gather, q);
    uint64_t top = 0;
    for (int i = 0; i < TS_SHARDS; i++) if (q->top[i] > top) top = q->top[i];
    t->top_stamp = top;
    for (int shift = 0; shift < 64 && (top >> shift) != 0; shift += 11) {
        q->shift = shift;
        mm_pool_for(pool, SQ_PARTS, 1, seq_count, q);
        size_t run = 0;
        for (int d = 0; d < 2048; d++)
            for (int p = 0; p < SQ_PARTS; p++) { const size_t c = cnt[(size_t)p * 2048 + d]; cnt[(size_t)p * 2048 + d] = run; run += c; }
        mm_pool_for(pool, SQ_PARTS, 1, seq_scatter, q);
        ent_t *tmp = q->a; q->a = q->b; q->b = tmp;
    }
    mm_pool_for(pool, (int64_t)n, 65536, seq_emit, q);
    free(q->a); free(q->b); free(q); free(cnt);
    t->n = t->cap = n;
    t->seq_valid = 1;
    return 0;
}
static int seq_build(mmh_tie_t *t) { return seq_build_mt(t, NULL); }

/* per read: its keys in the slot order of its own table */
typedef struct {
    mmh_tie_t *t;
    const mm_batch_t *batch;
    const mm_view_row_t *rows;
    const int64_t *first;          /* first row of every read (n_reads + 1 entries) */
    const uint8_t *const *klass;   /* per code: the 256-entry threshold class table of the mod it counts for */
    const char *const *codes; int n_codes;
    uint64_t serial0;              /* serial number of the batch's first read */
    int failed;
} readjob_t;

typedef struct { uint32_t gord, implicit, fq, m; uint32_t row; } callord_t;
static int callord_cmp(const void *a, const void *b) {
    const callord_t *x = (const callord_t *)a, *y = (const callord_t *)b;
    if (x->gord != y->gord) return x->gord < y->gord ? -1 : 1;
    if (x->implicit != y->implicit) return x->implicit < y->implicit ? -1 : 1;
    if (x->fq != y->fq) return x->fq < y->fq ? -1 : 1;
    if (x->m != y->m) return x->m < y->m ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}

/* The calls of a read into the order the reference met them.  They arrive by reference position, i.e. along the read (backwards for a
 * reverse read) with the groups mixed: a stable distribution by (group, implicit) and, where a group's run then falls instead of rising,
 * its reversal is the whole sort -- qsort with a comparison callback was a third of the replay's time.  Anything else (several code
 * letters a group, positions that neither rise nor fall) is sorted the general way. */
static void sort_calls(callord_t *ord, size_t n, int multi, callord_t **tmp, size_t *tmp_cap) {
    uint32_t gmax = 0;
    for (size_t i = 0; i < n; i++) if (ord[i].gord > gmax) gmax = ord[i].gord;
    if (multi || gmax >= 32 || n < 2) { qsort(ord, n, sizeof(callord_t), callord_cmp); return; }
    if (n > *tmp_cap) { free(*tmp); *tmp_cap = n * 2; *tmp = (callord_t *)malloc(sizeof(callord_t) * *tmp_cap); if (!*tmp) { *tmp_cap = 0; qsort(ord, n, sizeof(callord_t), callord_cmp); return; } }
    size_t cnt[65];
    memset(cnt, 0, sizeof cnt);
    for (size_t i = 0; i < n; i++) cnt[(ord[i].gord << 1 | ord[i].implicit) + 1]++;
    for (int b = 0; b < 64; b++) cnt[b + 1] += cnt[b];
    size_t at[64];
    memcpy(at, cnt, sizeof at);
    callord_t *t = *tmp;
    for (size_t i = 0; i < n; i++) t[at[ord[i].gord << 1 | ord[i].implicit]++] = ord[i];
    for (int b = 0; b < 64; b++) {
        callord_t *p = t + cnt[b];
        const size_t c = cnt[b + 1] - cnt[b];
        if (c < 2) continue;
        int up = 1, down = 1;
        for (size_t i = 1; i < c; i++) { if (p[i].fq <= p[i - 1].fq) up = 0; if (p[i].fq >= p[i - 1].fq) down = 0; }
        if (up) continue;
        if (down) { for (size_t i = 0, k = c - 1; i < k; i++, k--) { callord_t x = p[i]; p[i] = p[k]; p[k] = x; } continue; }
        qsort(p, c, sizeof(callord_t), callord_cmp);
    }
    memcpy(ord, t, sizeof(callord_t) * n);
}

/* letter index of device code `ci` inside MM group number `gord` of the read (0 for single-code groups) */
static uint32_t letter_index(const char *mm, uint32_t mm_len, uint32_t gord, const char *code) {
    uint32_t p = 0, g = 0;
    while (p < mm_len && g < gord) { while (p < mm_len && mm[p] != ';') p++; p++; g++; }
    if (p + 2 >= mm_len) return 0;
    uint32_t s = p + 2, e = s;
    while (e < mm_len && mm[e] != ',' && mm[e] != ';' && mm[e] != '?' && mm[e] != '.') e++;
    if (e <= s || (mm[s] >= '0' && mm[s] <= '9')) return 0;
    const size_t cl = strlen(code);
    for (uint32_t m = 0; s + m < e; m++)   /* the code of letter m is the string from m on (src/mod.c:1151) */
        if ((size_t)(e - s - m) == cl && memcmp(mm + s + m, code, cl) == 0) return m;
    return 0;
}

/* does any group of the read carry more than one code letter ("C+hm?")?  Then the letters of one token tie on everything
 * but their place in the header */
static int has_multi_letter_group(const char *mm, uint32_t mm_len) {
    uint32_t p = 0;
    while (p + 2 < mm_len) {
        uint32_t s = p + 2, e = s;
        while (e < mm_len && mm[e] != ',' && mm[e] != ';' && mm[e] != '?' && mm[e] != '.') e++;
        if (e - s > 1 && !(mm[s] >= '0' && mm[s] <= '9')) return 1;
        while (p < mm_len && mm[p] != ';') p++;
        p++;
    }
    return 0;
}

/* a worker thread's scratch, kept from one range of reads to the next (a range's 120 KB of records came from mmap and went back
 * with munmap, every range, on every thread: page faults and TLB shoot-downs were a third of the replay's time) */
typedef struct { callord_t *ord, *tmp_ord; size_t ord_cap, tmp_cap; trec_t *recs, *by; size_t recs_cap, by_cap; uint8_t *sh; size_t sh_cap; tkey_t *keys; uint32_t *hash; size_t keys_cap;
                 uint8_t *tab_used[2]; uint32_t *tab_id; } tscratch_t;
#define TS_TAB_CAP 8192u   /* buckets of a read's own table that the scratch holds (a bigger table goes to the heap, ktab_grow) */
static __thread tscratch_t g_ts;
static void read_range(void *arg, int64_t lo, int64_t hi) {
    readjob_t *j = (readjob_t *)arg;
    mmh_tie_t *t = j->t;
    tscratch_t *ts = &g_ts;
    callord_t *ord = ts->ord, *tmp_ord = ts->tmp_ord; size_t ord_cap = ts->ord_cap, tmp_cap = ts->tmp_cap;
    x31_piece_t mid[2][64];          /* "\t<strand>\t<code>\t" of the codes met in this range */
    uint8_t mid_have[2][64];
    memset(mid_have, 0, sizeof mid_have);
    const x31_piece_t tail0 = x31_piece("0\t-1", 4);
    trec_t *recs = ts->recs; size_t n_recs = 0, recs_cap = ts->recs_cap;   /* the keys of this range's reads with their stamps */
    for (int64_t r = lo; r < hi; r++) {
        const int64_t a = j->first[r], b = j->first[r + 1];
        if (a == b) continue;
        const mm_read_t *rd = &j->batch->reads[r];
        const char *mm = (const char *)j->batch->mm + rd->mm_off;
        if ((size_t)(b - a) > ord_cap) { ord_cap = (size_t)(b - a) * 2; free(ord); ord = (callord_t *)malloc(sizeof(callord_t) * ord_cap); ts->ord = ord; ts->ord_cap = ord ? ord_cap : 0; if (!ord) { j->failed = 1; return; } }
        size_t n = 0;
        const int multi = has_multi_letter_group(mm, rd->mm_len);
        for (int64_t i = a; i < b; i++) {
            const mm_view_row_t *w = &j->rows[i];
            const uint32_t implicit = w->read_pos >> 31, gord = w->read >> 21;
            if (w->code >= j->n_codes) { j->failed = 1; continue; }
            if (!implicit && j->klass[w->code][w->prob] == 0) continue;   /* ambiguous: never reaches the table (src/mod.c:1180-1191) */
            ord[n].gord = gord; ord[n].implicit = implicit; ord[n].fq = w->read_pos & 0x7FFFFFFFu; ord[n].m = 0; ord[n].row = (uint32_t)(i - a);
            n++;
        }
        if (n == 0) continue;
        /* rows arrive by position; the reference met them group by group, listed calls before implicit ones, along the read */
        if (multi) for (size_t i = 0; i < n; i++) ord[i].m = letter_index(mm, rd->mm_len, ord[i].gord, j->codes[j->rows[a + ord[i].row].code]);
        sort_calls(ord, n, multi, &tmp_ord, &tmp_cap);
        const size_t per = t->haplotypes ? 2 : 1;
        if (n * per > ts->keys_cap) {
            free(ts->keys); free(ts->hash);
            ts->keys_cap = n * per * 2;
            ts->keys = (tkey_t *)malloc(sizeof(tkey_t) * ts->keys_cap); ts->hash = (uint32_t *)malloc(sizeof(uint32_t) * ts->keys_cap);
            if (!ts->keys || !ts->hash) { free(ts->keys); free(ts->hash); ts->keys = NULL; ts->hash = NULL; ts->keys_cap = 0; j->failed = 1; continue; }
        }
        tkey_t *keys = ts->keys;
        uint32_t *hash = ts->hash;
        ktab_t tab;
        memset(&tab, 0, sizeof tab);
        if (!ts->tab_id) {
            ts->tab_used[0] = (uint8_t *)malloc(TS_TAB_CAP); ts->tab_used[1] = (uint8_t *)malloc(TS_TAB_CAP); ts->tab_id = (uint32_t *)malloc(sizeof(uint32_t) * TS_TAB_CAP);
            if (!ts->tab_used[0] || !ts->tab_used[1] || !ts->tab_id) { free(ts->tab_used[0]); free(ts->tab_used[1]); free(ts->tab_id); ts->tab_used[0] = ts->tab_used[1] = NULL; ts->tab_id = NULL; }
        }
        if (ts->tab_id) { tab.scr_used[0] = ts->tab_used[0]; tab.scr_used[1] = ts->tab_used[1]; tab.scr_id = ts->tab_id; tab.scr_cap = TS_TAB_CAP; }
        const char *contig = (rd->tid >= 0 && rd->tid < t->hdr->n_targets) ? t->hdr->target_name[rd->tid] : "*";
        const size_t clen = strlen(contig);
        int plain = 0;
        const uint32_t hc = x31_prefix(contig, clen, &plain);
        const int strand = (rd->flag & 0x10) ? 1 : 0;
        size_t nk = 0;
        for (size_t i = 0; i < n; i++) {
            const mm_view_row_t *w = &j->rows[a + ord[i].row];
            const char *code = j->codes[w->code];
            const size_t colen = strlen(code);
            for (size_t v = 0; v < per; v++) {   /* update_freq_map: the key with the haplotype, then the aggregate */
                tkey_t k;
                memset(&k, 0, sizeof k);
                k.tid = rd->tid; k.pos = w->pos; k.ins = t->insertions ? w->ins_offset : 0; k.code = (int16_t)w->code;
                k.strand = (rd->flag & 0x10) ? 1 : 0;
                k.hp = (int16_t)(t->haplotypes ? (v == 0 ? (int)rd->hp : -1) : -1);
                keys[nk] = k;
                if (plain && w->code < 64) {
                    if (!mid_have[strand][w->code]) {
                        char tmpc[MM_CODE_LEN + 4];
                        size_t tl = 0;
                        tmpc[tl++] = '\t'; tmpc[tl++] = strand ? '-' : '+'; tmpc[tl++] = '\t';
                        memcpy(tmpc + tl, code, colen < MM_CODE_LEN ? colen : MM_CODE_LEN - 1); tl += colen < MM_CODE_LEN ? colen : MM_CODE_LEN - 1;
                        tmpc[tl++] = '\t';
                        mid[strand][w->code] = x31_piece(tmpc, tl); mid_have[strand][w->code] = 1;
                    }
                    uint32_t h = x31_dec(hc, k.pos);
                    h = h * mid[strand][w->code].mul + mid[strand][w->code].add;
                    if (k.ins == 0 && k.hp == -1) h = h * tail0.mul + tail0.add;
                    else { h = x31_dec(h, k.ins); h = (h << 5) - h + (uint32_t)'\t'; h = x31_dec(h, k.hp); }
                    hash[nk] = h;
                } else hash[nk] = plain ? key_hash_from(hc, &k, code, colen) : key_hash(&k, contig, clen, code, colen);
                int pr = ktab_put(&tab, (uint32_t)nk, hash, keys);
                if (pr < 0) j->failed = 1;
                if (pr == 1) nk++;
            }
        }
        /* slot order of the read's table = the order merge_freq_maps offers its keys to the core table (src/mod.c:743-774: reads in
         * file order): the key in the w-th occupied slot of read number `serial` gets the stamp serial << 24 | w */
        (void)nk;
        size_t w2 = 0;
        if (n_recs + tab.size > recs_cap) {
            recs_cap = (n_recs + tab.size) * 2 + 1024;
            trec_t *nr = (trec_t *)realloc(recs, sizeof(trec_t) * recs_cap);
            if (!nr) { j->failed = 1; ktab_free(&tab); continue; }
            recs = nr; ts->recs = recs; ts->recs_cap = recs_cap;
        }
        for (uint32_t s = 0; s < tab.n_buckets; s++)
            if (tab.used[s]) {
                trec_t *q = &recs[n_recs++];
                q->k = keys[tab.id[s]]; q->hash = hash[tab.id[s]]; q->pad = 0; q->stamp = ((j->serial0 + (uint64_t)r) << 24) | (uint64_t)(w2 & 0xFFFFFFu);
                w2++;
            }
        if (w2) {   /* the read's last put (merge_freq_maps offers every key of the read to the core table, new there or not) */
            const uint64_t lp = ((j->serial0 + (uint64_t)r) << 24) | (uint64_t)((w2 - 1) & 0xFFFFFFu);
            uint64_t cur = __atomic_load_n(&t->last_put, __ATOMIC_RELAXED);
            while (lp > cur && !__atomic_compare_exchange_n(&t->last_put, &cur, lp, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
        ktab_free(&tab);
    }
    ts->tmp_ord = tmp_ord; ts->tmp_cap = tmp_cap;
    /* the range's keys go to their tables' waiting lists, table by table (a counting sort on the table's number first) */
    if (n_recs) {
        uint32_t cnt[TS_SHARDS + 1];
        memset(cnt, 0, sizeof cnt);
        if (n_recs > ts->sh_cap) { free(ts->sh); ts->sh_cap = n_recs * 2; ts->sh = (uint8_t *)malloc(ts->sh_cap); if (!ts->sh) ts->sh_cap = 0; }
        if (n_recs > ts->by_cap) { free(ts->by); ts->by_cap = n_recs * 2; ts->by = (trec_t *)malloc(sizeof(trec_t) * ts->by_cap); if (!ts->by) ts->by_cap = 0; }
        uint8_t *sh = ts->sh;
        trec_t *by = ts->by;
        if (!sh || !by) j->failed = 1;
        else {
            for (size_t i = 0; i < n_recs; i++) { sh[i] = (uint8_t)((tkey_mix(&recs[i].k) >> 56) & (TS_SHARDS - 1)); cnt[sh[i] + 1]++; }
            for (int i = 0; i < TS_SHARDS; i++) cnt[i + 1] += cnt[i];
            uint32_t at[TS_SHARDS];
            memcpy(at, cnt, sizeof at);
            for (size_t i = 0; i < n_recs; i++) by[at[sh[i]]++] = recs[i];
            for (int i = 0; i < TS_SHARDS; i++) {
                const size_t c = cnt[i + 1] - cnt[i];
                if (!c) continue;
                tshard_t *s = &t->shard[i];
                pthread_mutex_lock(&s->mu);
                if (s->stg_n + c > s->stg_cap) {
                    const size_t nc = (s->stg_n + c) * 2 + 256;
                    trec_t *ns = (trec_t *)realloc(s->stg, sizeof(trec_t) * nc);
                    if (ns) { s->stg = ns; s->stg_cap = nc; } else j->failed = 1;
                }
                if (s->stg_n + c <= s->stg_cap) { memcpy(s->stg + s->stg_n, by + cnt[i], sizeof(trec_t) * c); s->stg_n += c; }
                pthread_mutex_unlock(&s->mu);
            }
        }
    }
}

/* one table's waiting keys entered by one thread (every key keeps its smallest stamp: the order they arrive in says nothing) */
static void shard_merge(void *arg, int64_t lo, int64_t hi) {
    readjob_t *j = (readjob_t *)arg;
    for (int64_t i = lo; i < hi; i++) {
        tshard_t *s = &j->t->shard[i];
        for (size_t q = 0; q < s->stg_n; q++) {
            if (q + 8 < s->stg_n && s->slot_cap) __builtin_prefetch(&s->slot[(size_t)tkey_mix(&s->stg[q + 8].k) & (s->slot_cap - 1)], 0);
            if (stamp_add_locked(j->t, s, tkey_mix(&s->stg[q].k), &s->stg[q].k, s->stg[q].hash, s->stg[q].stamp)) j->failed = 1;
        }
        s->stg_n = 0;
    }
}

/* the first-insertion sequence so far, as opaque 16-byte keys and their hashes (a worker of `--devices` hands its own to the
 * parent); -1 when the replay has failed */
int64_t mmh_tie_export(const mmh_tie_t *tc, const void **keys, const uint32_t **hash) {
    mmh_tie_t *t = (mmh_tie_t *)tc;
    if (!t || t->failed || seq_build(t) != 0) return -1;
    *keys = t->keys; *hash = t->hash;
    return (int64_t)t->n;
}
/* ... and whether any put followed the last new key's (a worker's reads may end on keys met before) */
int64_t mmh_tie_export2(const mmh_tie_t *tc, const void **keys, const uint32_t **hash, int *put_after_last) {
    const int64_t n = mmh_tie_export(tc, keys, hash);
    if (put_after_last) *put_after_last = n >= 0 && tc->last_put > tc->top_stamp;
    return n;
}
/* the same sequence from rows and the order they were first entered in (the device-side replay's mm_tie_sequence): opaque 16-byte keys */
void mmh_tie_keys_from_rows(const mm_row_t *rows, const uint32_t *seq, int64_t n, void *keys16) {
    tkey_t *k = (tkey_t *)keys16;
    for (int64_t i = 0; i < n; i++) {
        const mm_row_t *r = &rows[seq ? seq[i] : (uint32_t)i];
        memset(&k[i], 0, sizeof k[i]);
        k[i].tid = r->tid; k[i].pos = r->pos; k[i].ins = r->ins_offset; k[i].code = r->code; k[i].hp = r->hp; k[i].strand = r->strand;
    }
}
/* ... appended to this sequence, in order (keys already in it keep their place) */
int mmh_tie_import2(mmh_tie_t *t, const void *keys, const uint32_t *hash, int64_t n, int put_after_last) {
    const uint64_t first = t ? t->reads_seen : 0;
    const int rc = mmh_tie_import(t, keys, hash, n);
    if (rc == 0 && put_after_last && n > 0) { const uint64_t lp = ((first + (uint64_t)n - 1) << 24) | 1u; if (lp > t->last_put) t->last_put = lp; }
    return rc;
}
int mmh_tie_import(mmh_tie_t *t, const void *keys, const uint32_t *hash, int64_t n) {
    if (!t || t->failed) return -1;
    const tkey_t *k = (const tkey_t *)keys;
    /* (a worker's sequence counts as the keys of so many reads of one key each: its order is kept, and it lies behind everything entered before) */
    for (int64_t i = 0; i < n; i++) if (stamp_add(t, &k[i], hash[i], (t->reads_seen + (uint64_t)i) << 24)) { t->failed = 1; return -1; }
    if (n > 0 && ((t->reads_seen + (uint64_t)n - 1) << 24) > t->last_put) t->last_put = (t->reads_seen + (uint64_t)n - 1) << 24;
    t->reads_seen += (uint64_t)n;
    return 0;
}

int mmh_tie_add_batch(mmh_tie_t *t, mm_pool_t *pool, const mm_batch_t *batch, const mm_view_row_t *rows, int64_t n,
                      const uint8_t *const *klass_of_code, const char *const *codes, int n_codes) {
    if (!t || t->failed) return -1;
    const int32_t nr = batch->n_reads;
    int64_t *first = (int64_t *)malloc(sizeof(int64_t) * ((size_t)nr + 1));
    if (!first) { t->failed = 1; return -1; }
    int64_t i = 0;
    for (int32_t r = 0; r <= nr; r++) {   /* rows come sorted by read */
        while (i < n && (int32_t)(rows[i].read & 0x1FFFFFu) < r) i++;
        first[r] = i;
    }
    first[nr] = n;
    readjob_t job;
    memset(&job, 0, sizeof job);
    job.t = t; job.batch = batch; job.rows = rows; job.first = first; job.klass = klass_of_code; job.codes = codes; job.n_codes = n_codes;
    job.serial0 = t->reads_seen;
    /* every read replays its own table and enters its keys with their stamps: merge_freq_maps' order (reads in batch order, every
     * read's keys in its table's slot order) is in the stamps, not in who gets there first */
    const int timing = getenv("MM_TIE_TIMING") != NULL;
    const double t0 = timing ? mmh_realtime() : 0;
    mm_pool_for(pool, nr, 16, read_range, &job);
    const double t1 = timing ? mmh_realtime() : 0;
    mm_pool_for(pool, TS_SHARDS, 1, shard_merge, &job);
    if (timing) fprintf(stderr, "[mmh_tie_add_batch] %d reads, %ld calls: the reads' own tables %.3f s, their keys into the 256 tables %.3f s\n", nr, (long)n, t1 - t0, mmh_realtime() - t1);
    t->reads_seen += (uint64_t)nr;
    if (job.failed) t->failed = 1;
    free(first);
    return t->failed ? -1 : 0;
}

typedef struct { mmh_tie_t *t; const mm_row_t *rows; mm_row_t *out; uint8_t *taken; uint32_t *idx; size_t cap; const sel_t *arr; int failed; } matchjob_t;
static void match_index(void *arg, int64_t lo, int64_t hi) {
    matchjob_t *m = (matchjob_t *)arg;
    for (int64_t i = lo; i < hi; i++) {
        tkey_t k;
        memset(&k, 0, sizeof k);
        k.tid = m->rows[i].tid; k.pos = m->rows[i].pos; k.ins = m->rows[i].ins_offset; k.code = m->rows[i].code; k.hp = m->rows[i].hp; k.strand = m->rows[i].strand;
        size_t s = (size_t)tkey_mix(&k) & (m->cap - 1);
        while (!__sync_bool_compare_and_swap(&m->idx[s], 0xFFFFFFFFu, (uint32_t)i)) s = (s + 1) & (m->cap - 1);
    }
}
static void match_rows(void *arg, int64_t lo, int64_t hi) {
    matchjob_t *m = (matchjob_t *)arg;
    for (int64_t p = lo; p < hi; p++) {
        const tkey_t *k = &m->t->keys[m->arr[p].id];
        size_t s = (size_t)tkey_mix(k) & (m->cap - 1);
        for (;;) {
            const uint32_t at = m->idx[s];
            if (at == 0xFFFFFFFFu) { m->failed = 1; break; }
            const mm_row_t *r = &m->rows[at];
            if (r->tid == k->tid && r->pos == k->pos && r->ins_offset == k->ins && r->code == k->code && r->hp == k->hp && r->strand == k->strand) {
                m->taken[at] = 1;   /* (two keys on one row would leave another row untaken: counted by the caller) */
                m->out[p] = *r;
                break;
            }
            s = (s + 1) & (m->cap - 1);
        }
    }
}

/* rows (any order, one per key) -> the order print_freq_output prints them in.  Returns 0, or -1 when the replay could not
 * be made (then the rows are left as they were). */
int mmh_tie_order_rows(mmh_tie_t *t, mm_row_t *rows, int64_t n) { return mmh_tie_order_rows_mt(t, NULL, rows, n); }
/* ... with the worker pool's help where the work is not the reference's own serial walk (pool may be NULL) */
int mmh_tie_order_rows_mt(mmh_tie_t *t, mm_pool_t *pool, mm_row_t *rows, int64_t n) {
    if (!t || t->failed) return -1;
    if (n == 0) return 0;
    const int timing = getenv("MM_TIE_TIMING") != NULL;
    double tp[5] = {0, 0, 0, 0, 0};
    tp[0] = mmh_realtime();
    if (seq_build_mt(t, pool) != 0) return -1;
    tp[1] = mmh_realtime();
    if ((size_t)n != t->n) return -1;   /* the replay saw another set of keys than the counters hold: do not guess */
    /* the core table: keys in first-insertion order */
    ktab_t core;
    memset(&core, 0, sizeof core);
    for (size_t i = 0; i < t->n; i++) {
        if (i + 12 < t->n && core.n_buckets) { const uint32_t pf = t->hash[i + 12] & (core.n_buckets - 1); __builtin_prefetch(&core.used[pf], 1); __builtin_prefetch(&core.id[pf], 1); }
        if (ktab_put_new(&core, (uint32_t)i, t->hash) != 1) { ktab_free(&core); return -1; }
    }
    /* a put behind the last new key's finds the table at its bound and grows it (khash.h kh_put looks at the bound before it looks for the key) */
    if (t->last_put > t->top_stamp && core.n_occupied >= core.upper && ktab_grow(&core, core.n_buckets + 1, t->hash)) { ktab_free(&core); return -1; }
    sel_t *arr = (sel_t *)malloc(sizeof(sel_t) * t->n);
    if (!arr) { ktab_free(&core); return -1; }
    size_t w = 0;
    for (uint32_t s = 0; s < core.n_buckets; s++)
        if (core.used[s]) {
            const tkey_t *k = &t->keys[core.id[s]];
            const int32_t rk = (k->tid >= 0 && k->tid < t->hdr->n_targets) ? t->rank[k->tid] : -1;
            arr[w].key = ((int64_t)rk << 32) + (int64_t)k->pos; arr[w].id = core.id[s]; arr[w].pad = 0;
            w++;
        }
    ktab_free(&core);
    tp[2] = mmh_realtime();
    if (intro_sort(w, arr)) { free(arr); return -1; }
    tp[3] = mmh_realtime();
    /* the GPU's row of every key: the rows' places entered into an index by all threads (an empty slot is claimed with a
     * compare-and-swap), every key's row looked up by all threads; that every row was taken exactly once is counted afterwards */
    mm_row_t *out = (mm_row_t *)malloc(sizeof(mm_row_t) * (size_t)n);
    uint8_t *taken = (uint8_t *)calloc((size_t)n, 1);
    size_t cap = 1;
    while (cap < (size_t)n * 2) cap <<= 1;
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    if (!out || !taken || !idx) { free(out); free(taken); free(idx); free(arr); return -1; }
    memset(idx, 0xFF, sizeof(uint32_t) * cap);
    matchjob_t mj = {t, rows, out, taken, idx, cap, arr, 0};
    if (pool) { mm_pool_for(pool, n, 65536, match_index, &mj); mm_pool_for(pool, (int64_t)w, 65536, match_rows, &mj); }
    else { match_index(&mj, 0, n); match_rows(&mj, 0, (int64_t)w); }
    int ok = !mj.failed && (int64_t)w == n;
    if (ok) { size_t c = 0; for (int64_t i = 0; i < n; i++) c += taken[i]; ok = c == (size_t)n; }
    if (ok) memcpy(rows, out, sizeof(mm_row_t) * (size_t)n);
    if (timing) fprintf(stderr, "[mmh_tie_order_rows] %ld keys: first-insertion sequence %.3f s, core table %.3f s, introsort %.3f s, rows to their keys %.3f s\n",
                        (long)n, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], mmh_realtime() - tp[3]);
    free(out); free(taken); free(idx); free(arr);
    return ok ? 0 : -1;
}
