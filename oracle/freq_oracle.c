/*
 * freq_oracle.c -- CPU restatement of minimod's `freq` hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call
 * this file.  The product path (minimod_amd/) never routes through it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement (freq and view modes) against the
 * reference's own golden files (reference test/expected/test3,4,5,5a,5b,5c,6,7,8,9,12,16 for freq and
 * test1,2,2a,2b,2c,2c_wild,10,11,15,17a for view) on the
 * reference's bundled BAMs, using the pseudo-references of tests/golden/make_fixtures.py, plus the
 * hand-built known-answer reads of SURVEY.md section 8(c).  The reference binary itself cannot be
 * built here: it needs htslib 1.9 (reference scripts/install-hts.sh:9, Makefile:29-30), which this
 * image lacks and which may not be replaced by a stand-in.
 *
 * Each function cites the reference code it restates (paths under /root/reference).  The code is
 * written from the behaviour, not copied: counters live in an integer-keyed open-addressing table
 * instead of the reference's string-keyed khash, so row ORDER among equal (contig,pos) is this
 * file's canonical order, not khash order.
 */
#include <ctype.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- batch layout (same bytes as include/minimod_hip.h mm_read_t; declared independently) ---- */
typedef struct {
    uint64_t cigar_off, seq_off, mm_off, ml_off;
    int32_t tid, pos;
    uint32_t l_qseq, n_cigar, mm_len, ml_len;
    uint16_t flag;
    uint8_t hp, rsvd;
    uint32_t rsvd2;
} orc_read_t;

typedef struct {
    int32_t tid, pos;
    int32_t strand;   /* 0 '+', 1 '-' */
    int32_t code;     /* index into the code-name table */
    int32_t ins_off;  /* 0..65535 */
    int32_t hp;       /* -1 = '*' */
    uint32_t n_called, n_mod;
} orc_row_t;

/* one row of `minimod view` (add_view_entry, mod.c:931-946; print_view_output, mod.c:560-626) */
typedef struct {
    int64_t read;     /* running index of the read over all processed batches */
    int32_t tid, pos, strand, code, ins_off, hp, read_pos;
    uint32_t prob;    /* ML byte, 0 for implicit calls */
} orc_view_row_t;

enum {
    ORC_OK = 0, ORC_E_HARDCLIP = 1, ORC_E_CIGAROP = 2, ORC_E_MMBASE = 3, ORC_E_MMSTRAND = 4,
    ORC_E_MMCODE = 5, ORC_E_MMEMPTY = 6, ORC_E_MMMIXED = 7, ORC_E_SKIPLEN = 8, ORC_E_SKIPVAL = 9,
    ORC_E_READPOS = 10, ORC_E_MLIDX = 11, ORC_E_NOCONTIG = 12, ORC_E_REFPOS = 13, ORC_E_QOVER = 14
};

#define MAX_MODS 64   /* (array sizes of this restatement: the reference counts its entries in a byte, src/minimod.h:114) */
#define MAX_CODES 256
#define CODE_LEN 32

typedef struct {
    char *name;
    int64_t len;
    uint8_t *fwd;            /* upper-cased, U->T (ref.c:73-78) */
    uint8_t *ctx[MAX_MODS];  /* is_context      (ref.c:205,217) */
    uint8_t *ctx_rev[MAX_MODS];
} contig_t;

typedef struct { uint64_t k0, k1; uint32_t n_called, n_mod; } slot_t; /* k1 == 0 => empty */

typedef struct {
    slot_t *s;
    uint64_t cap, n;
} map_t;

typedef struct {
    int n_mods;
    char req_code[MAX_MODS][CODE_LEN];
    char req_ctx[MAX_MODS][CODE_LEN];
    double thresh[MAX_MODS];
    int wildcard_idx; /* index of "*" in -c, or -1 (mod.c:1146) */
    int insertions, haplotypes;
    int n_contigs;
    contig_t *contigs; /* by tid */
    /* code-name table: ids 0..n_mods-1 are the -c entries; wildcard mode appends what reads carry */
    int n_codes;
    char code_name[MAX_CODES][CODE_LEN];
    pthread_mutex_t code_mu;
    map_t global;
    int err;
    int64_t err_read;
    int view;                 /* 1: collect view rows instead of counting */
    orc_view_row_t *vrows;
    int64_t n_vrows, cap_vrows, reads_seen;
    double merge_seconds;     /* time spent folding per-thread tables into the global one (merge_freq_maps' share) */
} orc_t;

/* ------------------------------------------------------------------ counter table */
static void map_init(map_t *m, uint64_t cap) {
    m->cap = cap; m->n = 0;
    m->s = (slot_t *)calloc(cap, sizeof(slot_t));
}
static inline uint64_t mix(uint64_t a, uint64_t b) {
    uint64_t x = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
static void map_add(map_t *m, uint64_t k0, uint64_t k1, uint32_t c, uint32_t d);
static void map_grow(map_t *m) {
    map_t n; map_init(&n, m->cap * 2);
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->s[i].k1) map_add(&n, m->s[i].k0, m->s[i].k1, m->s[i].n_called, m->s[i].n_mod);
    free(m->s); *m = n;
}
static void map_add(map_t *m, uint64_t k0, uint64_t k1, uint32_t c, uint32_t d) {
    if ((m->n + 1) * 10 > m->cap * 7) map_grow(m);
    uint64_t i = mix(k0, k1) & (m->cap - 1);
    for (;;) {
        slot_t *s = &m->s[i];
        if (!s->k1) { s->k0 = k0; s->k1 = k1; s->n_called = c; s->n_mod = d; m->n++; return; }
        if (s->k0 == k0 && s->k1 == k1) { s->n_called += c; s->n_mod += d; return; }
        i = (i + 1) & (m->cap - 1);
    }
}
/* key packing: k0 = tid<<32 | pos ; k1 = 1<<63 | strand<<40 | code<<32 | ins<<16 | (hp+1) */
static inline void key_pack(int tid, int pos, int strand, int code, int ins, int hp, uint64_t *k0, uint64_t *k1) {
    *k0 = ((uint64_t)(uint32_t)tid << 32) | (uint32_t)pos;
    *k1 = (1ull << 63) | ((uint64_t)strand << 40) | ((uint64_t)code << 32) | ((uint64_t)(ins & 0xFFFF) << 16) | (uint64_t)(hp + 1);
}

/* update_freq_map (mod.c:883-929): the key, and with a haplotype also the hp=-1 aggregate */
static inline void count_call(map_t *m, int tid, int pos, int ins, int code, int strand, int hp, int is_mod) {
    uint64_t k0, k1;
    key_pack(tid, pos, strand, code, ins, hp, &k0, &k1);
    map_add(m, k0, k1, 1, (uint32_t)is_mod);
    if (hp != -1) {
        key_pack(tid, pos, strand, code, ins, -1, &k0, &k1);
        map_add(m, k0, k1, 1, (uint32_t)is_mod);
    }
}

/* ------------------------------------------------------------------ lookup tables (mod.c:95-98) */
static int is_valid_base(int c) { return strchr("ACGTUNacgtun", c) != NULL && c != 0; }
static int base_class(int c) { /* base_idx_lookup, default 0 */
    switch (c) { case 'C': case 'c': return 1; case 'G': case 'g': return 2;
                 case 'T': case 't': case 'U': case 'u': return 3; case 'N': case 'n': return 4; default: return 0; }
}
static int base_complement(int c) { /* base_complement_lookup, default NUL */
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'U': return 'A'; case 'N': return 'N'; case 'a': return 't'; case 'c': return 'g';
                 case 'g': return 'c'; case 't': return 'a'; case 'u': return 'a'; case 'n': return 'n'; default: return 0; }
}
static const char NT16[] = "=ACMGRSVTWYHKDBN"; /* seq_nt16_str */
static inline int seq_char(const uint8_t *seq, uint32_t i) {
    uint8_t b = seq[i >> 1];
    return NT16[(i & 1) ? (b & 15) : (b >> 4)];
}

/* ------------------------------------------------------------------ set-up */
void *orc_create(int n_mods, const char **codes, const char **contexts, const double *thresh,
                 int insertions, int haplotypes, int n_contigs) {
    if (n_mods > MAX_MODS) return NULL;
    orc_t *o = (orc_t *)calloc(1, sizeof(orc_t));
    o->n_mods = n_mods;
    o->wildcard_idx = -1;
    for (int i = 0; i < n_mods; i++) {
        snprintf(o->req_code[i], CODE_LEN, "%s", codes[i]);
        snprintf(o->req_ctx[i], CODE_LEN, "%s", contexts[i]);
        o->thresh[i] = thresh[i];
        if (strcmp(codes[i], "*") == 0) o->wildcard_idx = i;
        snprintf(o->code_name[i], CODE_LEN, "%s", codes[i]);
    }
    o->n_codes = n_mods;
    o->insertions = insertions; o->haplotypes = haplotypes;
    o->n_contigs = n_contigs;
    o->contigs = (contig_t *)calloc(n_contigs > 0 ? n_contigs : 1, sizeof(contig_t));
    pthread_mutex_init(&o->code_mu, NULL);
    map_init(&o->global, 1 << 16);
    return o;
}

/* mark every position inside any occurrence of pat (ref.c:142-162; the KMP of ref.c:92-139 finds all,
 * overlapping, occurrences -- restated here as a direct scan) */
static void mark_context(const char *pat, const uint8_t *txt, int64_t n, uint8_t *out) {
    int64_t m = (int64_t)strlen(pat);
    if (m == 0 || m > n) return;
    for (int64_t s = 0; s + m <= n; s++) {
        if (txt[s] != (uint8_t)pat[0]) continue;
        int64_t j = 1;
        while (j < m && txt[s + j] == (uint8_t)pat[j]) j++;
        if (j == m) for (int64_t k = s; k < s + m; k++) out[k] = 1;
    }
}

/* load_ref (ref.c:46-89) normalisation + load_ref_contexts (ref.c:177-229) for one contig */
int orc_add_contig(void *h, int tid, const char *name, const uint8_t *raw, int64_t len) {
    orc_t *o = (orc_t *)h;
    if (tid < 0 || tid >= o->n_contigs) return -1;
    contig_t *c = &o->contigs[tid];
    c->name = strdup(name);
    c->len = len;
    c->fwd = (uint8_t *)malloc(len + 1);
    for (int64_t i = 0; i < len; i++) {
        int ch = toupper(raw[i]);
        c->fwd[i] = (uint8_t)(ch == 'U' ? 'T' : ch);
    }
    c->fwd[len] = 0;
    for (int i = 0; i < o->n_mods; i++) {
        c->ctx[i] = (uint8_t *)calloc(len + 1, 1);
        c->ctx_rev[i] = (uint8_t *)calloc(len + 1, 1);
        if (strcmp(o->req_ctx[i], "*") == 0) {
            memset(c->ctx[i], 1, len); memset(c->ctx_rev[i], 1, len);
        } else {
            char rc[CODE_LEN];
            int L = (int)strlen(o->req_ctx[i]);
            for (int j = 0; j < L; j++) rc[j] = (char)base_complement(o->req_ctx[i][L - 1 - j]);
            rc[L] = 0;
            mark_context(o->req_ctx[i], c->fwd, len, c->ctx[i]);
            mark_context(rc, c->fwd, len, c->ctx_rev[i]);
        }
    }
    return 0;
}
/* name-only contig (present in the BAM header, absent from the FASTA) */
int orc_name_contig(void *h, int tid, const char *name) {
    orc_t *o = (orc_t *)h;
    if (tid < 0 || tid >= o->n_contigs) return -1;
    if (!o->contigs[tid].name) o->contigs[tid].name = strdup(name);
    return 0;
}

static int code_id(orc_t *o, const char *s) {
    pthread_mutex_lock(&o->code_mu);
    int id = -1;
    for (int i = 0; i < o->n_codes; i++) if (strcmp(o->code_name[i], s) == 0) { id = i; break; }
    if (id < 0 && o->n_codes < MAX_CODES) { id = o->n_codes++; snprintf(o->code_name[id], CODE_LEN, "%s", s); }
    pthread_mutex_unlock(&o->code_mu);
    return id;
}

/* ------------------------------------------------------------------ per-read scratch */
typedef struct {
    int *aln, *ins, *ins_off, *skips;
    int *bases[5];
    uint32_t cap;
} scratch_t;
static void scratch_fit(scratch_t *s, uint32_t n) {
    if (n <= s->cap) return;
    s->cap = n * 2;
    s->aln = (int *)realloc(s->aln, sizeof(int) * s->cap);
    s->ins = (int *)realloc(s->ins, sizeof(int) * s->cap);
    s->ins_off = (int *)realloc(s->ins_off, sizeof(int) * s->cap);
    s->skips = (int *)realloc(s->skips, sizeof(int) * s->cap);
    for (int b = 0; b < 5; b++) s->bases[b] = (int *)realloc(s->bases[b], sizeof(int) * s->cap);
}
static void scratch_free(scratch_t *s) {
    free(s->aln); free(s->ins); free(s->ins_off); free(s->skips);
    for (int b = 0; b < 5; b++) free(s->bases[b]);
}

/* get_aln (mod.c:776-881): aligned pairs in ORIGINAL read (FASTQ) orientation; reverse reads walk
 * the CIGAR back to front with mirrored coordinates. */
static int walk_cigar(orc_t *o, const orc_read_t *rd, const uint32_t *cigar, scratch_t *s) {
    if (rd->tid < 0 || rd->tid >= o->n_contigs || !o->contigs[rd->tid].fwd) return ORC_E_NOCONTIG;
    const contig_t *c = &o->contigs[rd->tid];
    int rev = (rd->flag & 0x10) != 0;
    int L = (int)rd->l_qseq;
    int64_t span = 0;
    for (uint32_t i = 0; i < rd->n_cigar; i++) {
        int op = cigar[i] & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cigar[i] >> 4;
    }
    int pos = rd->pos;
    int end = (int)(pos + (span ? span : 1)); /* bam_endpos */
    for (int i = 0; i < L; i++) s->aln[i] = -1;
    if (o->insertions) for (int i = 0; i < L; i++) { s->ins[i] = -1; s->ins_off[i] = 0; }
    int read_pos = 0, ref_pos = pos;
    for (uint32_t ci = 0; ci < rd->n_cigar; ci++) {
        uint32_t w = rev ? cigar[rd->n_cigar - 1 - ci] : cigar[ci];
        int len = (int)(w >> 4), op = (int)(w & 15);
        int read_inc = 0, ref_inc = 0, aligned = 0, inserted = 0;
        if (op == 0 || op == 7 || op == 8) { aligned = 1; read_inc = 1; ref_inc = 1; }
        else if (op == 2 || op == 3) { ref_inc = 1; }
        else if (op == 1) { read_inc = 1; inserted = 1; }
        else if (op == 4) { read_inc = 1; }
        else if (op == 5) { return ORC_E_HARDCLIP; }
        else { return ORC_E_CIGAROP; }
        for (int j = 0; j < len; j++) {
            if (aligned) {
                if (read_pos >= L) return ORC_E_QOVER;
                s->aln[read_pos] = rev ? pos + end - ref_pos - 1 : ref_pos;
                if (ref_pos < 0 || ref_pos >= c->len) return ORC_E_REFPOS;
            }
            if (o->insertions && inserted) {
                if (read_pos >= L) return ORC_E_QOVER;
                s->ins[read_pos] = rev ? pos + end - ref_pos - 1 : ref_pos - 1;
                s->ins_off[read_pos] = rev ? len - j : j + 1;
            }
            read_pos += read_inc; ref_pos += ref_inc;
        }
    }
    return ORC_OK;
}

typedef struct {
    int n;              /* mod_codes_len (1 for ChEBI) */
    int gord;           /* ordinal of the group in the read's MM string */
    int has_nums;
    char codes[CODE_LEN];
    int req[CODE_LEN];  /* required-mod index per code letter, -1 = not requested */
    int cid[CODE_LEN];  /* output code id per code letter */
} group_codes_t;

/* one candidate call (explicit or implicit): filters + threshold + count.
 * mod.c:1140-1197 (explicit) and mod.c:1242-1284,1322-1364 (implicit). */
typedef struct { orc_view_row_t *v; int64_t n, cap; } vbuf_t;
static void vbuf_push(vbuf_t *b, orc_view_row_t r) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 1024; b->v = (orc_view_row_t *)realloc(b->v, sizeof(orc_view_row_t) * b->cap); }
    b->v[b->n++] = r;
}

static inline int emit_call(orc_t *o, map_t *m, const orc_read_t *rd, const contig_t *c, const group_codes_t *g,
                            int mb, int read_base, int ref_pos, int ins_off, int hp, int rev,
                            int explicit_call, int call_idx, int ml_start, const uint8_t *ml, vbuf_t *vb, int fq_pos) {
    for (int k = 0; k < g->n; k++) {
        int req = g->req[k];
        if (req < 0) continue;
        int all_ctx = strcmp(o->req_ctx[req], "*") == 0;
        int in_ctx = rev ? c->ctx_rev[req][ref_pos] : c->ctx[req][ref_pos];
        int matches = all_ctx || mb == 'N' || c->fwd[ref_pos] == read_base;
        if (!o->insertions && !(in_ctx && matches)) continue;
        int is_mod = 0;
        if (o->view) {   /* mod.c:1194-1196, :1281-1283: no threshold, probability 0 for implicit calls */
            uint32_t prob = 0;
            if (explicit_call) {
                int64_t ml_idx = (int64_t)ml_start + (int64_t)call_idx * g->n + k;
                if (ml_idx >= (int64_t)rd->ml_len) return ORC_E_MLIDX;
                prob = ml[ml_idx];
            }
            orc_view_row_t r;
            r.read = 0; r.tid = rd->tid; r.pos = ref_pos; r.strand = rev; r.code = g->cid[k]; r.ins_off = ins_off; r.hp = hp;
            r.read_pos = fq_pos; r.prob = prob;
            /* view mode 2 (tests of the product's tie-order replay): rows stay in call order, every call is kept, and the row
             * also says which group made the call and whether it was an implicit one */
            if (o->view == 2) r.prob = prob | ((uint32_t)(g->gord & 0xFF) << 8) | (explicit_call ? 0u : 0x80000000u);
            vbuf_push(vb, r);
            continue;
        }
        if (explicit_call) {
            int64_t ml_idx = (int64_t)ml_start + (int64_t)call_idx * g->n + k;
            if (ml_idx >= (int64_t)rd->ml_len) return ORC_E_MLIDX;
            double p = (double)((ml[ml_idx] + 0.5) / 256.0); /* THRESH_UINT8_TO_DBL, mod.c:56 */
            double t = o->thresh[req];
            if (p >= t) is_mod = 1;
            else if (p <= 1 - t) is_mod = 0;
            else continue;
        }
        count_call(m, rd->tid, ref_pos, ins_off, g->cid[k], rev, hp, is_mod);
    }
    return ORC_OK;
}

/* freq_view_single (mod.c:948-1370) for one read */
static int process_read(orc_t *o, map_t *m, const orc_read_t *rd, const uint32_t *cigar_pool,
                        const uint8_t *seq_pool, const uint8_t *mm_pool, const uint8_t *ml_pool, scratch_t *s, vbuf_t *vb) {
    const uint32_t *cigar = cigar_pool + rd->cigar_off;
    const uint8_t *seq = seq_pool + rd->seq_off;
    const char *mm = (const char *)(mm_pool + rd->mm_off);
    const uint8_t *ml = ml_pool + rd->ml_off;
    int L = (int)rd->l_qseq;
    int rev = (rd->flag & 0x10) != 0;
    int hp = o->haplotypes ? rd->hp : -1;
    scratch_fit(s, rd->l_qseq + 1);
    int e = walk_cigar(o, rd, cigar, s);
    if (e) return e;
    const contig_t *c = &o->contigs[rd->tid];

    int nb[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < L; i++) { int b = base_class(seq_char(seq, i)); s->bases[b][nb[b]++] = i; }

    int n = (int)rd->mm_len, i = 0, ml_start = 0, gord = 0;
    while (i < n) {
        group_codes_t g; memset(&g, 0, sizeof(g));
        g.gord = gord++;
        int flag = '.';
        if (!is_valid_base((unsigned char)mm[i])) return ORC_E_MMBASE;
        int modbase = mm[i] == 'U' ? 'T' : mm[i];
        i++;
        if (i < n) { if (mm[i] != '+' && mm[i] != '-') return ORC_E_MMSTRAND; i++; }
        int j = 0, has_alpha = 0;
        while (i < n && mm[i] != ',' && mm[i] != ';' && mm[i] != '?' && mm[i] != '.') {
            if (isdigit((unsigned char)mm[i])) g.has_nums = 1;
            else if (isalpha((unsigned char)mm[i])) has_alpha = 1;
            else return ORC_E_MMCODE;
            if (j < CODE_LEN - 1) g.codes[j] = mm[i];
            j++; i++;
        }
        if (j >= CODE_LEN) return ORC_E_MMCODE;
        g.codes[j] = 0;
        g.n = g.has_nums ? 1 : j;
        if (g.n <= 0) return ORC_E_MMEMPTY;
        if (g.has_nums && has_alpha) return ORC_E_MMMIXED;
        if (i < n && (mm[i] == '?' || mm[i] == '.')) { flag = mm[i]; i++; }
        int ns = 0;
        while (i < n && mm[i] != ';') {
            if (mm[i] == ',') { i++; continue; }
            int l = 0; long v = 0;
            while (i < n && mm[i] != ',' && mm[i] != ';') {
                if (!isdigit((unsigned char)mm[i])) return ORC_E_SKIPVAL;
                v = v * 10 + (mm[i] - '0');
                i++; l++;
                if (l >= 10) return ORC_E_SKIPLEN; /* assert(l < 10), mod.c:1080 */
            }
            s->skips[ns++] = (int)v;
            if (ns > L) return ORC_E_READPOS;
        }
        i++;
        /* required-code lookup per code letter (mod.c:1146-1160): wildcard first, then the C string
         * starting at letter k (so "hm" is looked up as "hm", then "m") */
        for (int k = 0; k < g.n; k++) {
            const char *name = g.has_nums ? g.codes : &g.codes[k];
            g.req[k] = -1; g.cid[k] = -1;
            if (o->wildcard_idx >= 0) { g.req[k] = o->wildcard_idx; g.cid[k] = code_id(o, name); }
            else for (int r = 0; r < o->n_mods; r++) if (strcmp(o->req_code[r], name) == 0) { g.req[k] = r; g.cid[k] = r; }
        }
        int mb = rev ? base_complement(modbase) : modbase;
        int idx = base_class(mb);
        int direct = (modbase == 'N');
        int rank = -1;
        for (int cidx = 0; cidx < ns; cidx++) {
            rank += s->skips[cidx] + 1;
            int read_pos;
            if (direct) read_pos = rev ? L - rank - 1 : rank;
            else {
                if (rank >= nb[idx]) return ORC_E_READPOS; /* the reference reads out of bounds here */
                read_pos = rev ? s->bases[idx][nb[idx] - rank - 1] : s->bases[idx][rank];
            }
            if (read_pos < 0 || read_pos >= L) return ORC_E_READPOS;
            int read_base = seq_char(seq, read_pos);
            int fq = rev ? L - read_pos - 1 : read_pos;
            int ref_pos = s->aln[fq];
            if (o->insertions && ref_pos == -1) ref_pos = s->ins[fq];
            if (ref_pos == -1) continue;
            int ins_off = o->insertions ? s->ins_off[fq] : 0;
            e = emit_call(o, m, rd, c, &g, mb, read_base, ref_pos, ins_off, hp, rev, 1, cidx, ml_start, ml, vb, fq);
            if (e) return e;
        }
        if (ns > 0) ml_start += ns * g.n; /* mod.c:1200 */
        if (flag == '.') {
            /* implicit calls: every base of the type not listed is called, unmodified (mod.c:1203-1367) */
            int prev = -1, r2 = -1;
            for (int cidx = 0; cidx <= ns; cidx++) {
                int hi;
                if (cidx < ns) { r2 += s->skips[cidx] + 1; hi = r2; } else hi = nb[idx];
                for (int sidx = prev + 1; sidx < hi; sidx++) {
                    int read_pos;
                    if (direct) read_pos = rev ? L - sidx - 1 : sidx;
                    else {
                        if (sidx >= nb[idx]) return ORC_E_READPOS;
                        read_pos = rev ? s->bases[idx][nb[idx] - sidx - 1] : s->bases[idx][sidx];
                    }
                    if (read_pos < 0 || read_pos >= L) return ORC_E_READPOS;
                    int read_base = seq_char(seq, read_pos);
                    int fq = rev ? L - read_pos - 1 : read_pos;
                    int ref_pos = s->aln[fq];
                    /* quirk kept: ins[] is indexed with the BAM-orientation position here (mod.c:1234,1314) */
                    if (o->insertions && ref_pos == -1) ref_pos = s->ins[read_pos];
                    if (ref_pos == -1) continue;
                    int ins_off = o->insertions ? s->ins_off[fq] : 0;
                    e = emit_call(o, m, rd, c, &g, mb, read_base, ref_pos, ins_off, hp, rev, 0, 0, 0, ml, vb, fq);
                    if (e) return e;
                }
                prev = hi;
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ batch driver (thread.c:50-158 restated
 * as a static block partition; per-thread tables are merged like merge_freq_maps, mod.c:743-774) */
typedef struct {
    orc_t *o; const orc_read_t *reads; int lo, hi;
    const uint32_t *cigar; const uint8_t *seq, *mm, *ml;
    map_t map; int err; int64_t err_read;
    vbuf_t rows;   /* view mode: this thread's rows, reads in order */
} job_t;

static int vrow_cmp(const void *a, const void *b) {
    const orc_view_row_t *x = (const orc_view_row_t *)a, *y = (const orc_view_row_t *)b;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    if (x->strand != y->strand) return x->strand - y->strand;
    if (x->code != y->code) return x->code - y->code;
    if ((x->ins_off & 0xFFFF) != (y->ins_off & 0xFFFF)) return (x->ins_off & 0xFFFF) - (y->ins_off & 0xFFFF);
    if (x->hp != y->hp) return x->hp - y->hp;
    return x->read < y->read ? -1 : (x->read > y->read);   /* `read` holds the emission order here */
}

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    scratch_t s; memset(&s, 0, sizeof(s));
    map_init(&j->map, 1 << 12);
    vbuf_t vb = {0};
    for (int i = j->lo; i < j->hi; i++) {
        vb.n = 0;
        int e = process_read(j->o, &j->map, &j->reads[i], j->cigar, j->seq, j->mm, j->ml, &s, &vb);
        if (e) { j->err = e; j->err_read = i; break; }
        if (j->o->view == 2 && vb.n) {   /* call order, nothing dropped */
            for (int64_t k = 0; k < vb.n; k++) { orc_view_row_t r = vb.v[k]; r.read = i; vbuf_push(&j->rows, r); }
        } else if (j->o->view && vb.n) {
            /* per read: sort by key, keep the first entry of every key (add_view_entry), rows by position */
            for (int64_t k = 0; k < vb.n; k++) vb.v[k].read = k;
            qsort(vb.v, (size_t)vb.n, sizeof(orc_view_row_t), vrow_cmp);
            for (int64_t k = 0; k < vb.n; k++) {
                if (k > 0) {
                    const orc_view_row_t *a = &vb.v[k - 1], *b = &vb.v[k];
                    if (a->pos == b->pos && a->strand == b->strand && a->code == b->code && (a->ins_off & 0xFFFF) == (b->ins_off & 0xFFFF) && a->hp == b->hp) continue;
                }
                orc_view_row_t r = vb.v[k];
                r.read = i;
                vbuf_push(&j->rows, r);
            }
        }
    }
    free(vb.v);
    scratch_free(&s);
    return NULL;
}

int orc_process(void *h, const orc_read_t *reads, int n, const uint32_t *cigar, const uint8_t *seq,
                const uint8_t *mm, const uint8_t *ml, int n_threads) {
    orc_t *o = (orc_t *)h;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n > 0 ? n : 1;
    job_t *jobs = (job_t *)calloc(n_threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].o = o; jobs[t].reads = reads; jobs[t].cigar = cigar; jobs[t].seq = seq; jobs[t].mm = mm; jobs[t].ml = ml;
        jobs[t].lo = (int)((int64_t)n * t / n_threads); jobs[t].hi = (int)((int64_t)n * (t + 1) / n_threads);
        if (n_threads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    int err = 0;
    if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int t = 0; t < n_threads; t++) {
        if (jobs[t].err && !err) { err = jobs[t].err; o->err = err; o->err_read = jobs[t].err_read; }
        if (o->view) {
            for (int64_t k = 0; k < jobs[t].rows.n; k++) {
                if (o->n_vrows == o->cap_vrows) { o->cap_vrows = o->cap_vrows ? o->cap_vrows * 2 : 4096; o->vrows = (orc_view_row_t *)realloc(o->vrows, sizeof(orc_view_row_t) * o->cap_vrows); }
                orc_view_row_t r = jobs[t].rows.v[k];
                r.read += o->reads_seen;
                o->vrows[o->n_vrows++] = r;
            }
            free(jobs[t].rows.v);
        }
        map_t *m = &jobs[t].map;
        for (uint64_t i = 0; i < m->cap; i++)
            if (m->s[i].k1) map_add(&o->global, m->s[i].k0, m->s[i].k1, m->s[i].n_called, m->s[i].n_mod);
        free(m->s);
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    o->merge_seconds += (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
    free(jobs); free(th);
    o->reads_seen += n;
    return err;
}

double orc_merge_seconds(void *h) { return ((orc_t *)h)->merge_seconds; }
void orc_set_view(void *h, int on) { ((orc_t *)h)->view = on; }
int64_t orc_n_view_rows(void *h) { return ((orc_t *)h)->n_vrows; }
void orc_view_rows(void *h, orc_view_row_t *out) { orc_t *o = (orc_t *)h; memcpy(out, o->vrows, sizeof(orc_view_row_t) * (size_t)o->n_vrows); }

int64_t orc_error_read(void *h) { return ((orc_t *)h)->err_read; }
int64_t orc_n_rows(void *h) { return (int64_t)((orc_t *)h)->global.n; }
int orc_n_codes(void *h) { return ((orc_t *)h)->n_codes; }
const char *orc_code_name(void *h, int id) { return ((orc_t *)h)->code_name[id]; }

static orc_t *g_sort_ctx;
/* row order: contig name in strcmp order then position (cmp_key_fast, mod.c:59-87); ties (which the
 * reference leaves in hash order) in the canonical order strand, code, ins_offset, haplotype with '*' last */
static int row_cmp(const void *a, const void *b) {
    const orc_row_t *x = (const orc_row_t *)a, *y = (const orc_row_t *)b;
    if (x->tid != y->tid) {
        const char *nx = g_sort_ctx->contigs[x->tid].name, *ny = g_sort_ctx->contigs[y->tid].name;
        int c = strcmp(nx ? nx : "", ny ? ny : "");
        if (c) return c;
        return x->tid < y->tid ? -1 : 1;
    }
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    if (x->strand != y->strand) return x->strand - y->strand;
    if (x->code != y->code) return x->code - y->code;
    if (x->ins_off != y->ins_off) return x->ins_off - y->ins_off;
    int hx = x->hp < 0 ? 1000 : x->hp, hy = y->hp < 0 ? 1000 : y->hp;
    return hx - hy;
}

void orc_rows(void *h, orc_row_t *out) {
    orc_t *o = (orc_t *)h;
    int64_t n = 0;
    for (uint64_t i = 0; i < o->global.cap; i++) {
        slot_t *s = &o->global.s[i];
        if (!s->k1) continue;
        orc_row_t *r = &out[n++];
        r->tid = (int32_t)(s->k0 >> 32); r->pos = (int32_t)(uint32_t)s->k0;
        r->strand = (int32_t)((s->k1 >> 40) & 1); r->code = (int32_t)((s->k1 >> 32) & 0xFF);
        r->ins_off = (int32_t)((s->k1 >> 16) & 0xFFFF); r->hp = (int32_t)(s->k1 & 0xFFFF) - 1;
        r->n_called = s->n_called; r->n_mod = s->n_mod;
    }
    g_sort_ctx = o;
    qsort(out, (size_t)n, sizeof(orc_row_t), row_cmp);
}

void orc_destroy(void *h) {
    orc_t *o = (orc_t *)h;
    for (int t = 0; t < o->n_contigs; t++) {
        contig_t *c = &o->contigs[t];
        free(c->name); free(c->fwd);
        for (int i = 0; i < o->n_mods; i++) { free(c->ctx[i]); free(c->ctx_rev[i]); }
    }
    free(o->contigs); free(o->global.s); free(o->vrows);
    pthread_mutex_destroy(&o->code_mu);
    free(o);
}
