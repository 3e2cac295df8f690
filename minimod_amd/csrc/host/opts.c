/* opts.c -- option parsing of `minimod freq`: -c / -m strings, -B numbers, threshold classes.
 * Behaviour follows reference src/mod.c:99-112 (default contexts), :204-326 (parse_mod_codes),
 * :328-398 (parse_mod_threshes), src/misc.c:74-87 (mm_parse_num). */
#include <ctype.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include "mmhost.h"

int mmh_log_level = 4;

double mmh_realtime(void) { struct timeval tp; gettimeofday(&tp, NULL); return tp.tv_sec + tp.tv_usec * 1e-6; }
double mmh_cputime(void) {
    struct rusage r; getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}
long mmh_peakrss(void) { struct rusage r; getrusage(RUSAGE_SELF, &r); return r.ru_maxrss * 1024; }

int64_t mmh_parse_num(const char *str) {
    char *p;
    double x = strtod(str, &p);
    if (*p == 'G' || *p == 'g') x *= 1e9;
    else if (*p == 'M' || *p == 'm') x *= 1e6;
    else if (*p == 'K' || *p == 'k') x *= 1e3;
    return (int64_t)(x + .499);
}

static const char *default_context(const char *code) {
    if (strlen(code) == 1) {
        switch (code[0]) {
            case '*': return "*";
            case 'm': case 'h': return "CG";
            case 'f': case 'c': case 'C': return "C";
            case 'g': case 'e': case 'b': case 'T': case 'U': return "T";
            case 'a': case 'A': return "A";
            case 'o': case 'G': return "G";
            case 'n': case 'N': return "N";
            default: break;
        }
    }
    return "CG";
}

int mmh_parse_mod_codes(const char *s, mmh_mods_t *out, char *err, size_t errlen) {
    memset(out, 0, sizeof(*out));
    size_t i = 0, n = strlen(s);
    while (i < n) {
        char code[64]; size_t j = 0;
        int has_alpha = 0, has_num = 0;
        while (i < n && s[i] != ',' && s[i] != '[') {
            char c = s[i];
            if (isalpha((unsigned char)c) || c == '*') has_alpha = 1;
            else if (isdigit((unsigned char)c)) has_num = 1;
            else { snprintf(err, errlen, "Invalid character %c in modification code in -c argument", c); return -1; }
            if (j < sizeof(code) - 1) code[j++] = c;
            i++;
        }
        code[j] = 0;
        if (has_alpha && has_num) { snprintf(err, errlen, "Modification code %s cannot contain both letters and numbers in -c argument", code); return -1; }
        if (j >= MM_CODE_LEN) { snprintf(err, errlen, "Modification code %s is longer than %d characters", code, MM_CODE_LEN - 1); return -1; }
        char ctx[64]; size_t k = 0;
        if (i < n && s[i] == '[') {
            i++;
            int star = 0;
            while (i < n && s[i] != ']') {
                char c = s[i];
                if (c == '*') star = 1;
                else if (!strchr("ACGTUNacgtun", c)) { snprintf(err, errlen, "Invalid character %c in context for modification code %s in -c argument", c, code); return -1; }
                c = (char)toupper((unsigned char)c);
                if (c == 'U') c = 'T';
                if (k < sizeof(ctx) - 1) ctx[k++] = c;
                i++;
            }
            if (i >= n) { snprintf(err, errlen, "Context not closed with a ] for modification code %s in -c argument", code); return -1; }
            ctx[k] = 0;
            if (star && k > 1) { snprintf(err, errlen, "Invalid context for modification code %s. * should be the only character within [ and ] in -c argument", code); return -1; }
            i++;
            if (i < n && s[i] == ',') i++;
        } else {
            snprintf(ctx, sizeof ctx, "%s", default_context(code));
            MMH_INFO("Context not provided for modification code %s in -c argument. Using %s", code, ctx);
            if (i < n && s[i] == ',') i++;
        }
        if (strlen(ctx) >= MM_CODE_LEN) { snprintf(err, errlen, "Context %s is longer than %d characters", ctx, MM_CODE_LEN - 1); return -1; }
        for (int t = 0; t < out->n_mods; t++)
            if (strcmp(out->code[t], code) == 0) { snprintf(err, errlen, "Duplicate modification code %s found in -c argument", code); return -1; }
        if (out->n_mods >= MM_MAX_MODS) { snprintf(err, errlen, "At most %d modification codes are supported", MM_MAX_MODS); return -1; }
        snprintf(out->code[out->n_mods], MM_CODE_LEN, "%s", code);
        snprintf(out->context[out->n_mods], MM_CODE_LEN, "%s", ctx);
        out->n_mods++;
    }
    return 0;
}

int mmh_parse_mod_threshes(const char *s, mmh_mods_t *m, char *err, size_t errlen) {
    int nt = 0;
    double d = 0.0;
    const char *p = s;
    while (*p) {
        char tok[64]; size_t j = 0;
        while (*p && *p != ',') { if (j < sizeof(tok) - 1) tok[j++] = *p; p++; }
        tok[j] = 0;
        errno = 0;
        d = atof(tok);
        if (errno != 0) { snprintf(err, errlen, "Invalid threshold. You entered %s", tok); return -1; }
        if (d < 0 || d > 1) { snprintf(err, errlen, "Modification threshold should be in the range 0.0 to 1.0. You entered %f", d); return -1; }
        if (nt < m->n_mods) {
            m->thresh[nt] = d;
            MMH_INFO("Modification code: %s, Context: %s, Threshold: %f", m->code[nt], m->context[nt], d);
        }
        nt++;
        if (!*p) break;
        p++;
    }
    if (nt == 1) {
        for (int i = 0; i < m->n_mods; i++) m->thresh[i] = d;   /* one -m value is broadcast, mod.c:385-393 */
    } else if (nt != m->n_mods) {
        snprintf(err, errlen, "Number of modification codes and thresholds do not match. Codes:%d, Thresholds:%d", m->n_mods, nt);
        return -1;
    }
    return 0;
}

void mmh_klass_lut(double thresh, uint8_t lut[256]) {
    for (int x = 0; x < 256; x++) {
        double p = (double)((x + 0.5) / 256.0);
        lut[x] = (uint8_t)(p >= thresh ? 3 : (p <= 1 - thresh ? 1 : 0));
    }
}

void mmh_fill_opts(const mmh_mods_t *m, int insertions, int haplotypes, int device, mm_freq_opts_t *o) {
    memset(o, 0, sizeof(*o));
    o->abi_version = MM_ABI_VERSION;
    o->n_mods = m->n_mods; o->insertions = insertions; o->haplotypes = haplotypes; o->device = device;
    for (int i = 0; i < m->n_mods; i++) {
        snprintf(o->mods[i].code, MM_CODE_LEN, "%s", m->code[i]);
        snprintf(o->mods[i].context, MM_CODE_LEN, "%s", m->context[i]);
        mmh_klass_lut(m->thresh[i], o->mods[i].klass);
    }
}
