/* main.c -- `minimod` command dispatch (reference src/main.c:46-98).  `freq` and `view` run on the GPU path; `summary`
 * (a census of MM group headers) stays on the host. */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mmhost.h"

static int print_usage(FILE *fp) {
    fprintf(fp, "Usage: minimod <command> [options]\n\n");
    fprintf(fp, "command:\n");
    fprintf(fp, "         freq       output base modification frequencies (MI355X hot path)\n");
    fprintf(fp, "         view       view base modifications (MI355X hot path)\n");
    fprintf(fp, "         summary    print the modification types of every read (host only)\n");
    return fp == stdout ? EXIT_SUCCESS : EXIT_FAILURE;
}

int main(int argc, char *argv[]) {
    double realtime0 = mmh_realtime();
    int ret = 1;
    if (argc < 2) return print_usage(stderr);
    if (strcmp(argv[1], "freq") == 0) ret = mmh_freq_main(argc - 1, argv + 1);
    else if (strcmp(argv[1], "mod-freq") == 0) { MMH_WARNING("%s", "mod-freq is deprecated. Use freq instead"); ret = mmh_freq_main(argc - 1, argv + 1); }
    else if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-V") == 0) { fprintf(stdout, "minimod %s\n", MMH_VERSION); exit(EXIT_SUCCESS); }
    else if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "-h") == 0) return print_usage(stdout);
    else if (strcmp(argv[1], "view") == 0) ret = mmh_view_main(argc - 1, argv + 1);
    else if (strcmp(argv[1], "summary") == 0) ret = mmh_summary_main(argc - 1, argv + 1);
    else { fprintf(stderr, "[minimod] Unrecognised command %s\n", argv[1]); return print_usage(stderr); }
    fprintf(stderr, "[%s] Version: %s\n", __func__, MMH_VERSION);
    fprintf(stderr, "[%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[%s] Real time: %.3f sec; CPU time: %.3f sec; Peak RAM: %.3f GB\n\n", __func__, mmh_realtime() - realtime0,
            mmh_cputime(), mmh_peakrss() / 1024.0 / 1024.0 / 1024.0);
    /* (freq_main.c, run_body's end) the output is complete and flushed: no exit handlers, the process's death frees what is left */
    if (!getenv("MM_FULL_TEARDOWN")) { fflush(NULL); mmh_leave_teardown_behind();   /* (exitpath.c: a process that held the GPU is reaped at once, its address space is taken apart behind it) */ _exit(ret); }
    return ret;
}
