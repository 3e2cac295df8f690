/* exitpath.c -- how a process that held the GPU leaves (minimod's main() and the workers of --devices). */
#define _GNU_SOURCE
#include <errno.h>
#include <sched.h>
#include <stdlib.h>
#include <unistd.h>

#include "mmhost.h"

int mmh_gpu_in_use = 0;   /* set once the HIP runtime is up in this process */

/* The last one out.  When a process that used the GPU dies, the kernel takes its address space apart before the caller's wait() returns: every
 * hardware queue's save area (173 MB of host memory a queue), the pinned staging, the driver's own mappings -- 0.10 - 0.19 s after the run's last
 * word (tools/exit_probe*.hip, MM_TIMELINE's "[outside]" line in bench.py --e2e-gbases), and nothing of it is the run's work: the output is written
 * and closed.  A helper that SHARES the address space (clone(CLONE_VM): no copy, no thread of this process) keeps it alive past this process's
 * death: the process is reaped at once, the helper -- which holds no descriptor but its end of a pipe, so that nobody's read of our stdout or stderr
 * waits for it -- sees the pipe close, leaves, and the address space is taken apart THEN, with nobody waiting.  The GPU's memory and queues are
 * given back those 0.1 - 0.2 s later (a run started right behind another finds the device busy with that either way).
 * Off with MM_SYNC_EXIT=1 (and with MM_FULL_TEARDOWN, which runs every destructor instead): bench.py reports both walls. */
static int exit_pipe_rd = -1;
static int last_one_out(void *arg) {
    (void)arg;
    for (int fd = 0; fd < 4096; fd++) if (fd != exit_pipe_rd) close(fd);
    char c;
    while (read(exit_pipe_rd, &c, 1) < 0 && errno == EINTR) { }
    _exit(0);
}
void mmh_leave_teardown_behind(void) {
    if (getenv("MM_SYNC_EXIT") || !mmh_gpu_in_use) return;   /* (a run that never touched the GPU dies in a millisecond; the helper ends as a zombie if nobody reaps orphans) */
    int pfd[2];
    if (pipe(pfd) != 0) return;
    const size_t stack_bytes = 256 * 1024;
    char *stack = (char *)malloc(stack_bytes);
    if (!stack) { close(pfd[0]); close(pfd[1]); return; }
    exit_pipe_rd = pfd[0];
    if (clone(last_one_out, stack + stack_bytes, CLONE_VM, NULL) < 0) { close(pfd[0]); close(pfd[1]); free(stack); return; }
    close(pfd[0]);   /* (the write end closes when this process is gone) */
}

