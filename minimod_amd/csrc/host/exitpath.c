/* exitpath.c -- how a process that held the GPU leaves (minimod's main() and the workers of --devices). */
#define _GNU_SOURCE
#include <errno.h>
#include <sched.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "mmhost.h"

int mmh_gpu_in_use = 0;   /* set once the HIP runtime is up in this process */

/* The last one out -- OFF unless MM_ASYNC_EXIT=1 (round 6: the default is the plain exit, and bench.py's end-to-end wall is the one that includes the
 * teardown).  When a process that used the GPU dies, the kernel takes its address space apart before the caller's wait() returns: every hardware queue's
 * save area (173 MB of host memory a queue), the pinned staging, the driver's own mappings -- 0.10 - 0.19 s after the run's last word (tools/exit_probe*.hip).
 * With MM_ASYNC_EXIT=1 a helper that SHARES the address space (clone(CLONE_VM): no copy, no thread of this process) keeps it alive past this process's
 * death: the process is reaped at once, the helper -- which holds no descriptor but the read end of a pipe -- sees the pipe close, leaves, and the address
 * space is taken apart then, with nobody waiting.  What that costs is why it is not the default: the GPU's memory and queues stay held 0.1 - 0.2 s by
 * something no caller can wait for (a scheduler that starts the next job when wait() returns may find the device short of memory), and a container
 * without an init keeps a zombie a run.  On the round-5 driver's box it was worth 2 ms. */
static int exit_pipe_rd = -1;
static int last_one_out(void *arg) {
    (void)arg;
    /* every descriptor but the pipe's read end, whatever its number (the write end among them: the helper must not hold its own EOF back) */
    if (exit_pipe_rd > 0) syscall(SYS_close_range, 0u, (unsigned)exit_pipe_rd - 1u, 0u);
    if (syscall(SYS_close_range, (unsigned)exit_pipe_rd + 1u, ~0u, 0u) != 0) {   /* (a kernel without close_range: /proc/self/fd says how far to go) */
        long top = sysconf(_SC_OPEN_MAX);
        if (top < 0 || top > (1 << 20)) top = 1 << 20;
        for (long fd = 0; fd < top; fd++) if (fd != exit_pipe_rd) close((int)fd);
    }
    char c;
    while (read(exit_pipe_rd, &c, 1) < 0 && errno == EINTR) { }
    _exit(0);
}
void mmh_leave_teardown_behind(void) {
    const char *e = getenv("MM_ASYNC_EXIT");
    if (!e || !*e || *e == '0' || getenv("MM_SYNC_EXIT") || !mmh_gpu_in_use) return;   /* (MM_SYNC_EXIT: rounds 5's switch, still honoured) */
    int pfd[2];
    if (pipe(pfd) != 0) return;
    const size_t stack_bytes = 256 * 1024;
    char *stack = (char *)malloc(stack_bytes);
    if (!stack) { close(pfd[0]); close(pfd[1]); return; }
    exit_pipe_rd = pfd[0];
    if (clone(last_one_out, stack + stack_bytes, CLONE_VM, NULL) < 0) { close(pfd[0]); close(pfd[1]); free(stack); return; }
    close(pfd[0]);   /* (the write end closes when this process is gone) */
}
