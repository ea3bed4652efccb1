/* Debug tooling, linked only into -DMMC_WATCH builds of the library (tools/build_variant.sh):
 * a hardware write watchpoint on one 8-byte word of host memory, armed on every thread of the
 * process (perf_event_open, PERF_TYPE_BREAKPOINT).  A thread that stores to the word gets a
 * signal and prints its own call stack, so a stray writer names itself.  Stores by a device
 * (DMA) or by the kernel do not trigger it: a changed word without a report points there. */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <execinfo.h>
#include <fcntl.h>
#include <linux/hw_breakpoint.h>
#include <linux/perf_event.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>

enum { MAX_THREADS = 256 };
static int g_fds[MAX_THREADS];
static int g_nfds = 0;
static volatile uint64_t *g_addr = NULL;
static int g_installed = 0, g_hits = 0, g_warned = 0;
static uint64_t g_expect = 0;
static int g_no_kernel = 0;
static int g_reported = 0;

static void on_hit(int sig, siginfo_t *info, void *ctx) {
    char line[160];
    void *bt[64];
    int n;
    (void)sig; (void)info;
    n = snprintf(line, sizeof(line), "\n==== mmc_watch: store to %p by thread %ld near pc %p, word now 0x%llx (hit %d) ====\n",
                 (void *)g_addr, (long)syscall(SYS_gettid), (void *)((ucontext_t *)ctx)->uc_mcontext.gregs[REG_RIP],
                 g_addr ? (unsigned long long)*g_addr : 0ull, ++g_hits);
    if (write(2, line, (size_t)n) < 0) return;
    backtrace_symbols_fd(bt, backtrace(bt, 64), 2);
}

void mmc_watch_disarm(void) {
    int i;
    for (i = 0; i < g_nfds; ++i) close(g_fds[i]);
    g_nfds = 0;
    g_addr = NULL;
}

int mmc_watch_hits(void) { return g_hits; }

/* Compares the watched word with the value it had when it was armed; names the first place where
 * it differs (for stores that the watchpoint cannot see: kernel mode, devices). */
void mmc_watch_check(const char *where) {
    int nthreads = 0;
    DIR *d;
    struct dirent *e;
    if (!g_addr || g_reported || *g_addr == g_expect) return;
    g_reported = 1;
    d = opendir("/proc/self/task");
    if (d) { while ((e = readdir(d)) != NULL) if (atoi(e->d_name) > 0) ++nthreads; closedir(d); }
    fprintf(stderr, "\n==== mmc_watch_check: word at %p is 0x%llx, was 0x%llx when armed; first seen at '%s'; "
            "watch hits so far %d; threads now %d, armed on %d ====\n", (void *)g_addr, (unsigned long long)*g_addr,
            (unsigned long long)g_expect, where, g_hits, nthreads, g_nfds);
}

int mmc_watch_arm(void *addr) {
    struct perf_event_attr pe;
    DIR *d;
    struct dirent *e;
    const int signo = SIGRTMIN + 4;
    mmc_watch_disarm();
    if (!g_installed) {
        struct sigaction sa;
        void *warm[4];
        backtrace(warm, 4);
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_hit;
        sa.sa_flags = SA_SIGINFO | SA_RESTART;
        sigemptyset(&sa.sa_mask);
        if (sigaction(signo, &sa, NULL) != 0) return -1;
        g_installed = 1;
    }
    g_addr = (volatile uint64_t *)addr;
    g_expect = *g_addr;
    g_reported = 0;
    memset(&pe, 0, sizeof(pe));
    pe.type = PERF_TYPE_BREAKPOINT;
    pe.size = sizeof(pe);
    pe.bp_type = HW_BREAKPOINT_W;
    pe.bp_addr = (uint64_t)(uintptr_t)addr;
    pe.bp_len = HW_BREAKPOINT_LEN_8;
    pe.sample_period = 1;
    pe.wakeup_events = 1;
    pe.exclude_kernel = 1;
    pe.exclude_hv = 1;
    d = opendir("/proc/self/task");
    if (!d) return -1;
    while ((e = readdir(d)) != NULL && g_nfds < MAX_THREADS) {
        struct f_owner_ex ow;
        int tid = atoi(e->d_name), fd;
        if (tid <= 0) continue;
        pe.exclude_kernel = g_no_kernel;
        fd = (int)syscall(SYS_perf_event_open, &pe, tid, -1, -1, 0);
        if (fd < 0 && !g_no_kernel) {          /* kernel-mode stores need perf_event_paranoid <= 1 */
            g_no_kernel = 1;
            pe.exclude_kernel = 1;
            fd = (int)syscall(SYS_perf_event_open, &pe, tid, -1, -1, 0);
        }
        if (fd < 0) {
            if (!g_warned) { fprintf(stderr, "mmc_watch: perf_event_open(tid %d): %s\n", tid, strerror(errno)); g_warned = 1; }
            continue;
        }
        ow.type = F_OWNER_TID;
        ow.pid = tid;
        if (fcntl(fd, F_SETFL, O_ASYNC) < 0 || fcntl(fd, F_SETSIG, signo) < 0 || fcntl(fd, F_SETOWN_EX, &ow) < 0) { close(fd); continue; }
        g_fds[g_nfds++] = fd;
    }
    closedir(d);
    { static int told = 0; if (!told) { fprintf(stderr, "mmc_watch: armed on %d threads (%s)\n", g_nfds, g_no_kernel ? "user-mode stores only" : "user- and kernel-mode stores"); told = 1; } }
    return g_nfds;
}

/* A canary chunk of the size that was seen corrupted (912 bytes: the block table of a six-block group
 * in the build that crashed): allocated where that table was allocated, filled with a pattern, its word
 * at offset 152 under the watchpoint, compared against the pattern at every check. */
enum { CANARY_BYTES = 912, CANARY_WATCH_OFFSET = 152 };
static unsigned char *g_canary = NULL;
static int g_canary_reported = 0;

void mmc_canary_new(void) {
    g_canary = (unsigned char *)malloc(CANARY_BYTES);
    if (!g_canary) return;
    memset(g_canary, 0x5a, CANARY_BYTES);
    g_canary_reported = 0;
    mmc_watch_arm(g_canary + CANARY_WATCH_OFFSET);
}

void mmc_canary_check(const char *where) {
    int i, shown = 0;
    if (!g_canary || g_canary_reported) return;
    for (i = 0; i < CANARY_BYTES; ++i)
        if (g_canary[i] != 0x5a) {
            if (!shown) fprintf(stderr, "\n==== mmc_canary: chunk %p changed, first seen at '%s' (watch hits so far %d):", (void *)g_canary, where, g_hits);
            if (shown < 64) fprintf(stderr, " [%d]=%02x", i, g_canary[i]);
            ++shown;
        }
    if (shown) { fprintf(stderr, " (%d bytes differ) ====\n", shown); g_canary_reported = 1; }
}

void mmc_canary_free(void) {
    mmc_canary_check("group destroy");
    mmc_watch_disarm();
    free(g_canary);
    g_canary = NULL;
}
