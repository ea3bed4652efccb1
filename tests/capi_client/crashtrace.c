/* Test tooling: prints the native call stack on SIGSEGV / SIGBUS / SIGABRT before handing the
 * signal on to the handler that was installed before (Python's faulthandler under pytest), so a
 * crash inside a shared library names its frames in the test log.  Loaded by tests/conftest.py. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_old[3];
static const int g_sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};

static void on_fatal(int sig, siginfo_t *info, void *ctx) {
    void *bt[96];
    static const char msg[] = "\n==== native backtrace (tests/capi_client/crashtrace.c) ====\n";
    int i;
    if (write(2, msg, sizeof(msg) - 1) < 0) _exit(128 + sig);
    backtrace_symbols_fd(bt, backtrace(bt, 96), 2);
    for (i = 0; i < 3; ++i)
        if (g_sigs[i] == sig) {
            sigaction(sig, &g_old[i], NULL);            /* put the previous handler back and re-raise */
            if ((g_old[i].sa_flags & SA_SIGINFO) && g_old[i].sa_sigaction) { g_old[i].sa_sigaction(sig, info, ctx); return; }
            if (g_old[i].sa_handler != SIG_DFL && g_old[i].sa_handler != SIG_IGN) { g_old[i].sa_handler(sig); return; }
        }
    raise(sig);
}

int crashtrace_install(void) {
    struct sigaction sa;
    int i;
    void *warm[4];
    backtrace(warm, 4);                                 /* loads libgcc now, not inside the handler */
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    for (i = 0; i < 3; ++i)
        if (sigaction(g_sigs[i], &sa, &g_old[i]) != 0) return -1;
    return 0;
}
