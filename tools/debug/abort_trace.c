/* LD_PRELOAD shim for chasing an abort() inside the library on a GPU box without a debugger: prints the C backtrace
 * of the aborting thread on stderr.  gcc -shared -fPIC -o abort_trace.so abort_trace.c ; run pytest with -p no:faulthandler. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "\n--- abort_trace: backtrace of the aborting thread ---\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
}
