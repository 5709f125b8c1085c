/* Test infrastructure (not product code): makes a native fault under pytest diagnosable.
 *
 * A SIGABRT / SIGSEGV / SIGBUS raised inside the HIP runtime or libcenterface_hip.so kills the interpreter; Python's faulthandler then
 * prints every Python thread's stack, which pushes the one useful line -- WHICH test, WHICH native frames -- out of the tail a CI
 * log keeps.  This handler is installed BEFORE faulthandler (tests/conftest.py), so it runs LAST: it prints the test id the
 * conftest last stored with cf_fault_note() and the native backtrace of the faulting thread (module + offset per frame), then
 * lets the default disposition end the process.  Also usable as an LD_PRELOAD (tools/diag/repro.sh).
 * Build: gcc -O1 -g -shared -fPIC -o _native_fault.so native_fault.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/syscall.h>

static char g_note[512] = "(no test id recorded)";
static int g_fd = 2;          /* where to write: the conftest passes a dup of the REAL stderr (pytest redirects fd 2 while a test runs) */

void cf_fault_note(const char* s) {
    if (!s) return;
    strncpy(g_note, s, sizeof g_note - 1);
    g_note[sizeof g_note - 1] = 0;
}

static void handler(int sig, siginfo_t* si, void* uc) {
    (void)uc;
    void* frames[96];
    char head[800];
    int n = snprintf(head, sizeof head, "\n=== native fault: signal %d in thread %ld of pid %d (si_addr %p) while running: %s ===\n", sig,
                     (long)syscall(SYS_gettid), (int)getpid(), si ? si->si_addr : (void*)0, g_note);
    if (n > 0 && write(g_fd, head, (size_t)n) < 0) { }
    int k = backtrace(frames, 96);
    backtrace_symbols_fd(frames, k, g_fd);
    n = snprintf(head, sizeof head, "=== native fault: end (test: %s) ===\n", g_note);
    if (n > 0 && write(g_fd, head, (size_t)n) < 0) { }
    signal(sig, SIG_DFL);
    raise(sig);
}

void cf_fault_install(int fd) {
    if (fd >= 0) g_fd = fd;
    void* warm[4];
    backtrace(warm, 4);                      /* loads libgcc now, not inside the handler */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}

__attribute__((constructor)) static void on_load(void) {
    if (getenv("CF_FAULT_PRELOAD")) cf_fault_install(2);      /* LD_PRELOAD use: install at load time */
}
