/* Diagnostic only (LD_PRELOAD): prints the NATIVE call stack of the thread that takes SIGSEGV / SIGBUS / SIGABRT to stderr
   (glibc backtrace_symbols_fd: async-signal-safe enough for a process that is dying anyway), then lets the default action run.
   Python's faulthandler shows where the interpreter was; this shows where inside libamdhip64 / libtorch the fault is.
   Build: gcc -O1 -g -shared -fPIC -o tools/native/libsegvbt.so tools/native/segv_backtrace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* ctx) {
    (void)ctx;
    static const char head[] = "\n==== native backtrace (tools/native/segv_backtrace.c) ====\n";
    void* frames[96];
    write(2, head, sizeof head - 1);
    if (si) {
        char buf[64] = "fault address: 0x";
        unsigned long a = (unsigned long)si->si_addr;
        int n = (int)strlen(buf);
        for (int s = 60; s >= 0; s -= 4) buf[n++] = "0123456789abcdef"[(a >> s) & 15];
        buf[n++] = '\n';
        write(2, buf, n);
    }
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND | SA_ONSTACK;
    static char stack[1 << 16];
    stack_t ss = { .ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0 };
    sigaltstack(&ss, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
