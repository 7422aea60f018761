/* debugging aid: a SIGSEGV handler that prints the native backtrace (module + offset; resolve with llvm-addr2line -e <module> <offset> on a -g build).
 * cc -shared -fPIC -o /tmp/segv_bt.so scripts/dev/segv_bt.c; ctypes.CDLL("/tmp/segv_bt.so").install() */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void handler(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}
void install(void) { signal(SIGSEGV, handler); signal(SIGABRT, handler); signal(SIGBUS, handler); }
