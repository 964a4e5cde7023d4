/* abort_bt.c -- TEST / DIAGNOSTIC INFRASTRUCTURE: who called abort()?  Loading this library (tests/conftest.py does, with ctypes;
 * LD_PRELOAD works too) installs a SIGABRT handler that prints the native backtrace of the aborting thread and then lets the
 * default action run.  It writes to a dup of stderr taken when it is loaded -- pytest's capture cannot swallow that -- and, when
 * ABORT_BT_LOG names a file, to that file too.  gcc -O1 -g -shared -fPIC -o abort_bt.so abort_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <stdlib.h>
#include <fcntl.h>
#include <sys/stat.h>
static int g_fd = 2, g_log = -1;
static struct sigaction g_prev;              /* whoever was there before (python's faulthandler): runs next */
static void on_abort(int sig)
{
  void *bt[64];
  const char m[] = "\n==== abort_bt: SIGABRT, native backtrace of the aborting thread ====\n";
  const int n = backtrace(bt, 64);
  if (write(g_fd, m, sizeof(m) - 1) < 0) {}
  backtrace_symbols_fd(bt, n, g_fd);
  if (g_log >= 0) { if (write(g_log, m, sizeof(m) - 1) < 0) {} backtrace_symbols_fd(bt, n, g_log); }
  /* what the process wrote to stderr last (the runtime's or glibc's own message): when fd 2 is a regular file -- pytest's capture --
   * its tail would die with the process */
  {
    struct stat st;
    static char buf[4096];
    if (fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      const off_t from = st.st_size > (off_t) sizeof(buf) ? st.st_size - (off_t) sizeof(buf) : 0;
      const ssize_t k = pread(2, buf, sizeof(buf), from);
      const char h[] = "==== abort_bt: the tail of the captured stderr ====\n";
      if (k > 0) {
        if (write(g_fd, h, sizeof(h) - 1) < 0 || write(g_fd, buf, (size_t) k) < 0) {}
        if (g_log >= 0) { if (write(g_log, h, sizeof(h) - 1) < 0 || write(g_log, buf, (size_t) k) < 0) {} }
      }
    }
  }
  sigaction(sig, &g_prev, NULL);
  raise(sig);
}
__attribute__((constructor)) static void init(void)
{
  const char *p = getenv("ABORT_BT_LOG");
  struct sigaction sa;
  void *warm[4];
  g_fd = dup(2);
  if (g_fd < 0) g_fd = 2;
  if (p && *p) g_log = open(p, O_WRONLY | O_CREAT | O_APPEND, 0644);
  (void) backtrace(warm, 4);                 /* (loads libgcc's unwinder now: not inside the handler) */
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_abort;
  sa.sa_flags = SA_NODEFER;
  sigaction(SIGABRT, &sa, &g_prev);
}
