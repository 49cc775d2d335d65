/* Test infrastructure: a SIGSEGV / SIGABRT / SIGBUS handler that writes the NATIVE call stack (and the load addresses of the libraries of this
 * repository) to stderr before the default action runs.  Python's faulthandler stops at the ctypes call; a crash inside libflucahip.so or
 * libfluca_host.so on the GPU box is otherwise a bare "Segmentation fault".  Loaded by tests/conftest.py when FLUCA_TEST_BACKTRACE=1. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static int out_fd = 2; /* FLUCA_TEST_BACKTRACE_FILE, opened at installation: pytest captures fd 2 and a crash loses what it holds */
static void put(const char *s) { (void)!write(out_fd, s, strlen(s)); }

static void on_fault(int sig)
{
  void *frames[64];
  put("\n==== native backtrace (tests/plugins/segv_trace.c) ====\n");
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, out_fd);
  put("==== mappings of the repository's libraries ====\n");
  const int fd = open("/proc/self/maps", O_RDONLY);
  if (fd >= 0) {
    static char buf[1 << 16];
    static char line[1024];
    ssize_t     got;
    size_t      len = 0;
    while ((got = read(fd, buf, sizeof buf)) > 0)
      for (ssize_t i = 0; i < got; ++i) {
        if (buf[i] != '\n' && len + 1 < sizeof line) { line[len++] = buf[i]; continue; }
        line[len] = 0;
        if ((strstr(line, "libfluca") || strstr(line, "librccl") || strstr(line, "inproc_comm") || strstr(line, "libamdhip")) && strstr(line, "r-xp")) { put(line); put("\n"); }
        len = 0;
      }
    close(fd);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

int segv_trace_install(const char *path)
{
  if (path && *path) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) out_fd = fd;
  }
  void *warm[4];
  (void)backtrace(warm, 4); /* loads libgcc now: not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fault;
  sa.sa_flags   = SA_NODEFER | SA_RESETHAND;
  sigemptyset(&sa.sa_mask);
  int rc = 0;
  rc |= sigaction(SIGSEGV, &sa, NULL);
  rc |= sigaction(SIGBUS, &sa, NULL);
  rc |= sigaction(SIGABRT, &sa, NULL);
  return rc;
}
