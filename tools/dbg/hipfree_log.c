/* LD_PRELOAD aid: log every hipFree (who unmaps device memory, and when) with a short native backtrace to $HIPFREE_LOG.
 *   gcc -shared -fPIC -o /tmp/hipfree_log.so tools/dbg/hipfree_log.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>

typedef int (*hipFree_t)(void*);
static int g_fd = -2, g_n = 0;

int hipFree(void* p) {
  static hipFree_t real = 0;
  if (!real) real = (hipFree_t)dlsym(RTLD_NEXT, "hipFree");
  if (g_fd == -2) {
    const char* path = getenv("HIPFREE_LOG");
    g_fd = path ? open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : -1;
  }
  if (g_fd >= 0 && p) {
    char line[96];
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    int n = snprintf(line, sizeof line, "hipFree #%d %p t=%ld.%03ld\n", ++g_n, p, (long)ts.tv_sec, ts.tv_nsec / 1000000);
    (void)!write(g_fd, line, (size_t)n);
    if (g_n <= 6 || g_n % 200 == 0) {
      void* bt[24];
      int k = backtrace(bt, 24);
      backtrace_symbols_fd(bt, k, g_fd);
    }
  }
  return real(p);
}
