/* LD_PRELOAD aid: when the process calls abort(), save the tail of whatever file fd 2 points at (under pytest's fd capture: the
 * temporary file that swallows the HSA runtime's "Memory access fault by GPU ..." line) to $ABORT_TEE_OUT, then abort for real.
 *   gcc -shared -fPIC -o /tmp/abort_tee.so tools/dbg/abort_tee.c -ldl
 *   ABORT_TEE_OUT=gpurun_out/abort_stderr.txt LD_PRELOAD=/tmp/abort_tee.so python -m pytest ... */
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

void abort(void) {
  const char* out = getenv("ABORT_TEE_OUT");
  int in = open("/proc/self/fd/2", O_RDONLY);
  int o = open(out ? out : "/tmp/abort_stderr.txt", O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (in >= 0 && o >= 0) {
    off_t end = lseek(in, 0, SEEK_END);
    off_t start = end > 16384 ? end - 16384 : 0;
    lseek(in, start, SEEK_SET);
    char buf[4096];
    ssize_t n;
    while ((n = read(in, buf, sizeof buf)) > 0) (void)!write(o, buf, (size_t)n);
    const char* tail = "\n[abort_tee] abort() called\n";
    (void)!write(o, tail, strlen(tail));
  }
  if (o >= 0) close(o);
  raise(SIGABRT);
  _exit(134);
}
