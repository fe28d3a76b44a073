/* abort_trace.c -- development aid: LD_PRELOAD this and a process that dies of SIGABRT / SIGSEGV prints the native call stack of the dying thread first
   (the GPU suite once died inside a library call with no message at all).  gcc -shared -fPIC -O1 -g scripts/abort_trace.c -o scripts/bin/libabort_trace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_signal(int sig)
{
  void* frames[64];
  const char head[] = "\n==== native stack at the fatal signal ====\n";
  /* (not to descriptor 2: pytest has it redirected into a capture file while a test runs) */
  int fd = open("/tmp/abort_trace.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) fd = 2;
  (void)!write(fd, head, sizeof(head) - 1);
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, fd);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void)
{
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_signal;
  sa.sa_flags = SA_NODEFER | SA_RESETHAND;
  sigaction(SIGABRT, &sa, 0);
}
