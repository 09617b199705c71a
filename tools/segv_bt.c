// Debug aid: prints the native call stack on SIGSEGV (loaded with ctypes when DM_SEGV_BT=1), then re-raises.
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
  void* frames[64];
  int n = backtrace(frames, 64);
  const char* msg = "\n==== native backtrace (SIGSEGV) ====\n";
  write(2, msg, strlen(msg));
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
void dm_install_segv_bt(void) {
  static char alt[1 << 16];
  stack_t ss;
  ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = handler;
  sa.sa_flags = SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
}
