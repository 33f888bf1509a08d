/* bench.py's safety net -- bench-only code, not part of any product library (round-4 advisor).
 * The multi-rank variants bench.py tries after its plain-launch measurement (captured collectives, peer stores) have never run across GPUs;
 * a GPU memory fault in one of them ends the process through abort() inside the HSA runtime, from where no Python handler runs.
 * gsfm_crash_line_arm() keeps the JSON line measured BEFORE the variant; SIGABRT / SIGSEGV / SIGBUS / SIGFPE then write that line -- with the
 * signal number patched into its "crashed_variant": {..., "signal": 00} field, so the line itself says that and how the variant died -- to `fd`
 * and leave (async-signal-safe: write + _exit only).  gsfm_crash_line_disarm() restores the default dispositions.
 * Built by __graft_entry__.build() into globalsfmpy_amd/libgsfm_benchguard.so. */
#include <signal.h>
#include <string.h>
#include <unistd.h>

static char g_line[1 << 16];
static volatile size_t g_len = 0, g_sig_off = 0;
static volatile int g_fd = -1;

static void crash_handler(int sig) {
  if (g_fd >= 0 && g_len) {
    if (g_sig_off && g_sig_off + 2 <= g_len) { g_line[g_sig_off] = sig >= 10 ? (char)('0' + (sig / 10) % 10) : ' ';   /* (a leading zero is not JSON) */ g_line[g_sig_off + 1] = (char)('0' + sig % 10); }
    ssize_t r = write(g_fd, g_line, g_len); (void)r;
  }
  _exit(0);   /* the line that was measured stands; the variant's death is IN the line */
}

/* sig_off: offset of the two-digit placeholder inside `line` (0 = none) */
int gsfm_crash_line_arm(int fd, const char* line, size_t len, size_t sig_off) {
  static const int sigs[4] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE};
  if (len >= sizeof(g_line)) return 1;
  memcpy(g_line, line, len);
  g_len = len; g_fd = fd; g_sig_off = sig_off;
  for (int k = 0; k < 4; ++k) signal(sigs[k], crash_handler);
  return 0;
}
void gsfm_crash_line_disarm(void) {
  static const int sigs[4] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE};
  for (int k = 0; k < 4; ++k) signal(sigs[k], SIG_DFL);
  g_fd = -1; g_len = 0; g_sig_off = 0;
}
