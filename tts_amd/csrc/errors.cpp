// Thread-local last-error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#include "../../include/ttsx.h"

static thread_local char g_err[512] = "no error";

void ttsx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ttsx_last_error_string(void) { return g_err; }
extern "C" int ttsx_version(void) { return TTSX_VERSION; }

// One team kernel at a time per process: a team needs most of the chip's CUs resident together, and two teams that
// were dispatched into each other would both wait for CUs the other holds until their bounded waits give up.
// Held from the launch until the run's stream synchronisation.  (Per process, not per device: the launches of two
// GPUs do not compete, but a request thread holds the lock for a few milliseconds at most.)
std::mutex& ttsx_team_mutex() {
  static std::mutex m;
  return m;
}

// Testing hook: kinds of bounded in-launch wait whose next run is to report an expired wait.
static std::atomic<unsigned> g_fail_waits{0};
extern "C" int ttsx_debug_fail_next_wait(unsigned kinds) {
  if (kinds & ~15u) {
    ttsx_set_error("ttsx_debug_fail_next_wait: kinds=%u holds bits outside the four TTSX_WAIT_ values", kinds);
    return TTSX_ERR_ARG;
  }
  g_fail_waits.fetch_or(kinds);
  return TTSX_OK;
}
bool ttsx_take_wait_failure(unsigned kind) { return (g_fail_waits.fetch_and(~kind) & kind) != 0; }
