// Version and thread-local error string of the C ABI.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <cstring>

static thread_local char g_err[512] = "";

extern "C" void mgx_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mgx_last_error(void) { return g_err; }
extern "C" int mgx_version(void) { return 100; }
