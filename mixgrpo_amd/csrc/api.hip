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
// 100 * major + minor of the C ABI; NEGATIVE for a diagnostic build (common.h): such a library computes wrong results on
// purpose and the host binding refuses to load it
#ifdef MGX_DIAGNOSTIC_BUILD
extern "C" int mgx_version(void) { return -101; }
#else
extern "C" int mgx_version(void) { return 101; }
#endif
