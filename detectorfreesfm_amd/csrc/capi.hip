// Library-level entry points of libdfsfm_hip.so (version, error text).
#include "common.h"
#include <cstdio>

namespace dfsfm {
namespace {
thread_local char g_err[256] = "";
}
void set_last_error(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
}  // namespace dfsfm

extern "C" int dfsfm_version(void) { return 1; }
extern "C" const char* dfsfm_last_error_string(void) { return dfsfm::g_err; }
