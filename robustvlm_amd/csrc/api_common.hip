// Error plumbing + version of the C ABI.
#include "common.h"

namespace rvlm {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace rvlm

extern "C" const char* rvlm_last_error(void) { return rvlm::g_last_error.c_str(); }
extern "C" int rvlm_version(void) { return RVLM_VERSION; }
