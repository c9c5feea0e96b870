// Error plumbing of the C ABI (include/mxdenoise.h).
#include <string>

#include "../../include/mxdenoise.h"
#include "common.h"

namespace mx {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace mx

extern "C" const char* mx_last_error(void) { return mx::g_err.c_str(); }
extern "C" int mx_version(void) { return 1; }
