// Error plumbing of the C ABI: a thread-local message behind az_last_error().
#include <string>

#include "../../include/azamd.h"

namespace az {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace az

extern "C" const char* az_last_error(void) { return az::g_last_error.c_str(); }
extern "C" int az_version(void) { return 1; }
