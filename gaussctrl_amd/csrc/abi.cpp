// abi.cpp -- error channel and version of libgaussctrl_hip.so.
#include "common.h"
#include <stdarg.h>
#include <string.h>

namespace gc {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gc

extern "C" {
const char *gc_last_error_string(void) { return gc::g_err; }
int gc_abi_version(void) { return 1; }
}
