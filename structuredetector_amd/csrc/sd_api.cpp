// Error string + version for libsdnet_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sdnet_hip.h"

namespace sd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sd

extern "C" {
int sd_version(void) { return 1; }
const char* sd_last_error(void) { return sd::g_err; }
}
