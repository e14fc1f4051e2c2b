// pmx_core.hip — error reporting and library-level queries.
#include <stdarg.h>

#include "pmx_common.h"

namespace pmx {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pmx

extern "C" const char *pmx_last_error(void) { return pmx::g_err; }
extern "C" int pmx_version(void) { return 100; }
extern "C" int pmx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
