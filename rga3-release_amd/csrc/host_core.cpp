// Error plumbing and version for librga3_hip.so (C ABI: include/rga3_hip.h).  Host-only, no HIP headers (also built under the CPU sanitizers).
#include "errors.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace rga3 {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code == 0 ? RGA3_EINVAL : code;
}

}  // namespace rga3

extern "C" int rga3_version(void) { return 1; }

extern "C" int rga3_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return 0;
    size_t len = strlen(rga3::g_err);
    if (len >= n) len = n - 1;
    memcpy(buf, rga3::g_err, len);
    buf[len] = 0;
    return (int)len;
}
