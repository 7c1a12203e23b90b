// Error convention of librga3_hip.so: 0 = ok, negative code otherwise; the message is thread-local (rga3_last_error).
// No HIP headers: shared by the device sources (through common.h) and by the host-only C++ files that also build under the CPU sanitizers.
#pragma once
#include <stdint.h>
#include <stddef.h>

#include "../../include/rga3_hip.h"

namespace rga3 {

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define RGA3_CHECK_ARG(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::rga3::fail(RGA3_EINVAL, __VA_ARGS__); \
    } while (0)

}  // namespace rga3
