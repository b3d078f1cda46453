// Shared host-side helpers for libnrf.so (error reporting, launch checks).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nrf.h"

namespace nrf {

char *last_error_buf();  // thread-local, 512 bytes

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NRF_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NRF_OK;
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;

inline int grid_for(int64_t work_items, int per_block, int max_blocks = 256 * 8) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace nrf

#define NRF_REQUIRE(cond, ...)                                        \
    do {                                                              \
        if (!(cond)) return nrf::fail(NRF_ERR_INVALID, __VA_ARGS__);  \
    } while (0)
