// Shared helpers for the gfx950 kernels of libmrcnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mrcnn_hip.h"

namespace mrcnn {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define MRCNN_HIP_TRY(expr)                                              \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            mrcnn::set_error("%s: %s", #expr, hipGetErrorString(e_));    \
            return 1;                                                    \
        }                                                                \
    } while (0)

#define MRCNN_REQUIRE(cond, ...)                                         \
    do {                                                                 \
        if (!(cond)) {                                                   \
            mrcnn::set_error(__VA_ARGS__);                               \
            return 2;                                                    \
        }                                                                \
    } while (0)

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mrcnn
