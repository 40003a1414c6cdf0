// Shared host-side helpers of libhydra_mi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/hydra_mi.h"

void hm_set_error(const char *fmt, ...);

#define HM_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            hm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                         __FILE__, __LINE__);                                         \
            return HM_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define HM_ARG(cond, ...)                \
    do {                                 \
        if (!(cond)) {                   \
            hm_set_error(__VA_ARGS__);   \
            return HM_ERR_ARG;           \
        }                                \
    } while (0)

static inline int hm_cdiv(int a, int b) { return (a + b - 1) / b; }
