// Shared host-side helpers of libhydra_mi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
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

// Every device allocation of the library.  HYDRA_MI_POISON=<mask> (development aid) fills those of the classes in the
// mask -- 1 the filter's context, 2 a flow handle, 4 hm_dev_alloc (the caller's buffers: frame ring, flow planes) --
// with 0xFF bytes: NaNs as floating point, -1 as integers, so that a kernel that reads memory nobody has written shows
// at once and every time, not only when the allocator hands back a block an earlier handle left its numbers in.
#ifndef HM_ALLOC_CLASS
#define HM_ALLOC_CLASS 1
#endif
static inline hipError_t hm_malloc(void **p, size_t bytes, int cls = HM_ALLOC_CLASS)
{
    static const int poison = getenv("HYDRA_MI_POISON") ? atoi(getenv("HYDRA_MI_POISON")) : 0;
    // HYDRA_MI_POISON_ONLY=k: of the allocations of those classes (counted per translation unit, from 0) only the k-th
    // -- to find the buffer a failure under HYDRA_MI_POISON comes from
    static const int only = getenv("HYDRA_MI_POISON_ONLY") ? atoi(getenv("HYDRA_MI_POISON_ONLY")) : -1;
    static int count = 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && (poison & cls)) {
        const int k = count++;
        if (only < 0 || only == k) {
            e = hipMemset(*p, 0xFF, bytes);
            if (e == hipSuccess) e = hipDeviceSynchronize();      // before anything a non-blocking stream writes there
        }
        if (only == k) fprintf(stderr, "[hydra_mi] poisoned allocation %d: %zu bytes\n", k, bytes);
    }
    return e;
}
