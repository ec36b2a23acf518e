// common.h -- shared device/host helpers for libeld_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eld_amd.h"

#define ELD_WAVE 64

// Kernel launch: the per-thread "last error" of the HIP runtime is shared with every other user of the runtime in the process
// (PyTorch's own probes and event queries leave hipErrorNotReady / hipErrorNoDevice behind), so it is cleared right before
// the launch; ELD_LAUNCH_CHECK() after it then reports this launch only.
#define ELD_LAUNCH(...)                            \
    do {                                           \
        (void)hipGetLastError();                   \
        hipLaunchKernelGGL(__VA_ARGS__);           \
    } while (0)

#define ELD_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// unsigned division by a launch-uniform divisor without the ~30-instruction udiv expansion
struct FastDiv {
    uint32_t d, m, s;
};

static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    uint32_t l = 0;
    while ((1ull << l) < f.d) ++l;
    f.s = l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - f.d)) / f.d + 1);
    return f;
}

__device__ __forceinline__ uint32_t fdiv_u32(uint32_t n, FastDiv f) {
    return (uint32_t)(((uint64_t)__umulhi(n, f.m) + n) >> f.s);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
