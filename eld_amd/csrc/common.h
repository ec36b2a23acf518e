// common.h -- shared device/host helpers for libeld_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/eld_amd.h"

#define ELD_WAVE 64

// Ablation switches (ELD_CONV_DBG / ELD_NOISE_DBG) and the s_memtime stage profiler (eld_debug_conv_prof) are DEVELOPER tools
// for tools/: they change numerical results by design.  Production builds compile them out (ELD_DEV_TOOLS=0, the default);
// build with HIPCC_EXTRA=-DELD_DEV_TOOLS=1 to get them back.
#ifndef ELD_DEV_TOOLS
#define ELD_DEV_TOOLS 0
#endif
#define ELD_DBG(a) (ELD_DEV_TOOLS ? (a).dbg : 0)
#define ELD_PROF(a) (ELD_DEV_TOOLS ? (a).prof : nullptr)

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

// compute units of the CURRENT device (persistent-grid sizing); read once per device, immutable afterwards
static inline int eld_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t p;
        int n = (hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember which devices a kernel has been prepared on
struct EldAttrOnce {
    bool done[64] = {false};
    template <typename K>
    int ensure(K kern, size_t lds_bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!done[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return (int)e;
            done[dev] = true;
        }
        return 0;
    }
};
