// api.hip -- ABI bookkeeping entry points of libeld_amd.so.
#include "common.h"

extern "C" int eld_abi_version(void) { return ELD_ABI_VERSION; }

// ELD_SRC_HASH: first 16 hex digits of the SHA-256 over csrc/*.hip, csrc/*.h and include/eld_amd.h, passed by the build (__graft_entry__.build) --
// what bench.py compares with the hash recorded beside the committed PMC traffic figures (profiles/traffic.json), so that figures measured on
// another kernel set are never reported for this one
#ifndef ELD_SRC_HASH
#define ELD_SRC_HASH "unknown"
#endif
extern "C" const char* eld_build_info(void) {
    return "libeld_amd gfx950 (CDNA4) HIP " __VERSION__ " src=" ELD_SRC_HASH;
}

extern "C" const char* eld_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case ELD_EINVAL: return "ELD_EINVAL: bad shape, flag combination or null pointer";
        case ELD_ENOTSUP: return "ELD_ENOTSUP: not implemented in this build";
        case ELD_EWS: return "ELD_EWS: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown eld_amd error";
    }
}
