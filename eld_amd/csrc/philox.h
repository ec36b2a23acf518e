// philox.h -- Philox4x32-R and the sampler's counter layout (device side).
// The layout is documented in oracle/philox_ref.py, which is the bit-exact CPU statement of it.
//
// Rounds: the sampler runs Philox4x32-7 (ELD_PHILOX_ROUNDS).  Salmon et al. (SC'11, table 2) list 7 rounds as the smallest
// Crush-resistant count for Philox-4x32 (it passes BigCrush) and 10 as the default with a safety margin; Random123 ships both and
// its kat_vectors pin both (tests/test_oracle_golden.py).  The 32x32->64 multiplies are quarter rate on gfx950 and were 43 % of
// the full model's VALU time at 10 rounds, so the three rounds of margin cost ~10 % of the sampler for no statistical property
// any test of this repository (or TestU01) can see.  -DELD_PHILOX_ROUNDS=10 restores cuRAND/rocRAND's count; oracle/philox_ref.py
// ROUNDS must be changed with it.
#pragma once
#include "common.h"

enum : uint32_t {
    STREAM_ROW = 0,     // index = sensor row         -> words 0,1 -> one normal
    STREAM_TL = 1,      // index = group (elem/4)     -> word j    -> Tukey-lambda uniform
    STREAM_QUANT = 2,   // index = group              -> word j    -> quantisation uniform
    STREAM_NREAD = 3,   // index = group              -> 4 normals ('g')
    STREAM_NSHOT = 4,   // index = group              -> 4 normals ('p')
    STREAM_POIS_U = 5,  // index = group              -> word j    -> Poisson attempt-0 U
    STREAM_POIS_V = 6,  // index = group              -> word j    -> Poisson attempt-0 V
    STREAM_POIS_R = 7,  // index = element, iter      -> (U,V),(U,V) retries
};

struct PhiloxKey {
    uint32_t k0, k1;
};

#ifndef ELD_PHILOX_ROUNDS
#define ELD_PHILOX_ROUNDS 7
#endif

template <int ROUNDS = ELD_PHILOX_ROUNDS>
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, PhiloxKey key) {
    uint32_t k0 = key.k0, k1 = key.k1;   // wave-uniform: the key schedule lives on the scalar ALU
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: integer
        // multiplies are quarter-rate and they are half of the sampler's VALU time
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

struct SamplerRng {
    PhiloxKey key;
    uint32_t sid_lo, sid_hi;
    __device__ __forceinline__ uint4 words(uint32_t index, uint32_t stream, uint32_t iter = 0) const {
        return philox4x32<ELD_PHILOX_ROUNDS>(index, sid_lo, sid_hi, stream | (iter << 8), key);
    }
};

// (w>>9)*2^-23 + 2^-24 = (2k+1)*2^-24 in (0,1): exactly representable, u01(~w) == 1-u01(w)
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 9) * 0x1p-23f + 0x1p-24f; }
// (w>>8)*2^-24 in [0,1)
__device__ __forceinline__ float u01_co(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }

// two words -> two standard normals; sin/cos of 2*pi*u are the hardware v_sin/v_cos (argument in turns)
__device__ __forceinline__ float2 box_muller(uint32_t wa, uint32_t wb) {
    const float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u01(wa)));     // hardware sqrt: an RNG transform, not reference arithmetic
    const float t = u01_co(wb);
    return make_float2(r * __builtin_amdgcn_cosf(t), r * __builtin_amdgcn_sinf(t));
}
