// noise.hip -- fused per-pixel physics-based noise sampler for packed-raw Bayer tensors (gfx950).
//
// Replaces NoiseModelBase.__call__ (reference noise.py:149-170), batched over images, plus the
// withheld ELD terms (Tukey-lambda read, row, quantisation, colour bias).
//
// Design (HBM-streaming kernel, 8 B of algorithmic traffic per raw pixel):
//   * grid = (ceil(groups_per_image / GROUPS_PER_BLOCK), N): one image per blockIdx.y so the
//     per-image parameter record is wave-uniform (scalar loads, SGPR-resident);
//   * a "group" is 4 consecutive elements of the flattened (C,H,W) image = one 16-byte lane access;
//     a wave touches 1 KiB contiguous per load/store instruction;
//   * one Philox4x32-7 counter stream per lane (philox.h: why 7 rounds): counters are (group|element|row index, global
//     sample id, stream), so the result is independent of grid shape, wave mapping and GPU count;
//   * row noise: the block's rows' normals are drawn once into LDS and broadcast to the pixels;
//   * the float32 op sequence is exactly the reference's (one rounding per op, FMA contraction
//     OFF for this translation unit), so with injected variates the output is bit-identical to
//     the reference's NumPy evaluation.
#include <stdlib.h>
#include "philox.h"
#include "poisson_alias_table.h"

#pragma clang fp contract(off)

#define NOISE_THREADS 256
#define NOISE_ITERS 4
#define GROUPS_PER_BLOCK (NOISE_THREADS * NOISE_ITERS)
#define ELEMS_PER_BLOCK (GROUPS_PER_BLOCK * 4)
#define MAX_LDS_ROWS 256
#define RUNTIME_FLAGS 0xFFFFFFFFu
#define POIS_TABLE_LAM 32.0f   // below: alias table of Poisson(floor(lam)) + inversion at the fractional rate; at or above: PTRS
#define RES_STEPS 5            // branch-free inversion steps of the fractional-rate draw (P(more) < 6e-4 at a rate below 1)
#define PTRS_DENSE_MIN 13      // lanes of a wave with lam >= 32 in one pixel slot: above -> PTRS inline, else -> queue
#define QP_CAP 768             // open PTRS draws per 4096 pixels (16 B entries: 12 of 64 lanes at most take the queue route); overflow is resolved in place
#define NOISE_WAVES (NOISE_THREADS / 64)
#define QP_WAVE (QP_CAP / NOISE_WAVES)      // round 4: every wave owns a slice of the queue and drains it itself -- no workgroup barrier around the Poisson phases
#define PASS_GROUPS 1           // 4-pixel groups a lane carries through one pass of phase 1 (1: 4 pixels -> <= 128 VGPRs, 4 workgroups per CU)

struct NoiseArgs {
    const void* in;
    float* out;
    const EldNoiseParams* params;
    const float* inject;
    float* dump;
    size_t total;          // N*C*H*W (plane stride of inject/dump)
    size_t in_stride, out_stride;   // elements between consecutive images of in / out (chw when dense)
    uint32_t chw, ngroups, C, H, W;
    FastDiv divW, divH;    // divW divides by W/4 in the vector kernel, by W in the scalar kernel
    uint32_t flags, in_dtype;
    PhiloxKey key;
    uint32_t dbg;          // ablation switches (dev builds, env ELD_NOISE_DBG): 1 skip the queue drain, 4 PTRS draws always take the queue route, 8 skip phase 3 RNG
};

// ---------------------------------------------------------------------------------------------
// Poisson(lam), exact, float32 (replaces the internals of np.random.poisson that noise.py:159 calls; the CPU statement of
// exactly this word usage is oracle/noise_ref.py::_poisson_philox):
//   lam < 32   X = A_n(w0) + Inv(lam - n, u01(w1)),  n = floor(lam): the sum of independent Poisson(n) and Poisson(lam - n) draws.
//              A_n = one lookup in Walker's alias table of Poisson(n) (LDS copy of POIS_ALIAS, oracle/gen_poisson_alias.py) with
//              the whole word w0; Inv = CDF inversion at a rate below 1: RES_STEPS branch-free steps, the < 6e-4 leftovers and the
//              table's tail outcome {A >= 63} (< 4e-7) finish in a rarely taken loop.  ~50 VALU per pixel, no divergence.
//   lam >= 32  PTRS transformed rejection (Hoermann 1993, the large-rate sampler of NumPy's legacy RandomState): attempt 0 takes
//              U = u01(w0), V = u01(w1); attempts 2c+1, 2c+2 the words of retry call c of the element (stream POIS_R).  Where
//              most lanes of a wave are in this regime the attempt runs inline and only rejections are queued; where few are
//              (an image whose rate peaks just above 32) the draws go to an LDS queue that is drained with dense lanes.  Both
//              routes evaluate the same function of (lam, element): results do not depend on the route.
// w0 / w1 = word (element % 4) of the group's STREAM_POIS_U / STREAM_POIS_V Philox calls.
// ---------------------------------------------------------------------------------------------
// The variate transforms below are NOT part of the reference's arithmetic (they replace NumPy's RNG internals), so they
// may use FMA contraction and the hardware reciprocal / square root; only the op chain of noise.py:155-169 in phase 3
// has to round like NumPy.
#pragma clang fp contract(fast)
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float log1pmx(float x) {   // log1p(x) - x without cancellation
    if (fabsf(x) < 0.125f) {
        float s = 0.f;
        s = s * x + (1.f / 9.f);
        s = s * x + (-1.f / 8.f);
        s = s * x + (1.f / 7.f);
        s = s * x + (-1.f / 6.f);
        s = s * x + (1.f / 5.f);
        s = s * x + (-1.f / 4.f);
        s = s * x + (1.f / 3.f);
        s = s * x + (-1.f / 2.f);
        return s * x * x;
    }
    return __logf(1.0f + x) - x;
}

// -lam + k*log(lam) - lgamma(k+1), Stirling form (k >= 1), stable in float32
__device__ __forceinline__ float pois_logpmf(float k, float lam) {
    if (k < 0.5f) return -lam;
    const float rk = frcp(k), rk2 = rk * rk;
    const float x = (lam - k) * rk;
    const float corr = rk * ((1.f / 12.f) - rk2 * ((1.f / 360.f) - rk2 * (1.f / 1260.f)));
    return k * log1pmx(x) - 0.5f * __logf(6.2831853071795865f * k) - corr;
}

struct Ptrs {
    float a, b, invalpha, vr;
    __device__ __forceinline__ void init(float lam) {
        const float slam = __builtin_amdgcn_sqrtf(lam);
        b = 0.931f + 2.53f * slam;
        a = -0.059f + 0.02483f * b;
        invalpha = 1.1239f + 1.1328f * frcp(b - 3.4f);
        vr = 0.9277f - 3.6224f * frcp(b - 2.0f);
    }
    __device__ __forceinline__ float k_of(float lam, float U01, float& us) const {
        const float U = U01 - 0.5f;
        us = 0.5f - fabsf(U);
        return floorf((2.0f * a * frcp(us) + b) * U + lam + 0.43f);
    }
    __device__ __forceinline__ bool slow_accept(float lam, float k, float us, float V) const {
        if ((k < 0.f) || (us < 0.013f && V > us)) return false;
        const float lhs = __logf(V) + __logf(invalpha) - __logf(a * frcp(us * us) + b);
        return lhs <= pois_logpmf(k, lam);
    }
};

// finish a PTRS draw: `fresh` -> attempt 0 with (w0, w1) first; then attempts 2c+1, 2c+2 from retry call c of this element
__device__ __noinline__ float ptrs_resolve(float lam, uint32_t elem, const SamplerRng& rng, bool fresh, uint32_t w0, uint32_t w1) {
    Ptrs P;
    P.init(lam);
    float k = 0.f;
    if (fresh) {
        float us;
        k = P.k_of(lam, u01(w0), us);
        const float V = u01(w1);
        if ((us >= 0.07f && V <= P.vr) || P.slow_accept(lam, k, us, V)) return k;
    }
    for (uint32_t call = 0; call < 64u; ++call) {
        const uint4 r = rng.words(elem, STREAM_POIS_R, call);
        float us;
        k = P.k_of(lam, u01(r.x), us);
        float V = u01(r.y);
        if ((us >= 0.07f && V <= P.vr) || P.slow_accept(lam, k, us, V)) return k;
        k = P.k_of(lam, u01(r.z), us);
        V = u01(r.w);
        if ((us >= 0.07f && V <= P.vr) || P.slow_accept(lam, k, us, V)) return k;
    }
    return fmaxf(k, 0.f);
}

// unit-scale Tukey-lambda quantile from one word: u = u01(w), 1-u = u01(~w) (exact complement); inv_lam = 1/lam (per image)
__device__ __forceinline__ float tukey_lambda(uint32_t w, float lam, float inv_lam) {
    const float lu = __builtin_amdgcn_logf(u01(w)), lv = __builtin_amdgcn_logf(u01(~w));
    if (lam == 0.0f) return (lu - lv) * 0.6931471805599453f;
    return (__builtin_amdgcn_exp2f(lam * lu) - __builtin_amdgcn_exp2f(lam * lv)) * inv_lam;
}
#pragma clang fp contract(off)

// rare tails of the small-rate draw.  k0 == 63: the alias table returned its tail outcome {X >= 63}: sample the conditional tail of
// Poisson(n) by inversion from 63 with a fresh uniform (retry stream, call 64: never used by PTRS retries, which stop at 63).
// r > 0: the fractional-rate inversion is not finished after RES_STEPS steps (state p, r): keep stepping.
__device__ __noinline__ int pois_tail_alias(int n, uint32_t elem, const SamplerRng& rng) {
    const uint4 w = rng.words(elem, STREAM_POIS_R, 64u);
    float q = POIS_TAIL_Q0[n], t = u01(w.x) - q;
    int k = 63;
    while (t > 0.f && k < 255) {
        ++k;
        q = q * ((float)n / (float)k);
        t = t - q;
    }
    return k;
}
__device__ __noinline__ int pois_tail_res(float d, float p, float r) {
    int k = RES_STEPS;
    for (int it = RES_STEPS + 1; r > 0.f && it <= 96; ++it) {
        ++k;
        p = p * (d * (1.0f / (float)it));
        r = r - p;
    }
    return k;
}

// Poisson count and the 9 spare bits of the V word share an LDS word (counts stay far below 2^23: beyond that float32 PTRS has no integer resolution)
__device__ __forceinline__ uint32_t pack_cnt(float k, uint32_t wv) { return ((uint32_t)fminf(k, 8388607.0f) << 9) | (wv & 511u); }

// a / b rounded to nearest, bit for bit what NumPy's float32 division gives, for a divisor that is constant over the image:
// rb = RN(1 / b) (one IEEE division per wave) and two Markstein refinements -- q0 = RN(a rb); r = a - b q (exact, FMA); q' = RN(q + r rb).
// After the first step q is a faithful quotient, after the second it is the correctly rounded one (Markstein 1990; Muller et al.,
// Handbook of Floating-Point Arithmetic, sec. 4.7) provided nothing underflows: operands below 2^-100 take the hardware sequence.
// 5 full-rate ops instead of ~11 (v_div_scale x2, v_rcp, 4 FMA, v_div_fmas, v_div_fixup) -- these two divisions were a third of
// the streaming models' VALU work.
__device__ __forceinline__ float div_rn(float a, float b, float rb) {
    if (__builtin_expect(fabsf(a) < 0x1p-100f && a != 0.f, 0)) return a / b;
    const float q0 = a * rb;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), rb, q0);
    return __builtin_fmaf(__builtin_fmaf(-b, q1, a), rb, q1);
}

__device__ __forceinline__ uint32_t pick(const uint4& w, int j) { return j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w; }

__device__ __forceinline__ float row_normal(uint32_t srow, const SamplerRng& rng) {
    const uint4 w = rng.words(srow, STREAM_ROW);
    return box_muller(w.x, w.y).x;
}

template <bool VEC>
__device__ __forceinline__ void load_y4(const NoiseArgs& a, size_t img_off /* n * in_stride */, uint32_t e0, uint32_t nvalid, float (&y)[4]) {
    if (VEC) {
        if (a.in_dtype == ELD_IN_U16) {
            const ushort4 q = *reinterpret_cast<const ushort4*>(static_cast<const uint16_t*>(a.in) + img_off + e0);
            y[0] = (float)q.x / 65535.0f; y[1] = (float)q.y / 65535.0f; y[2] = (float)q.z / 65535.0f; y[3] = (float)q.w / 65535.0f;
        } else {
            const float4 q = *reinterpret_cast<const float4*>(static_cast<const float*>(a.in) + img_off + e0);
            y[0] = q.x; y[1] = q.y; y[2] = q.z; y[3] = q.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = 0.f;
            if ((uint32_t)j < nvalid)
                y[j] = (a.in_dtype == ELD_IN_U16) ? (float)static_cast<const uint16_t*>(a.in)[img_off + e0 + j] / 65535.0f
                                                  : static_cast<const float*>(a.in)[img_off + e0 + j];
        }
    }
    if (a.in_dtype == ELD_IN_U16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fminf(fmaxf(y[j], 0.f), 1.f);   // lmdb_dataset.py:39
    }
}

template <bool VEC, uint32_t TFLAGS, bool DEBUG>
__global__ __launch_bounds__(NOISE_THREADS) void noise_kernel(const NoiseArgs a) {
    constexpr bool MAYBE_P = (TFLAGS == RUNTIME_FLAGS) || (TFLAGS & ELD_SHOT_POISSON);
    __shared__ float s_row[MAX_LDS_ROWS];
    __shared__ uint32_t s_cnt[MAYBE_P ? ELEMS_PER_BLOCK : 1];     // (count << 9) | low 9 bits of the pixel's POIS_V word (free: u01 takes w >> 9)
    __shared__ uint4 s_qp[MAYBE_P ? QP_CAP : 1];
    __shared__ uint32_t s_tab[MAYBE_P ? POIS_TAB_N * POIS_TAB_ENT : 1];
    __shared__ uint32_t s_qn[NOISE_WAVES];
    const uint32_t flags = (TFLAGS == RUNTIME_FLAGS) ? a.flags : TFLAGS;
    const uint32_t n = blockIdx.y;
    const EldNoiseParams P = a.params[n];             // wave-uniform -> scalar loads
    SamplerRng rng;
    rng.key = a.key;
    rng.sid_lo = P.sample_id_lo;
    rng.sid_hi = P.sample_id_hi;

    const uint32_t tid = threadIdx.x;
    const uint32_t g_begin = blockIdx.x * GROUPS_PER_BLOCK;
    const uint32_t g_end = min(g_begin + GROUPS_PER_BLOCK, a.ngroups);
    const size_t img_off = (size_t)n * a.chw;                  // debug planes (inject / dump) are dense
    const size_t in_off = (size_t)n * a.in_stride, out_off = (size_t)n * a.out_stride;
    const bool inject = DEBUG && a.inject != nullptr;
    const bool do_pois = MAYBE_P && (flags & ELD_SHOT_POISSON) && !inject;
    const float S = P.saturation, ratio = P.ratio, K = P.K;

    const uint32_t wave = tid >> 6;
    if (do_pois) {
        if (tid < NOISE_WAVES) s_qn[tid] = 0;
#pragma unroll
        for (int i = 0; i < POIS_TAB_N * POIS_TAB_ENT / NOISE_THREADS; ++i) s_tab[tid + i * NOISE_THREADS] = POIS_ALIAS[tid + i * NOISE_THREADS];
    }

    // ---- row normals of this block's rows -> LDS ------------------------------------------------
    uint32_t r_first = 0;
    bool lds_rows = false;
    if ((flags & ELD_ROW) && !inject) {
        const uint32_t e_first = g_begin * 4u, e_last = min(g_end * 4u, a.chw) - 1u;
        r_first = VEC ? fdiv_u32(g_begin, a.divW) : fdiv_u32(e_first, a.divW);
        const uint32_t r_last = VEC ? fdiv_u32(g_end - 1u, a.divW) : fdiv_u32(e_last, a.divW);
        const uint32_t nrows = r_last - r_first + 1u;
        lds_rows = nrows <= MAX_LDS_ROWS;
        if (lds_rows) {
            for (uint32_t t = tid; t < nrows; t += NOISE_THREADS) {
                const uint32_t r = r_first + t;
                const uint32_t c = fdiv_u32(r, a.divH), h = r - c * a.H;
                s_row[t] = row_normal(2u * h + (c >> 1), rng);
            }
        }
    }
    __syncthreads();

    // ================================ phase 1 + 2: Poisson counts -> s_cnt ================================
    if (do_pois) {
        // the rate is the reference's op chain, op for op: ((y * S) / ratio) / K with correctly rounded divisions (noise.py:155-159; oracle:
        // poisson_lambda).  Rounds 2-3 formed it as y * c with one per-image constant (10 of ~250 VALU operations per pixel saved, the rate an
        // ulp off the reference's); round 4 pays them: the only arithmetic of the pinned models that was not the reference's is gone.
        const float pr_ratio = 1.0f / ratio, pr_K = 1.0f / K;
#pragma unroll 1
        for (int half = 0; half < NOISE_ITERS / PASS_GROUPS; ++half) {
            constexpr int PE = 4 * PASS_GROUPS;                 // pixels a lane carries through this pass
            float lam[PE], p[PE], r[PE];
            uint32_t w[PE], wv[PE];
            int kk[PE];
            bool ok[PE];
#pragma unroll
            for (int gi = 0; gi < PASS_GROUPS; ++gi) {
                const uint32_t g = g_begin + (half * PASS_GROUPS + gi) * NOISE_THREADS + tid;
                const bool gv = g < g_end;
                const uint32_t e0 = g * 4u;
                const uint32_t nvalid = gv ? (VEC ? 4u : min(4u, a.chw - e0)) : 0u;
                float y[4] = {0.f, 0.f, 0.f, 0.f};
                uint4 wd = make_uint4(0, 0, 0, 0), wd2 = make_uint4(0, 0, 0, 0);
                if (gv) {
                    load_y4<VEC>(a, in_off, e0, nvalid, y);
                    wd = rng.words(g, STREAM_POIS_U);
                    wd2 = rng.words(g, STREAM_POIS_V);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = gi * 4 + j;
                    ok[e] = (uint32_t)j < nvalid;
                    lam[e] = fmaxf(div_rn(div_rn(y[j] * S, ratio, pr_ratio), K, pr_K), 0.f);
                    w[e] = pick(wd, j);
                    wv[e] = pick(wd2, j);
                }
            }
            // ---- lam < 32: alias table of Poisson(floor(lam)) + branch-free inversion at the fractional rate ----------------
            uint32_t big = 0, tail = 0;                 // bit e: pixel slot e is in the PTRS regime / needs one of the rare tail loops
#pragma unroll
            for (int e = 0; e < PE; ++e) {
                const bool small = ok[e] && lam[e] < POIS_TABLE_LAM;
                if (ok[e] && !small) big |= 1u << e;
                const float ls = small ? lam[e] : 0.f;
                const int n = (int)ls;
                const float d = ls - (float)n;                              // exact
                const uint32_t j = w[e] >> 26;
                const uint32_t ent = s_tab[n * POIS_TAB_ENT + (int)j];
                const uint32_t k0 = (w[e] << 6) < (ent & 0xFFFFFFC0u) ? j : (ent & 63u);
                p[e] = __expf(-d);
                r[e] = u01(wv[e]) - p[e];
                int k1 = 0;
#pragma unroll
                for (int it = 1; it <= RES_STEPS; ++it) {                   // once r <= 0 it stays there (p > 0): no select needed
                    k1 += r[e] > 0.f ? 1 : 0;
                    p[e] = p[e] * (d * (1.0f / (float)it));
                    r[e] = r[e] - p[e];
                }
                kk[e] = (int)k0 + k1;
                if (small && (k0 == 63u || r[e] > 0.f)) tail |= 1u << e;
            }
            if (__any(tail != 0u)) {                                        // < 1e-3 of the draws
#pragma unroll
                for (int e = 0; e < PE; ++e) {
                    if (!(tail & (1u << e))) continue;
                    const uint32_t le = (uint32_t)(half * PASS_GROUPS + (e >> 2)) * (NOISE_THREADS * 4u) + tid * 4u + (uint32_t)(e & 3);
                    const int n = (int)lam[e];
                    int k0 = kk[e] - RES_STEPS, k1 = RES_STEPS;             // r > 0 after the last step: every step counted
                    if (!(r[e] > 0.f)) { k0 = 63; k1 = kk[e] - 63; }        // only the alias draw hit its tail outcome
                    if (k0 == 63) k0 = pois_tail_alias(n, g_begin * 4u + le, rng);
                    if (r[e] > 0.f) k1 = pois_tail_res(lam[e] - (float)n, p[e], r[e]);
                    kk[e] = k0 + k1;
                }
            }
            // ---- lam >= 32: PTRS.  Per pixel slot the wave decides: many lanes -> attempt 0 inline, few -> queue ------------------
            uint32_t pendF = 0, pendR = 0;              // fresh draw queued / attempt 0 rejected, retries queued
            if (__any(big != 0u)) {
#pragma unroll
                for (int e = 0; e < PE; ++e) {
                    const bool mine = (big >> e) & 1u;
                    const int nbig = __popcll(__ballot(mine));
                    if (nbig == 0) continue;
                    if (nbig < PTRS_DENSE_MIN || (ELD_DBG(a) & 4)) { if (mine) pendF |= 1u << e; continue; }
                    if (mine) {
                        Ptrs T;
                        T.init(lam[e]);
                        float us;
                        const float k0 = T.k_of(lam[e], u01(w[e]), us);
                        const float V = u01(wv[e]);
                        bool acc = us >= 0.07f && V <= T.vr;
                        if (!acc) acc = T.slow_accept(lam[e], k0, us, V);
                        kk[e] = (int)k0;
                        if (!acc) pendR |= 1u << e;
                    }
                }
            }
            // ---- one LDS atomic per lane reserves the queue slots of its open draws ----------------------------------------
            const uint32_t pend = pendF | pendR;
            uint32_t base = 0;
            if (pend) base = atomicAdd(&s_qn[wave], (uint32_t)__popc(pend));
#pragma unroll
            for (int e = 0; e < PE; ++e) {
                if (!ok[e]) continue;
                const uint32_t le = (uint32_t)(half * PASS_GROUPS + (e >> 2)) * (NOISE_THREADS * 4u) + tid * 4u + (uint32_t)(e & 3);
                if (pend & (1u << e)) {
                    const bool fresh = (pendF >> e) & 1u;
                    const uint32_t pos = base + (uint32_t)__popc(pend & ((1u << e) - 1u));
                    if (pos < QP_WAVE) s_qp[wave * QP_WAVE + pos] = make_uint4(le | (fresh ? 0x80000000u : 0u), __float_as_uint(lam[e]), w[e], wv[e]);
                    else s_cnt[le] = pack_cnt(ptrs_resolve(lam[e], g_begin * 4u + le, rng, fresh, w[e], wv[e]), wv[e]);
                } else {
                    s_cnt[le] = pack_cnt((float)kk[e], wv[e]);
                }
            }
        }
        // ---- phase 2: the wave drains ITS OWN queue slice with dense lanes ----------------------------------------------------
        // Everything a wave reads from here on (its queue slice, the counts of its own pixels) was written by the wave itself: LDS operations of a
        // wave execute in order, so no workgroup barrier is needed -- only the compiler must keep the order (round 3 had two barriers here, and
        // a third of the wave time of the full model was spent parked at them waiting for the slowest wave's rejection loops).
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t nP = (ELD_DBG(a) & 1) ? 0u : min(s_qn[wave], (uint32_t)QP_WAVE);
        for (uint32_t q = tid & 63u; q < nP; q += 64u) {
            const uint4 en = s_qp[wave * QP_WAVE + q];
            const uint32_t le = en.x & 0x7FFFFFFFu;
            s_cnt[le] = pack_cnt(ptrs_resolve(__uint_as_float(en.y), g_begin * 4u + le, rng, (en.x >> 31) != 0u, en.z, en.w), en.w);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ================================ phase 3: the rest of the model + reference arithmetic ===================
    const float g_sigma = fmaxf(P.g_scale, 1e-10f);
    const float r_ratio = 1.0f / ratio, r_S = 1.0f / S;          // correctly rounded reciprocals (IEEE division, once per wave)
    const float inv_tl_lambda = P.tl_lambda != 0.f ? 1.0f / P.tl_lambda : 0.f;
    auto phase3 = [&](int it, const float* ypre) {      // ypre: the group's input, already loaded (streaming models), or null
        const uint32_t g = g_begin + it * NOISE_THREADS + tid;
        if (g >= g_end) return;
        const uint32_t e0 = g * 4u;
        const uint32_t nvalid = VEC ? 4u : min(4u, a.chw - e0);
        const uint32_t le0 = (uint32_t)it * (NOISE_THREADS * 4u) + tid * 4u;

        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (ypre) { y[0] = ypre[0]; y[1] = ypre[1]; y[2] = ypre[2]; y[3] = ypre[3]; }
        else if (!do_pois || !(flags & ELD_SHOT_POISSON) || DEBUG) load_y4<VEC>(a, in_off, e0, nvalid, y);
        float4 cnt4 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 cw = make_uint4(0, 0, 0, 0);
        if (do_pois) {
            cw = *reinterpret_cast<const uint4*>(&s_cnt[le0]);
            cnt4 = make_float4((float)(cw.x >> 9), (float)(cw.y >> 9), (float)(cw.z >> 9), (float)(cw.w >> 9));
        }
        // full model: the quantisation uniform is made of bits the other draws leave over (low 9 of the Tukey-lambda word, low 9 of the
        // Poisson V word: u01 uses w >> 9) -- 18 bits, one Philox call per 4 pixels saved
        const bool uq_borrow = do_pois && (flags & ELD_READ_TL) && (flags & ELD_QUANT);

        // one Philox call per needed stream per group
        uint4 w_tl, w_q;
        float nrd[4], nsh[4];
        if (!inject) {
            if (flags & ELD_READ_TL) w_tl = (ELD_DBG(a) & 8) ? make_uint4(g, g * 3u, g * 5u, g * 7u) : rng.words(g, STREAM_TL);
            if ((flags & ELD_QUANT) && !uq_borrow) w_q = (ELD_DBG(a) & 8) ? make_uint4(g, g * 3u, g * 5u, g * 7u) : rng.words(g, STREAM_QUANT);
            if (flags & ELD_READ_GAUSS) {
                const uint4 w = rng.words(g, STREAM_NREAD);
                const float2 p0 = box_muller(w.x, w.y), p1 = box_muller(w.z, w.w);
                nrd[0] = p0.x; nrd[1] = p0.y; nrd[2] = p1.x; nrd[3] = p1.y;
            }
            if (flags & ELD_SHOT_GAUSS) {
                const uint4 w = rng.words(g, STREAM_NSHOT);
                const float2 p0 = box_muller(w.x, w.y), p1 = box_muller(w.z, w.w);
                nsh[0] = p0.x; nsh[1] = p0.y; nsh[2] = p1.x; nsh[3] = p1.y;
            }
        }

        uint32_t r_vec = 0;
        if (VEC && (flags & (ELD_ROW | ELD_CBIAS))) r_vec = fdiv_u32(g, a.divW);

        float z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!VEC && (uint32_t)j >= nvalid) { z[j] = 0.f; continue; }
            const uint32_t e = e0 + j;
            const size_t ge = img_off + e;
            uint32_t r = r_vec;
            if (!VEC && (flags & (ELD_ROW | ELD_CBIAS))) r = fdiv_u32(e, a.divW);

            float v_cnt = 0.f, v_nshot = 0.f, v_nread = 0.f, v_tl = 0.f, v_nrow = 0.f, v_uq = 0.f;
            float zz;
            if (flags & ELD_SHOT_POISSON) {       // noise.py:158-159
                v_cnt = inject ? a.inject[ELD_PLANE_COUNT * a.total + ge] : (j == 0 ? cnt4.x : j == 1 ? cnt4.y : j == 2 ? cnt4.z : cnt4.w);
                zz = v_cnt * K;
            } else {
                const float y1 = y[j] * S;        // noise.py:155
                const float y2 = div_rn(y1, ratio, r_ratio);      // noise.py:156  (y1 / ratio, correctly rounded)
                if (flags & ELD_SHOT_GAUSS) {     // noise.py:160-161
                    v_nshot = inject ? a.inject[ELD_PLANE_NSHOT * a.total + ge] : nsh[j];
                    zz = y2 + v_nshot * __builtin_sqrtf(fmaxf(K * y2, 1e-10f));
                } else {                          // noise.py:162-163
                    zz = y2;
                }
            }
            if (flags & ELD_READ_GAUSS) {         // noise.py:165-166
                v_nread = inject ? a.inject[ELD_PLANE_NREAD * a.total + ge] : nrd[j];
                zz = zz + v_nread * g_sigma;
            }
            if (flags & ELD_READ_TL) {
                v_tl = inject ? a.inject[ELD_PLANE_TL * a.total + ge] : tukey_lambda(pick(w_tl, j), P.tl_lambda, inv_tl_lambda);
                zz = zz + v_tl * P.tl_scale;
            }
            if (flags & ELD_ROW) {
                if (inject) {
                    v_nrow = a.inject[ELD_PLANE_NROW * a.total + ge];
                } else if (lds_rows) {
                    v_nrow = s_row[r - r_first];
                } else {
                    const uint32_t c = fdiv_u32(r, a.divH), h = r - c * a.H;
                    v_nrow = row_normal(2u * h + (c >> 1), rng);
                }
                zz = zz + v_nrow * P.row_scale;
            }
            if (flags & ELD_QUANT) {
                if (inject) v_uq = a.inject[ELD_PLANE_UQ * a.total + ge];
                else if (uq_borrow) v_uq = (float)(((pick(w_tl, j) & 511u) << 9) | (pick(cw, j) & 511u)) * 0x1p-18f;
                else v_uq = u01_co(pick(w_q, j));
                zz = zz + (v_uq - 0.5f) * P.q_step;
            }
            if (flags & ELD_CBIAS) {
                const uint32_t c = fdiv_u32(r, a.divH);
                zz = zz + P.color_bias[c & 3u];
            }
            zz = zz * ratio;                      // noise.py:168
            zz = div_rn(zz, S, r_S);              // noise.py:169  (zz / S, correctly rounded)
            if (flags & ELD_CLIP) zz = fmaxf(fminf(zz, 1.0f), 0.0f);   // sid_dataset.py:277
            z[j] = zz;

            if (DEBUG && a.dump != nullptr) {
                a.dump[ELD_PLANE_COUNT * a.total + ge] = v_cnt;
                a.dump[ELD_PLANE_NSHOT * a.total + ge] = v_nshot;
                a.dump[ELD_PLANE_NREAD * a.total + ge] = v_nread;
                a.dump[ELD_PLANE_TL * a.total + ge] = v_tl;
                a.dump[ELD_PLANE_NROW * a.total + ge] = v_nrow;
                a.dump[ELD_PLANE_UQ * a.total + ge] = v_uq;
            }
        }

        if (VEC) {
            // written once, never read again by this kernel: nontemporal (streaming) store keeps the output out of the L2's way
            typedef float f4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((f4v){z[0], z[1], z[2], z[3]}, reinterpret_cast<f4v*>(a.out + out_off + e0));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((uint32_t)j < nvalid) a.out[out_off + e0 + j] = z[j];
        }
    };
    if constexpr (MAYBE_P) {
#pragma unroll 1
        for (int it = 0; it < NOISE_ITERS; ++it) phase3(it, nullptr);
    } else {                                     // streaming models ('g', 'pg', scale only): all four 16-byte loads of a lane in flight at once
        float ypre[NOISE_ITERS][4];
        if constexpr (VEC) {
            // branch-free straight-line loads (out-of-range groups re-read the block's last group; their results are dropped): the
            // four requests of a lane leave back to back instead of load -> wait -> store per group
            const uint32_t g_last = g_end - 1u;
            if (a.in_dtype == ELD_IN_U16) {
                ushort4 q[NOISE_ITERS];
#pragma unroll
                for (int it = 0; it < NOISE_ITERS; ++it)
                    q[it] = *reinterpret_cast<const ushort4*>(static_cast<const uint16_t*>(a.in) + in_off + (size_t)min(g_begin + it * NOISE_THREADS + tid, g_last) * 4u);
#pragma unroll
                for (int it = 0; it < NOISE_ITERS; ++it) {
                    const float v[4] = {(float)q[it].x / 65535.0f, (float)q[it].y / 65535.0f, (float)q[it].z / 65535.0f, (float)q[it].w / 65535.0f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) ypre[it][j] = fminf(fmaxf(v[j], 0.f), 1.f);      // lmdb_dataset.py:39
                }
            } else {
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v q[NOISE_ITERS];
#pragma unroll
                for (int it = 0; it < NOISE_ITERS; ++it)
                    q[it] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(static_cast<const float*>(a.in) + in_off + (size_t)min(g_begin + it * NOISE_THREADS + tid, g_last) * 4u));
#pragma unroll
                for (int it = 0; it < NOISE_ITERS; ++it) { ypre[it][0] = q[it][0]; ypre[it][1] = q[it][1]; ypre[it][2] = q[it][2]; ypre[it][3] = q[it][3]; }
            }
#pragma unroll
            for (int it = 0; it < NOISE_ITERS; ++it) phase3(it, ypre[it]);
        } else {
#pragma unroll 1
            for (int it = 0; it < NOISE_ITERS; ++it) phase3(it, nullptr);
        }
    }
}

template <bool VEC, uint32_t TFLAGS, bool DEBUG>
static int launch_noise(const NoiseArgs& a, int N, hipStream_t st) {
    dim3 grid((a.ngroups + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK, N);
    ELD_LAUNCH((noise_kernel<VEC, TFLAGS, DEBUG>), grid, dim3(NOISE_THREADS), 0, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

extern "C" int eld_noise_forward_strided(const void* in, int in_dtype, size_t in_image_stride, float* out, size_t out_image_stride,
                                         const EldNoiseParams* params, int N, int C, int H, int W, uint32_t flags, uint64_t seed,
                                         const float* inject, float* dump, void* stream);

extern "C" int eld_noise_forward(const void* in, int in_dtype, float* out, const EldNoiseParams* params,
                                 int N, int C, int H, int W, uint32_t flags, uint64_t seed,
                                 const float* inject, float* dump, void* stream) {
    const size_t chw = (size_t)(C > 0 ? C : 0) * (H > 0 ? H : 0) * (W > 0 ? W : 0);
    return eld_noise_forward_strided(in, in_dtype, chw, out, chw, params, N, C, H, W, flags, seed, inject, dump, stream);
}

extern "C" int eld_noise_forward_strided(const void* in, int in_dtype, size_t in_image_stride, float* out, size_t out_image_stride,
                                         const EldNoiseParams* params, int N, int C, int H, int W, uint32_t flags, uint64_t seed,
                                         const float* inject, float* dump, void* stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return ELD_EINVAL;
    if (in_dtype != ELD_IN_F32 && in_dtype != ELD_IN_U16) return ELD_EINVAL;
    if ((flags & ELD_SHOT_POISSON) && (flags & ELD_SHOT_GAUSS)) return ELD_EINVAL;   // 'P' wins in the parser (noise.py:158-160)
    if ((flags & (ELD_ROW | ELD_CBIAS)) && C != 4) return ELD_EINVAL;
    const size_t chw = (size_t)C * H * W;
    if (N == 0 || chw == 0) return 0;               // empty input: nothing to do (reference returns an empty array)
    if (!in || !out || !params) return ELD_EINVAL;
    if (chw >= (1ull << 32) - 4) return ELD_ENOTSUP;

    NoiseArgs a;
    a.in = in; a.out = out; a.params = params; a.inject = inject; a.dump = dump;
    a.total = (size_t)N * chw;
    if (out_image_stride < chw && N > 1) return ELD_EINVAL;          // outputs must not overlap (inputs may: stride 0 = one clean image)
    a.in_stride = in_image_stride; a.out_stride = out_image_stride;
    a.chw = (uint32_t)chw;
    a.ngroups = (uint32_t)((chw + 3) / 4);
    a.C = C; a.H = H; a.W = W;
    a.divH = make_fastdiv((uint32_t)H);
    a.flags = flags; a.in_dtype = (uint32_t)in_dtype;
    a.key.k0 = (uint32_t)seed; a.key.k1 = (uint32_t)(seed >> 32);
    a.dbg = 0;
#if ELD_DEV_TOOLS
    { static const int dbg = [] { const char* e = getenv("ELD_NOISE_DBG"); return e ? atoi(e) : 0; }(); a.dbg = (uint32_t)dbg; }
#endif

    const size_t in_align = (in_dtype == ELD_IN_U16) ? 8 : 16;
    const bool vec = (W % 4 == 0) && ((uintptr_t)in % in_align == 0) && ((uintptr_t)out % 16 == 0) && in_image_stride % 4 == 0 && out_image_stride % 4 == 0;
    a.divW = make_fastdiv(vec ? (uint32_t)W / 4u : (uint32_t)W);
    hipStream_t st = as_stream(stream);
    const bool debug = inject != nullptr || dump != nullptr;
    if (debug) return vec ? launch_noise<true, RUNTIME_FLAGS, true>(a, N, st) : launch_noise<false, RUNTIME_FLAGS, true>(a, N, st);
    if (!vec) return launch_noise<false, RUNTIME_FLAGS, false>(a, N, st);
    // compile-time specialisations of the hot model strings (dead terms and their registers vanish)
    constexpr uint32_t FULL = ELD_SHOT_POISSON | ELD_READ_TL | ELD_ROW | ELD_QUANT;   // 'PGRU' -- BASELINE.json config 2
    constexpr uint32_t PG = ELD_SHOT_POISSON | ELD_READ_GAUSS;                        // 'Pg'   -- config 1
    switch (flags) {
        case FULL: return launch_noise<true, FULL, false>(a, N, st);
        case FULL | ELD_CLIP: return launch_noise<true, FULL | ELD_CLIP, false>(a, N, st);
        case PG: return launch_noise<true, PG, false>(a, N, st);
        case PG | ELD_CLIP: return launch_noise<true, PG | ELD_CLIP, false>(a, N, st);
        case ELD_READ_GAUSS: return launch_noise<true, ELD_READ_GAUSS, false>(a, N, st);
        case ELD_READ_GAUSS | ELD_CLIP: return launch_noise<true, ELD_READ_GAUSS | ELD_CLIP, false>(a, N, st);
        case ELD_SHOT_GAUSS | ELD_READ_GAUSS: return launch_noise<true, ELD_SHOT_GAUSS | ELD_READ_GAUSS, false>(a, N, st);                       // 'pg'
        case ELD_SHOT_GAUSS | ELD_READ_GAUSS | ELD_CLIP: return launch_noise<true, ELD_SHOT_GAUSS | ELD_READ_GAUSS | ELD_CLIP, false>(a, N, st);
        case 0u: return launch_noise<true, 0u, false>(a, N, st);                                                                                     // '' (scale only)
        case ELD_CLIP: return launch_noise<true, ELD_CLIP, false>(a, N, st);
        default: return launch_noise<true, RUNTIME_FLAGS, false>(a, N, st);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void philox_words_kernel(uint32_t* out, uint32_t n, uint32_t index0, SamplerRng rng, uint32_t stream, uint32_t iter) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 w = rng.words(index0 + i, stream, iter);
    reinterpret_cast<uint4*>(out)[i] = w;
}

extern "C" int eld_philox_rounds(void) { return ELD_PHILOX_ROUNDS; }

extern "C" int eld_philox_words(uint32_t* out, uint32_t n, uint32_t index0, uint64_t sample_id,
                                uint32_t stream, uint32_t iter, uint64_t seed, void* stream_h) {
    if (n == 0) return 0;
    if (!out) return ELD_EINVAL;
    SamplerRng rng;
    rng.key.k0 = (uint32_t)seed; rng.key.k1 = (uint32_t)(seed >> 32);
    rng.sid_lo = (uint32_t)sample_id; rng.sid_hi = (uint32_t)(sample_id >> 32);
    ELD_LAUNCH(philox_words_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream_h), out, n, index0, rng, stream, iter);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Bayer pack / unpack (RawPacker, noise.py:10-20, 66-81).  Pure index maps, bit-exact.
//   packed[n][0] = mosaic[2y][2x]   packed[n][1] = mosaic[2y][2x+1]
//   packed[n][2] = mosaic[2y+1][2x+1] packed[n][3] = mosaic[2y+1][2x]
// ---------------------------------------------------------------------------------------------
template <bool PACK>
__global__ void bayer_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int h, int w) {
    const size_t hw = (size_t)h * w;
    const size_t total = (size_t)N * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, p = i - n * hw;
        const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
        const size_t W2 = 2 * (size_t)w;
        const size_t m = n * 4 * hw + (size_t)(2 * y) * W2 + 2 * x;    // mosaic[n][2y][2x]
        const size_t q = n * 4 * hw + p;                                 // packed[n][0][y][x]
        if (PACK) {
            const float2 top = *reinterpret_cast<const float2*>(src + m);
            const float2 bot = *reinterpret_cast<const float2*>(src + m + W2);
            dst[q] = top.x; dst[q + hw] = top.y; dst[q + 2 * hw] = bot.y; dst[q + 3 * hw] = bot.x;
        } else {
            *reinterpret_cast<float2*>(dst + m) = make_float2(src[q], src[q + hw]);
            *reinterpret_cast<float2*>(dst + m + W2) = make_float2(src[q + 3 * hw], src[q + 2 * hw]);
        }
    }
}

static int bayer_launch(bool pack, const float* src, float* dst, int N, int h, int w, void* stream) {
    if (N < 0 || h < 0 || w < 0) return ELD_EINVAL;
    const size_t total = (size_t)N * h * w;
    if (total == 0) return 0;
    if (!src || !dst) return ELD_EINVAL;
    const int blocks = (int)min((total + 255) / 256, (size_t)8192);
    if (pack) ELD_LAUNCH(bayer_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, N, h, w);
    else ELD_LAUNCH(bayer_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, N, h, w);
    ELD_LAUNCH_CHECK();
    return 0;
}

extern "C" int eld_pack_bayer(const float* mosaic, float* packed, int N, int h, int w, void* stream) {
    return bayer_launch(true, mosaic, packed, N, h, w, stream);
}
extern "C" int eld_unpack_bayer(const float* packed, float* mosaic, int N, int h, int w, void* stream) {
    return bayer_launch(false, packed, mosaic, N, h, w, stream);
}

// ---------------------------------------------------------------------------------------------
// X-Trans pack / unpack (RawPacker.pack_raw_xtrans / unpack_raw_xtrans, noise.py:22-64, 83-127).  Pure index maps, bit-exact:
// the 6x6 colour cell <-> 9 planes at 1/3 resolution.  Planes 0..4: packed (2a + pi, 2b + pj) <-> cell (a, b), position
// XT_RC[c][pi][pj]; planes 5..8: packed (i, j) <-> 3x3 block (i, j), position XT_RC3[c - 5].  One thread per packed element
// (the packed side is the coalesced one: 4 B per lane, consecutive j); the 36 positions of a cell are covered exactly once.
// ---------------------------------------------------------------------------------------------
__constant__ unsigned char XT_RC[5][2][2][2] = {
    {{{0, 0}, {0, 4}}, {{3, 1}, {3, 3}}},
    {{{0, 2}, {0, 5}}, {{3, 2}, {3, 5}}},
    {{{0, 1}, {0, 3}}, {{3, 0}, {3, 4}}},
    {{{1, 2}, {2, 5}}, {{5, 2}, {4, 5}}},
    {{{2, 2}, {1, 5}}, {{4, 2}, {5, 5}}},
};
__constant__ unsigned char XT_RC3[4][2] = {{1, 0}, {1, 1}, {2, 0}, {2, 1}};

template <bool PACK>
__global__ __launch_bounds__(256) void xtrans_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int Hm, int Wm) {
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w, total = 9 * hw, msz = (size_t)Hm * Wm;
    const float* s = src + (size_t)n * (PACK ? msz : total);
    float* d = dst + (size_t)n * (PACK ? total : msz);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e / hw);
        const int r = (int)(e - (size_t)c * hw);
        const int i = r / w, j = r - i * w;
        int row, col;
        if (c < 5) { row = 6 * (i >> 1) + XT_RC[c][i & 1][j & 1][0]; col = 6 * (j >> 1) + XT_RC[c][i & 1][j & 1][1]; }
        else { row = 3 * i + XT_RC3[c - 5][0]; col = 3 * j + XT_RC3[c - 5][1]; }
        const size_t m = (size_t)row * Wm + col;
        if (PACK) d[e] = s[m]; else d[m] = s[e];
    }
}

// mosaic float32 [N, Hm, Wm] -> packed float32 [N, 9, 2*(Hm/6), 2*(Wm/6)] (rows / columns beyond the last whole 6x6 cell are ignored,
// noise.py:25-26)
extern "C" int eld_pack_xtrans(const float* mosaic, float* packed, int N, int Hm, int Wm, void* stream) {
    if (N < 0 || Hm < 0 || Wm < 0) return ELD_EINVAL;
    const int h = 2 * (Hm / 6), w = 2 * (Wm / 6);
    const size_t total = (size_t)9 * h * w;
    if (N == 0 || total == 0) return 0;
    if (!mosaic || !packed) return ELD_EINVAL;
    dim3 grid((unsigned)min((total + 255) / 256, (size_t)4096), N);
    ELD_LAUNCH(xtrans_kernel<true>, grid, dim3(256), 0, as_stream(stream), mosaic, packed, h, w, Hm, Wm);
    ELD_LAUNCH_CHECK();
    return 0;
}

// packed float32 [N, 9, h, w] -> mosaic float32 [N, 3h, 3w] (every mosaic element is written: no zero fill needed)
extern "C" int eld_unpack_xtrans(const float* packed, float* mosaic, int N, int h, int w, void* stream) {
    if (N < 0 || h < 0 || w < 0) return ELD_EINVAL;
    const size_t total = (size_t)9 * h * w;
    if (N == 0 || total == 0) return 0;
    if (!mosaic || !packed) return ELD_EINVAL;
    dim3 grid((unsigned)min((total + 255) / 256, (size_t)4096), N);
    ELD_LAUNCH(xtrans_kernel<false>, grid, dim3(256), 0, as_stream(stream), packed, mosaic, h, w, 3 * h, 3 * w);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Sensor mosaic -> packed, normalised raw (pack_raw_bayer, dataset/sid_dataset.py:172-196), one pass:
//   out[k][y][x] = clip((float(im[2y+oy_k][2x+ox_k]) - black[k]) / (white - black[k]), 0, 1),  k = R, G1, B, G2 = raw_pattern codes 0..3
// float32 throughout as NumPy evaluates it (uint16 -> float32 exact, one rounding per op, true division).  2 B read + 4 B written
// per sensor pixel.  A lane handles two horizontally adjacent packed positions of all four channels: one 8-byte read per mosaic row.
// ---------------------------------------------------------------------------------------------
struct PackRawArgs { int oy[4], ox[4]; float black[4], denom[4]; };

__global__ __launch_bounds__(256) void pack_raw_kernel(const uint16_t* __restrict__ im, float* __restrict__ out, int h, int w, PackRawArgs p) {
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w, W2 = 2 * (size_t)w;
    const uint16_t* src = im + (size_t)n * 4 * hw;
    float* dst = out + (size_t)n * 4 * hw;
    const int wp = (w + 1) / 2;                                     // position pairs per packed row
    const size_t total = (size_t)h * wp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / wp), x = 2 * (int)(i - (size_t)y * wp);
        const bool two = x + 1 < w;
        uint16_t q[2][4];                                           // q[row][col] of the 2 x 4 mosaic block
        const uint16_t* r0 = src + (size_t)(2 * y) * W2 + 2 * x;
        if (two && ((W2 & 3) == 0)) {
            const ushort4 a = *reinterpret_cast<const ushort4*>(r0), b = *reinterpret_cast<const ushort4*>(r0 + W2);
            q[0][0] = a.x; q[0][1] = a.y; q[0][2] = a.z; q[0][3] = a.w; q[1][0] = b.x; q[1][1] = b.y; q[1][2] = b.z; q[1][3] = b.w;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const bool ok = c < 2 || two; q[0][c] = ok ? r0[c] : 0; q[1][c] = ok ? r0[W2 + c] : 0; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v0 = ((float)q[p.oy[k]][p.ox[k]] - p.black[k]) / p.denom[k];
            const float v1 = ((float)q[p.oy[k]][2 + p.ox[k]] - p.black[k]) / p.denom[k];
            float* o = dst + (size_t)k * hw + (size_t)y * w + x;
            o[0] = fminf(fmaxf(v0, 0.f), 1.f);
            if (two) o[1] = fminf(fmaxf(v1, 0.f), 1.f);
        }
    }
}

extern "C" int eld_pack_raw_bayer_u16(const uint16_t* mosaic, float* packed, int N, int h, int w, const int* raw_pattern, const float* black_level,
                                      float white_point, void* stream) {
    if (N < 0 || h < 0 || w < 0 || !raw_pattern || !black_level) return ELD_EINVAL;
    if (N == 0 || h == 0 || w == 0) return 0;
    if (!mosaic || !packed) return ELD_EINVAL;
    PackRawArgs p;
    bool seen[4] = {false, false, false, false};
    for (int i = 0; i < 4; ++i) {                                    // np.where(raw_pattern == k): position of colour code k in the 2x2 cell
        const int k = raw_pattern[i];
        if (k < 0 || k > 3 || seen[k]) return ELD_EINVAL;
        seen[k] = true; p.oy[k] = i >> 1; p.ox[k] = i & 1;
    }
    for (int k = 0; k < 4; ++k) { p.black[k] = black_level[k]; p.denom[k] = white_point - black_level[k]; }
    const size_t total = (size_t)h * ((w + 1) / 2);
    dim3 grid((unsigned)min((total + 255) / 256, (size_t)4096), N);
    ELD_LAUNCH(pack_raw_kernel, grid, dim3(256), 0, as_stream(stream), mosaic, packed, h, w, p);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Augmentation (sid_dataset.py:344-352): out = transpose?(flipW?(flipH?(x))) per image, optional clip (:354).
//   no transpose: out[c][i][j] = x[c][fh(i)][fw(j)]      transpose: out[c][i][j] = x[c][fh(j)][fw(i)]
// ---------------------------------------------------------------------------------------------
template <typename TI>
__global__ void augment_kernel(const TI* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ aug, int C, int H, int W, uint32_t flags) {
    const int n = blockIdx.y;
    const int bits = aug ? aug[n] : 0;
    const bool fh = bits & 1, fw = bits & 2, tr = (bits & 4) && !(flags & ELD_AUG_NOTRANSPOSE);
    const size_t chw = (size_t)C * H * W, hw = (size_t)H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < chw; e += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / hw);
        const int r = (int)(e - (size_t)c * hw);
        const int i = r / W, j = r - i * W;               // output coordinates (H == W when tr)
        const int a = tr ? j : i, b = tr ? i : j;         // coordinates in the flipped image
        const int sy = fh ? H - 1 - a : a, sx = fw ? W - 1 - b : b;
        float v;
        if constexpr (sizeof(TI) == 2) v = fminf(fmaxf((float)in[(size_t)n * chw + (size_t)c * hw + (size_t)sy * W + sx] / 65535.0f, 0.f), 1.f);   // lmdb_dataset.py:38-39
        else v = in[(size_t)n * chw + (size_t)c * hw + (size_t)sy * W + sx];
        if (flags & ELD_CLIP) v = fmaxf(fminf(v, 1.0f), 0.0f);
        out[(size_t)n * chw + e] = v;
    }
}

template <typename TI>
static int augment_launch(const TI* in, float* out, const int32_t* aug, int N, int C, int H, int W, uint32_t flags, void* stream, bool need_aug) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return ELD_EINVAL;
    const size_t chw = (size_t)C * H * W;
    if (N == 0 || chw == 0) return 0;
    if (!in || !out || (need_aug && !aug) || (const void*)in == (const void*)out) return ELD_EINVAL;
    if (!aug) flags |= ELD_AUG_NOTRANSPOSE;
    // a transposed member of a batched tensor needs H == W; with ELD_AUG_NOTRANSPOSE the caller vouches that no image has bit 4
    // set (the kernel then ignores that bit) and any H, W is accepted
    if (H != W && !(flags & ELD_AUG_NOTRANSPOSE)) return ELD_ENOTSUP;
    dim3 grid((unsigned)min((chw + 255) / 256, (size_t)4096), N);
    ELD_LAUNCH(augment_kernel<TI>, grid, dim3(256), 0, as_stream(stream), in, out, aug, C, H, W, flags);
    ELD_LAUNCH_CHECK();
    return 0;
}

extern "C" int eld_augment(const float* in, float* out, const int32_t* aug, int N, int C, int H, int W, uint32_t flags, void* stream) {
    return augment_launch<float>(in, out, aug, N, C, H, W, flags, stream, true);
}

extern "C" int eld_augment_u16(const uint16_t* in, float* out, const int32_t* aug, int N, int C, int H, int W, uint32_t flags, void* stream) {
    return augment_launch<uint16_t>(in, out, aug, N, C, H, W, flags, stream, false);
}
