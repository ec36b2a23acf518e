// unet_misc.hip -- the HBM-bound pieces of the U-Net step (gfx950): layout conversion at the two
// ends of the network, 2x2 max-pool forward/backward, the 1x1 head (conv10_1) forward/backward,
// weight packing, column sums, L1 loss and Adam.  Every kernel streams its operands once with
// 16-byte lane accesses.  Reference ops: models/arch/Unet.py:13,46,51-63,88 (pool, conv10_1),
// models/losses.py:32 (L1Loss), models/ELD_model.py:400-401,475 (Adam).
#include "unet_misc.h"

// ------------------------------------------------------------------------------------------------
// NCHW (C <= 16 planes) -> NHWC with 16 channels (zero padded): input of conv1_1
// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc16_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, size_t HW) {
    const size_t total = (size_t)N * HW;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t n = p / HW, q = p - n * HW;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = c < C ? x[(n * C + c) * HW + q] : 0.f;
        float4* o = reinterpret_cast<float4*>(y + p * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
}

int launch_nchw_to_nhwc16(const float* x, float* y, int N, int C, int H, int W, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    ELD_LAUNCH(nchw_to_nhwc16_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)16384)), dim3(256), 0, st, x, y, N, C, (size_t)H * W);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 2x2 max pool, NHWC.  One thread = one output pixel x 4 channels (float4).
// ------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int Ho, int Wo, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const float4* src = reinterpret_cast<const float4*>(in + (((size_t)n * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C) + c4;
        const float4 a = src[0], b = src[C4], c = src[(size_t)Wi * C4], d = src[(size_t)Wi * C4 + C4];
        float4 r;
        r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
        r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
        r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
        r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
        reinterpret_cast<float4*>(out)[i] = r;
    }
}

int launch_maxpool_fwd(const float* in, float* out, int N, int Ho, int Wo, int C, hipStream_t st) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (!total) return 0;
    ELD_LAUNCH(maxpool_fwd_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)32768)), dim3(256), 0, st, in, out, N, Ho, Wo, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

// Backward of pool fused with the skip-connection add and the LeakyReLU slope of the pooled tensor:
//   g[pos] = (route(dp)[pos] + skip[pos]) * slope(act[pos])
// route: the whole gradient goes to the FIRST maximum in row-major window order (what torch's CPU
// max_pool2d backward does; all-equal window -> [1,0,0,0]).
// one channel of one 2x2 window: routed pool gradient + skip gradient, times the LeakyReLU slope
#define POOL_BWD_1(F)                                                                          \
    {                                                                                          \
        const float mx = fmaxf(fmaxf(a.F, b.F), fmaxf(c.F, d.F));                              \
        const int sel = a.F == mx ? 0 : (b.F == mx ? 1 : (c.F == mx ? 2 : 3));                 \
        ga.F = ((sel == 0 ? gp.F : 0.f) + sa.F) * lrelu_slope(a.F);                            \
        gb.F = ((sel == 1 ? gp.F : 0.f) + sb.F) * lrelu_slope(b.F);                            \
        gc.F = ((sel == 2 ? gp.F : 0.f) + sc.F) * lrelu_slope(c.F);                            \
        gd.F = ((sel == 3 ? gp.F : 0.f) + sd.F) * lrelu_slope(d.F);                            \
    }

__global__ void maxpool_bwd_kernel(const float* __restrict__ act, const float* __restrict__ dp, const float* __restrict__ skip,
                                   float* __restrict__ g, int N, int Ho, int Wo, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const size_t base = ((((size_t)n * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C) / 4 + c4;
        const size_t o1 = C4, o2 = (size_t)Wi * C4, o3 = o2 + C4;
        const float4* A = reinterpret_cast<const float4*>(act);
        const float4 a = A[base], b = A[base + o1], c = A[base + o2], d = A[base + o3];
        const float4 gp = reinterpret_cast<const float4*>(dp)[i];
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa, sc = sa, sd = sa;
        if (skip) {
            const float4* S = reinterpret_cast<const float4*>(skip);
            sa = S[base]; sb = S[base + o1]; sc = S[base + o2]; sd = S[base + o3];
        }
        float4 ga, gb, gc, gd;
        POOL_BWD_1(x) POOL_BWD_1(y) POOL_BWD_1(z) POOL_BWD_1(w)
        float4* G = reinterpret_cast<float4*>(g);
        G[base] = ga; G[base + o1] = gb; G[base + o2] = gc; G[base + o3] = gd;
    }
}

// The same from CODES instead of the saved un-pooled tensor (round 6): the pool's backward needs two facts about it, which window element took the maximum and
// each element's LeakyReLU slope class -- 2 bits per pooled element (ConvArgs::pool_codes_out) and 2 bits per element (ConvArgs::codes_out), written by the
// forward epilogue that produced the tensor -- 0.31 bytes per element instead of 4: this kernel moves 9.3 B per element instead of 13 and is HBM-bound.
// C % 32 == 0.  Same selections and slopes as maxpool_bwd_kernel, so the same bits.
__global__ void maxpool_bwd_codes_kernel(const unsigned* __restrict__ pool_codes, const unsigned* __restrict__ slope_codes, const float* __restrict__ dp,
                                         const float* __restrict__ skip, float* __restrict__ g, int N, int Ho, int Wo, int C) {
    const int C4 = C / 4, CB = C >> 5;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t pp = i / C4;                                      // pooled pixel (n, yo, xo)
        size_t p = pp;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const int cb = 4 * c4, blk = cb >> 5, e0 = cb & 31;
        const int q = e0 >> 3, hi = (e0 >> 2) & 1;                     // channels 8q + 4hi + j, j = 0..3 <-> code elements 4q + j
        const unsigned pw = pool_codes[(pp * CB + blk) * 2 + hi] >> (4 * q);
        const size_t pix0 = ((size_t)n * 2 * Ho + 2 * yo) * Wi + 2 * xo;
        const size_t pixs[4] = {pix0, pix0 + 1, pix0 + Wi, pix0 + Wi + 1};
        const float4 gp = reinterpret_cast<const float4*>(dp)[i];
        const float gpv[4] = {gp.x, gp.y, gp.z, gp.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned sw = slope_codes[(pixs[k] * CB + blk) * 2 + hi] >> (8 * q);
            float4 sk = make_float4(0.f, 0.f, 0.f, 0.f);
            if (skip) sk = reinterpret_cast<const float4*>(skip)[(pixs[k] * C) / 4 + c4];
            const float skv[4] = {sk.x, sk.y, sk.z, sk.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sel = 2 * (int)((pw >> j) & 1u) + (int)((pw >> (16 + j)) & 1u);
                o[j] = ((sel == k ? gpv[j] : 0.f) + skv[j]) * slope_of_code(sw, j);
            }
            reinterpret_cast<float4*>(g)[(pixs[k] * C) / 4 + c4] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

int launch_maxpool_bwd_codes(const unsigned* pool_codes, const unsigned* slope_codes, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C,
                             hipStream_t st) {
    if (C % 32) return ELD_EINVAL;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (!total) return 0;
    ELD_LAUNCH(maxpool_bwd_codes_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)32768)), dim3(256), 0, st, pool_codes, slope_codes, dp, skip, g, N, Ho, Wo, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_maxpool_bwd(const float* act, const float* dp, const float* skip, float* g, int N, int Ho, int Wo, int C, hipStream_t st) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (!total) return 0;
    ELD_LAUNCH(maxpool_bwd_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)32768)), dim3(256), 0, st, act, dp, skip, g, N, Ho, Wo, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// conv10_1: 1x1, 32 -> OC (<= 4), no activation; NHWC in, NCHW out (Unet.py:46,88)
// ------------------------------------------------------------------------------------------------
// LPP lanes per pixel (8 x 4 fp32 channels / 4 x 8 bf16 channels: a wave reads whole pixels, 1 KiB contiguous per instruction); the 32-channel sum
// is formed as per-lane fma chains + a butterfly over the pixel's lanes -- the SAME order as the fused training head (head_train_kernel), so the
// inference output and the training step's output are the same bits.
__device__ __forceinline__ float quad_xor_add(float v, int xor2) {
    const int i = __float_as_int(v);
    return v + __int_as_float(xor2 ? __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true));
}
// out[o] - b[o] for the pixel whose CPL channels av[] this lane holds (wq[o][j] = W[o][CPL cq + j]); every lane of the pixel returns the same bits
template <int CPL>
__device__ __forceinline__ void head_dot(const float (&av)[CPL], const float (&wq)[4][CPL], float (&po)[4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) sacc = fmaf(av[j], wq[o][j], sacc);
        sacc = quad_xor_add(sacc, 0);
        sacc = quad_xor_add(sacc, 1);
        if (CPL == 4) sacc += __shfl_xor(sacc, 4, 64);
        po[o] = sacc;
    }
}
template <typename T, int CPL>
__device__ __forceinline__ void head_load(const T* act, size_t pc, int cq, float (&av)[CPL]) {
    if constexpr (sizeof(T) == 4) {
        const float4 a4 = reinterpret_cast<const float4*>(act + pc * 32)[cq];
        av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
    } else {
        const uint4 a4 = reinterpret_cast<const uint4*>(act + pc * 32)[cq];
        const float4 lo = unpack_bf4(make_uint2(a4.x, a4.y)), hi4 = unpack_bf4(make_uint2(a4.z, a4.w));
        av[0] = lo.x; av[1] = lo.y; av[2] = lo.z; av[3] = lo.w; av[4] = hi4.x; av[5] = hi4.y; av[6] = hi4.z; av[7] = hi4.w;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ out, int N, size_t HW, int OC) {
    constexpr int CPL = sizeof(T) == 4 ? 4 : 8, LPP = 32 / CPL, PPB = 256 / LPP;
    const int tid = threadIdx.x, cq = tid & (LPP - 1), o_ld = tid & 3;
    float wq[4][CPL];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < CPL; ++j) wq[o][j] = o < OC ? w[o * 32 + CPL * cq + j] : 0.f;
    const float bo = o_ld < OC ? b[o_ld] : 0.f;
    const size_t total = (size_t)N * HW;
    const size_t stride = (size_t)gridDim.x * PPB;
    size_t p = (size_t)blockIdx.x * PPB + (tid / LPP);
    const size_t iters = (total + stride - 1) / stride;               // every lane of a pixel group runs the same trip count (the exchanges need its mates)
    for (size_t itn = 0; itn < iters; ++itn, p += stride) {
        const bool ok = p < total;
        const size_t pc = ok ? p : total - 1;
        float av[CPL], po[4];
        head_load<T, CPL>(in, pc, cq, av);
        head_dot<CPL>(av, wq, po);
        const float mine = (o_ld == 0 ? po[0] : (o_ld == 1 ? po[1] : (o_ld == 2 ? po[2] : po[3]))) + bo;
        const size_t n = pc / HW, q = pc - n * HW;
        if (ok && cq < 4 && o_ld < OC) out[(n * OC + o_ld) * HW + q] = mine;
    }
}

int launch_head_fwd(const float* in, const float* w, const float* b, float* out, int N, int H, int W, int OC, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    ELD_LAUNCH(head_fwd_kernel<float>, dim3((unsigned)min((total + 31) / 32, (size_t)16384)), dim3(256), 0, st, in, w, b, out, N, (size_t)H * W, OC);
    ELD_LAUNCH_CHECK();
    return 0;
}

// backward: g[p][c] = (sum_o W[o][c] d[o][p]) * slope(act[p][c]);  dW[o][c] = sum_p d[o][p] act[p][c];  db[o] = sum_p d[o][p]
// per-block partials [nblocks][132] -> reduced by head_bwd_reduce_kernel in fixed order.
#define HEAD_BLOCKS 1024
// Eight lanes per pixel, one channel quad each: a wave reads / writes 8 whole pixels = 1 KiB contiguous per instruction (the one-thread-per-
// pixel layout touched 64 cache lines per load).  The pixel's four output gradients are read once (lane o of each quad of lanes reads plane o)
// and handed round with quad-permute DPP moves.  A lane keeps dW[o][its 4 channels] (16 sums) and, in quad 0, db[o].
__device__ __forceinline__ float quad_bcast(float v, int src) {
    const int i = __float_as_int(v);
    switch (src) {
        case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x00, 0xF, 0xF, true));
        case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x55, 0xF, 0xF, true));
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xAA, 0xF, 0xF, true));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xFF, 0xF, 0xF, true));
    }
}

template <typename T>       // T = float or bf16_t: element type of the saved activation and of the gradient written (math in fp32 either way)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ act, const float* __restrict__ w,
                                                       T* __restrict__ g, float* __restrict__ part, int N, size_t HW, int OC) {
    __shared__ float red[4][132];
    const int tid = threadIdx.x, cq = tid & 7, o_ld = tid & 3;
    float wq[4][4];                                   // w[o][4cq + j]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[o][j] = o < OC ? w[o * 32 + 4 * cq + j] : 0.f;
    float dw[4][4];
    float db = 0.f;                                   // lanes with cq < 4 sum plane o_ld = cq
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[o][j] = 0.f;
    const size_t total = (size_t)N * HW;
    const size_t stride = (size_t)gridDim.x * 32;
    size_t p = (size_t)blockIdx.x * 32 + (tid >> 3);
    const size_t iters = (total + stride - 1) / stride;               // every lane of a quad runs the same trip count (DPP needs its quad mates)
    for (size_t itn = 0; itn < iters; ++itn, p += stride) {
        const bool ok = p < total;
        const size_t pc = ok ? p : total - 1;
        const size_t n = pc / HW, q = pc - n * HW;
        float dl = 0.f;
        if (ok && o_ld < OC) dl = dout[(n * OC + o_ld) * HW + q];
        if (cq < 4) db += dl;
        float d[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) d[o] = quad_bcast(dl, o);
        if (!ok) continue;
        float4 a;
        if constexpr (sizeof(T) == 4) a = reinterpret_cast<const float4*>(act + pc * 32)[cq];
        else a = unpack_bf4(reinterpret_cast<const uint2*>(act + pc * 32)[cq]);
        const float av[4] = {a.x, a.y, a.z, a.w};
        float gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) { s = fmaf(wq[o][j], d[o], s); dw[o][j] = fmaf(d[o], av[j], dw[o][j]); }
            gv[j] = s * lrelu_slope(av[j]);
        }
        if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(g + pc * 32)[cq] = make_float4(gv[0], gv[1], gv[2], gv[3]);
        else reinterpret_cast<uint2*>(g + pc * 32)[cq] = pack_bf4(make_float4(gv[0], gv[1], gv[2], gv[3]));
    }
    // block reduction in a fixed order: lanes sharing a channel quad (cq, cq+8, ...) by xor-shuffles, then the 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = dw[o][j];
            for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < 8) red[wave][o * 32 + 4 * lane + j] = v;
        }
    {
        float v = db;
        for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < 4) red[wave][128 + lane] = v;
    }
    __syncthreads();
    if (tid < 132)
        part[(size_t)blockIdx.x * 132 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// 132 outputs, 16 lanes each over the per-block partials; fixed shuffle tree
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nblocks, int OC) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = (blockIdx.x * 4 + wave) * 4 + (lane & 3), slice = lane >> 2;
    float s = 0.f;
    if (t < 132)
        for (int b = slice; b < nblocks; b += 16) s += part[(size_t)b * 132 + t];
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
    if (slice != 0 || t >= 132) return;
    if (t < 128) { if (t / 32 < OC) dw[t] = s; } else if (t - 128 < OC) db[t - 128] = s;
}

size_t head_bwd_ws_floats() { return (size_t)HEAD_BLOCKS * 132; }

int launch_head_bwd(const float* dout, const float* act, const float* w, float* g, float* dw, float* db, float* part,
                    int N, int H, int W, int OC, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    const int nb = (int)min((total + 31) / 32, (size_t)HEAD_BLOCKS);
    ELD_LAUNCH(head_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, dout, act, w, g, part, N, (size_t)H * W, OC);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(head_bwd_reduce_kernel, dim3((132 + 15) / 16), dim3(256), 0, st, part, dw, db, nb, OC);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// column sums of an NHWC matrix [P][C] -> [C]  (bias gradient of the transposed convs)
// ------------------------------------------------------------------------------------------------
#define COLSUM_BLOCKS 2048
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ part, size_t P, int C) {
    // thread t handles channel (t % C4)*4.. of pixels t / C4 + k*(256/C4 * gridDim)
    const int C4 = C / 4;
    const int ppb = 256 / C4;                  // pixels per block-iteration (C4 <= 128 -> ppb >= 2); threads beyond ppb*C4 idle
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < ppb) {
        const size_t step = (size_t)gridDim.x * ppb;
        size_t p = (size_t)blockIdx.x * ppb + pl;
        for (; p + 3 * step < P; p += 4 * step) {            // four independent 16-byte loads in flight per lane
            const float4 v0 = reinterpret_cast<const float4*>(x + p * C)[c4], v1 = reinterpret_cast<const float4*>(x + (p + step) * C)[c4];
            const float4 v2 = reinterpret_cast<const float4*>(x + (p + 2 * step) * C)[c4], v3 = reinterpret_cast<const float4*>(x + (p + 3 * step) * C)[c4];
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; p < P; p += step) {
            const float4 v = reinterpret_cast<const float4*>(x + p * C)[c4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    __shared__ float4 sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ppb; ++k) { const float4 v = sh[k * C4 + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        reinterpret_cast<float4*>(part + (size_t)blockIdx.x * C)[threadIdx.x] = t;
    }
}

// 64 channels per workgroup (lanes) x 16 waves over the per-block partials, 4 loads in flight, fixed-order LDS combine
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblocks, int C) {
    __shared__ float sh[16 * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C) {
        int b = w;
        for (; b + 48 < nblocks; b += 64)
            s += (part[(size_t)b * C + c] + part[(size_t)(b + 16) * C + c]) + (part[(size_t)(b + 32) * C + c] + part[(size_t)(b + 48) * C + c]);
        for (; b < nblocks; b += 16) s += part[(size_t)b * C + c];
    }
    sh[w * 64 + lane] = s;
    __syncthreads();
    if (w == 0 && c < C) {
        for (int ww = 1; ww < 16; ++ww) s += sh[ww * 64 + lane];
        out[c] = s;
    }
}

size_t colsum_ws_floats(int C) { return (size_t)COLSUM_BLOCKS * C; }

// out[c] = sum_b part[b][c]: second stage alone, for partials another kernel produced (wgrad_kernel's xbpart)
int launch_colsum_reduce(const float* part, float* out, int nblocks, int C, hipStream_t st) {
    if (nblocks <= 0 || C <= 0) return 0;
    ELD_LAUNCH(colsum_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, out, nblocks, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_colsum(const float* x, float* out, float* part, size_t P, int C, hipStream_t st) {
    if (C % 4 || C > 512) return ELD_EINVAL;
    const int ppb = 256 / (C / 4);
    const int nb = (int)min((P + ppb - 1) / ppb, (size_t)COLSUM_BLOCKS);
    if (nb == 0) return 0;
    ELD_LAUNCH(colsum_kernel, dim3(nb), dim3(256), 0, st, x, part, P, C);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(colsum_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, out, nb, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight packing (tiny: 7.76 M parameters in total)
//   PACK_CONV_FWD : dst[t][co][ci_p] = src[co][ci][t]        (zero for ci >= Cin)       src OIHW
//   PACK_CONV_BWD : dst[t][ci][co]   = src[co][ci][T-1-t]    (flipped taps, transposed)
//   PACK_CONVT_FWD: dst[tap*Cout+co][ci] = src[ci][co][tap]                              src (Cin,Cout,2,2)
//   PACK_CONVT_BWD: dst[tap][ci][co]     = src[ci][co][tap]
// ------------------------------------------------------------------------------------------------
// Pre-split slab layout of conv_x3d_kernel (conv_x3.hip): element (tap t, GEMM column n of N, k-channel c of K) of the logical packed
// weight B[t][n][c] is cut EXACTLY into its three bf16 pieces (top 16 bits, top 16 bits of the exact remainder, exact rest -- the same
// cut split_store makes) and stored at   slab(ky, c/16, n/BN) + row(kx, n%BN) * 112 B + piece * 32 B + (c%16) * 2 B.
__device__ __forceinline__ void x3_store(float* dst, int t, int n, int c, int N, int K, int BN, float v) {
    const int NCH = K >> 4, NB = N / BN;
    const int ky = t / 3, kx = t - 3 * ky;
    const size_t slab = (size_t)(ky * NCH + (c >> 4)) * NB + n / BN;
    bf16_t* d = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(dst) + slab * x3_slab_stride(BN)) + (size_t)(kx * BN + (n % BN)) * 56 + (c & 15);
    const unsigned x = __float_as_uint(v);
    const float r = v - __uint_as_float(x & 0xFFFF0000u);
    const unsigned y = __float_as_uint(r);
    const float q = r - __uint_as_float(y & 0xFFFF0000u);
    d[0] = (bf16_t)(x >> 16); d[16] = (bf16_t)(y >> 16); d[32] = (bf16_t)(__float_as_uint(q) >> 16);
}

// Slab layout of conv_bfd_kernel (conv_bfd.hip): element (tap t, GEMM column n of N, k-channel c of K) of the logical packed weight
// B[t][n][c] as bf16 at   slab(ky, c/32, n/BN) + slot * 16 B + (c%8) * 2 B,   slot = 4 P + (octet ^ ((P >> 2) & 3)),  P = kx*BN + n%BN,
// octet = (c%32)/8: the exact (XOR-swizzled) LDS image of a stage's slab, so it travels by linear 1 KiB LDS-DMA pieces.
__device__ __forceinline__ void bfd_store(float* dst, int t, int n, int c, int N, int K, int BN, float v) {
    const int NCH = K >> 5, NB = N / BN;
    const int ky = t / 3, kx = t - 3 * ky;
    const size_t slab = (size_t)(ky * NCH + (c >> 5)) * NB + n / BN;
    const int P = kx * BN + (n % BN), o = (c & 31) >> 3;
    const int slot = 4 * P + (o ^ ((P >> 2) & 3));
    reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(dst) + slab * bfd_slab_bytes(BN))[slot * 8 + (c & 7)] = f2bf(v);
}

// Slab layout of conv_bfg_kernel (conv_bfg.hip): element (GEMM column n of N, k index c of K) of the logical packed weight B[n][c] as bf16 at
//   slab(c/32, n/BN) + slot * 16 B + (c%8) * 2 B,   slot = 4 n' + (octet ^ ((n' >> 2) & 3)),  n' = n % BN, octet = (c%32)/8
__device__ __forceinline__ void bfg_store(float* dst, int n, int c, int N, int K, int BN, float v) {
    const int NB = N / BN, np = n % BN;
    const size_t slab = (size_t)(c >> 5) * NB + n / BN;
    const int slot = 4 * np + (((c & 31) >> 3) ^ ((np >> 2) & 3));
    reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(dst) + slab * bfg_slab_bytes(BN))[slot * 8 + (c & 7)] = f2bf(v);
}

__global__ void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int kind, int Cout, int Cin, int Cinp, int T, int x3bn) {
    size_t total;
    if (kind == PACK_CONV_FWD) total = (size_t)T * Cout * Cinp; else total = (size_t)T * Cout * Cin;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (kind == PACK_CONV_FWD) {
        const int ci = (int)(i % Cinp); const int co = (int)((i / Cinp) % Cout); const int t = (int)(i / ((size_t)Cinp * Cout));
        const float v = ci < Cin ? src[((size_t)co * Cin + ci) * T + t] : 0.f;
        if (x3bn) x3_store(dst, t, co, ci, Cout, Cinp, x3bn, v); else dst[i] = v;
    } else if (kind == PACK_CONV_BWD) {
        const int co = (int)(i % Cout); const int ci = (int)((i / Cout) % Cin); const int t = (int)(i / ((size_t)Cout * Cin));
        const float v = src[((size_t)co * Cin + ci) * T + (T - 1 - t)];
        if (x3bn) x3_store(dst, t, ci, co, Cin, Cout, x3bn, v); else dst[i] = v;
    } else if (kind == PACK_CONVT_FWD) {
        const int ci = (int)(i % Cin); const int co = (int)((i / Cin) % Cout); const int t = (int)(i / ((size_t)Cin * Cout));
        dst[i] = src[((size_t)ci * Cout + co) * T + t];
    } else {
        const int co = (int)(i % Cout); const int ci = (int)((i / Cout) % Cin); const int t = (int)(i / ((size_t)Cout * Cin));
        dst[i] = src[((size_t)ci * Cout + co) * T + t];
    }
}

int launch_pack(const float* src, float* dst, int kind, int Cout, int Cin, int Cinp, int T, hipStream_t st, int x3bn) {
    const size_t total = (size_t)T * Cout * (kind == PACK_CONV_FWD ? Cinp : Cin);
    ELD_LAUNCH(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, kind, Cout, Cin, Cinp, T, x3bn);
    ELD_LAUNCH_CHECK();
    return 0;
}

// max |x| of a tensor into *slot (atomic max on the bit pattern; slot zeroed by the caller) -- operand bound of the two-piece
// fp16 product scheme for tensors whose producer does not report it
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ slot) {
    float mx = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) mx = fmaxf(mx, fabsf(x[n4 * 4 + threadIdx.x]));
    amax_accumulate(slot, mx);
}

int launch_absmax(const float* x, size_t n, float* slot, hipStream_t st) {
    if (!n) return 0;
    const unsigned nb = (unsigned)min((n / 4 + 255) / 256 + 1, (size_t)2048);
    ELD_LAUNCH(absmax_kernel, dim3(nb), dim3(256), 0, st, x, n, slot);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ---- eight consecutive k-channels (c0 .. c0 + 7, c0 % 8 == 0) of one (tap, column) per thread: in every packed layout these are 16 contiguous bytes ----
__device__ __forceinline__ uint4 bf8(const float (&v)[8]) { return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])); }
__device__ __forceinline__ unsigned hi16_pair(float lo, float hi) { return (__float_as_uint(hi) & 0xFFFF0000u) | (__float_as_uint(lo) >> 16); }
// x3_store for the eight values: the three pieces as three 16-byte stores
__device__ __forceinline__ void x3_store8(float* dst, int t, int n, int c0, int N, int K, int BN, const float (&v)[8]) {
    const int NCH = K >> 4, NB = N / BN;
    const int ky = t / 3, kx = t - 3 * ky;
    const size_t slab = (size_t)(ky * NCH + (c0 >> 4)) * NB + n / BN;
    bf16_t* d = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(dst) + slab * x3_slab_stride(BN)) + (size_t)(kx * BN + (n % BN)) * 56 + (c0 & 15);
    float r[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r[j] = v[j] - __uint_as_float(__float_as_uint(v[j]) & 0xFFFF0000u);
        q[j] = r[j] - __uint_as_float(__float_as_uint(r[j]) & 0xFFFF0000u);
    }
    *reinterpret_cast<uint4*>(d) = make_uint4(hi16_pair(v[0], v[1]), hi16_pair(v[2], v[3]), hi16_pair(v[4], v[5]), hi16_pair(v[6], v[7]));
    *reinterpret_cast<uint4*>(d + 16) = make_uint4(hi16_pair(r[0], r[1]), hi16_pair(r[2], r[3]), hi16_pair(r[4], r[5]), hi16_pair(r[6], r[7]));
    *reinterpret_cast<uint4*>(d + 32) = make_uint4(hi16_pair(q[0], q[1]), hi16_pair(q[2], q[3]), hi16_pair(q[4], q[5]), hi16_pair(q[6], q[7]));
}
__device__ __forceinline__ void bfd_store8(float* dst, int t, int n, int c0, int N, int K, int BN, const float (&v)[8]) {
    const int NCH = K >> 5, NB = N / BN;
    const int ky = t / 3, kx = t - 3 * ky;
    const size_t slab = (size_t)(ky * NCH + (c0 >> 5)) * NB + n / BN;
    const int P = kx * BN + (n % BN), o = (c0 & 31) >> 3;
    const int slot = 4 * P + (o ^ ((P >> 2) & 3));
    reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + slab * bfd_slab_bytes(BN))[slot] = bf8(v);
}
__device__ __forceinline__ void bfg_store8(float* dst, int n, int c0, int N, int K, int BN, const float (&v)[8]) {
    const int NB = N / BN, np = n % BN;
    const size_t slab = (size_t)(c0 >> 5) * NB + n / BN;
    const int slot = 4 * np + (((c0 & 31) >> 3) ^ ((np >> 2) & 3));
    reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + slab * bfg_slab_bytes(BN))[slot] = bf8(v);
}

// all layers of the network in ONE launch: block b belongs to job j with first_block[j] <= b < first_block[j+1].  A thread packs eight consecutive
// k-channels of one (tap, column) -- 16 contiguous bytes per piece in every layout -- and the threads of a wave walk the SOURCE tensor in its own order
// (OIHW / IOHW: the taps fastest), so that the eight strided loads of a wave cover whole cache lines between them:
//   PACK_CONV_FWD   k = ci (source stride T):        thread = (co, ci / 8, t)
//   PACK_CONV_BWD   k = co (source stride Cin T):    thread = (co / 8, ci, source tap)
//   PACK_CONVT_FWD  k = ci (source stride Cout T):   thread = (ci / 8, co, tap)
//   PACK_CONVT_BWD  k = tap * Cout + co (stride T):  thread = (ci, co / 8, tap)
__global__ __launch_bounds__(256) void pack_all_kernel(const PackJobs jobs, const float* __restrict__ params, float* __restrict__ ws, float* __restrict__ amax) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].first_block) ++j;
    const PackJob J = jobs.job[j];
    const int Cout = J.Cout, Cin = J.Cin, Cinp = J.Cinp, T = J.T;
    const size_t groups = (size_t)T * Cout * (J.kind == PACK_CONV_FWD ? Cinp : Cin) / 8;
    const size_t i = (size_t)(blockIdx.x - J.first_block) * blockDim.x + threadIdx.x;
    const bool live = i < groups;
    const float* src = params + J.src_off;
    float* dst = ws + J.dst_off;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    int t = 0, n = 0, c0 = 0;          // tap, column and first k-channel of the packed matrix B[t][n][c]
    size_t plain = 0;                  // index of the vector in the plain (un-slabbed) layout
    if (!live) {
    } else if (J.kind == PACK_CONV_FWD) {
        const int G = Cinp >> 3;
        t = (int)(i % T); const int g = (int)((i / T) % G); n = (int)(i / ((size_t)T * G)); c0 = 8 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c0 + e < Cin) v[e] = src[((size_t)n * Cin + c0 + e) * T + t];
        plain = ((size_t)t * Cout + n) * Cinp + c0;
    } else if (J.kind == PACK_CONV_BWD) {
        const int ts = (int)(i % T); n = (int)((i / T) % Cin); c0 = 8 * (int)(i / ((size_t)T * Cin)); t = T - 1 - ts;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[((size_t)(c0 + e) * Cin + n) * T + ts];
        plain = ((size_t)t * Cin + n) * Cout + c0;
    } else if (J.kind == PACK_CONVT_FWD) {
        t = (int)(i % T); const int co = (int)((i / T) % Cout); c0 = 8 * (int)(i / ((size_t)T * Cout)); n = t * Cout + co;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[((size_t)(c0 + e) * Cout + co) * T + t];
        plain = (size_t)n * Cin + c0;
    } else {
        const int G = Cout >> 3;
        t = (int)(i % T); const int g = (int)((i / T) % G); n = (int)(i / ((size_t)T * G)); c0 = t * Cout + 8 * g;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[((size_t)n * Cout + 8 * g + e) * T + t];
        plain = ((size_t)t * Cin + n) * Cout + 8 * g;
    }
    if (live) {
        if (J.bf16 && J.bfdbn && J.kind == PACK_CONV_FWD) bfd_store8(dst, t, n, c0, Cout, Cinp, J.bfdbn, v);
        else if (J.bf16 && J.bfdbn && J.kind == PACK_CONV_BWD) bfd_store8(dst, t, n, c0, Cin, Cout, J.bfdbn, v);
        else if (J.bf16 && J.bfgbn && J.kind == PACK_CONVT_FWD) bfg_store8(dst, n, c0, T * Cout, Cin, J.bfgbn, v);          // n = tap*Cout + co, k = ci
        else if (J.bf16 && J.bfgbn && J.kind == PACK_CONVT_BWD) bfg_store8(dst, n, c0, Cin, T * Cout, J.bfgbn, v);          // n = ci, k = tap*Cout + co
        else if (J.bf16) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + plain) = bf8(v);
        else if (J.x3bn && J.kind == PACK_CONV_FWD) x3_store8(dst, t, n, c0, Cout, Cinp, J.x3bn, v);
        else if (J.x3bn && J.kind == PACK_CONV_BWD) x3_store8(dst, t, n, c0, Cin, Cout, J.x3bn, v);
        else {
            *reinterpret_cast<float4*>(dst + plain) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(dst + plain + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    if (amax) {      // per-layer weight bound (all lanes take part)
        float m = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
        amax_accumulate(amax + J.amax_slot, m);
    }
}

int launch_pack_all(PackJobs& jobs, const float* params, float* ws, hipStream_t st, float* amax) {
    int blocks = 0;
    for (int j = 0; j < jobs.n; ++j) {
        PackJob& J = jobs.job[j];
        const size_t groups = (size_t)J.T * J.Cout * (J.kind == PACK_CONV_FWD ? J.Cinp : J.Cin) / 8;      // Cinp % 16 == 0, every other channel count % 32 == 0
        J.first_block = blocks;
        blocks += (int)((groups + 255) / 256);
    }
    if (!blocks) return 0;
    ELD_LAUNCH(pack_all_kernel, dim3(blocks), dim3(256), 0, st, jobs, params, ws, amax);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// L1 loss (mean |out - target|) forward + backward in one pass; two-pass deterministic reduction.
//   dout = sign(out - target) * scale / numel     (torch: sign(0) = 0)
// ------------------------------------------------------------------------------------------------
#define L1_BLOCKS 1024
template <int MSE>
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ out, const float* __restrict__ tgt, float* __restrict__ dout,
                                                 float* __restrict__ part, size_t n, float gscale) {
    float s = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(out)[i], b = reinterpret_cast<const float4*>(tgt)[i];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        if (MSE) s += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        else s += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
        if (dout && MSE) {
            reinterpret_cast<float4*>(dout)[i] = make_float4(2.0f * d0 * gscale, 2.0f * d1 * gscale, 2.0f * d2 * gscale, 2.0f * d3 * gscale);
        } else if (dout) {
            float4 g;
            g.x = d0 > 0.f ? gscale : (d0 < 0.f ? -gscale : 0.f);
            g.y = d1 > 0.f ? gscale : (d1 < 0.f ? -gscale : 0.f);
            g.z = d2 > 0.f ? gscale : (d2 < 0.f ? -gscale : 0.f);
            g.w = d3 > 0.f ? gscale : (d3 < 0.f ? -gscale : 0.f);
            reinterpret_cast<float4*>(dout)[i] = g;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float d = out[i] - tgt[i];
        s += MSE ? d * d : fabsf(d);
        if (dout) dout[i] = MSE ? 2.0f * d * gscale : (d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f));
    }
    __shared__ float sh[4];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void l1_reduce_kernel(const float* __restrict__ part, float* __restrict__ loss, int nblocks, float inv_n) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)part[b];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(sh[0] * (double)inv_n);
}

size_t l1_ws_floats() { return L1_BLOCKS; }

int launch_l1(const float* out, const float* tgt, float* dout, float* loss, float* part, size_t n, float grad_scale, hipStream_t st) {
    if (n == 0) return ELD_EINVAL;
    const int nb = (int)min((n / 4 + 255) / 256 + 1, (size_t)L1_BLOCKS);
    ELD_LAUNCH(l1_kernel<0>, dim3(nb), dim3(256), 0, st, out, tgt, dout, part, n, grad_scale / (float)n);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(l1_reduce_kernel, dim3(1), dim3(256), 0, st, part, loss, nb, 1.0f / (float)n);
    ELD_LAUNCH_CHECK();
    return 0;
}

// nn.MSELoss (models/losses.py:34): mean (out - target)^2, gradient 2 (out - target) * grad_scale / n
int launch_mse(const float* out, const float* tgt, float* dout, float* loss, float* part, size_t n, float grad_scale, hipStream_t st) {
    if (n == 0) return ELD_EINVAL;
    const int nb = (int)min((n / 4 + 255) / 256 + 1, (size_t)L1_BLOCKS);
    ELD_LAUNCH(l1_kernel<1>, dim3(nb), dim3(256), 0, st, out, tgt, dout, part, n, grad_scale / (float)n);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(l1_reduce_kernel, dim3(1), dim3(256), 0, st, part, loss, nb, 1.0f / (float)n);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam over one flat parameter buffer (torch.optim.Adam semantics, amsgrad off, weight decay wd
// added to the gradient):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float step_size, float b1, float b2, float eps, float wd, float bc2_sqrt, float gscale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * gscale;
    if (wd != 0.f) gi = gi + wd * p[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = __builtin_sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// scalars are formed in double on the host exactly as torch/optim/adam.py does, then rounded to float
int launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps, double wd,
                int step, double gscale, hipStream_t st) {
    if (!n) return 0;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    ELD_LAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, (float)(lr / bc1), (float)b1,
                       (float)b2, (float)eps, (float)wd, (float)sqrt(bc2), (float)gscale);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bf16 activation variants of the streaming kernels (forward path): 4 channels = 8 bytes per lane access
// ------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int N, int Ho, int Wo, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const uint2* src = reinterpret_cast<const uint2*>(in + (((size_t)n * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C) + c4;
        const float4 a = unpack_bf4(src[0]), b = unpack_bf4(src[C4]), c = unpack_bf4(src[(size_t)Wi * C4]), d = unpack_bf4(src[(size_t)Wi * C4 + C4]);
        float4 r;
        r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
        r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
        r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
        r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
        reinterpret_cast<uint2*>(out)[i] = pack_bf4(r);
    }
}

int launch_maxpool_fwd_bf16(const bf16_t* in, bf16_t* out, int N, int Ho, int Wo, int C, hipStream_t st) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (!total) return 0;
    ELD_LAUNCH(maxpool_fwd_bf16_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)32768)), dim3(256), 0, st, in, out, N, Ho, Wo, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_head_fwd_bf16(const bf16_t* in, const float* w, const float* b, float* out, int N, int H, int W, int OC, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    ELD_LAUNCH(head_fwd_kernel<bf16_t>, dim3((unsigned)min((total + 63) / 64, (size_t)16384)), dim3(256), 0, st, in, w, b, out, N, (size_t)H * W, OC);
    ELD_LAUNCH_CHECK();
    return 0;
}

// ---- backward pieces with bf16 activations / gradients (fp32 math, fp32 weight-gradient outputs) -----------------------
__global__ void maxpool_bwd_bf16_kernel(const bf16_t* __restrict__ act, const bf16_t* __restrict__ dp, const bf16_t* __restrict__ skip,
                                        bf16_t* __restrict__ g, int N, int Ho, int Wo, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)N * Ho * Wo * C4;
    const int Wi = 2 * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t p = i / C4;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const size_t base = ((((size_t)n * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C) / 4 + c4;
        const size_t o1 = C4, o2 = (size_t)Wi * C4, o3 = o2 + C4;
        const uint2* A = reinterpret_cast<const uint2*>(act);
        const float4 a = unpack_bf4(A[base]), b = unpack_bf4(A[base + o1]), c = unpack_bf4(A[base + o2]), d = unpack_bf4(A[base + o3]);
        const float4 gp = unpack_bf4(reinterpret_cast<const uint2*>(dp)[i]);
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa, sc = sa, sd = sa;
        if (skip) {
            const uint2* S = reinterpret_cast<const uint2*>(skip);
            sa = unpack_bf4(S[base]); sb = unpack_bf4(S[base + o1]); sc = unpack_bf4(S[base + o2]); sd = unpack_bf4(S[base + o3]);
        }
        float4 ga, gb, gc, gd;
        POOL_BWD_1(x) POOL_BWD_1(y) POOL_BWD_1(z) POOL_BWD_1(w)
        uint2* G = reinterpret_cast<uint2*>(g);
        G[base] = pack_bf4(ga); G[base + o1] = pack_bf4(gb); G[base + o2] = pack_bf4(gc); G[base + o3] = pack_bf4(gd);
    }
}

int launch_maxpool_bwd_bf16(const bf16_t* act, const bf16_t* dp, const bf16_t* skip, bf16_t* g, int N, int Ho, int Wo, int C, hipStream_t st) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (!total) return 0;
    ELD_LAUNCH(maxpool_bwd_bf16_kernel, dim3((unsigned)min((total + 255) / 256, (size_t)32768)), dim3(256), 0, st, act, dp, skip, g, N, Ho, Wo, C);
    ELD_LAUNCH_CHECK();
    return 0;
}

// bf16 head backward: FOUR lanes per pixel, eight channels (16 bytes) each -- a wave reads / writes 16 whole pixels = 1 KiB contiguous per
// instruction (head_bwd_kernel<float>'s eight-lane layout at 8 bytes per lane measured slower than one thread per pixel).  The pixel's four
// output gradients are read once (lane o of the quad reads plane o) and handed round with quad-permute DPP moves; a lane keeps
// dW[o][its 8 channels] (32 sums) and db[its plane].
__global__ __launch_bounds__(256) void head_bwd_bf16_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ act, const float* __restrict__ w,
                                                            bf16_t* __restrict__ g, float* __restrict__ part, int N, size_t HW, int OC) {
    __shared__ float red[4][132];
    const int tid = threadIdx.x, cq = tid & 3;
    float wq[4][8];                                   // w[o][8cq + j]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) wq[o][j] = o < OC ? w[o * 32 + 8 * cq + j] : 0.f;
    float dw[4][8];
    float db = 0.f;                                   // lane cq sums plane cq
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) dw[o][j] = 0.f;
    const size_t total = (size_t)N * HW;
    const size_t stride = (size_t)gridDim.x * 64;
    size_t p = (size_t)blockIdx.x * 64 + (tid >> 2);
    const size_t iters = (total + stride - 1) / stride;               // every lane of a quad runs the same trip count (DPP needs its quad mates)
    for (size_t itn = 0; itn < iters; ++itn, p += stride) {
        const bool ok = p < total;
        const size_t pc = ok ? p : total - 1;
        const size_t n = pc / HW, q = pc - n * HW;
        float dl = 0.f;
        if (ok && cq < OC) dl = dout[(n * OC + cq) * HW + q];
        db += dl;
        float d[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) d[o] = quad_bcast(dl, o);
        if (!ok) continue;
        const uint4 av4 = reinterpret_cast<const uint4*>(act + pc * 32)[cq];
        const float4 alo = unpack_bf4(make_uint2(av4.x, av4.y)), ahi = unpack_bf4(make_uint2(av4.z, av4.w));
        const float av[8] = {alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w};
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) { sacc = fmaf(wq[o][j], d[o], sacc); dw[o][j] = fmaf(d[o], av[j], dw[o][j]); }
            gv[j] = sacc * lrelu_slope(av[j]);
        }
        const uint2 g0 = pack_bf4(make_float4(gv[0], gv[1], gv[2], gv[3])), g1 = pack_bf4(make_float4(gv[4], gv[5], gv[6], gv[7]));
        reinterpret_cast<uint4*>(g + pc * 32)[cq] = make_uint4(g0.x, g0.y, g1.x, g1.y);
    }
    // block reduction in a fixed order: lanes sharing a channel octet (cq, cq+4, ...) by xor-shuffles, then the 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = dw[o][j];
            for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < 4) red[wave][o * 32 + 8 * lane + j] = v;
        }
    {
        float v = db;
        for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < 4) red[wave][128 + lane] = v;
    }
    __syncthreads();
    if (tid < 132)
        part[(size_t)blockIdx.x * 132 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

int launch_head_bwd_bf16(const float* dout, const bf16_t* act, const float* w, bf16_t* g, float* dw, float* db, float* part,
                         int N, int H, int W, int OC, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    const int nb = (int)min((total + 63) / 64, (size_t)HEAD_BLOCKS);
    ELD_LAUNCH(head_bwd_bf16_kernel, dim3(nb), dim3(256), 0, st, dout, act, w, g, part, N, (size_t)H * W, OC);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(head_bwd_reduce_kernel, dim3((132 + 15) / 16), dim3(256), 0, st, part, dw, db, nb, OC);
    ELD_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, size_t P, int C) {
    const int C4 = C / 4;
    const int ppb = 256 / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < ppb) {
        const size_t step = (size_t)gridDim.x * ppb;
        size_t p = (size_t)blockIdx.x * ppb + pl;
        for (; p + 3 * step < P; p += 4 * step) {
            const uint2 q0 = reinterpret_cast<const uint2*>(x + p * C)[c4], q1 = reinterpret_cast<const uint2*>(x + (p + step) * C)[c4];
            const uint2 q2 = reinterpret_cast<const uint2*>(x + (p + 2 * step) * C)[c4], q3 = reinterpret_cast<const uint2*>(x + (p + 3 * step) * C)[c4];
            const float4 v0 = unpack_bf4(q0), v1 = unpack_bf4(q1), v2 = unpack_bf4(q2), v3 = unpack_bf4(q3);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; p < P; p += step) {
            const float4 v = unpack_bf4(reinterpret_cast<const uint2*>(x + p * C)[c4]);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    __shared__ float4 sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < ppb; ++k) { const float4 v = sh[k * C4 + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        reinterpret_cast<float4*>(part + (size_t)blockIdx.x * C)[threadIdx.x] = t;
    }
}

int launch_colsum_bf16(const bf16_t* x, float* out, float* part, size_t P, int C, hipStream_t st) {
    if (C % 4 || C > 512) return ELD_EINVAL;
    const int ppb = 256 / (C / 4);
    const int nb = (int)min((P + ppb - 1) / ppb, (size_t)COLSUM_BLOCKS);
    if (nb == 0) return 0;
    ELD_LAUNCH(colsum_bf16_kernel, dim3(nb), dim3(256), 0, st, x, part, P, C);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(colsum_reduce_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, part, out, nb, C);
    ELD_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Fused training head: conv10_1 forward (Unet.py:46,88) + nn.L1Loss / nn.MSELoss (models/losses.py:30-34) + the head's backward in ONE pass over
// conv9_2's output.  The three separate kernels read that 32-channel full-resolution tensor twice and move the output gradient through HBM
// (head_fwd, l1_kernel, head_bwd: 1.34 ms per 8-frame bf16 step); here every pixel is read once:
//      out[o]  = b[o] + sum_c W[o][c] act[c]                     (written: NCHW fp32, the network's output)
//      d[o]    = dLoss/dout[o] from (out[o] - target[o])          (sign * gscale for L1, 2 * diff * gscale for MSE; never stored)
//      g[c]    = (sum_o W[o][c] d[o]) * slope(act[c])             (written: gradient of conv9_2's pre-activation output)
//      dW, db, loss: per-block partials -> head_bwd_reduce_kernel / l1_reduce_kernel (fixed order)
// Layouts as head_bwd_kernel: LPP lanes per pixel (8 x 4 fp32 channels / 4 x 8 bf16 channels), lane o of each quad owns output plane o.
// The forward sum is head_fwd_kernel's (per-lane fma chains + a butterfly over the pixel's lanes): the output equals the unfused path's bit for bit;
// the loss differs from l1_kernel's in the order of its partial sums only.
// ------------------------------------------------------------------------------------------------
template <typename T, int MSE>
__global__ __launch_bounds__(256) void head_train_kernel(const T* __restrict__ act, const float* __restrict__ w, const float* __restrict__ b,
                                                         const float* __restrict__ tgt, float* __restrict__ out, T* __restrict__ g, float* __restrict__ part,
                                                         float* __restrict__ lpart, int N, size_t HW, int OC, float gscale) {
    constexpr int CPL = sizeof(T) == 4 ? 4 : 8, LPP = 32 / CPL, PPB = 256 / LPP;      // channels per lane, lanes per pixel, pixels per block pass
    __shared__ float red[4][133];
    const int tid = threadIdx.x, cq = tid & (LPP - 1), o_ld = tid & 3;
    float wq[4][CPL];                                 // w[o][CPL cq + j]
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < CPL; ++j) wq[o][j] = o < OC ? w[o * 32 + CPL * cq + j] : 0.f;
    const float bo = o_ld < OC ? b[o_ld] : 0.f;
    float dw[4][CPL];
    float db = 0.f, ls = 0.f;                         // lanes with cq < 4 sum plane o_ld = cq
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < CPL; ++j) dw[o][j] = 0.f;
    const size_t total = (size_t)N * HW;
    const size_t stride = (size_t)gridDim.x * PPB;
    size_t p = (size_t)blockIdx.x * PPB + (tid / LPP);
    const size_t iters = (total + stride - 1) / stride;               // every lane of a pixel group runs the same trip count (the exchanges need its mates)
    for (size_t itn = 0; itn < iters; ++itn, p += stride) {
        const bool ok = p < total;
        const size_t pc = ok ? p : total - 1;
        const size_t n = pc / HW, q = pc - n * HW;
        float av[CPL], po[4];
        head_load<T, CPL>(act, pc, cq, av);
        head_dot<CPL>(av, wq, po);                   // forward (the order of head_fwd_kernel: same bits)
        const float mine = (o_ld == 0 ? po[0] : (o_ld == 1 ? po[1] : (o_ld == 2 ? po[2] : po[3]))) + bo;
        // loss and its gradient on this lane's plane
        float dl = 0.f;
        if (ok && o_ld < OC) {
            const size_t idx = (n * OC + o_ld) * HW + q;
            const float diff = mine - tgt[idx];
            if (cq < 4) { out[idx] = mine; ls += MSE ? diff * diff : fabsf(diff); }
            dl = MSE ? 2.0f * diff * gscale : (diff > 0.f ? gscale : (diff < 0.f ? -gscale : 0.f));
        }
        if (cq < 4) db += dl;
        float d[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) d[o] = quad_bcast(dl, o);
        if (!ok) continue;
        // backward
        float gv[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) { sacc = fmaf(wq[o][j], d[o], sacc); dw[o][j] = fmaf(d[o], av[j], dw[o][j]); }
            gv[j] = sacc * lrelu_slope(av[j]);
        }
        if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(g + pc * 32)[cq] = make_float4(gv[0], gv[1], gv[2], gv[3]);
        else {
            const uint2 g0 = pack_bf4(make_float4(gv[0], gv[1], gv[2], gv[3])), g1 = pack_bf4(make_float4(gv[4], gv[5], gv[6], gv[7]));
            reinterpret_cast<uint4*>(g + pc * 32)[cq] = make_uint4(g0.x, g0.y, g1.x, g1.y);
        }
    }
    // block reduction in a fixed order: lanes sharing a channel group by xor-shuffles, then the 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float v = dw[o][j];
            for (int off = LPP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < LPP) red[wave][o * 32 + CPL * lane + j] = v;
        }
    {
        float v = db;
        for (int off = LPP; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < 4) red[wave][128 + lane] = v;
        float l = ls;
        for (int off = 1; off < 64; off <<= 1) l += __shfl_xor(l, off, 64);
        if (lane == 0) red[wave][132] = l;
    }
    __syncthreads();
    if (tid < 132) part[(size_t)blockIdx.x * 132 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    if (tid == 132) lpart[blockIdx.x] = red[0][132] + red[1][132] + red[2][132] + red[3][132];
}

size_t head_train_ws_floats() { return (size_t)HEAD_BLOCKS * 133; }
static int head_train_blocks(size_t total, int bf16) { return (int)min((total + (bf16 ? 63 : 31)) / (bf16 ? 64 : 32), (size_t)HEAD_BLOCKS); }

// part: head_train_ws_floats() floats that must survive until launch_head_train_reduce (the backward) has run
int launch_head_train(const void* act, int bf16, const float* w, const float* b, const float* tgt, float* out, void* g, float* part, float* loss,
                      int N, int H, int W, int OC, int mse, float grad_scale, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return ELD_EINVAL;
    const int nb = head_train_blocks(total, bf16);
    float* lpart = part + (size_t)HEAD_BLOCKS * 132;
    const float n = (float)(total * (size_t)OC);
    const float gs = grad_scale / n;
    const size_t HW = (size_t)H * W;
    if (bf16) {
        if (mse) { ELD_LAUNCH((head_train_kernel<bf16_t, 1>), dim3(nb), dim3(256), 0, st, (const bf16_t*)act, w, b, tgt, out, (bf16_t*)g, part, lpart, N, HW, OC, gs); }
        else { ELD_LAUNCH((head_train_kernel<bf16_t, 0>), dim3(nb), dim3(256), 0, st, (const bf16_t*)act, w, b, tgt, out, (bf16_t*)g, part, lpart, N, HW, OC, gs); }
    } else {
        if (mse) { ELD_LAUNCH((head_train_kernel<float, 1>), dim3(nb), dim3(256), 0, st, (const float*)act, w, b, tgt, out, (float*)g, part, lpart, N, HW, OC, gs); }
        else { ELD_LAUNCH((head_train_kernel<float, 0>), dim3(nb), dim3(256), 0, st, (const float*)act, w, b, tgt, out, (float*)g, part, lpart, N, HW, OC, gs); }
    }
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(l1_reduce_kernel, dim3(1), dim3(256), 0, st, lpart, loss, nb, 1.0f / n);
    ELD_LAUNCH_CHECK();
    return 0;
}

// the head's dW / db from the partials launch_head_train left in `part`
int launch_head_train_reduce(const float* part, float* dw, float* db, int N, int H, int W, int OC, int bf16, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    if (!total) return 0;
    ELD_LAUNCH(head_bwd_reduce_kernel, dim3((132 + 15) / 16), dim3(256), 0, st, part, dw, db, head_train_blocks(total, bf16), OC);
    ELD_LAUNCH_CHECK();
    return 0;
}
