// conv_first.hip -- the first layer of the U-Net (conv1_1: Cin <= 4 raw planes -> 32 channels, 3x3, +bias +LeakyReLU),
// forward and weight/bias gradient, on the exact-fp32 MFMA (gfx950).  Reads the network input as it arrives (NCHW
// planes, models/arch/Unet.py:49) and writes NHWC, so no layout-conversion pass and no channel padding: the
// contraction is K = 9*Cin (36 for packed raw), not the 144 of the generic 16-channel path.
//   forward : D[co][pixel]  = sum_k W[co][k] * patch[k][pixel],   k = (c, dy, dx) in OIHW order
//   wgrad   : dW[co][k]     = sum_pixels g[pixel][co] * patch[k][pixel];  db[co] = sum_pixels g[pixel][co]
// Both keep the input tile as a zero-padded halo [Cin][TH+2][34] in LDS; a patch element is one ds_read_b32 at
// (lane-constant tap offset + pixel offset), conflict-free along the 32 pixels of a row.
#include "unet_misc.h"

#define FTH 8
#define FTW 32
#define FHP ((FTH + 2) * (FTW + 2))      // halo pixels per plane

// The zero-padded halo [CIN][FTH+2][34] of a tile travels HBM -> registers -> LDS: all of a thread's loads are issued together, one tile
// ahead (they are in flight during the previous tile's MFMA phase and epilogue), and go to LDS between the two barriers of the next step.
// (A load -> ds_write loop costs one HBM round trip per iteration: 6 serialised latencies per tile were 3/4 of these kernels' time.)
template <int CIN>
struct FirstHalo {
    static constexpr int IT = (CIN * FHP + 255) / 256;
    float v[IT];
    __device__ __forceinline__ void load(const float* __restrict__ x, int img, int y0, int x0, int H, int W) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int u = threadIdx.x + it * 256;
            const int c = u / FHP, hp = u - c * FHP;
            const int hy = hp / (FTW + 2), hx = hp - hy * (FTW + 2);
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            v[it] = 0.f;
            if (u < CIN * FHP && gy >= 0 && gy < H && gx >= 0 && gx < W) v[it] = x[((size_t)(img * CIN + c) * H + gy) * W + gx];
        }
    }
    __device__ __forceinline__ void store(float* halo) const {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int u = threadIdx.x + it * 256;
            if (u < CIN * FHP) halo[u] = v[it];
        }
    }
};

template <int CIN, typename TO>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             TO* __restrict__ out, int N, int H, int W, int lrelu) {
    constexpr int K = 9 * CIN, KS = (K + 1) / 2;
    __shared__ float halo[CIN * FHP];
    __shared__ float wl[2 * KS * 32];            // [k][co], zero padded to 2*KS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, hi = lane >> 5;
    for (int u = tid; u < 2 * KS * 32; u += 256) {
        const int k = u >> 5, co = u & 31;
        wl[u] = k < K ? w[co * K + k] : 0.f;     // OIHW: w[co][c][dy][dx] = w[co*K + k]
    }
    const int tiles_x = (W + FTW - 1) / FTW, tiles_y = (H + FTH - 1) / FTH;
    const int total = tiles_x * tiles_y * N;
    // lane-constant halo offsets of this lane's k values: k = 2s + hi
    int koff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + hi;
        const int kk = k < K ? k : 0;
        const int c = kk / 9, r = kk - c * 9;
        koff[s] = c * FHP + (r / 3) * (FTW + 2) + (r % 3);
    }
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * hi), b1 = *reinterpret_cast<const float4*>(bias + 8 + 4 * hi),
                 b2 = *reinterpret_cast<const float4*>(bias + 16 + 4 * hi), b3 = *reinterpret_cast<const float4*>(bias + 24 + 4 * hi);
    FirstHalo<CIN> hr;
    auto prefetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        hr.load(x, img, ty * FTH, tx * FTW, H, W);
    };
    if ((int)blockIdx.x < total) prefetch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int y0 = ty * FTH, x0 = tx * FTW;
        __syncthreads();
        hr.store(halo);
        __syncthreads();
        if (t + (int)gridDim.x < total) prefetch(t + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float a = wl[(2 * s + hi) * 32 + m];                       // A[co = m][k]
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = wave * 2 + r;
                const float b = halo[koff[s] + row * (FTW + 2) + m];        // B[k][pixel = m]: x[c][row+dy][m+dx] in halo coords
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
            }
        }
        const int xx = x0 + m;
        if (xx < W) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int y = y0 + wave * 2 + r;
                if (y >= H) continue;
                TO* dst = out + ((size_t)(img * H + y) * W + xx) * 32 + 4 * hi;
                const float4 bs[4] = {b0, b1, b2, b3};
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = make_float4(acc[r][4 * q] + bs[q].x, acc[r][4 * q + 1] + bs[q].y, acc[r][4 * q + 2] + bs[q].z, acc[r][4 * q + 3] + bs[q].w);
                    if (lrelu) { v[q].x = fmaxf(0.2f * v[q].x, v[q].x); v[q].y = fmaxf(0.2f * v[q].y, v[q].y); v[q].z = fmaxf(0.2f * v[q].z, v[q].z); v[q].w = fmaxf(0.2f * v[q].w, v[q].w); }
                }
                if constexpr (sizeof(TO) == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(dst + 8 * q) = v[q];
                } else {                                     // bf16: 16-byte stores of whole 8-channel groups (conv.h bf16_pair_swap)
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(dst + 16 * j + 4 * hi) = bf16_pair_swap(pack_bf4(v[2 * j]), pack_bf4(v[2 * j + 1]));
                }
            }
        }
    }
}

template <typename TO>
static int launch_first_t(const float* x, const float* w, const float* bias, TO* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st) {
    const int tiles = ((W + FTW - 1) / FTW) * ((H + FTH - 1) / FTH) * N;
    if (tiles <= 0) return 0;
    const int grid = tiles < 2048 ? tiles : 2048;
    switch (Cin) {
        case 1: ELD_LAUNCH((conv_first_fwd_kernel<1, TO>), dim3(grid), dim3(256), 0, st, x, w, bias, out, N, H, W, lrelu); break;
        case 2: ELD_LAUNCH((conv_first_fwd_kernel<2, TO>), dim3(grid), dim3(256), 0, st, x, w, bias, out, N, H, W, lrelu); break;
        case 3: ELD_LAUNCH((conv_first_fwd_kernel<3, TO>), dim3(grid), dim3(256), 0, st, x, w, bias, out, N, H, W, lrelu); break;
        case 4: ELD_LAUNCH((conv_first_fwd_kernel<4, TO>), dim3(grid), dim3(256), 0, st, x, w, bias, out, N, H, W, lrelu); break;
        default: return ELD_ENOTSUP;
    }
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_conv_first_fwd(const float* x, const float* w, const float* bias, float* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st) {
    return launch_first_t<float>(x, w, bias, out, N, Cin, H, W, lrelu, st);
}
int launch_conv_first_fwd_bf16(const float* x, const float* w, const float* bias, bf16_t* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st) {
    return launch_first_t<bf16_t>(x, w, bias, out, N, Cin, H, W, lrelu, st);
}

// ------------------------------------------------------------------------------------------------------------------
// weight / bias gradient.  MFMA: A[i = co][k = pixel] = g[pixel][co], B[k = pixel][j] = patch value of tap-channel j at
// that pixel (j < K; two 32-wide j tiles cover K <= 36).  Each wave owns 64 pixels of the 256-pixel tile; the workgroup
// walks tiles persistently, reduces its 4 waves through LDS and writes one partial [2][32][32] (+ 32 bias sums).
// ------------------------------------------------------------------------------------------------------------------
#define FW_BLOCKS 1024
template <int CIN, typename TG>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const TG* __restrict__ g, const float* __restrict__ x, float* __restrict__ part,
                                                               int N, int H, int W) {
    constexpr int K = 9 * CIN;
    __shared__ float halo[CIN * FHP];
    __shared__ float gl[FTH * FTW * 32];         // [pixel][co]  32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, hi = lane >> 5;
    const int tiles_x = (W + FTW - 1) / FTW, tiles_y = (H + FTH - 1) / FTH;
    const int total = tiles_x * tiles_y * N;
    int joff[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = jt * 32 + m;
        const int jj = j < K ? j : 0;
        const int c = jj / 9, r = jj - c * 9;
        joff[jt] = j < K ? c * FHP + (r / 3) * (FTW + 2) + (r % 3) : -1;
    }
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[jt][i] = 0.f;
    float bsum = 0.f;
    FirstHalo<CIN> hr;
    float4 gr[8];                                // this thread's 8 units of the 256-pixel x 32-channel gradient tile
    auto prefetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int y0 = ty * FTH, x0 = tx * FTW;
        hr.load(x, img, y0, x0, H, W);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = threadIdx.x + it * 256;
            const int lp = u >> 3, part4 = u & 7;
            const int gy = y0 + lp / FTW, gx = x0 + lp % FTW;
            gr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W) {
                const TG* gp = g + ((size_t)(img * H + gy) * W + gx) * 32 + part4 * 4;
                if constexpr (sizeof(TG) == 4) gr[it] = *reinterpret_cast<const float4*>(gp);
                else gr[it] = unpack_bf4(*reinterpret_cast<const uint2*>(gp));
            }
        }
    };
    if ((int)blockIdx.x < total) prefetch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        __syncthreads();
        hr.store(halo);
#pragma unroll
        for (int it = 0; it < 8; ++it) *reinterpret_cast<float4*>(gl + (tid + it * 256) * 4) = gr[it];
        __syncthreads();
        if (t + (int)gridDim.x < total) prefetch(t + gridDim.x);
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
            const int lp = wave * 64 + s + hi * 32;          // rows 2*wave (hi=0) and 2*wave+1 (hi=1), column s
            const int py = lp / FTW, px = lp - py * FTW;
            const float a = gl[lp * 32 + m];
            bsum += a;
            const int pofs = py * (FTW + 2) + px;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const float b = joff[jt] >= 0 ? halo[joff[jt] + pofs] : 0.f;
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[jt], 0, 0, 0);
            }
        }
    }
    // reduce the 4 waves through LDS (fixed order), wave 0 writes the partial
    __syncthreads();
    float* red = gl;                              // [3 waves][2][16][64]
    if (wave > 0) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wave - 1) * 2 + jt) * 16 + i) * 64 + lane] = acc[jt][i];
    }
    float* bred = gl + 3 * 2 * 16 * 64;
    bred[wave * 64 + lane] = bsum;
    __syncthreads();
    if (wave == 0) {
        float* p = part + (size_t)blockIdx.x * (2 * 32 * 32 + 32);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[jt][i];
                for (int wv = 0; wv < 3; ++wv) v += red[((wv * 2 + jt) * 16 + i) * 64 + lane];
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;             // co
                p[(jt * 32 + row) * 32 + m] = v;                              // [jt][co][j%32]
            }
        if (lane < 32) {
            float b = 0.f;
            for (int wv = 0; wv < 4; ++wv) b += bred[wv * 64 + lane] + bred[wv * 64 + 32 + lane];
            p[2 * 32 * 32 + lane] = b;
        }
    }
}

// dW[co][k] (OIHW, k < K) and db[co] from the per-workgroup partials, fixed order: 16 outputs x 16 partial slices per
// workgroup, 4 loads in flight per lane, slices combined through LDS
__global__ __launch_bounds__(256) void conv_first_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int nblocks, int K) {
    constexpr int NOUT = 2 * 32 * 32 + 32;
    __shared__ float sh[16][16];
    const int tl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tl;
    float s = 0.f;
    if (t < NOUT) {
        int b = sl;
        for (; b + 48 < nblocks; b += 64)
            s += (part[(size_t)b * NOUT + t] + part[(size_t)(b + 16) * NOUT + t]) + (part[(size_t)(b + 32) * NOUT + t] + part[(size_t)(b + 48) * NOUT + t]);
        for (; b < nblocks; b += 16) s += part[(size_t)b * NOUT + t];
    }
    sh[sl][tl] = s;
    __syncthreads();
    if (sl != 0 || t >= NOUT) return;
    for (int k = 1; k < 16; ++k) s += sh[k][tl];
    if (t < 2 * 32 * 32) {
        const int jt = t / 1024, co = (t / 32) % 32, j = jt * 32 + (t % 32);
        if (j < K) dw[co * K + j] = s;
    } else {
        db[t - 2 * 32 * 32] = s;
    }
}

size_t conv_first_wgrad_ws_floats() { return (size_t)FW_BLOCKS * (2 * 32 * 32 + 32); }

template <typename TG>
static int launch_first_wgrad_t(const TG* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st) {
    const int tiles = ((W + FTW - 1) / FTW) * ((H + FTH - 1) / FTH) * N;
    if (tiles <= 0) return 0;
    const int grid = tiles < FW_BLOCKS ? tiles : FW_BLOCKS;
    switch (Cin) {
        case 1: ELD_LAUNCH((conv_first_wgrad_kernel<1, TG>), dim3(grid), dim3(256), 0, st, g, x, part, N, H, W); break;
        case 2: ELD_LAUNCH((conv_first_wgrad_kernel<2, TG>), dim3(grid), dim3(256), 0, st, g, x, part, N, H, W); break;
        case 3: ELD_LAUNCH((conv_first_wgrad_kernel<3, TG>), dim3(grid), dim3(256), 0, st, g, x, part, N, H, W); break;
        case 4: ELD_LAUNCH((conv_first_wgrad_kernel<4, TG>), dim3(grid), dim3(256), 0, st, g, x, part, N, H, W); break;
        default: return ELD_ENOTSUP;
    }
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(conv_first_wgrad_reduce_kernel, dim3((2 * 32 * 32 + 32 + 15) / 16), dim3(256), 0, st, part, dw, db, grid, 9 * Cin);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_conv_first_wgrad(const float* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st) {
    return launch_first_wgrad_t<float>(g, x, dw, db, part, N, Cin, H, W, st);
}
int launch_conv_first_wgrad_bf16(const bf16_t* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st) {
    return launch_first_wgrad_t<bf16_t>(g, x, dw, db, part, N, Cin, H, W, st);
}
