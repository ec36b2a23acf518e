// conv_first.hip -- the first layer of the U-Net (conv1_1: Cin <= 4 raw planes -> 32 channels, 3x3, +bias +LeakyReLU),
// forward and weight/bias gradient, on the exact-fp32 MFMA (gfx950).  Reads the network input as it arrives (NCHW
// planes, models/arch/Unet.py:49) and writes NHWC, so no layout-conversion pass and no channel padding: the
// contraction is K = 9*Cin (36 for packed raw), not the 144 of the generic 16-channel path.
//   forward : D[co][pixel]  = sum_k W[co][k] * patch[k][pixel],   k = (c, dy, dx) in OIHW order
//   wgrad   : dW[co][k]     = sum_pixels g[pixel][co] * patch[k][pixel];  db[co] = sum_pixels g[pixel][co]
// Both keep the input tile as a zero-padded halo [Cin][TH+2][34] in LDS; a patch element is one ds_read_b32 at
// (lane-constant tap offset + pixel offset), conflict-free along the 32 pixels of a row.
#include <type_traits>
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

// ------------------------------------------------------------------------------------------------------------------
// The 4-channel first layer on the bf16 MFMA.  K is laid out as 3 kernel rows x (4 horizontal slots x 4 channels) = 48 = three 16-k blocks,
// slot 3 carrying zero weights: a lane's B operand of a block (its pixel m, k-half hi) is then the 16 contiguous bytes of halo pixels
// m + 2hi and m + 2hi + 1 in a [pixel][4 channels] bf16 plane -- one 8-byte-aligned read, no gather.  fp32 values enter as exact bf16
// pieces (NPC = 3: the three-piece cut of conv_x3.hip, six products, fp32-exact; NPC = 2 for the bf16 network: two pieces, three
// products, 2^-16), cut ONCE per halo pixel while the tile is staged.  9 (18) MFMAs of 32 cycles per row of 32 pixels instead of 18 of 64
// on v_mfma_f32_32x32x2_f32, and 6 (9) LDS reads instead of 36.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
template <int NPC>
__device__ __forceinline__ void first_cut(float v, bf16_t (&p)[NPC]) {      // v = p[0] + p[1] (+ p[2]) with truncated pieces (exact for NPC = 3)
    float r = v;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const unsigned u = __float_as_uint(r) & 0xFFFF0000u;
        p[i] = (bf16_t)(u >> 16);
        r -= __uint_as_float(u);
    }
}

template <typename TO, int NPC>
__global__ __launch_bounds__(256) void conv_first_fwd_mma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 TO* __restrict__ out, int N, int H, int W, int lrelu, unsigned* __restrict__ codes) {
    constexpr int CIN = 4, HPX = FHP + 2;                     // + 2 pixels: slot 3 of the last halo row reads one pixel past it (zero weights, finite data)
    __shared__ __attribute__((aligned(16))) bf16_t halo[NPC][HPX][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, hi = lane >> 5;
    for (int u = tid; u < NPC * 2 * 4; u += 256) (&halo[0][0][0])[(u / 8) * HPX * 4 + FHP * 4 + (u & 7)] = 0;      // the pad pixels of every plane
    // A operand: rows = output channel m; k = 8 hi + 4 dxo + c of kernel row dy  <->  w[m][c][dy][2 hi + dxo]  (OIHW)
    fbf16x8 wq[NPC][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        bf16_t pc[8][NPC];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int dx = 2 * hi + (j >> 2), c = j & 3;
            first_cut<NPC>(dx < 3 ? w[m * 36 + c * 9 + dy * 3 + dx] : 0.f, pc[j]);
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const uint4 q = make_uint4((unsigned)pc[0][i] | ((unsigned)pc[1][i] << 16), (unsigned)pc[2][i] | ((unsigned)pc[3][i] << 16),
                                       (unsigned)pc[4][i] | ((unsigned)pc[5][i] << 16), (unsigned)pc[6][i] | ((unsigned)pc[7][i] << 16));
            wq[i][dy] = __builtin_bit_cast(fbf16x8, q);
        }
    }
    const int tiles_x = (W + FTW - 1) / FTW, tiles_y = (H + FTH - 1) / FTH;
    const int total = tiles_x * tiles_y * N;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * hi), b1 = *reinterpret_cast<const float4*>(bias + 8 + 4 * hi),
                 b2 = *reinterpret_cast<const float4*>(bias + 16 + 4 * hi), b3 = *reinterpret_cast<const float4*>(bias + 24 + 4 * hi);
    // halo staging: a thread owns halo pixels tid and tid + 256 (all four channels), fetched one tile ahead
    float hv[2][CIN];
    auto prefetch = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hp = tid + it * 256;
            const int hy = hp / (FTW + 2), hx = hp - hy * (FTW + 2);
            const int gy = ty * FTH + hy - 1, gx = tx * FTW + hx - 1;
            const bool ok = hp < FHP && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) hv[it][c] = ok ? x[((size_t)(img * CIN + c) * H + gy) * W + gx] : 0.f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int hp = tid + it * 256;
            if (hp >= FHP) continue;
            bf16_t pc[CIN][NPC];
#pragma unroll
            for (int c = 0; c < CIN; ++c) first_cut<NPC>(hv[it][c], pc[c]);
#pragma unroll
            for (int i = 0; i < NPC; ++i)
                *reinterpret_cast<uint2*>(&halo[i][hp][0]) = make_uint2((unsigned)pc[0][i] | ((unsigned)pc[1][i] << 16), (unsigned)pc[2][i] | ((unsigned)pc[3][i] << 16));
        }
    };
    if ((int)blockIdx.x < total) prefetch(blockIdx.x);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int y0 = ty * FTH, x0 = tx * FTW;
        __syncthreads();
        stage();
        __syncthreads();
        if (t + (int)gridDim.x < total) prefetch(t + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                fbf16x8 xb[NPC];
                const int hp = (wave * 2 + r + dy) * (FTW + 2) + m + 2 * hi;
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(&halo[i][hp][0]), hi2 = *reinterpret_cast<const uint2*>(&halo[i][hp + 1][0]);
                    xb[i] = __builtin_bit_cast(fbf16x8, make_uint4(lo.x, lo.y, hi2.x, hi2.y));
                }
                if constexpr (NPC == 3) {       // six piece products, smallest first (conv_x3.hip)
                    constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
                    constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int q = 0; q < 6; ++q) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[WI[q]][dy], xb[XI[q]], acc[r], 0, 0, 0);
                } else {
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0][dy], xb[1], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1][dy], xb[0], acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0][dy], xb[0], acc[r], 0, 0, 0);
                }
            }
        // epilogue in the full-line layouts of conv.h (every lane takes part in the exchanges; the stores are predicated)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int y = y0 + wave * 2 + r;                 // wave-uniform
            const bool yok = y < H;
            const float4 bs[4] = {b0, b1, b2, b3};
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = make_float4(acc[r][4 * q] + bs[q].x, acc[r][4 * q + 1] + bs[q].y, acc[r][4 * q + 2] + bs[q].z, acc[r][4 * q + 3] + bs[q].w);
                if (lrelu) { v[q].x = fmaxf(0.2f * v[q].x, v[q].x); v[q].y = fmaxf(0.2f * v[q].y, v[q].y); v[q].z = fmaxf(0.2f * v[q].z, v[q].z); v[q].w = fmaxf(0.2f * v[q].w, v[q].w); }
            }
            TO* blk = out + ((size_t)(img * H + (yok ? y : 0)) * W + x0) * 32;
            if constexpr (sizeof(TO) == 4) {
                if (codes != nullptr && yok && x0 + m < W) {             // slope codes of the finished values (conv.h ConvArgs::codes_out): word (pixel, hi) of the one 32-channel block
                    unsigned cwd = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        cwd |= (slope_code(v[q].x) | (slope_code(v[q].y) << 2) | (slope_code(v[q].z) << 4) | (slope_code(v[q].w) << 6)) << (8 * q);
                    codes[((size_t)(img * H + y) * W + x0 + m) * 2 + hi] = cwd;
                }
                f32_line_store(v, reinterpret_cast<float*>(blk), 32, lane, yok, W - x0);
            } else {
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf4(v[q]);
                if (codes != nullptr && yok && x0 + m < W) {             // the bf16 network's codes come from the ROUNDED values (conv.h slope_codes_bf16)
                    codes[((size_t)(img * H + y) * W + x0 + m) * 2 + hi] = slope_codes_bf16(pk);
                }
                uint4 s0, s1;
                bf16_line_swap(pk, s0, s1);
                const int lp = lane & 15;
                bf16_t* row = reinterpret_cast<bf16_t*>(blk) + 8 * bf16_line_group(lane);
                if (yok && x0 + lp < W) *reinterpret_cast<uint4*>(row + lp * 32) = s0;
                if (yok && x0 + lp + 16 < W) *reinterpret_cast<uint4*>(row + (lp + 16) * 32) = s1;
            }
        }
    }
}

static int first_mma_on() { static const int mma = [] { const char* e = getenv("ELD_FIRST_MMA"); return e ? atoi(e) : 1; }(); return mma; }      // ELD_FIRST_MMA=0: the K = 36 kernel on the fp32 MFMA for every Cin
bool conv_first_writes_codes(int Cin) { return Cin == 4 && first_mma_on(); }      // which launches honour the `codes` argument of launch_conv_first_fwd
template <typename TO>
static int launch_first_t(const float* x, const float* w, const float* bias, TO* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st, unsigned* codes = nullptr) {
    const int tiles = ((W + FTW - 1) / FTW) * ((H + FTH - 1) / FTH) * N;
    if (tiles <= 0) return 0;
    const int grid = tiles < 2048 ? tiles : 2048;
    const int mma = first_mma_on();
    if (Cin == 4 && mma) {                        // packed Bayer raw: the bf16-MFMA kernel (fp32 output: exact three-piece products; bf16 output: two pieces)
        ELD_LAUNCH((conv_first_fwd_mma_kernel<TO, sizeof(TO) == 4 ? 3 : 2>), dim3(grid), dim3(256), 0, st, x, w, bias, out, N, H, W, lrelu, codes);
        ELD_LAUNCH_CHECK();
        return 0;
    }
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

int launch_conv_first_fwd(const float* x, const float* w, const float* bias, float* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st, unsigned* codes) {
    return launch_first_t<float>(x, w, bias, out, N, Cin, H, W, lrelu, st, codes);
}
int launch_conv_first_fwd_bf16(const float* x, const float* w, const float* bias, bf16_t* out, int N, int Cin, int H, int W, int lrelu, hipStream_t st, unsigned* codes) {
    return launch_first_t<bf16_t>(x, w, bias, out, N, Cin, H, W, lrelu, st, codes);
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

// ------------------------------------------------------------------------------------------------------------------
// bf16-gradient variant (BASELINE configs[2]) on the bf16 MFMA: the fp32 MFMA above needs 64 cycles per 2 pixels and was the whole kernel
// (1.3 ms: 223 GFLOP at the 157 TF/s fp32 rate); v_mfma_f32_32x32x16_bf16 covers 16 pixels in 32 cycles.
//   A[i = co][k = pixel]  = g[pixel][co]          bf16 as stored; the tile's [256 pixels][32 co] image comes HBM -> LDS by LDS-DMA (double buffered)
//                                                   and is read transposed by ds_read_b64_tr_b16 (as wgrad_kernel<bf16_t> does)
//   B[k = pixel][j]       = patch value of tap-channel j at that pixel, as TWO bf16 pieces of the fp32 input (x = hi + lo to 2^-17: the raw input keeps
//                           16 significant bits; the gradient operand is bf16 anyway): acc += g * x_hi, acc += g * x_lo
// A lane's B operand is 8 horizontally consecutive pixels of one (channel, dy, dx): to make that ONE aligned 16-byte LDS read the halo is stored three
// times, shifted by dx = 0, 1, 2 ([piece][dx][c][row][40] bf16, 19 KB) -- 6 two-byte stores per halo element, once per tile, instead of 16 two-byte reads
// per MFMA.  Partials and their reduction are the fp32 kernel's ([2][co][j % 32] + 32 bias sums per workgroup, fixed order).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 fw_bf16x8 __attribute__((ext_vector_type(8)));
typedef short fw_s16x4 __attribute__((ext_vector_type(4)));
typedef int fw_i32x4 __attribute__((ext_vector_type(4)));
#define FXW 40                                   // padded row of a shifted halo copy (bf16 elements; 80 B keeps 16-byte alignment)

__device__ __forceinline__ void fw_dma16(fw_i32x4 rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_bf16_kernel(const bf16_t* __restrict__ g, const float* __restrict__ x, float* __restrict__ part,
                                                                    int N, int H, int W) {
    constexpr int K = 9 * CIN;
    constexpr int XS = 2 * 3 * CIN * (FTH + 2) * FXW;                     // bf16 elements of the shifted halo copies
    __shared__ __attribute__((aligned(16))) bf16_t gl[2][FTH * FTW * 32];                            // 2 x 16 KB
    __shared__ __attribute__((aligned(16))) bf16_t xs[XS];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (W + FTW - 1) / FTW, tiles_y = (H + FTH - 1) / FTH;
    const int total = tiles_x * tiles_y * N;
    // this lane's two B columns: tap-channel j = jt*32 + m -> (c, dy, dx); offset of its operand for tile row 0, pixel 0, piece 0 (bf16 elements)
    int boff[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = jt * 32 + m;
        const int c = j / 9, r9 = j - c * 9, dy = r9 / 3, dx = r9 - dy * 3;
        boff[jt] = j < K ? ((dx * CIN + c) * (FTH + 2) + dy) * FXW : -1;
    }
    const int gi = lane & 15, gg = lane >> 4;
    const int gq_off = (8 * hi + (gi >> 2)) * 32 + (gg & 1) * 16 + (gi & 3) * 4;          // ds_read_b64_tr_b16 addressing (conv_wgrad.hip)
    const unsigned lds_gl = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) bf16_t*)&gl[0][0]);
    const unsigned long long gbase = (unsigned long long)g;
    const size_t gbytes = (size_t)N * H * W * 64;
    const fw_i32x4 rsrc_g = {(int)(unsigned)gbase, (int)((unsigned)(gbase >> 32) & 0xFFFFu), (int)(gbytes < 0xFFFFFFF0ull ? gbytes : 0xFFFFFFF0ull), 0x00020000};
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[jt][i] = 0.f;
    float bsum = 0.f;
    FirstHalo<CIN> hr;
    auto prefetch = [&](int t, int buf) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int y0 = ty * FTH, x0 = tx * FTW;
        hr.load(x, img, y0, x0, H, W);
#pragma unroll
        for (int it = 0; it < 4; ++it) {                                  // 16 pieces of 1 KiB = 16 pixels x 64 B each: piece p = (row p >> 1, pixels 16 (p & 1) ..)
            const int p = wave + it * 4;
            const int gy = y0 + (p >> 1), gx = x0 + 16 * (p & 1) + (lane >> 2);
            const bool ok = gy < H && gx < W;
            const unsigned voff = ok ? (unsigned)(((size_t)(img * H + gy) * W + gx) * 64 + (lane & 3) * 16) : 0xFFFFFFF0u;
            fw_dma16(rsrc_g, voff, lds_gl + (unsigned)(buf * (FTH * FTW * 64) + p * 1024));
        }
    };
    int buf = 0;
    if ((int)blockIdx.x < total) prefetch(blockIdx.x, 0);
    for (int t = blockIdx.x; t < total; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the gradient tile (and its halo loads) have landed
        __syncthreads();                                        // everybody's; the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < FirstHalo<CIN>::IT; ++it) {      // halo -> two bf16 pieces -> the three shifted copies
            const int u = tid + it * 256;
            if (u >= CIN * FHP) continue;
            const int c = u / FHP, hp = u - c * FHP;
            const int hy = hp / (FTW + 2), hx = hp - hy * (FTW + 2);
            const float v = hr.v[it];
            const unsigned vb = __float_as_uint(v);
            const bf16_t ph = (bf16_t)(vb >> 16), pl = f2bf(v - __uint_as_float(vb & 0xFFFF0000u));
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = hx - dx;
                if (xx < 0 || xx >= FTW) continue;
                const int o = ((dx * CIN + c) * (FTH + 2) + hy) * FXW + xx;
                xs[o] = ph;
                xs[o + 3 * CIN * (FTH + 2) * FXW] = pl;
            }
        }
        __syncthreads();
        if (t + (int)gridDim.x < total) prefetch(t + gridDim.x, buf ^ 1);
        const bf16_t* gt = &gl[buf][0];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                        // 16 consecutive pixels of one row per k-step; the wave owns rows 2 wave, 2 wave + 1
            const int row = 2 * wave + (ks >> 1), col0 = 16 * (ks & 1);
            const bf16_t* p0 = gt + (row * FTW + col0) * 32 + gq_off;
            const fw_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fw_s16x4*)p0);
            const fw_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fw_s16x4*)(p0 + 4 * 32));
            const fw_bf16x8 ga = __builtin_bit_cast(fw_bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
            {
                const uint4 q = __builtin_bit_cast(uint4, ga);
                bsum += __uint_as_float(q.x << 16) + __uint_as_float(q.x & 0xFFFF0000u) + __uint_as_float(q.y << 16) + __uint_as_float(q.y & 0xFFFF0000u)
                      + __uint_as_float(q.z << 16) + __uint_as_float(q.z & 0xFFFF0000u) + __uint_as_float(q.w << 16) + __uint_as_float(q.w & 0xFFFF0000u);
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                uint4 bh = make_uint4(0, 0, 0, 0), bl = make_uint4(0, 0, 0, 0);
                if (boff[jt] >= 0) {
                    const bf16_t* pb = xs + boff[jt] + row * FXW + col0 + 8 * hi;
                    bh = *reinterpret_cast<const uint4*>(pb);
                    bl = *reinterpret_cast<const uint4*>(pb + 3 * CIN * (FTH + 2) * FXW);
                }
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, __builtin_bit_cast(fw_bf16x8, bh), acc[jt], 0, 0, 0);
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, __builtin_bit_cast(fw_bf16x8, bl), acc[jt], 0, 0, 0);
            }
        }
    }
    // reduce the 4 waves through LDS (fixed order), wave 0 writes the partial -- as conv_first_wgrad_kernel
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(&gl[0][0]);            // [3 waves][2][16][64] floats = 24 KB of the 32 KB
    if (wave > 0) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wave - 1) * 2 + jt) * 16 + i) * 64 + lane] = acc[jt][i];
    }
    float* bred = red + 3 * 2 * 16 * 64;
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

// ------------------------------------------------------------------------------------------------------------------
// fp32-gradient variant for the three-piece product scheme (conv_fp32_algo 1, the default fp32 path) on the bf16 MFMA: conv_first_wgrad_kernel's fp32
// MFMAs take 64 cycles per 2 pixels and are its whole time (1.21 ms per 8 frames; a conflict-free tile layout did not move it).  As conv_x3.hip does for
// the other layers, both operands are cut EXACTLY into three bf16 pieces (v = p1 + p2 + p3, 8 + 8 + 8 significant bits) while they are staged and a
// 16-pixel k-step accumulates the six products g1x1 + g1x2 + g2x1 + g1x3 + g2x2 + g3x1 (the dropped ones are below 2^-23 of the product):
//   A[i = co][k = pixel]  the gradient tile, fp32 [256 pixels][32 co] from HBM through registers (loaded one tile ahead), cut into three [pixel][32] bf16
//                         planes in LDS and read transposed by ds_read_b64_tr_b16 exactly like conv_first_wgrad_bf16_kernel's single plane;
//   B[k = pixel][j]       the three shifted halo copies of conv_first_wgrad_bf16_kernel, with three pieces instead of two.
// The bias gradient is summed from the fp32 registers while they are cut (a thread always stages the same four channels).  Partials and their
// reduction are the other kernels' ([2][co][j % 32] + 32 bias sums per workgroup, fixed order).  LDS: 48 KB + 28.8 KB (dynamic), two workgroups per CU.
// Measured: 1.21 -> 0.96 ms per 8 frames (3.5 GB: 3.6 TB/s).  The matrix work is now 0.24 ms of that; what is left is the VALU of the cuts and of the
// nine two-byte halo stores per input element (≈ 850 VALU instructions per wave and tile against 48 MFMAs).
// ------------------------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_x3_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ part,
                                                                     int N, int H, int W) {
    constexpr int K = 9 * CIN;
    constexpr int GPL = FTH * FTW * 32;                                    // bf16 elements of one gradient piece plane
    constexpr int XPL = 3 * CIN * (FTH + 2) * FXW;                         // bf16 elements of the three shifted halo copies of one piece
    extern __shared__ __attribute__((aligned(16))) bf16_t x3lds[];
    bf16_t* gl = x3lds;                                                    // [3 pieces][256 pixels][32 co]
    bf16_t* xs = x3lds + 3 * GPL;                                          // [3 pieces][dx][c][row][FXW]
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (W + FTW - 1) / FTW, tiles_y = (H + FTH - 1) / FTH;
    const int total = tiles_x * tiles_y * N;
    int boff[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = jt * 32 + m;
        const int c = j / 9, r9 = j - c * 9, dy = r9 / 3, dx = r9 - dy * 3;
        boff[jt] = j < K ? ((dx * CIN + c) * (FTH + 2) + dy) * FXW : -1;
    }
    const int gi = lane & 15, gg = lane >> 4;
    const int gq_off = (8 * hi + (gi >> 2)) * 32 + (gg & 1) * 16 + (gi & 3) * 4;          // ds_read_b64_tr_b16 addressing (conv_wgrad.hip)
    f32x16 acc[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[jt][i] = 0.f;
    float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f);      // bias sums of channels 4 (tid & 7) .. + 3 over this thread's pixels
    // Two tiles of loads in flight: a tile's gradient / halo loads are issued TWO tiles ahead, right after the buffer they reuse has been cut (one tile of
    // lead measured 1.02 ms per 8 frames on a 144 ms-per-step board, two tiles 0.96 ms on a 140 ms one: worth about 3 %; 256 VGPRs, no spills).
    FirstHalo<CIN> hr[2];
    float4 gr[2][8];                                   // a thread's 8 units of the 256-pixel x 32-channel gradient tile, two tiles
    auto prefetch = [&](int t, auto B) {
        constexpr int b = decltype(B)::value;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
        const int y0 = ty * FTH, x0 = tx * FTW;
        hr[b].load(x, img, y0, x0, H, W);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int u = threadIdx.x + it * 256;
            const int lp = u >> 3, part4 = u & 7;
            const int gy = y0 + lp / FTW, gx = x0 + lp % FTW;
            gr[b][it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W) gr[b][it] = *reinterpret_cast<const float4*>(g + ((size_t)(img * H + gy) * W + gx) * 32 + part4 * 4);
        }
    };
    auto top = [](float v) { return __uint_as_float(__float_as_uint(v) & 0xFFFF0000u); };
    auto hp = [](float lo, float hi_) { return __builtin_amdgcn_perm(__float_as_uint(hi_), __float_as_uint(lo), 0x07060302u); };      // the two top halves as a bf16 pair
    const int G = (int)gridDim.x;
    auto tile = [&](int t, auto B) {
        constexpr int b = decltype(B)::value;
        __syncthreads();                                        // the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < 8; ++it) {                        // gradient tile -> three bf16 planes
            const int u = tid + it * 256;
            const float4 v = gr[b][it];
            bs4.x += v.x; bs4.y += v.y; bs4.z += v.z; bs4.w += v.w;
            const float4 r = make_float4(v.x - top(v.x), v.y - top(v.y), v.z - top(v.z), v.w - top(v.w));
            const float4 q = make_float4(r.x - top(r.x), r.y - top(r.y), r.z - top(r.z), r.w - top(r.w));
            bf16_t* d = gl + u * 4;                             // [pixel u >> 3][channels 4 (u & 7) ..]
            *reinterpret_cast<uint2*>(d) = make_uint2(hp(v.x, v.y), hp(v.z, v.w));
            *reinterpret_cast<uint2*>(d + GPL) = make_uint2(hp(r.x, r.y), hp(r.z, r.w));
            *reinterpret_cast<uint2*>(d + 2 * GPL) = make_uint2(hp(q.x, q.y), hp(q.z, q.w));
        }
#pragma unroll
        for (int it = 0; it < FirstHalo<CIN>::IT; ++it) {      // halo -> three bf16 pieces -> the three shifted copies
            const int u = tid + it * 256;
            if (u >= CIN * FHP) continue;
            const int c = u / FHP, hp_ = u - c * FHP;
            const int hy = hp_ / (FTW + 2), hx = hp_ - hy * (FTW + 2);
            const float v = hr[b].v[it];
            const float r = v - top(v), q = r - top(r);
            const bf16_t p1 = (bf16_t)(__float_as_uint(v) >> 16), p2 = (bf16_t)(__float_as_uint(r) >> 16), p3 = (bf16_t)(__float_as_uint(q) >> 16);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xx = hx - dx;
                if (xx < 0 || xx >= FTW) continue;
                const int o = ((dx * CIN + c) * (FTH + 2) + hy) * FXW + xx;
                xs[o] = p1;
                xs[o + XPL] = p2;
                xs[o + 2 * XPL] = p3;
            }
        }
        __syncthreads();
        if (t + 2 * G < total) prefetch(t + 2 * G, B);
        constexpr int GI[6] = {0, 1, 2, 0, 1, 0};
        constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                        // 16 consecutive pixels of one row per k-step; the wave owns rows 2 wave, 2 wave + 1
            const int row = 2 * wave + (ks >> 1), col0 = 16 * (ks & 1);
            fw_bf16x8 ga[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                const bf16_t* p0 = gl + pc * GPL + (row * FTW + col0) * 32 + gq_off;
                const fw_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fw_s16x4*)p0);
                const fw_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fw_s16x4*)(p0 + 4 * 32));
                ga[pc] = __builtin_bit_cast(fw_bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                uint4 bx[3] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
                if (boff[jt] >= 0) {
                    const bf16_t* pb = xs + boff[jt] + row * FXW + col0 + 8 * hi;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) bx[pc] = *reinterpret_cast<const uint4*>(pb + pc * XPL);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[GI[q]], __builtin_bit_cast(fw_bf16x8, bx[XI[q]]), acc[jt], 0, 0, 0);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    if ((int)blockIdx.x < total) prefetch(blockIdx.x, B0());
    if ((int)blockIdx.x + G < total) prefetch(blockIdx.x + G, B1());
    for (int t = blockIdx.x; t < total; t += 2 * G) {
        tile(t, B0());
        if (t + G < total) tile(t + G, B1());
    }
    // reduce the 4 waves through LDS (fixed order), wave 0 writes the partial -- as conv_first_wgrad_kernel
    __syncthreads();
    float* red = reinterpret_cast<float*>(x3lds);                 // [3 waves][2][16][64] floats = 24 KB, then [256 threads][4] bias sums = 4 KB
    if (wave > 0) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(((wave - 1) * 2 + jt) * 16 + i) * 64 + lane] = acc[jt][i];
    }
    float4* bred = reinterpret_cast<float4*>(red + 3 * 2 * 16 * 64);
    bred[tid] = bs4;
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
        if (lane < 32) {                                                      // channel `lane`: threads lane / 4 + 8 k hold its sums (component lane % 4)
            const float* bf = reinterpret_cast<const float*>(bred);
            float b = 0.f;
            for (int k = 0; k < 32; ++k) b += bf[((lane >> 2) + 8 * k) * 4 + (lane & 3)];
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

// x3: the call runs under the three-piece product scheme (conv_fp32_algo 1): the packed-raw layer (Cin = 4) goes to the bf16-MFMA kernel
int launch_conv_first_wgrad(const float* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st, bool x3) {
    const int tiles = ((W + FTW - 1) / FTW) * ((H + FTH - 1) / FTH) * N;
    if (tiles <= 0) return 0;
    if (!x3 || Cin != 4) return launch_first_wgrad_t<float>(g, x, dw, db, part, N, Cin, H, W, st);      // fp32-MFMA kernel
    constexpr size_t lds_bytes = (size_t)(3 * FTH * FTW * 32 + 3 * 3 * 4 * (FTH + 2) * FXW) * sizeof(bf16_t);
    auto kern = conv_first_wgrad_x3_kernel<4>;
    static EldAttrOnce once;
    { const int rc = once.ensure(kern, lds_bytes); if (rc) return rc; }
    const int resident = 2 * eld_num_cus();                  // 77 KB of LDS per workgroup: two per CU; the persistent grid must not exceed what is co-resident
    const int cap = resident < FW_BLOCKS ? resident : FW_BLOCKS;
    const int grid = tiles < cap ? tiles : cap;
    ELD_LAUNCH(kern, dim3(grid), dim3(256), lds_bytes, st, g, x, part, N, H, W);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(conv_first_wgrad_reduce_kernel, dim3((2 * 32 * 32 + 32 + 15) / 16), dim3(256), 0, st, part, dw, db, grid, 9 * Cin);
    ELD_LAUNCH_CHECK();
    return 0;
}
int launch_conv_first_wgrad_bf16(const bf16_t* g, const float* x, float* dw, float* db, float* part, int N, int Cin, int H, int W, hipStream_t st) {
    const int tiles = ((W + FTW - 1) / FTW) * ((H + FTH - 1) / FTH) * N;
    if (tiles <= 0) return 0;
    if (Cin != 4 || (size_t)N * H * W * 64 >= 0xFFFFFFF0ull) return launch_first_wgrad_t<bf16_t>(g, x, dw, db, part, N, Cin, H, W, st);      // generic fp32-MFMA kernel
    const int resident = 3 * eld_num_cus();                  // 51 KB of LDS per workgroup: three per CU; the persistent grid must not exceed what is co-resident
    const int cap = resident < FW_BLOCKS ? resident : FW_BLOCKS;
    const int grid = tiles < cap ? tiles : cap;
    ELD_LAUNCH((conv_first_wgrad_bf16_kernel<4>), dim3(grid), dim3(256), 0, st, g, x, part, N, H, W);
    ELD_LAUNCH_CHECK();
    ELD_LAUNCH(conv_first_wgrad_reduce_kernel, dim3((2 * 32 * 32 + 32 + 15) / 16), dim3(256), 0, st, part, dw, db, grid, 9 * Cin);
    ELD_LAUNCH_CHECK();
    return 0;
}
