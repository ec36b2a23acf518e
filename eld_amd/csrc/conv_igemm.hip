// conv_igemm.hip -- implicit-GEMM convolution on the exact-fp32 MFMA (gfx950), NHWC float32.
//
// One kernel family for: conv3x3 forward (+bias +LeakyReLU epilogue), conv3x3 backward-data (same
// kernel on flipped/transposed packed weights, epilogue multiplies by the LeakyReLU slope of the
// saved activation and splits the channel range for virtual concats), transposed-conv 2x2/s2
// forward (1x1 mode, scatter epilogue) and its backward-data (2x2 gather mode).
// Replaces nn.Conv2d / nn.ConvTranspose2d + torch.max(0.2x,x) of models/arch/Unet.py:11-46,48-91,102-104
// and their autograd backward.
//
// Tiling (per 256-thread workgroup = 4 waves):
//   * output tile = TH rows x 32 columns of pixels x BN output channels, TH = 4*RPW;
//     wave w owns rows [w*RPW, (w+1)*RPW) x all BN channels: RPW x (BN/32) accumulator tiles of
//     32x32 (v_mfma_f32_32x32x2_f32: A = 32 pixels x 2 k, B = 2 k x 32 channels, 16 acc VGPRs each);
//   * K loop = chunks of CK=16 input channels; per chunk the (TH+2)x34 input halo tile and the
//     [taps][BN][16] weight slab are staged in LDS once and reused by all taps (9x input reuse from
//     LDS instead of HBM/L2);
//   * LDS rows are [pixel][16 ch + 4 pad] (20-word stride): a lane's 8 channels for the k-halves
//     trick below are two ds_read_b128, and 20*p mod 64 hits 16 distinct 4-word slots for any 16
//     pixels distinct mod 16 -> conflict-free for the b128 lane groups;
//   * k-halves: MFMA lane l supplies k = l>>5.  Lanes 0-31 walk channels [0,8) of the chunk, lanes
//     32-63 walk [8,16): each MFMA consumes channel s from the low half and 8+s from the high half.
//     The summation order over k is a permutation of the reference's -- results agree to fp32
//     round-off, not bitwise (tests compare against torch fp32 and fp64).
// fp32 MFMA issues at 64 cycles/instruction/SIMD, so one chunk (9 taps x 8 k-steps x RPW*BN/32 tiles)
// is >= 18k cycles of matrix work per wave against ~60 KB of staged operands: the kernel is
// MFMA-bound and a plain stage -> barrier -> compute -> barrier loop with >= 2 workgroups per CU
// (LDS <= 80 KB each) keeps the matrix pipe busy while the other workgroup stages.
#include "conv.h"

#define CK 16
#define PS 20           // LDS pixel stride in words (16 + 4 pad)
#define TW 32

template <int MODE, int RPW>
struct Geo {
    static constexpr int TH = 4 * RPW;
    static constexpr int TAPS = MODE == CONV_3X3 ? 9 : (MODE == CONV_1X1 ? 1 : 4);
    static constexpr int A_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : (MODE == CONV_1X1 ? TH * TW : 4 * TH * TW);
};

template <int MODE, int BN, int RPW>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    using G = Geo<MODE, RPW>;
    constexpr int TH = G::TH, TAPS = G::TAPS, A_PIX = G::A_PIX, NT = BN / 32;
    constexpr int A_WORDS = A_PIX * PS, B_ROWS = TAPS * BN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsA = lds;
    float* ldsB = lds + A_WORDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int NB = a.Nout / BN;
    const int bid = blockIdx.x;
    const int nb = bid % NB;
    int tile = bid / NB;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int img = tile / a.tiles_y;
    const int Cin = a.C0 + a.C1;
    const int y0 = ty * TH, x0 = tx * TW;
    const int Hs = MODE == CONV_GATHER2X2 ? 2 * a.H : a.H;     // source dims
    const int Ws = MODE == CONV_GATHER2X2 ? 2 * a.W : a.W;

    f32x16 acc[RPW][NT];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][t][i] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += CK) {
        const float* src;
        int Cs, cs;
        if (c0 < a.C0) { src = a.in0; Cs = a.C0; cs = c0; } else { src = a.in1; Cs = a.C1; cs = c0 - a.C0; }
        __syncthreads();
        // ---- stage A: input tile, 4 float4 per pixel ------------------------------------------
        for (int u = tid; u < A_PIX * 4; u += 256) {
            const int hp = u >> 2, part = u & 3;
            int gy, gx;
            bool ok;
            if (MODE == CONV_3X3) {
                const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
                gy = y0 + hy - 1; gx = x0 + hx - 1;
                ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            } else if (MODE == CONV_1X1) {
                const int py = hp / TW, px = hp - py * TW;
                gy = y0 + py; gx = x0 + px;
                ok = gy < a.H && gx < a.W;
            } else {
                const int tap = hp / (TH * TW), lp = hp - tap * (TH * TW);
                const int py = lp / TW, px = lp - py * TW;
                ok = (y0 + py) < a.H && (x0 + px) < a.W;
                gy = 2 * (y0 + py) + (tap >> 1); gx = 2 * (x0 + px) + (tap & 1);
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(src + ((size_t)(img * Hs + gy) * Ws + gx) * Cs + cs + part * 4);
            *reinterpret_cast<float4*>(ldsA + hp * PS + part * 4) = v;
        }
        // ---- stage B: weights [taps][BN][16] ----------------------------------------------------
        for (int u = tid; u < B_ROWS * 4; u += 256) {
            const int row = u >> 2, part = u & 3;
            const int tap = row / BN, n = row - tap * BN;
            const float4 v = *reinterpret_cast<const float4*>(a.wp + ((size_t)tap * a.Nout + nb * BN + n) * Cin + c0 + part * 4);
            *reinterpret_cast<float4*>(ldsB + row * PS + part * 4) = v;
        }
        __syncthreads();
        // ---- MFMA over taps x 8 k-steps -----------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            int aoff[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int row = wave * RPW + r;
                if (MODE == CONV_3X3) aoff[r] = ((row + tap / 3) * (TW + 2) + m + tap % 3) * PS + hi * 8;
                else if (MODE == CONV_1X1) aoff[r] = (row * TW + m) * PS + hi * 8;
                else aoff[r] = (tap * TH * TW + row * TW + m) * PS + hi * 8;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 av[RPW], bv[NT];
#pragma unroll
                for (int r = 0; r < RPW; ++r) av[r] = *reinterpret_cast<const float4*>(ldsA + aoff[r] + q * 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const float4*>(ldsB + ((tap * BN + t * 32 + m) * PS + hi * 8) + q * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        const float af = kk == 0 ? av[r].x : kk == 1 ? av[r].y : kk == 2 ? av[r].z : av[r].w;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const float bf = kk == 0 ? bv[t].x : kk == 1 ? bv[t].y : kk == 2 ? bv[t].z : bv[t].w;
                            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[r][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: D layout col = lane&31 (channel), row = (i&3) + 8*(i>>2) + 4*hi (pixel x) ------------
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int y = y0 + wave * RPW + r;
        if (y >= a.H) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = nb * BN + t * 32 + m;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int x = x0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (x >= a.W) continue;
                float v = acc[r][t][i];
                const size_t pix = (size_t)(img * a.H + y) * a.W + x;
                if (a.epi == EPI_FWD) {
                    v += a.bias[n];
                    if (a.lrelu) v = fmaxf(0.2f * v, v);
                    a.out0[pix * a.Nout + n] = v;
                } else if (a.epi == EPI_CONVT_FWD) {
                    const int tap = n / a.Cout_t, co = n - tap * a.Cout_t;
                    const size_t op = (size_t)(img * 2 * a.H + 2 * y + (tap >> 1)) * (2 * a.W) + 2 * x + (tap & 1);
                    a.out0[op * a.Cout_t + co] = v + a.bias[co];
                } else {
                    if (n < a.split) {
                        const size_t idx = pix * a.split + n;
                        if (a.act0) v *= lrelu_slope(a.act0[idx]);
                        a.out0[idx] = v;
                    } else {
                        const size_t idx = pix * (a.Nout - a.split) + (n - a.split);
                        if (a.act1) v *= lrelu_slope(a.act1[idx]);
                        a.out1[idx] = v;
                    }
                }
            }
        }
    }
}

template <int MODE, int BN, int RPW>
static int launch_t(ConvArgs a, hipStream_t st) {
    using G = Geo<MODE, RPW>;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    const size_t lds_bytes = (size_t)(G::A_PIX + G::TAPS * BN) * PS * sizeof(float);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * a.N * (a.Nout / BN);
    if (blocks <= 0) return 0;
    if (blocks > 0x7fffffffLL) return ELD_ENOTSUP;
    auto kern = conv_igemm_kernel<MODE, BN, RPW>;
    static bool attr_set = false;     // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_conv(const ConvArgs& a, int mode, hipStream_t st) {
    const int Cin = a.C0 + a.C1;
    if (Cin % CK || a.C0 % CK || a.Nout % 32) return ELD_EINVAL;
    if (a.epi == EPI_GRAD && (a.split % 32)) return ELD_EINVAL;
    const bool n64 = a.Nout % 64 == 0;
    switch (mode) {
        case CONV_3X3: return n64 ? launch_t<CONV_3X3, 64, 2>(a, st) : launch_t<CONV_3X3, 32, 2>(a, st);
        case CONV_1X1: return n64 ? launch_t<CONV_1X1, 64, 2>(a, st) : launch_t<CONV_1X1, 32, 2>(a, st);
        case CONV_GATHER2X2: return n64 ? launch_t<CONV_GATHER2X2, 64, 1>(a, st) : launch_t<CONV_GATHER2X2, 32, 1>(a, st);
    }
    return ELD_EINVAL;
}
