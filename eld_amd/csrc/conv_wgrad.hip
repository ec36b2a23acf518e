// conv_wgrad.hip -- weight-gradient reductions of the U-Net on the exact-fp32 MFMA (gfx950).
//
//      P[tap][i][j] = sum over pixels p of  G[p][i] * X[p (+) tap][j]
//   conv3x3 (MODE 0):   G = grad wrt the conv's pre-activation output (CA = Cout), X = the conv's
//                       input with the 3x3 shift (CB = Cin)            ->  dW[co][ci][dy][dx]
//   convT 2x2/s2 (MODE 2): G = the layer's input (CA = Cin), X = grad of its output gathered at
//                       (2y+dy, 2x+dx) (CB = Cout)                     ->  dW[ci][co][dy][dx]
// plus, for free, the bias gradient sum_p G[p][i].  Replaces autograd's conv weight/bias backward
// for models/arch/Unet.py:11-46.
//
// GEMM view: M = i (32 per MFMA tile), N = j (32), K = pixels.  v_mfma_f32_32x32x2_f32 takes
// A[i][k] = G[pixel_k][i] and B[k][j] = X[pixel_k (+) tap][j]: with NHWC both are 32 consecutive
// floats of one pixel across lanes 0-31 -> conflict-free ds_read_b32, no transposes.
// Workgroup = 4 waves = WCO (i-tiles) x WPIX (pixel slices); each wave keeps all TAPS 32x32
// accumulators (9*16 = 144 VGPRs) and walks its slice of every spatial tile assigned to the
// workgroup (persistent over tiles: the K reduction stays in registers), so the only HBM writes are
// one partial per workgroup.  Partials are summed by a second kernel in a fixed order
// (run-to-run bit-stable, no atomics).
#include "conv.h"

#define TW 32
#define JB 32     // j (shifted-operand channel) tile

template <int MODE, int WCO, int TH>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int TAPS = MODE == CONV_3X3 ? 9 : 4;
    constexpr int WPIX = 4 / WCO;
    constexpr int COB = 32 * WCO;
    constexpr int TPIX = TH * TW;
    constexpr int PW = TPIX / WPIX;        // pixels per wave per tile
    constexpr int KS = PW / 2;             // MFMA k-steps per wave per tile
    constexpr int X_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : 4 * TPIX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsG = lds;                     // [TPIX][COB]
    float* ldsX = lds + TPIX * COB;        // [X_PIX][JB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, hi = lane >> 5;
    const int wco = wave % WCO, wpix = wave / WCO;
    const int IB = a.CA / COB, JBn = a.CBp / JB;
    int bid = blockIdx.x;
    const int jb = bid % JBn; bid /= JBn;
    const int ib = bid % IB;
    const int ps = bid / IB;
    const int i0 = ib * COB, j0 = jb * JB;
    const int CB = a.C0 + a.C1;
    const int Hx = MODE == CONV_GATHER2X2 ? 2 * a.H : a.H;
    const int Wx = MODE == CONV_GATHER2X2 ? 2 * a.W : a.W;
    // source of the j tile (virtual concat); channels >= CB are zero padding
    const float* xsrc; int Cs, cs;
    if (j0 < a.C0) { xsrc = a.x0; Cs = a.C0; cs = j0; } else { xsrc = a.x1; Cs = a.C1; cs = j0 - a.C0; }
    const int jvalid = min(JB, CB - j0 > 0 ? (j0 < a.C0 ? a.C0 - j0 : CB - j0) : 0);   // channels of this tile that exist

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;

    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int ntiles = tiles_per_img * a.N;
    // register-staged pipeline: tile t+1's global loads are in flight during tile t's MFMA phase
    constexpr int G_UNITS = TPIX * (COB / 4), X_UNITS = X_PIX * (JB / 4);
    constexpr int G_IT = (G_UNITS + 255) / 256, X_IT = (X_UNITS + 255) / 256;
    float4 rg[G_IT], rx[X_IT];
    auto load_tile = [&](int tile) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int u = tid + it * 256;
            const int lp = u / (COB / 4), part = u - lp * (COB / 4);
            const int py = lp / TW, px = lp - py * TW;
            const int gy = y0 + py, gx = x0 + px;
            rg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < G_UNITS && gy < a.H && gx < a.W)
                rg[it] = *reinterpret_cast<const float4*>(a.g + ((size_t)(img * a.H + gy) * a.W + gx) * a.CA + i0 + part * 4);
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int u = tid + it * 256;
            const int hp = u / (JB / 4), part = u - hp * (JB / 4);
            int gy, gx; bool ok;
            if (MODE == CONV_3X3) {
                const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
                gy = y0 + hy - 1; gx = x0 + hx - 1;
                ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            } else {
                const int tap = hp / TPIX, lp = hp - tap * TPIX;
                const int py = lp / TW, px = lp - py * TW;
                ok = (y0 + py) < a.H && (x0 + px) < a.W;
                gy = 2 * (y0 + py) + (tap >> 1); gx = 2 * (x0 + px) + (tap & 1);
            }
            rx[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < X_UNITS && ok && part * 4 < jvalid)
                rx[it] = *reinterpret_cast<const float4*>(xsrc + ((size_t)(img * Hx + gy) * Wx + gx) * Cs + cs + part * 4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int u = tid + it * 256;
            if (u < G_UNITS) *reinterpret_cast<float4*>(ldsG + u * 4) = rg[it];          // [lp][COB] is unit-linear
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int u = tid + it * 256;
            if (u < X_UNITS) *reinterpret_cast<float4*>(ldsX + u * 4) = rx[it];          // [hp][JB] is unit-linear
        }
    };

    if (ps < ntiles) load_tile(ps);
    for (int tile = ps; tile < ntiles; tile += a.psplit) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (tile + a.psplit < ntiles) load_tile(tile + a.psplit);
        // ---- K loop over this wave's pixels: lanes 0-31 take pixel s, lanes 32-63 pixel s + KS ---------
#pragma unroll 1
        for (int s = 0; s < KS; ++s) {
            const int lp = wpix * PW + s + hi * KS;
            const float av = ldsG[lp * COB + wco * 32 + m];
            bsum += av;
            const int py = lp / TW, px = lp - py * TW;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                int xp;
                if (MODE == CONV_3X3) xp = (py + t / 3) * (TW + 2) + px + t % 3;
                else xp = t * TPIX + lp;
                const float bv = ldsX[xp * JB + m];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }

    // ---- reduce the WPIX pixel slices through LDS, one tap at a time; slice 0 writes the partial ---------
    __syncthreads();
    float* red = lds;                                  // [(WPIX-1)][WCO][16][64]
    const size_t pbase = (size_t)ps * TAPS * a.CA * a.CBp;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        if (WPIX > 1) {
            if (wpix > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((wpix - 1) * WCO + wco) * 16 + i) * 64 + lane] = acc[t][i];
            }
            __syncthreads();
            if (wpix == 0) {
                for (int w = 0; w < WPIX - 1; ++w)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += red[((w * WCO + wco) * 16 + i) * 64 + lane];
            }
            __syncthreads();
        }
        if (wpix == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
                a.part[pbase + ((size_t)t * a.CA + i0 + wco * 32 + row) * a.CBp + j0 + m] = acc[t][i];
            }
        }
    }
    if (a.bpart && jb == 0) {
        red[wave * 64 + lane] = bsum;
        __syncthreads();
        if (tid < COB) {
            const int wc = tid / 32, mm = tid % 32;
            float s = 0.f;
            for (int wp = 0; wp < WPIX; ++wp) s += red[(wp * WCO + wc) * 64 + mm] + red[(wp * WCO + wc) * 64 + 32 + mm];
            a.bpart[(size_t)ps * a.CA + i0 + tid] = s;
        }
    }
}

template <int MODE, int WCO, int TH>
static int launch_w(WgradArgs a, hipStream_t st) {
    constexpr int COB = 32 * WCO;
    constexpr int X_PIX = MODE == CONV_3X3 ? (TH + 2) * (TW + 2) : 4 * TH * TW;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    size_t lds_bytes = (size_t)(TH * TW * COB + X_PIX * JB) * sizeof(float);
    const size_t red_bytes = (size_t)4 * 16 * 64 * sizeof(float);
    if (lds_bytes < red_bytes) lds_bytes = red_bytes;
    const long long blocks = (long long)(a.CA / COB) * (a.CBp / JB) * a.psplit;
    if (blocks <= 0) return 0;
    auto kern = wgrad_kernel<MODE, WCO, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds_bytes, st, a);
    ELD_LAUNCH_CHECK();
    return 0;
}

int launch_wgrad(const WgradArgs& a, int mode, hipStream_t st) {
    if (a.CA % 32 || a.CBp % 32 || a.C0 % 4 || a.C1 % 4 || a.psplit < 1) return ELD_EINVAL;
    if (a.C1 > 0 && a.C0 % 32) return ELD_EINVAL;
    const bool c64 = a.CA % 64 == 0;
    if (mode == CONV_3X3) return c64 ? launch_w<CONV_3X3, 2, 4>(a, st) : launch_w<CONV_3X3, 1, 4>(a, st);
    if (mode == CONV_GATHER2X2) return c64 ? launch_w<CONV_GATHER2X2, 2, 2>(a, st) : launch_w<CONV_GATHER2X2, 1, 2>(a, st);
    return ELD_EINVAL;
}

// out[(i*CBr + j)*T + tap] = sum_s part[s][tap][i][j]; fixed summation order -> deterministic
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ wgrad,
                                    float* __restrict__ bgrad, int psplit, int T, int CA, int CBp, int CBr) {
    const size_t plane = (size_t)T * CA * CBp;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < plane) {
        const int j = (int)(idx % CBp);
        const int i = (int)((idx / CBp) % CA);
        const int t = (int)(idx / ((size_t)CBp * CA));
        if (j < CBr) {
            float s = 0.f;
            for (int p = 0; p < psplit; ++p) s += part[p * plane + idx];
            wgrad[((size_t)i * CBr + j) * T + t] = s;
        }
    }
    if (bgrad && idx < (size_t)CA) {
        float s = 0.f;
        for (int p = 0; p < psplit; ++p) s += bpart[(size_t)p * CA + idx];
        bgrad[idx] = s;
    }
}

int launch_wgrad_reduce(const float* part, const float* bpart, float* wgrad, float* bgrad, int psplit, int T, int CA,
                        int CBp, int CBr, hipStream_t st) {
    const size_t plane = (size_t)T * CA * CBp;
    const unsigned blocks = (unsigned)((plane + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, bpart, wgrad, bgrad, psplit, T, CA, CBp, CBr);
    ELD_LAUNCH_CHECK();
    return 0;
}
